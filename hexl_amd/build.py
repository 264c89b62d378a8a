"""Builds hexl_amd/lib/libhexl_amd.so (C-ABI + gfx950 kernels) and
hexl_amd/lib/libhexl.so (the intel::hexl C++ shim over the C-ABI) with hipcc.

In-tree build: the .so files travel with the repo snapshot to the GPU box
(they are git-ignored, not gpurun-ignored).  hipcc cross-compiles gfx950
without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# ntt_kernels.hip is compiled once per arithmetic policy (-DHEXL_AMD_TU=0..7: the kernels of
# that policy) plus once for the dispatch and the process-wide state (-DHEXL_AMD_TU=-1): the
# template instantiations are disjoint between policies and compile in parallel.
NTT_UNITS = [("ntt_kernels.hip", f"-DHEXL_AMD_TU={tu}", f"ntt_kernels.tu{tu if tu >= 0 else 'd'}")
             for tu in (2, 5, 6, 3, 4, 1, 7, 0, -1)]  # the slowest units first
CORE_SOURCES = NTT_UNITS + ["eltwise_kernels.hip", "keyswitch_kernels.hip", "capi.cpp",
                            "number_theory.cpp", "workspace.cpp"]
SHIM_SOURCES = ["hexl_shim.cpp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-command-line-argument",
          f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _headers():
    hs = []
    for d, _, fs in os.walk(os.path.join(ROOT, "include")):
        hs += [os.path.join(d, f) for f in fs]
    hs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return hs


def _compile(unit, obj_dir=None, extra=()):
    """unit: a source file name, or (source, extra flag, object base name)"""
    src, flag, base = unit if isinstance(unit, tuple) else (unit, None, os.path.basename(unit))
    obj = os.path.join(obj_dir or OBJ, base + ".o")
    path = os.path.join(CSRC, src)
    if _newer(obj, [path] + _headers()):
        return obj
    cmd = [HIPCC, f"--offload-arch={ARCH}"] + COMMON + list(extra) + ([flag] if flag else []) + ["-c", path, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x")
        cmd.insert(2, "c++")
        cmd.remove(f"--offload-arch={ARCH}")
        cmd += ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    return obj


def build_variant(tag, extra_flags, units=None):
    """Developer A/B builds: tools/libhexl_amd_<tag>.so = the core library compiled with extra
    -D flags (objects under lib/obj_<tag>/; HEXL_AMD_LIB selects it in the Python binding)."""
    obj_dir = os.path.join(LIB, "obj_" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda u: _compile(u, obj_dir, extra_flags), CORE_SOURCES))
    out = os.path.join(ROOT, "tools", f"libhexl_amd_{tag}.so")
    subprocess.check_call([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = CORE_SOURCES + [s for s in SHIM_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 4)) as ex:
        objs = dict(zip(srcs, ex.map(_compile, srcs)))
    core = os.path.join(LIB, "libhexl_amd.so")
    core_objs = [objs[s] for s in CORE_SOURCES]
    if not _newer(core, core_objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", core] + core_objs
        subprocess.check_call(cmd)
    shim = os.path.join(LIB, "libhexl.so")
    shim_objs = [objs[s] for s in SHIM_SOURCES if s in objs]
    if shim_objs and not _newer(shim, shim_objs + [core]):
        cmd = [HIPCC, "-shared", "-fPIC", "-o", shim] + shim_objs + [
            f"-L{LIB}", "-lhexl_amd", "-Wl,-rpath,$ORIGIN"]
        subprocess.check_call(cmd)
    # the debug flavour of the shim (the reference's CMake builds `hexl_debug` from the same
    # sources with HEXL_DEBUG, hexl/CMakeLists.txt:68-72): element-wise bound checks that throw
    if shim_objs:
        dbg = os.path.join(LIB, "libhexl_debug.so")
        shim_src = os.path.join(CSRC, SHIM_SOURCES[0])
        if not _newer(dbg, [shim_src, core] + _headers()):
            cmd = [HIPCC, "-x", "c++"] + COMMON + ["-DHEXL_DEBUG", "-I/opt/rocm/include",
                   "-D__HIP_PLATFORM_AMD__", "-shared", shim_src, "-o", dbg, f"-L{LIB}",
                   "-lhexl_amd", "-Wl,-rpath,$ORIGIN"]
            subprocess.check_call(cmd)
    # the static flavour: the reference's DEFAULT build is a static library (CMakeLists.txt:61
    # HEXL_SHARED_LIB OFF -> hexl/CMakeLists.txt:53-57 add_library(hexl STATIC ...)): libhexl.a =
    # the shim + the C-ABI + the kernels (every object carries its own gfx950 code object and
    # registers it at start-up: no device link step), to be linked with libamdhip64
    if shim_objs:
        static = os.path.join(LIB, "libhexl.a")
        if not _newer(static, shim_objs + core_objs):
            if os.path.exists(static):
                os.remove(static)
            subprocess.check_call(["ar", "rcs", static] + shim_objs + core_objs)
    if verbose:
        print("built", core, "and", shim if shim_objs else "(no shim yet)")
    return core


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # build.py --variant TAG -DFLAG ...
        print("built", build_variant(sys.argv[2], sys.argv[3:]))
    else:
        build(verbose=True)
