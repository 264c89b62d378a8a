"""Packaging parity: the install tree answers find_package(HEXL 1.2.5)
and pkg-config like the reference's (hexl/CMakeLists.txt:114-215)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def prefix(tmp_path_factory):
    from hexl_amd.install import install
    return install(str(tmp_path_factory.mktemp("prefix")))


@pytest.fixture(scope="module")
def static_prefix(tmp_path_factory):
    """The reference's DEFAULT flavour: HEXL_SHARED_LIB OFF, HEXL::hexl a static library
    (/root/reference/CMakeLists.txt:61, hexl/CMakeLists.txt:53-57)."""
    from hexl_amd.install import install
    return install(str(tmp_path_factory.mktemp("static_prefix")), static=True)


def test_install_tree_layout(prefix):
    for rel in ("include/hexl/hexl.hpp", "include/hexl/ntt/ntt.hpp", "include/hexl_amd.h",
                "lib/libhexl.so", "lib/libhexl_amd.so", "lib/cmake/hexl-1.2.5/HEXLConfig.cmake",
                "lib/cmake/hexl-1.2.5/HEXLConfigVersion.cmake",
                "lib/cmake/hexl-1.2.5/HEXLTargets.cmake", "lib/pkgconfig/hexl.pc"):
        assert os.path.exists(os.path.join(prefix, rel)), rel
    pc = open(os.path.join(prefix, "lib", "pkgconfig", "hexl.pc")).read()
    assert "Version: 1.2.5" in pc and "-lhexl" in pc


def test_static_flavour_links_without_the_shared_objects(static_prefix, tmp_path):
    """find_package(HEXL 1.2.5) + HEXL::hexl against the static install tree: the consumer carries
    the shim, the C-ABI and the kernels itself (libhexl.a) and depends on the HIP runtime only."""
    if shutil.which("cmake") is None:
        pytest.skip("cmake not available")
    assert os.path.exists(os.path.join(static_prefix, "lib", "libhexl.a"))
    targets = open(os.path.join(static_prefix, "lib", "cmake", "hexl-1.2.5", "HEXLTargets.cmake")).read()
    assert "HEXL::hexl_static" in targets and "INTERFACE_LINK_LIBRARIES HEXL::hexl_static" in targets
    r, exe = _configure_and_build(static_prefix, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    deps = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libhexl" not in deps and "libamdhip64" in deps, deps
    pc = open(os.path.join(static_prefix, "lib", "pkgconfig", "hexl.pc")).read()
    assert "/libhexl.a" in pc and "-lamdhip64" in pc and "-lhexl_amd" not in pc and "-lhexl " not in pc


def _pc_flags(prefix):
    """Cflags / Libs of the installed hexl.pc with its variables expanded (what `pkg-config
    --cflags --libs hexl` prints; pkg-config itself is not in this image)."""
    var, out = {}, {}
    for line in open(os.path.join(prefix, "lib", "pkgconfig", "hexl.pc")):
        line = line.strip()
        if "=" in line and ":" not in line.split("=")[0]:
            k, v = line.split("=", 1)
            var[k] = v
        elif ":" in line:
            k, v = line.split(":", 1)
            out[k] = v.strip()
    def expand(t):
        for _ in range(4):
            for k, v in var.items():
                t = t.replace("${%s}" % k, v)
        return t.split()
    return expand(out["Cflags"]), expand(out["Libs"])


@pytest.mark.parametrize("flavour", ["shared", "static"])
def test_pkg_config_consumer_links_the_flavour_it_names(flavour, prefix, static_prefix, tmp_path):
    """A consumer built from hexl.pc alone: the shared tree depends on libhexl.so + libhexl_amd.so,
    the static tree on neither (libhexl.so sits next to libhexl.a there and must not win)."""
    tree = static_prefix if flavour == "static" else prefix
    cflags, libs = _pc_flags(tree)
    exe = tmp_path / "consumer"
    cmd = ["g++", "-std=c++17", "-O1"] + cflags + [
        os.path.join(ROOT, "tests", "cpp", "consumer", "consumer.cpp")] + libs + [
        "-pthread", "-Wl,-rpath," + os.path.join(tree, "lib"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr
    deps = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    if flavour == "static":
        assert "libhexl" not in deps and "libamdhip64" in deps, deps
    else:
        assert "libhexl.so" in deps and "libhexl_amd.so" in deps, deps


def test_every_umbrella_header_of_the_reference_is_installed(prefix, tmp_path):
    """hexl/include/hexl/hexl.hpp:6-26 names 21 headers; all but the FFT-like and LR mat-vec
    experiments (out of scope) exist in the install tree, plus ntt-cache.hpp and
    locks.hpp; the consumer that uses each compiles in the Release and the HEXL_DEBUG flavour."""
    for rel in ("logging/logging.hpp", "experimental/seal/dyadic-multiply-internal.hpp",
                "experimental/seal/key-switch-internal.hpp", "experimental/seal/ntt-cache.hpp",
                "experimental/seal/locks.hpp", "experimental/seal/dyadic-multiply.hpp",
                "experimental/seal/key-switch.hpp", "util/check.hpp", "util/compiler.hpp",
                "util/defines.hpp", "util/types.hpp", "util/util.hpp", "ntt/ntt.hpp",
                "number-theory/number-theory.hpp", "eltwise/eltwise-reduce-mod.hpp"):
        assert os.path.exists(os.path.join(prefix, "include", "hexl", rel)), rel
    src = os.path.join(ROOT, "tests", "cpp", "consumer", "consumer.cpp")
    for flags, lib in (([], "-lhexl"), (["-DHEXL_DEBUG"], "-lhexl_debug")):
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror"] + flags +
                           ["-I" + os.path.join(prefix, "include"), src,
                            "-L" + os.path.join(prefix, "lib"), lib, "-lhexl_amd", "-pthread",
                            "-o", str(tmp_path / ("consumer" + lib))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_static_consumer_runs_on_gpu(static_prefix, tmp_path):
    if shutil.which("cmake") is None:
        pytest.skip("cmake not available")
    r, exe = _configure_and_build(static_prefix, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "consumer OK" in out.stdout, out.stdout + out.stderr


def _configure_and_build(prefix, tmp_path, version="1.2.5"):
    src = tmp_path / "src"
    shutil.copytree(os.path.join(ROOT, "tests", "cpp", "consumer"), src)
    if version != "1.2.5":
        p = src / "CMakeLists.txt"
        p.write_text(p.read_text().replace("HEXL 1.2.5", "HEXL " + version))
    build = tmp_path / "build"
    r = subprocess.run(["cmake", "-S", str(src), "-B", str(build),
                        "-DCMAKE_PREFIX_PATH=" + prefix, "-DCMAKE_BUILD_TYPE=Release"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        return r, None
    r = subprocess.run(["cmake", "--build", str(build)], capture_output=True, text=True)
    return r, build / "consumer"


def test_find_package_consumer_builds(prefix, tmp_path):
    if shutil.which("cmake") is None:
        pytest.skip("cmake not available")
    r, exe = _configure_and_build(prefix, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(exe)


def test_find_package_is_exact_version(prefix, tmp_path):
    """COMPATIBILITY ExactVersion (hexl/CMakeLists.txt:180-183): 1.2.4 must not resolve."""
    if shutil.which("cmake") is None:
        pytest.skip("cmake not available")
    r, _ = _configure_and_build(prefix, tmp_path, version="1.2.4")
    assert r.returncode != 0


@pytest.mark.gpu
def test_find_package_consumer_runs_on_gpu(prefix, tmp_path):
    if shutil.which("cmake") is None:
        pytest.skip("cmake not available")
    r, exe = _configure_and_build(prefix, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "consumer OK" in out.stdout, out.stdout + out.stderr
