"""CPU-side checks of the product library (no GPU compute calls):
the C-ABI shared library loads, exports every symbol include/hexl_amd.h
declares, the host number-theory entry points agree with the oracle and the
reference's KATs, and compute entry points fail loudly without a GPU."""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "hexl_kat.json")))


@pytest.fixture(scope="module")
def hx():
    import hexl_amd
    return hexl_amd


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hexl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hexl_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hx):
    lib = C.CDLL(hx.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/hexl_amd.h but not exported"
    assert sorted(hx.C_ABI_SYMBOLS) == syms


def test_header_is_plain_c(tmp_path):
    """include/hexl_amd.h is the FFI boundary: it must compile as C (a cgo / JNI / ctypes stub
    binds it), with nothing but <stdint.h> / <stddef.h> types in the signatures."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "hexl_amd.h"\n'
                   "int probe(void) {\n"
                   "  hexl_amd_ntt* plan = 0; const hexl_amd_ntt* plans[1]; uint8_t slot[1] = {0};\n"
                   "  uint64_t bad = 0; void* p = 0;\n"
                   "  plans[0] = plan;\n"
                   "  return hexl_amd_ntt_forward_map(plans, 1, slot, 1, 1, 0, 0, 0, 1, 1, 0) +\n"
                   "         hexl_amd_check_bounds(0, 0, 1, &bad) + hexl_amd_host_alloc(&p, 0) +\n"
                   "         hexl_amd_pointer_kind(p) + hexl_amd_release_workspaces();\n"
                   "}\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(src)])
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hexl_amd.h")).read(), flags=re.S)
    assert "torch" not in text and "hip" not in text.replace("hexl_amd", "")  # outside comments


def test_shim_library_exports_reference_api():
    """libhexl.so carries the intel::hexl symbols a HEXL caller links against."""
    path = os.path.join(ROOT, "hexl_amd", "lib", "libhexl.so")
    assert os.path.exists(path), "build with python hexl_amd/build.py"
    import subprocess
    out = subprocess.check_output(["nm", "-DC", "--defined-only", path], text=True)
    for needle in ("intel::hexl::NTT::ComputeForward(unsigned long*, unsigned long const*, "
                   "unsigned long, unsigned long)",
                   "intel::hexl::NTT::ComputeInverse(",
                   "intel::hexl::NTT::NTT(unsigned long, unsigned long, std::shared_ptr",
                   "intel::hexl::NTT::CheckArguments(",
                   "intel::hexl::EltwiseAddMod(unsigned long*, unsigned long const*, unsigned long "
                   "const*, unsigned long, unsigned long)",
                   "intel::hexl::EltwiseAddMod(unsigned long*, unsigned long const*, unsigned long, "
                   "unsigned long, unsigned long)",
                   "intel::hexl::EltwiseSubMod(", "intel::hexl::EltwiseMultMod(",
                   "intel::hexl::EltwiseFMAMod(", "intel::hexl::EltwiseReduceMod(",
                   "intel::hexl::DyadicMultiply(", "intel::hexl::KeySwitch(",
                   "intel::hexl::EltwiseCmpAdd(unsigned long*, unsigned long const*, unsigned long, "
                   "intel::hexl::CMPINT, unsigned long, unsigned long)",
                   "intel::hexl::EltwiseCmpSubMod(unsigned long*, unsigned long const*, unsigned "
                   "long, unsigned long, intel::hexl::CMPINT, unsigned long, unsigned long)",
                   "intel::hexl::MinimalPrimitiveRoot(", "intel::hexl::GeneratePrimes(",
                   "intel::hexl::IsPrime(", "intel::hexl::InverseMod(", "intel::hexl::PowMod(",
                   "intel::hexl::ReverseBits(", "intel::hexl::mallocStrategy"):
        assert needle in out, needle
    # the shim reaches the kernels only through the C-ABI: no HIP symbols of its own
    und = subprocess.check_output(["nm", "-D", "--undefined-only", path], text=True)
    assert "hexl_amd_ntt_forward" in und and "hip" not in und.lower().replace("hexl_amd", "")


def test_host_number_theory_matches_kats(hx):
    nt = KAT["number_theory"]
    for m, x, y, e in nt["multiply_mod"]:
        assert hx.MultiplyMod(x, y, m) == e
    for m, b, e, r in nt["pow_mod"]:
        assert hx.PowMod(b, e, m) == r
    for m, root, deg, r in nt["is_primitive_root"]:
        assert hx.IsPrimitiveRoot(root, deg, m) == r
    for m, deg, r in nt["minimal_primitive_root"]:
        assert hx.MinimalPrimitiveRoot(deg, m) == r
    for x, m, r in nt["inverse_mod"]:
        assert hx.InverseMod(x, m) == r
    for x, w, r in nt["reverse_bits"]:
        assert hx.ReverseBits(x, w) == r
    for n, r in nt["is_prime"]:
        assert hx.IsPrime(n) == r
    for c in KAT["generate_primes_survey_probe"]["cases"]:
        assert hx.GeneratePrimes(*c["args"]) == c["out"]
    for c in KAT["ntt_minimal_root_survey_probe"]["cases"]:
        assert hx.MinimalPrimitiveRoot(2 * c["n"], c["q"]) == c["w"]


def test_host_number_theory_matches_oracle(hx):
    from oracle import hexl_oracle as ho
    import random
    rng = random.Random(7)
    for bits in (20, 33, 49, 54, 60, 61):
        for n in (16, 1024, 65536):
            if n.bit_length() + 4 > bits:
                continue  # too few candidates == 1 mod 2n in that range
            for small in (True, False):
                assert hx.GeneratePrimes(3, bits, small, n) == ho.generate_primes(3, bits, small, n)
        q = ho.generate_primes(1, bits, True, 1024)[0]
        assert hx.MinimalPrimitiveRoot(2048, q) == ho.minimal_primitive_root(2048, q)
        for _ in range(50):
            x, y = rng.randrange(q), rng.randrange(1, q)
            assert hx.MultiplyMod(x, y, q) == ho.multiply_mod(x, y, q)
            assert hx.InverseMod(y, q) == ho.inverse_mod(y, q)
            assert hx.PowMod(x, y, q) == ho.pow_mod(x, y, q)
            for shift in (32, 52, 64):
                assert hx.MultiplyFactor(x, shift, q) == ho.multiply_factor(x, shift, q)
    assert hx.NTT.CheckArguments(1024, 0xffffee001)
    assert not hx.NTT.CheckArguments(1000, 0xffffee001)
    assert not hx.NTT.CheckArguments(1024, 0xffffee001 + 2048)  # == 1 mod 2N but composite?
    assert not hx.NTT.CheckArguments(1 << 21, 0xffffee001)


def test_compute_fails_loudly_without_gpu(hx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hx.HexlAmdError):
        hx.NTT(1024, 0xffffee001)
    x = torch.zeros(8, dtype=torch.int64)
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseAddMod(x, x, x, 8, 769)
    # straight through the C-ABI as well: plan creation needs a device
    h = C.c_void_p()
    rc = hx.lib.hexl_amd_ntt_create(C.byref(h), 1024, 0xffffee001, 0, -1)
    assert rc != 0 and hx.lib.hexl_amd_last_error()


def test_no_product_file_touches_the_oracle():
    """The oracle is test infrastructure: nothing under hexl_amd/ or include/ may
    reference it."""
    for base in ("hexl_amd", "include"):
        for d, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    text = open(os.path.join(d, f), errors="ignore").read()
                    assert "hexl_oracle" not in text and "oracle/" not in text, os.path.join(d, f)


def test_bench_uses_the_oracle_in_its_cpu_baseline_legs_only():
    """bench.py may import / call the oracle only as the reported CPU baseline (`cpu_baseline`,
    and its per-call leg for `host_path`): never inside what is measured as the product."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    allowed = {"cpu_baseline", "cpu_per_call_baseline"}
    for node in tree.body:
        uses = [n for n in ast.walk(node)
                if (isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle"))
                or (isinstance(n, ast.Import) and any(a.name.startswith("oracle") for a in n.names))]
        if uses:
            assert isinstance(node, ast.FunctionDef) and node.name in allowed, getattr(node, "name", node)
    # and tests/cpp/multi_device.cpp, the C++ caller bench.py --launcher threads runs, is C-ABI only
    text = open(os.path.join(ROOT, "tests", "cpp", "multi_device.cpp")).read()
    assert "hip/" not in text and "oracle" not in text and '#include "hexl_amd.h"' in text


def test_c_abi_header_is_plain_c(tmp_path):
    """include/hexl_amd.h is a C header: C99, pedantic, no C++ or HIP types, and links against
    the library from a C translation unit."""
    import subprocess
    src = tmp_path / "c_abi_check.c"
    src.write_text('#include "hexl_amd.h"\n'
                   'int main(void) { int n = 0; (void)hexl_amd_device_count(&n);\n'
                   '  return hexl_amd_last_error() == 0; }\n')
    exe = tmp_path / "c_abi_check"
    lib = os.path.join(ROOT, "hexl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic",
                           "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L" + lib, "-lhexl_amd", "-Wl,-rpath," + lib])
    assert os.path.exists(exe)


def test_every_tuning_key_and_counter_is_documented_in_the_header():
    """hexl_amd_set_tuning / hexl_amd_get_counter accept exactly the keys include/hexl_amd.h documents:
    every `strcmp(key, "...")` of the library's sources is named in the header, and the header names
    no key the sources do not know (removed knobs are listed as removed, in parentheses)."""
    import re
    csrc = os.path.join(ROOT, "hexl_amd", "csrc")
    accepted = set()
    for name in ("capi.cpp", "ntt_kernels.hip"):
        accepted |= set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', open(os.path.join(csrc, name)).read()))
    assert {"fp64", "fp64_long", "lazy_family", "h60", "tile13", "bigtile", "walk14", "host_bounce_kb",
            "host_direct_copy", "host_copy_threads", "host_poll", "ks_graph", "ks_fuse", "ks_mac_onestep", "ks_graph_replays", "ks_graph_captures", "ks_eager", "host_polls", "host_poll_timeouts"} <= accepted
    header = open(os.path.join(ROOT, "include", "hexl_amd.h")).read()
    documented = set(re.findall(r'^ \*   "([a-z0-9_]+)"', header, re.M)) | set(
        re.findall(r'"(ks_[a-z_]+|host_polls|host_poll_timeouts)"', header))
    assert accepted <= documented, sorted(accepted - documented)
    assert documented <= accepted, sorted(documented - accepted)
