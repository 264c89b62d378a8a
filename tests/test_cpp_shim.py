"""Runs the C++ drop-in API test (tests/cpp/test_shim.cpp: the reference's KATs
through intel::hexl::NTT / Eltwise* from libhexl.so, host buffers) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_shim.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_shim")
LIB = os.path.join(ROOT, "hexl_amd", "lib")


def build_exe():
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= os.path.getmtime(SRC):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{os.path.join(ROOT, 'include')}", SRC,
                           f"-L{LIB}", "-lhexl", "-lhexl_amd",
                           "-Wl,-rpath," + LIB, "-pthread", "-o", EXE])


def test_cpp_api_compiles_against_headers():
    """CPU: the public headers are self-contained and the test links."""
    build_exe()
    assert os.path.exists(EXE)


def write_key_switch_kat(path):
    """The KeySwitch known answer (tests/golden/hexl_kat.json) as plain numbers for the C++ test."""
    import json
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "hexl_kat.json")))[
        "key_switch"]["cases"][0]
    n, D, C = case["n"], case["decomp_modulus_size"], case["key_component_count"]
    words = [n, D, case["key_modulus_size"], case["rns_modulus_size"], C]
    words += case["moduli"] + case["modswitch_factors"]
    for k in case["keys"]:
        words += k
    words += case["input"][:C * D * n] + case["t_target"] + case["out"][:C * D * n]
    with open(path, "w") as f:
        f.write("\n".join(str(w) for w in words) + "\n")


@pytest.mark.gpu
def test_cpp_shim_kats_on_gpu(tmp_path):
    build_exe()
    kat = str(tmp_path / "key_switch_kat.txt")
    write_key_switch_kat(kat)
    env = dict(os.environ, LD_LIBRARY_PATH=LIB + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([EXE, kat], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C++ shim checks passed" in r.stdout


DBG_SRC = os.path.join(ROOT, "tests", "cpp", "test_shim_debug.cpp")
DBG_EXE = os.path.join(ROOT, "tests", "cpp", "test_shim_debug")


def build_debug_exe():
    if os.path.exists(DBG_EXE) and os.path.getmtime(DBG_EXE) >= os.path.getmtime(DBG_SRC):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-DHEXL_DEBUG",
                           f"-I{os.path.join(ROOT, 'include')}", DBG_SRC, f"-L{LIB}",
                           "-lhexl_debug", "-lhexl_amd", "-Wl,-rpath," + LIB, "-pthread", "-o",
                           DBG_EXE])


def test_cpp_debug_flavour_links():
    """CPU: the HEXL_DEBUG flavour of the shim (libhexl_debug.so) exists and the test links."""
    assert os.path.exists(os.path.join(LIB, "libhexl_debug.so"))
    build_debug_exe()


@pytest.mark.gpu
def test_cpp_debug_contract_on_gpu():
    """The reference's TEST(NTT, bad_input) (test/test-ntt.cpp:20-94) and the eltwise bad-input
    blocks against libhexl_debug.so: out-of-range elements throw, legal inputs do not."""
    build_debug_exe()
    env = dict(os.environ, LD_LIBRARY_PATH=LIB + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([DBG_EXE], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all debug-contract checks passed" in r.stdout


REF_EXAMPLE_SRC = "/root/reference/example/example.cpp"
REF_EXAMPLE_EXE = os.path.join(ROOT, "oracle", "_ref", "hexl_example")


def test_reference_example_compiles_unmodified():
    """CPU, in the build container only: the reference's own example program
    (example/example.cpp: every Eltwise* entry point and an NTT round trip through
    "hexl/hexl.hpp") compiles and links, unmodified and where it lies, against this repo's
    headers and libhexl.so (oracle/Makefile `ref-example`; nothing of it enters the repo)."""
    if not os.path.exists(REF_EXAMPLE_SRC):
        pytest.skip("reference tree not present")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref-example"])
    assert os.path.exists(REF_EXAMPLE_EXE)


@pytest.mark.gpu
def test_reference_example_runs_on_gpu():
    """The binary built above runs on the GPU: its eight examples complete and none of its own
    comparisons against the reference's expected vectors reports a mismatch."""
    if not os.path.exists(REF_EXAMPLE_EXE):
        pytest.skip("oracle/_ref/hexl_example was not built (needs the reference tree at build time)")
    r = subprocess.run([REF_EXAMPLE_EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Not equal" not in r.stdout, r.stdout
    assert r.stdout.count("Done running") == 8, r.stdout
