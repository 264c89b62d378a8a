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


@pytest.mark.gpu
def test_cpp_shim_kats_on_gpu():
    build_exe()
    env = dict(os.environ, LD_LIBRARY_PATH=LIB + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([EXE], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C++ shim checks passed" in r.stdout
