"""The device arithmetic (hexl_amd/csrc/modarith.h) replayed on the CPU.

modarith.h compiles for the host as well; tests/cpp/host_arith_check.cpp runs
whole forward / inverse networks through the same butterfly, finish and ladder
functions the HIP kernels inline, checks every intermediate against the range
the policy promises and every output against the oracle.  No GPU needed.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_arithmetic_on_host(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    exe = str(tmp_path / "host_arith_check")
    subprocess.run(
        ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "hexl_amd", "csrc"),
         "-I" + os.path.join(ROOT, "oracle"),
         os.path.join(ROOT, "tests", "cpp", "host_arith_check.cpp"), "-o", exe,
         "-L" + os.path.join(ROOT, "oracle"), "-lhexl_oracle",
         "-Wl,-rpath," + os.path.join(ROOT, "oracle")],
        check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "host_arith_check OK" in out.stdout


def test_lds_access_patterns_are_conflict_free(tmp_path):
    """Every LDS access of every tile geometry the library instantiates, walked on the CPU through
    the kernels' own index functions (hexl_amd/csrc/tile_geometry.h): no bank conflicts, the swizzle
    is a permutation of the tile and linear over XOR (the kernels form addresses as a0 ^ constant)."""
    exe = str(tmp_path / "lds_conflict_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "hexl_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "lds_conflict_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "lds_conflict_check OK" in out.stdout
