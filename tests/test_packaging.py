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
    assert "-lhexl -L" in pc and "-lamdhip64" in pc and "-lhexl_amd" not in pc


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
