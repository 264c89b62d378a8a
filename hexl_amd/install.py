"""Install layout of the drop-in library -- packaging parity with the reference
(hexl/CMakeLists.txt:114-215, cmake/hexl/HEXLConfig.cmake.in, pkgconfig/hexl.pc.in):

    <prefix>/include/hexl/**                    the intel::hexl headers
    <prefix>/include/hexl_amd.h                 the C-ABI
    <prefix>/lib/libhexl.so, libhexl_amd.so, libhexl_debug.so (the HEXL_DEBUG flavour: the
                                                reference's debug builds name theirs hexl_debug,
                                                hexl/CMakeLists.txt:68-72; target HEXL::hexl_debug)
    <prefix>/lib/libhexl.a                      the static flavour (shim + C-ABI + kernels in one
                                                archive): the reference's DEFAULT build is static
                                                (CMakeLists.txt:61 HEXL_SHARED_LIB OFF ->
                                                hexl/CMakeLists.txt:53-57)
    <prefix>/lib/cmake/hexl-1.2.5/HEXLConfig.cmake, HEXLConfigVersion.cmake, HEXLTargets.cmake
    <prefix>/lib/pkgconfig/hexl.pc

so that `find_package(HEXL 1.2.5)` + `target_link_libraries(app HEXL::hexl)` (what SEAL /
OpenFHE do) or `pkg-config --cflags --libs hexl` resolve to this build unchanged.
The package version is the reference's, 1.2.5 with ExactVersion compatibility
(hexl/CMakeLists.txt:180-183), because that is what its consumers ask for.

Which of the two `HEXL::hexl` is follows the reference's switch: `install(prefix)` /
`python -m hexl_amd.install <prefix>` gives the shared library (HEXL_SHARED_LIB=ON), `install(prefix,
static=True)` / `--static` the static archive (the reference's default); the other flavour is
always there as HEXL::hexl_static / HEXL::hexl_shared.

    python -m hexl_amd.install <prefix> [--static]
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
VERSION = "1.2.5"

CONFIG = """# HEXLConfig.cmake -- MI355X-native build of the intel/hexl hot path.
# Defines, like the reference's package (cmake/hexl/HEXLConfig.cmake.in):
#   HEXL_FOUND, HEXL_VERSION, HEXL_VERSION_MAJOR, HEXL_VERSION_MINOR, HEXL_VERSION_PATCH
# and the imported target HEXL::hexl.
include(${CMAKE_CURRENT_LIST_DIR}/HEXLTargets.cmake)
if(TARGET HEXL::hexl)
    set(HEXL_FOUND TRUE)
    message(STATUS "HEXL (hexl_amd, MI355X) found")
else()
    message(STATUS "HEXL not found")
endif()
set(HEXL_VERSION "%(version)s")
set(HEXL_VERSION_MAJOR "%(major)s")
set(HEXL_VERSION_MINOR "%(minor)s")
set(HEXL_VERSION_PATCH "%(patch)s")
set(HEXL_DEBUG "OFF")
"""

TARGETS = """# HEXLTargets.cmake -- imported targets of the hexl_amd install tree.
get_filename_component(_hexl_prefix "${CMAKE_CURRENT_LIST_DIR}/../../.." ABSOLUTE)
if(NOT TARGET HEXL::hexl_amd)
    add_library(HEXL::hexl_amd SHARED IMPORTED)
    set_target_properties(HEXL::hexl_amd PROPERTIES
        IMPORTED_LOCATION "${_hexl_prefix}/lib/libhexl_amd.so"
        IMPORTED_NO_SONAME TRUE
        INTERFACE_INCLUDE_DIRECTORIES "${_hexl_prefix}/include")
endif()
if(NOT TARGET HEXL::hexl_shared)
    add_library(HEXL::hexl_shared SHARED IMPORTED)
    set_target_properties(HEXL::hexl_shared PROPERTIES
        IMPORTED_LOCATION "${_hexl_prefix}/lib/libhexl.so"
        IMPORTED_NO_SONAME TRUE
        INTERFACE_INCLUDE_DIRECTORIES "${_hexl_prefix}/include"
        INTERFACE_COMPILE_FEATURES cxx_std_17
        INTERFACE_LINK_LIBRARIES HEXL::hexl_amd)
endif()
if(NOT TARGET HEXL::hexl_static AND EXISTS "${_hexl_prefix}/lib/libhexl.a")
    # shim + C-ABI + kernels in one archive (the reference's default: a static library,
    # hexl/CMakeLists.txt:53-57); the HIP runtime is the one dependency left
    find_library(_hexl_hip amdhip64 HINTS %(rocm_lib)s ENV ROCM_PATH PATH_SUFFIXES lib)
    add_library(HEXL::hexl_static STATIC IMPORTED)
    set_target_properties(HEXL::hexl_static PROPERTIES
        IMPORTED_LOCATION "${_hexl_prefix}/lib/libhexl.a"
        INTERFACE_INCLUDE_DIRECTORIES "${_hexl_prefix}/include"
        INTERFACE_COMPILE_FEATURES cxx_std_17
        INTERFACE_LINK_LIBRARIES "${_hexl_hip};pthread;dl")
endif()
if(NOT TARGET HEXL::hexl)
    # HEXL_SHARED_LIB of this install tree: %(shared_lib)s
    add_library(HEXL::hexl INTERFACE IMPORTED)
    set_target_properties(HEXL::hexl PROPERTIES INTERFACE_LINK_LIBRARIES HEXL::hexl_%(flavour)s)
endif()
if(NOT TARGET HEXL::hexl_debug AND EXISTS "${_hexl_prefix}/lib/libhexl_debug.so")
    # the debug flavour: element-wise bound checks that throw (HEXL_CHECK_BOUNDS)
    add_library(HEXL::hexl_debug SHARED IMPORTED)
    set_target_properties(HEXL::hexl_debug PROPERTIES
        IMPORTED_LOCATION "${_hexl_prefix}/lib/libhexl_debug.so"
        IMPORTED_NO_SONAME TRUE
        INTERFACE_INCLUDE_DIRECTORIES "${_hexl_prefix}/include"
        INTERFACE_COMPILE_FEATURES cxx_std_17
        INTERFACE_COMPILE_DEFINITIONS HEXL_DEBUG
        INTERFACE_LINK_LIBRARIES HEXL::hexl_amd)
endif()
unset(_hexl_prefix)
"""

# write_basic_package_version_file(... COMPATIBILITY ExactVersion) semantics
VERSION_FILE = """set(PACKAGE_VERSION "%(version)s")
if(PACKAGE_FIND_VERSION VERSION_EQUAL PACKAGE_VERSION)
    set(PACKAGE_VERSION_EXACT TRUE)
    set(PACKAGE_VERSION_COMPATIBLE TRUE)
elseif(NOT PACKAGE_FIND_VERSION)
    set(PACKAGE_VERSION_COMPATIBLE TRUE)
else()
    set(PACKAGE_VERSION_COMPATIBLE FALSE)
endif()
"""

PKGCONFIG = """prefix=%(prefix)s
libdir=${prefix}/lib
includedir=${prefix}/include

Name: Intel HEXL (hexl_amd, MI355X-native hot path)
Version: %(version)s
Description: Drop-in for the NTT and element-wise modular arithmetic of Intel HEXL on AMD MI355X.

Libs: %(pc_hexl)s %(pc_libs)s
Libs.private: -L%(rocm_lib)s -lamdhip64 -lpthread -ldl
Cflags: -I${includedir}
"""


def install(prefix, static=False):
    prefix = os.path.abspath(prefix)
    major, minor, patch = VERSION.split(".")
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    subst = dict(version=VERSION, major=major, minor=minor, patch=patch, prefix=prefix,
                 rocm_lib=rocm_lib, flavour="static" if static else "shared",
                 shared_lib="OFF" if static else "ON",
                 # static: name the archive by path -- libhexl.so is installed next to it and
                 # GNU ld prefers the shared object for -lhexl
                 pc_hexl="${libdir}/libhexl.a" if static else "-L${libdir} -lhexl",
                 pc_libs=f"-L{rocm_lib} -lamdhip64 -lpthread -ldl" if static else "-lhexl_amd")
    inc = os.path.join(prefix, "include")
    lib = os.path.join(prefix, "lib")
    cmk = os.path.join(lib, "cmake", "hexl-" + VERSION)
    pkg = os.path.join(lib, "pkgconfig")
    for d in (inc, lib, cmk, pkg):
        os.makedirs(d, exist_ok=True)
    shutil.copytree(os.path.join(ROOT, "include", "hexl"), os.path.join(inc, "hexl"),
                    dirs_exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "hexl_amd.h"), inc)
    for name in ("libhexl.so", "libhexl_amd.so"):
        src = os.path.join(HERE, "lib", name)
        if not os.path.exists(src):
            raise RuntimeError(f"{src} is missing: run python hexl_amd/build.py first")
        shutil.copy(src, lib)
    for name in ("libhexl_debug.so", "libhexl.a"):
        src = os.path.join(HERE, "lib", name)
        if os.path.exists(src):
            shutil.copy(src, lib)
        elif static and name == "libhexl.a":
            raise RuntimeError(f"{src} is missing: run python hexl_amd/build.py first")
    open(os.path.join(cmk, "HEXLConfig.cmake"), "w").write(CONFIG % subst)
    open(os.path.join(cmk, "HEXLTargets.cmake"), "w").write(TARGETS % subst)
    open(os.path.join(cmk, "HEXLConfigVersion.cmake"), "w").write(VERSION_FILE % subst)
    open(os.path.join(pkg, "hexl.pc"), "w").write(PKGCONFIG % subst)
    return prefix


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--static"]
    if len(args) != 1:
        sys.exit(__doc__)
    print("installed to", install(args[0], static="--static" in sys.argv[1:]))
