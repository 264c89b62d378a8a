"""Install layout of the drop-in library -- packaging parity with the reference
(hexl/CMakeLists.txt:114-215, cmake/hexl/HEXLConfig.cmake.in, pkgconfig/hexl.pc.in):

    <prefix>/include/hexl/**                    the intel::hexl headers
    <prefix>/include/hexl_amd.h                 the C-ABI
    <prefix>/lib/libhexl.so, libhexl_amd.so, libhexl_debug.so (the HEXL_DEBUG flavour: the
                                                reference's debug builds name theirs hexl_debug,
                                                hexl/CMakeLists.txt:68-72; target HEXL::hexl_debug)
    <prefix>/lib/cmake/hexl-1.2.5/HEXLConfig.cmake, HEXLConfigVersion.cmake, HEXLTargets.cmake
    <prefix>/lib/pkgconfig/hexl.pc

so that `find_package(HEXL 1.2.5)` + `target_link_libraries(app HEXL::hexl)` (what SEAL /
OpenFHE do) or `pkg-config --cflags --libs hexl` resolve to this build unchanged.
The package version is the reference's, 1.2.5 with ExactVersion compatibility
(hexl/CMakeLists.txt:180-183), because that is what its consumers ask for.

    python -m hexl_amd.install <prefix>
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
if(NOT TARGET HEXL::hexl)
    add_library(HEXL::hexl SHARED IMPORTED)
    set_target_properties(HEXL::hexl PROPERTIES
        IMPORTED_LOCATION "${_hexl_prefix}/lib/libhexl.so"
        IMPORTED_NO_SONAME TRUE
        INTERFACE_INCLUDE_DIRECTORIES "${_hexl_prefix}/include"
        INTERFACE_COMPILE_FEATURES cxx_std_17
        INTERFACE_LINK_LIBRARIES HEXL::hexl_amd)
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

Libs: -L${libdir} -lhexl -lhexl_amd
Cflags: -I${includedir}
"""


def install(prefix):
    prefix = os.path.abspath(prefix)
    major, minor, patch = VERSION.split(".")
    subst = dict(version=VERSION, major=major, minor=minor, patch=patch, prefix=prefix)
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
    dbg = os.path.join(HERE, "lib", "libhexl_debug.so")
    if os.path.exists(dbg):
        shutil.copy(dbg, lib)
    open(os.path.join(cmk, "HEXLConfig.cmake"), "w").write(CONFIG % subst)
    open(os.path.join(cmk, "HEXLTargets.cmake"), "w").write(TARGETS)
    open(os.path.join(cmk, "HEXLConfigVersion.cmake"), "w").write(VERSION_FILE % subst)
    open(os.path.join(pkg, "hexl.pc"), "w").write(PKGCONFIG % subst)
    return prefix


if __name__ == "__main__":
    if len(sys.argv) != 2:
        sys.exit(__doc__)
    print("installed to", install(sys.argv[1]))
