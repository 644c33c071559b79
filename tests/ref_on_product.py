"""TEST INFRASTRUCTURE: build recipes for the two "reference sources, unchanged, on top of the product" artefacts
(tests/_build/, git-ignored, shipped to the GPU box with the snapshot).  Used by tests/conftest.py and by
__graft_entry__.build(); they need /root/reference, so they are built in the container and only run on the GPU box."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/surfel_fusion/src"
PKG = os.path.join(ROOT, "densesurfelmapping_amd")
OUT = os.path.join(ROOT, "tests", "_build")
_LINK = ["-L" + PKG, "-l:libdsm_hip.so", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]


def _run(cmd):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps if os.path.exists(d))


def build_ros_node_on_product():
    """surfel_fusion/src/ros_node.cpp, in place and unchanged, against include/ros_compat."""
    from densesurfelmapping_amd import build
    lib = build.build_library()
    out = os.path.join(OUT, "ros_node_on_product")
    src = [os.path.join(REF_SRC, "ros_node.cpp"), os.path.join(ROOT, "tests", "ros_shims", "ros_shim_bus.cpp")]
    deps = src + [lib, os.path.join(ROOT, "include", "dsm_surfel_map.hpp"), os.path.join(ROOT, "include", "ros_compat", "surfel_map.h"),
                  os.path.join(ROOT, "oracle", "shims", "ros", "ros.h")]
    if _stale(out, deps):
        os.makedirs(OUT, exist_ok=True)
        _run(["g++", "-std=c++11", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "ros_compat"),
              "-I" + os.path.join(ROOT, "tests", "ros_shims"), "-I" + os.path.join(ROOT, "oracle", "shims")] + src + ["-o", out] + _LINK)
    return out


def build_ref_map_on_product():
    """surfel_fusion/src/surfel_map.{h,cpp}, in place and unchanged, with FusionFunctions = the product's facade
    (include/engine_compat), behind the C driver of oracle/ref_map_driver.cpp."""
    from densesurfelmapping_amd import build
    lib = build.build_library()
    out = os.path.join(OUT, "libdsm_ref_map_on_product.so")
    src = os.path.join(ROOT, "oracle", "ref_map_driver.cpp")
    deps = [src, lib, os.path.join(ROOT, "include", "dsm_fusion_functions.hpp"), os.path.join(ROOT, "include", "engine_compat", "fusion_functions.h"),
            os.path.join(REF_SRC, "surfel_map.cpp"), os.path.join(ROOT, "oracle", "shims", "Eigen", "Eigen")]
    if _stale(out, deps):
        os.makedirs(OUT, exist_ok=True)
        _run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++11", "-O3", "-pthread", "-fPIC", "-shared", "-ffp-contract=off",
              "-ftrivial-auto-var-init=zero", "-fno-access-control", "-w", "-DDSM_ORACLE_QUIET", "-DDSM_ORACLE_DEFERRED_THREADS",
              "-I" + os.path.join(ROOT, "include", "engine_compat"), "-I" + os.path.join(ROOT, "oracle", "shims"), "-I" + REF_SRC,
              "-o", out, src] + _LINK)
    return out
