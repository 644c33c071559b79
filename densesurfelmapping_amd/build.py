"""Build the HIP shared library in-tree (gfx950 only).

    python -m densesurfelmapping_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off is mandatory: results must match the
reference's non-fused IEEE arithmetic bit for bit (surfel_fusion/CMakeLists.txt:7-8 builds the
reference without -march / -ffast-math).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libdsm_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["dsm_kernels.hip", "dsm_api.hip", "dsm_surfel_map.cpp"]
HEADERS = ["dsm_math.h", "dsm_device.h", "dsm_k_common.h", "dsm_k_superpixel.h", "dsm_k_planes.h", "dsm_k_map.h", os.path.join("..", "..", "include", "dsm.h"),
           os.path.join("..", "..", "include", "dsm_surfel_map.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, defines=(), out: str = LIB_PATH) -> str:
    """The product library; `defines` / `out` build an instrumented copy beside it (tools/wave_stamps.py: DSM_WAVE_STAMPS=1)."""
    if not force and not defines and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value"] + ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


MERGE_LIB_PATH = os.path.join(HERE, "libdsm_merge_rccl.so")


def build_merge_library(force: bool = False, verbose: bool = False) -> str:
    """include/dsm_merge.h: the RCCL merge of the final clouds for C++ hosts -- a library of its own (links librccl), so that
    libdsm_hip.so carries no RCCL dependency."""
    src = os.path.join(CSRC, "dsm_merge_rccl.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "dsm_merge.h"), os.path.join(HERE, "..", "include", "dsm.h")]
    if not force and os.path.exists(MERGE_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(MERGE_LIB_PATH) for d in deps):
        return MERGE_LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-x", "hip", src, "-lrccl", "-o", MERGE_LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(MERGE_LIB_PATH + ".tmp", MERGE_LIB_PATH)
    return MERGE_LIB_PATH


def build_merge_test(out: str) -> str:
    """tests/cpp/merge_rccl_test.cpp against the merge library (run by the -m gpu suite)"""
    root = os.path.dirname(HERE)
    build_merge_library()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", os.path.join(root, "tests", "cpp", "merge_rccl_test.cpp"),
                    "-L" + HERE, "-ldsm_merge_rccl", "-lrccl", "-Wl,-rpath," + HERE, "-o", out], check=True)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_merge_library(force="--force" in sys.argv, verbose=True))
