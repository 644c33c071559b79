"""pytest configuration: `gpu` marker, CPU-side checker builds, shared helpers."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # no test may hang a run: a stuck stream wait is a failure after 15 minutes (the longest test takes one), not a wait for
    # the driver's own limit.  pytest-timeout's thread method: a signal would not interrupt a call that is stuck inside the
    # HIP runtime.  (Only if the plugin is installed and nobody chose a timeout on the command line.)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900
        if not getattr(config.option, "timeout_method", None):
            config.option.timeout_method = "thread"


def _run(cmd, **kw):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, **kw)


@pytest.fixture(scope="session")
def oracle_built():
    """C restatement always; the reference TU only where /root/reference exists (prebuilt otherwise)."""
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-s", "-C", odir, "all"])
    if os.path.isdir("/root/reference/surfel_fusion/src"):
        _run(["make", "-s", "-C", odir, "ref"])
    return odir


@pytest.fixture(scope="session")
def hostemu_lib():
    out = os.path.join(ROOT, "tests", "_build", "libhostemu.so")
    src = os.path.join(ROOT, "tests", "hostemu.cpp")
    deps = [src, os.path.join(ROOT, "densesurfelmapping_amd", "csrc", "dsm_math.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out])
    return out


def fields_equal(a, b):
    """Bit-exact comparison of two structured arrays, NaN == NaN (sign/payload of a NaN is not
    defined by the reference's arithmetic).  Returns a list of (field, n_mismatch)."""
    assert a.dtype == b.dtype
    bad = []
    if len(a) != len(b):
        return [("len", abs(len(a) - len(b)))]
    for f in a.dtype.names:
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            same = (x.view("u4") == y.view("u4")) | (np.isnan(x) & np.isnan(y))
        else:
            same = x == y
        n = int((~same).sum())
        if n:
            bad.append((f, n))
    return bad


def fields_close(a, b, rtol=1e-4):
    """north_star tolerance: float attributes within 1e-4 relative, NaN masks identical,
    integer fields exact."""
    assert a.dtype == b.dtype and len(a) == len(b)
    for f in a.dtype.names:
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            assert np.array_equal(np.isnan(x), np.isnan(y)), f
            m = ~np.isnan(x)
            assert np.allclose(x[m], y[m], rtol=rtol, atol=1e-6), f
        else:
            assert np.array_equal(x, y), f


@pytest.fixture(scope="session")
def node_hostemu_lib(oracle_built):
    """The product's node-level host logic (csrc/dsm_surfel_map.cpp) over a CPU stand-in for the engine
    (tests/node_hostemu.cpp + the C restatement oracle): lets the pose-graph code run without a GPU."""
    out = os.path.join(ROOT, "tests", "_build", "libnode_hostemu.so")
    deps = [os.path.join(ROOT, "tests", "node_hostemu.cpp"),
            os.path.join(ROOT, "densesurfelmapping_amd", "csrc", "dsm_surfel_map.cpp"),
            os.path.join(ROOT, "include", "dsm_surfel_map.h"), os.path.join(ROOT, "include", "dsm.h"),
            os.path.join(oracle_built, "liboracle_port.so")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-Wno-subobject-linkage",
              "-I" + os.path.join(ROOT, "include"), deps[0], "-o", out, "-L" + oracle_built, "-l:liboracle_port.so",
              "-Wl,-rpath," + oracle_built])
    return out


from ref_on_product import REF_SRC, build_ref_map_on_product, build_ros_node_on_product  # noqa: E402


def _prebuilt_or_skip(out):
    if not os.path.exists(out):
        pytest.skip("reference sources not present and no prebuilt " + os.path.basename(out))
    return out


@pytest.fixture(scope="session")
def ros_node_on_product():
    """The reference's own surfel_fusion/src/ros_node.cpp, compiled IN PLACE and UNCHANGED, with include/ros_compat in
    front of it: `SurfelMap surfel_map(nh)` and its subscriber bindings are the product's node class.  ROS itself is
    shimmed (oracle/shims + tests/ros_shims: a message pump fed from a recorded log).  Built where the reference
    exists (tests/ref_on_product.py); the binary travels to the GPU box."""
    out = os.path.join(ROOT, "tests", "_build", "ros_node_on_product")
    return build_ros_node_on_product() if os.path.isdir(REF_SRC) else _prebuilt_or_skip(out)


@pytest.fixture(scope="session")
def ref_map_on_product():
    """INTEGRATION.md §2 as a build: the reference's surfel_map.{h,cpp} compiled in place and unchanged with
    include/engine_compat in front (its `#include <fusion_functions.h>` resolves to the product's facade), linked against
    libdsm_hip.so -- the reference's node class running on the HIP engine through the drop-in call."""
    out = os.path.join(ROOT, "tests", "_build", "libdsm_ref_map_on_product.so")
    return build_ref_map_on_product() if os.path.isdir(REF_SRC) else _prebuilt_or_skip(out)
