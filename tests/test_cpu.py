"""CPU-side tests (no GPU): the oracle against the reference TU and the golden vectors, the GPU
formulation replayed serially on the host (tests/hostemu.cpp), compaction and fixed-point
brute-force checks, the C ABI surface, and the multi-rank merge on gloo."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, fields_equal


@pytest.fixture(scope="module")
def ob(oracle_built):
    from oracle import bindings
    return bindings


@pytest.fixture(scope="module")
def synth():
    from densesurfelmapping_amd import synth
    return synth


# ------------------------------------------------------------------ oracle pinning
def test_port_oracle_matches_golden(ob, synth):
    """oracle/dsm_oracle.c vs vectors produced by the reference's own fusion_functions.cpp."""
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    for case in meta["cases"]:
        cam = getattr(synth, case["camera"])
        scene = synth.Scene(**case["scene"])
        orc = ob.PortOracle(cam)
        local = np.zeros(0, ob.SURFEL_DTYPE)
        for (t, img, dep, pose, ref), want in zip(synth.sequence(cam, scene, case["frames"]), case["per_frame"]):
            local, k = orc.fuse_map(ref, img, dep, pose, local)
            assert k == want["n_new"] and len(local) == want["n_local"], (case["name"], t)
            assert hashlib.sha256(orc.labels().tobytes()).hexdigest() == want["labels_sha256"], (case["name"], t)
            assert int(orc.seeds()["stable"].sum()) == want["n_stable"]
        ref_map = np.load(os.path.join(ROOT, "tests", "golden", case["final_map"]))
        assert not fields_equal(local, ref_map), case["name"]


def test_port_oracle_matches_reference_tu(ob, synth):
    """Live comparison with oracle/_ref (only where it was built: needs /root/reference or the prebuilt .so)."""
    if not ob.have_ref("serial"):
        pytest.skip("oracle/_ref not built")
    # TINY_RAGGED: (size mod 8) > 4, the reference touches superpixel_seeds[-1] for the border pixels (undefined
    # behaviour, harmless in this build: heap bytes in front of the vector); the port pins that record to zero
    for cam, scene, n in ((synth.TINY, synth.Scene(seed=3), 40), (synth.KITTI_1241, synth.Scene(seed=4), 3),
                          (synth.TINY_RAGGED, synth.Scene(seed=3), 12)):
        ref, port = ob.RefOracle(cam), ob.PortOracle(cam)
        lr = np.zeros(0, ob.SURFEL_DTYPE)
        lp = lr.copy()
        for t, img, dep, pose, ridx in synth.sequence(cam, scene, n):
            lr, kr = ref.fuse_map(ridx, img, dep, pose, lr)
            lp, kp = port.fuse_map(ridx, img, dep, pose, lp)
            assert kr == kp
            assert np.array_equal(ref.labels(), port.labels())
            assert not fields_equal(ref.seeds(), port.seeds())
            assert not fields_equal(lr, lp)


def test_reference_quirk_states(ob, synth):
    """State-level check of the two scheduling quirks on the reference TU vs the port:
    random stable flags + random labels, and an unstable seed that owns no pixel mid-chunk."""
    if not ob.have_ref("serial"):
        pytest.skip("oracle/_ref not built")
    cam = synth.TINY
    img, dep, _ = synth.render(cam, synth.Scene(seed=8), 0)
    rng = np.random.default_rng(0)
    for trial in range(6):
        ref, port = ob.RefOracle(cam), ob.PortOracle(cam)
        for o in (ref, port):
            o.set_frame(img, dep)
            o.stage("initialize_seeds")
            o.stage("update_pixels")
            o.stage("update_seeds")
        seeds = ref.seeds()
        labels = ref.labels()
        seeds["stable"] = rng.random(len(seeds)) < 0.5
        victim = int(rng.integers(0, len(seeds)))
        seeds["stable"][victim] = 0
        gw = cam.width // 8
        neighbour = victim + 1 if (victim % gw) + 1 < gw else victim - 1
        labels[labels == victim] = neighbour  # victim now owns nothing
        for o in (ref, port):
            o.set_seeds(seeds)
            o.set_labels(labels)
            o.stage("update_seeds")
        assert not fields_equal(ref.seeds(), port.seeds())
        for o in (ref, port):
            o.stage("update_pixels")
        assert np.array_equal(ref.labels(), port.labels())
        assert not fields_equal(ref.seeds(), port.seeds())


# ------------------------------------------------------------------ GPU formulation on the host
class Emu:
    def __init__(self, path, cam):
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.emu_create.restype = vp
        lib.emu_create.argtypes = [C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_int]
        lib.emu_destroy.argtypes = [vp]
        lib.emu_fuse_map.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, C.c_int, vp]
        lib.emu_get_labels.argtypes = [vp, vp]
        lib.emu_get_seeds.argtypes = [vp, vp]
        lib.emu_set_order_salt.argtypes = [vp, C.c_int]
        lib.emu_compact.argtypes = [vp, C.c_int, vp, C.c_int]
        self.lib, self.cam = lib, cam
        self.h = lib.emu_create(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.far, cam.near, int(cam.rgbd))
        self.S = (cam.width // 8) * (cam.height // 8)

    def fuse_map(self, dtype, ref, img, dep, pose, local):
        cap = len(local) + self.S
        buf = np.zeros(cap, dtype)
        buf[: len(local)] = local
        n, k = C.c_int(len(local)), C.c_int(0)
        pcm = np.ascontiguousarray(np.asarray(pose, np.float32).T).ravel()
        rc = self.lib.emu_fuse_map(self.h, ref, img.ctypes.data, img.strides[0], dep.ctypes.data, dep.strides[0],
                                   pcm.ctypes.data, buf.ctypes.data, C.byref(n), cap, C.byref(k))
        assert rc == 0
        return buf[: n.value].copy(), k.value

    def labels(self):
        out = np.zeros((self.cam.height, self.cam.width), np.int32)
        self.lib.emu_get_labels(self.h, out.ctypes.data)
        return out

    def seeds(self, dtype):
        out = np.zeros(self.S, dtype)
        self.lib.emu_get_seeds(self.h, out.ctypes.data)
        return out


# the reference's real feed (publisher.py:37-40): disparity-quantised depth, +inf at zero disparity, saturated / 8-level image
STEREO_SCENE = dict(stereo=True, saturate_above=150.0, intensity_levels=8)
TUM_SCENE = dict(seed=7, tum=True, frames_per_period=24, intensity_noise=8.0, checker=25.0, n_boxes=6)


@pytest.mark.parametrize("camera,frames,salt,stereo", [("TINY", 48, 0, None), ("TINY", 48, 977, None), ("KITTI_1226", 3, 12345, None),
                                                        ("VGA_RGBD", 2, 0, None), ("TINY_RAGGED", 30, 0, None), ("TINY_RAGGED", 30, 555, None),
                                                        ("TINY", 30, 0, "inf"), ("TINY", 30, 311, "zero"), ("KITTI_1226", 3, 0, "inf"),
                                                        ("NODE_CAM_RGBD", 24, 0, "tum"), ("NODE_CAM_RGBD", 24, 4242, "tum"), ("VGA_RGBD", 3, 0, "tum")])
def test_gpu_formulation_on_host(ob, synth, hostemu_lib, camera, frames, salt, stereo):
    """dsm_math.h + the tmin/worklist fixed point, staged seed commit, 20-lane Gauss-Newton and the
    parallel-exact compaction, executed serially (and in scrambled order when salt != 0), bit-equal to
    the oracle.  `stereo`: the same on the reference's kind of input -- depth = bf / quantised disparity with +inf (or 0)
    where the disparity is 0, few grey levels -- where seeds end with an infinite or NaN mean depth and the filtered
    pick meets ten times as many near-ties."""
    cam = getattr(synth, camera)
    scene = synth.Scene(seed=5, scale=0.12, step=0.05) if cam.rgbd else synth.Scene()
    if stereo == "tum":  # BASELINE configs[3]'s kind of input: Kinect-quantised uint16 / 5000 depth, zero in shadows and blobs
        scene = synth.Scene(**TUM_SCENE)
    elif stereo:
        scene = synth.Scene(zero_disparity_inf=stereo == "inf", **STEREO_SCENE)
    emu, orc = Emu(hostemu_lib, cam), ob.PortOracle(cam)
    emu.lib.emu_set_order_salt(emu.h, salt)
    le = np.zeros(0, ob.SURFEL_DTYPE)
    lo = le.copy()
    for t, img, dep, pose, ref in synth.sequence(cam, scene, frames, keyframe_every=4 if stereo == "tum" else 5):
        le, ke = emu.fuse_map(ob.SURFEL_DTYPE, ref, img, dep, pose, le)
        lo, ko = orc.fuse_map(ref, img, dep, pose, lo)
        assert ke == ko, t
        assert np.array_equal(emu.labels(), orc.labels()), t
        assert not fields_equal(emu.seeds(ob.SEED_DTYPE), orc.seeds()), t
        assert not fields_equal(le, lo), t
    # the filtered pick of k_assign (dsm_math.h, pick_seed_fast), run beside the reference's on every pixel of every sweep:
    # it never answers differently, its error bound holds on every candidate cost, and it answers nearly always
    st = (C.c_longlong * 24)()
    emu.lib.emu_fast_pick_stats.argtypes = [C.c_void_p, C.c_void_p]
    emu.lib.emu_fast_pick_stats(emu.h, st)
    pixels, unsure, mismatches, checked, violations = list(st)[:5]
    assert pixels >= frames * 3 * cam.width * cam.height * 0.9 and checked > pixels
    assert mismatches == 0 and violations == 0, (mismatches, violations)
    assert unsure < (0.06 if stereo else 0.02) * pixels, (unsure, pixels)
    # the exact-sum claim behind k_seed_fit's tree-ordered Jacobian sums (tools/_exp/r04_fit_tree.patch, row16_sum): fp32-product terms
    # spanning <= 21 binades sum exactly in double in ANY order -- on every qualifying sum of every fitted seed a 16-way
    # tree gives the ordered sum's bits, and nearly every all-core step qualifies
    ex = (C.c_longlong * 6)()
    emu.lib.emu_exact_sum_stats.argtypes = [C.c_void_p, C.c_void_p]
    emu.lib.emu_exact_sum_stats(emu.h, ex)
    assert ex[3] == 0, f"{ex[3]} qualifying sums whose tree-order value differs from the ordered one"
    assert stereo or ex[2] == 0 or ex[1] >= 0.95 * ex[2], (ex[1], ex[2])  # (a statistic of the smooth scenes; NaN / inf planes never qualify)
    # k_seed_fit's free-order steps (round 6; dsm_math.h, gn_sum_is_exact): steps 2..5 of a seed whose Huber classes stand take
    # their Jacobian sums in four interleaved parts wherever a cheap test proves that no addition of any order rounds --
    # never a different sum, and nearly always a pass
    ct = (C.c_longlong * 3)()
    emu.lib.emu_cert_stats.argtypes = [C.c_void_p, C.c_void_p]
    emu.lib.emu_cert_stats(emu.h, ct)
    assert ct[2] == 0, f"{ct[2]} steps whose sums pass the exactness test and still differ from the ordered sums"
    assert ct[0] == 0 or ct[1] >= 0.99 * ct[0], (ct[1], ct[0])
    print(f"free-order steps: {ct[1]} of {ct[0]} qualifying steps 2..5 pass the exactness test for all four sums; every one equals the ordered sums")
    print(f"exact sums: steps 2..5 whose Jacobian sums all span <= 21 binades: {ex[1]} of {ex[2]}; step-1 (H and J): {ex[0]} of {st[8]} seeds; tree == ordered on all")
    print(f"plane fit: {st[7]} of {st[8]} seeds with a residual outside the Huber core at step 1, {st[9]} with a class change later")
    print(f"pick_seed_fast: {unsure} of {pixels} pixels unsure ({100.0 * unsure / pixels:.3f} %), {checked} costs within their bound")


def test_parallel_compaction_bruteforce(ob, hostemu_lib):
    """k_hole_scan + k_compact's closed form vs the serial loop of surfel_map.cpp:1077-1109 (oracle),
    including tail holes interleaved with live elements and K <, =, > k."""
    emu = C.CDLL(hostemu_lib)
    port = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_port.so"))
    emu.emu_compact.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    port.dsmo_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(42)
    for trial in range(3000):
        m = int(rng.integers(0, 40))
        k_new = int(rng.integers(0, 12))
        a = np.zeros(m + k_new + 1, ob.SURFEL_DTYPE)
        a["px"][:m] = np.arange(m) + 1
        a["update_times"][:m] = (rng.random(m) > rng.uniform(0, 1)).astype(np.int32)
        if m and trial % 3 == 0:
            a["update_times"][max(0, m - int(rng.integers(1, 6))): m] = 0  # force tail holes
        fresh = np.zeros(max(k_new, 1), ob.SURFEL_DTYPE)
        fresh["px"][:k_new] = 1000 + np.arange(k_new)
        fresh["update_times"][:k_new] = 1
        b = a.copy()
        n_a = C.c_int(m)
        assert port.dsmo_compact(a.ctypes.data, C.byref(n_a), len(a), fresh.ctypes.data, k_new) == 0
        n_b = emu.emu_compact(b.ctypes.data, m, fresh.ctypes.data, k_new)
        assert n_a.value == n_b, trial
        assert np.array_equal(a["px"][:n_b], b["px"][:n_b]), trial
        assert (b["update_times"][:n_b] != 0).all()


def test_tabled_inverse_is_the_closed_form(hostemu_lib):
    """The index/sign table the kernel evaluates with one lane per cofactor == inverse4, bit for bit."""
    emu = C.CDLL(hostemu_lib)
    emu.emu_inverse4d.argtypes = [C.c_void_p] * 3
    rng = np.random.default_rng(9)
    for trial in range(2000):
        a = rng.normal(size=16) * 10.0 ** rng.integers(-3, 4)
        if trial % 4 == 0:  # symmetric positive definite, like the damped Gauss-Newton Hessian
            m = rng.normal(size=(4, 6))
            a = (m @ m.T + 5 * np.eye(4)).ravel()
        a = np.ascontiguousarray(a, np.float64)
        x, y = np.zeros(16), np.zeros(16)
        emu.emu_inverse4d(a.ctypes.data, x.ctypes.data, y.ctypes.data)
        assert x.tobytes() == y.tobytes(), trial


def test_div_by_100_is_the_ieee_quotient(hostemu_lib):
    """The 3-operation x/100.0 of dsm_math.h == the IEEE quotient on random floats of every exponent."""
    emu = C.CDLL(hostemu_lib)
    emu.emu_div100_mismatches.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(4)
    bits = rng.integers(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = np.abs(x[np.isfinite(x)])
    x = np.concatenate([x, (rng.uniform(-255, 255, 2_000_000).astype(np.float32)) ** 2])
    x = np.ascontiguousarray(x, np.float32)
    assert emu.emu_div100_mismatches(x.ctypes.data, len(x)) == 0


def test_double_typed_compares_in_fp32(hostemu_lib):
    """dsm_math.h compares floats against the reference's double thresholds in fp32 (flt_below / flt_above) and takes
    the Newton step of the robust mean as an fp32 divide: same outcome as the double expressions for every float
    around every threshold the path uses, and on random sums of every exponent."""
    emu = C.CDLL(hostemu_lib)
    emu.emu_threshold_mismatches.argtypes = [C.c_double, C.c_int]
    for c in (0.01, 0.05, 0.1, 0.2, 0.4, 0.8, 0.5, 0.25, 1.0, 0.3, 1e-3, 7.0):
        assert emu.emu_threshold_mismatches(c, 5000) == 0, c
    emu.emu_newton_step_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(9)
    bits = rng.integers(0, 2 ** 32, 3_000_000, dtype=np.uint64).astype(np.uint32)
    a = np.concatenate([bits.view(np.float32), rng.normal(0, 40, 3_000_000).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3e38], np.float32)])
    a = np.ascontiguousarray(a, np.float32)
    n_core = np.ascontiguousarray(rng.integers(0, 257, len(a)), np.int32)
    assert emu.emu_newton_step_mismatches(a.ctypes.data, n_core.ctypes.data, len(a)) == 0


def test_fuse_expressions_in_fp32(hostemu_lib):
    """k_fuse_surfels evaluates the depth tolerance and the renormalisation of the fused normal in fp32 where the
    reference's double expressions allow it exactly (dsm_math.h): equal bits on depths across and beyond the working
    range, for both constant sets and several focal lengths; a focal length for which BASELINE * f is no float falls
    back to the double form (and is equal trivially)."""
    emu = C.CDLL(hostemu_lib)
    emu.emu_fuse_fp32_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    rng = np.random.default_rng(21)
    n = 2_000_000
    z = np.concatenate([rng.uniform(0.5, 30.0, n), rng.uniform(1e-3, 200.0, n // 4), 10.0 ** rng.uniform(-3, 3, n // 4)]).astype(np.float32)
    bits = rng.integers(0, 2 ** 32, len(z), dtype=np.uint64).astype(np.uint32)
    a = np.ascontiguousarray(bits.view(np.float32))
    b = np.ascontiguousarray(np.concatenate([rng.uniform(1e-3, 4.0, len(z) // 2), 10.0 ** rng.uniform(-30, 30, len(z) - len(z) // 2)]).astype(np.float32))
    z = np.ascontiguousarray(z)
    for rgbd, focal, expect_fp32 in ((0, 707.0912, 1), (0, 718.856, 1), (0, 1400.0, 1), (1, 525.0, 1), (1, 517.3, 0), (0, 0.3337, 1)):
        used = C.c_int(-1)
        assert emu.emu_fuse_fp32_mismatches(z.ctypes.data, a.ctypes.data, b.ctypes.data, len(z), rgbd, focal, C.byref(used)) == 0, (rgbd, focal)
        assert used.value == expect_fp32, (rgbd, focal, used.value)


def test_stable_skip_fixed_point_bruteforce():
    """The tmin fixed point of k_assign/k_resolve/k_apply vs the reference's sequential scan
    (FF.cpp:400,445,450) on random (old label, pick, stable) instances."""
    rng = np.random.default_rng(1)
    INF = 2 ** 31 - 1
    for trial in range(400):
        n_seed = int(rng.integers(2, 12))
        n_pix = int(rng.integers(1, 200))
        old = rng.integers(0, n_seed, n_pix)
        pick = rng.integers(0, n_seed, n_pix)
        stable0 = rng.random(n_seed) < rng.uniform(0.2, 0.95)
        # reference: sequential
        st = stable0.copy()
        lab_ref = old.copy()
        for p in range(n_pix):
            if st[lab_ref[p]]:
                continue
            lab_ref[p] = pick[p]
            st[pick[p]] = False
        # GPU formulation
        tmin = np.where(stable0, INF, -1).astype(np.int64)
        work = []
        for p in rng.permutation(n_pix):
            l, c = old[p], pick[p]
            if tmin[l] == -1:
                tmin[c] = min(tmin[c], p)
            elif c != l and tmin[c] != -1:
                work.append(p)
        changed = True
        while changed:
            changed = False
            for p in work:
                if tmin[old[p]] < p and tmin[pick[p]] > p:
                    tmin[pick[p]] = p
                    changed = True
        lab = np.where(tmin[old] < np.arange(n_pix), pick, old)
        assert np.array_equal(lab, lab_ref), trial
        assert np.array_equal(tmin == INF, st), trial


# ------------------------------------------------------------------ ABI surface
def test_c_abi_exports_every_declared_symbol():
    """include/dsm.h <-> libdsm_hip.so <-> api.ABI_SYMBOLS agree (load only, no compute without a GPU)."""
    from densesurfelmapping_amd import api, build
    build.build_library()
    from densesurfelmapping_amd import surfel_map
    lib = C.CDLL(api.LIB_PATH)
    for hdr, symbols in (("dsm.h", api.ABI_SYMBOLS), ("dsm_surfel_map.h", surfel_map.ABI_SYMBOLS)):
        header = open(os.path.join(ROOT, "include", hdr)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # prose in comments mentions functions of the other header
        declared = set(re.findall(r"\b(dsm_[a-z_0-9]+)\s*\(", header))
        assert declared == set(symbols), (hdr, declared ^ set(symbols))
        for name in declared:
            assert hasattr(lib, name), name
    lib.dsm_abi_version.restype = C.c_int
    assert lib.dsm_abi_version() == 4
    # include/dsm_merge.h: the RCCL merge for C++ hosts, a library of its own
    mlib = C.CDLL(build.build_merge_library())
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dsm_merge.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(dsm_[a-z_0-9]+)\s*\(", header))
    assert declared == {"dsm_merge_clouds_rccl", "dsm_merge_last_error"} and all(hasattr(mlib, n) for n in declared)
    mlib.dsm_merge_clouds_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    assert mlib.dsm_merge_clouds_rccl(None, 1, 0, None, 0, None, 0, None, None) == -1  # (argument checks come before any device call)
    assert C.sizeof(api._Config) == 88  # 8 x 4 B + 4 doubles + 5 x 4 B, padded to 8


def test_public_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: include/dsm.h, dsm_surfel_map.h and dsm_merge.h compile as C99 (no C++ in the signatures),
    with warnings as errors, in a translation unit that calls across all three."""
    src = tmp_path / "c_abi_check.c"
    src.write_text('''#include "dsm.h"
#include "dsm_surfel_map.h"
#include "dsm_merge.h"
int use(dsm_handle *h, const uint8_t *const *images, const size_t *is, const float *const *depths, const size_t *ds, uint8_t *di, float *dd) {
    int32_t pitch = 0;
    if (dsm_frame_pitch(h, &pitch)) return 1;
    return dsm_host_pack_frames(1, 8, 8, images, is, depths, ds, di, (size_t)pitch, (size_t)pitch * 8, dd, (size_t)pitch * 4, (size_t)pitch * 32)
         + dsm_merge_clouds_rccl(0, 1, 0, 0, 0, 0, 0, 0, 0);
}
''')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_host_pack_frames():
    """dsm_host_pack_frames (host code only: no GPU needed): the caller's frames -- contiguous and strided rows -- land in the
    slot layout row for row, pad bytes untouched; bad steps and null planes are refused before anything is copied."""
    from densesurfelmapping_amd import api, build
    build.build_library()
    lib = api.load_library()
    rng = np.random.default_rng(1)
    n, w, h = 9, 166, 103
    pitch = (w + 63) // 64 * 64
    wide_i, wide_d = rng.integers(0, 256, (h, w + 9), dtype=np.uint8), rng.random((h, w + 5), dtype=np.float32)
    ims = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(n - 1)] + [wide_i[:, 3:3 + w]]
    dps = [rng.random((h, w), dtype=np.float32) for _ in range(n - 1)] + [wide_d[:, 2:2 + w]]
    dst_i, dst_d = np.full((n, h, pitch), 7, np.uint8), np.full((n, h, pitch), -1, np.float32)

    def call(n_, ims_, dps_, img_step=pitch):
        ip, dp = (C.c_void_p * len(ims_))(*[a.ctypes.data if a is not None else None for a in ims_]), (C.c_void_p * len(dps_))(*[a.ctypes.data for a in dps_])
        ist = (C.c_size_t * len(ims_))(*[a.strides[0] if a is not None else 0 for a in ims_])
        dst = (C.c_size_t * len(dps_))(*[a.strides[0] for a in dps_])
        return lib.dsm_host_pack_frames(n_, w, h, ip, ist, dp, dst, C.c_void_p(dst_i.ctypes.data), img_step, pitch * h,
                                        C.c_void_p(dst_d.ctypes.data), pitch * 4, pitch * h * 4)
    assert call(n, ims, dps) == 0
    for i in range(n):
        assert np.array_equal(dst_i[i, :, :w], ims[i]) and np.array_equal(dst_d[i, :, :w], dps[i]), i
    assert (dst_i[:, :, w:] == 7).all() and (dst_d[:, :, w:] == -1).all(), "pad bytes were written"
    assert call(0, ims, dps) == 0
    assert call(n, ims, dps, img_step=w - 1) == -1 and "row step" in lib.dsm_last_error(None).decode()
    assert call(n, ims[:-1] + [None], dps) == -1 and "frame 8" in lib.dsm_last_error(None).decode()
    # the Python mirror: PinnedFrames.set_many needs page-locked memory (a GPU runtime); its argument marshalling is the call above


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from densesurfelmapping_amd import api, synth
    with pytest.raises(api.DsmError) as ei:
        api.FusionFunctions.from_camera(synth.TINY)
    assert ei.value.code == -2  # DSM_E_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "densesurfelmapping_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("the oracle", "").replace("oracle's", "").replace("oracle uses", ""), \
                    os.path.join(dirpath, f)


# ------------------------------------------------------------------ multi-rank merge (gloo, 2 ranks)
def test_bench_reads_the_committed_profiles():
    """bench.py's `traffic` and `valu_issue` come from rocprofv3 --pmc passes committed under profiles/ (counters
    cannot be read from inside the run): the files it names exist, parse, and hold the kernels it looks up; and the
    kernel-trace summaries DESIGN.md cites are there."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    b, src = bench.pmc_traffic("k_update_seeds", bench.PMC_TRAFFIC_SINGLE)
    assert src == "r06_pmc_traffic.json" and 4_000_000 < b < 40_000_000
    # a batched stage may be several kernels (the lane-per-seed forms): their traffic is summed; the record is of the
    # launches the timed region makes (32 subsequences per launch) and is not used for any other batch size
    pair = bench.BATCHED_KERNELS_OF_STAGE["update_seeds_1"]
    b32, src32 = bench.pmc_traffic(pair, bench.PMC_TRAFFIC_BATCHED, 32)
    assert src32 == "r06_pmc_traffic_batched.json" and 32 * 4_000_000 < b32 < 32 * 40_000_000
    assert bench.pmc_traffic(pair, bench.PMC_TRAFFIC_BATCHED, 8) == (None, None)
    for stage, kernels in bench.BATCHED_KERNELS_OF_STAGE.items():
        assert bench.pmc_traffic(kernels, bench.PMC_TRAFFIC_BATCHED, 32)[0], (stage, kernels)
    v = bench.valu_issue(28000.0)
    assert v and 8e6 < v["valu_wave_insts_per_frame"] <= 10.8e6 and 0.3 < v["frac"] < 1.2, "VERDICT r05 next #4: <= 10.8 M wave-instructions per frame"
    assert "batched over 32" in v["source"]  # (the launches the timed region makes, not a launch-of-8 pass scaled)
    assert bench.pmc_8m("k_fuse_surfels", 200.0)["traffic"]["source"].startswith("profiles/r06_") and bench.pmc_8m("k_warp", 120.0)["traffic"]["read_bytes"] > 3e8
    for name in ("r06_kernel_trace_batch8x1.md", "r06_kernel_trace_batch32x1.md", "r06_kernel_trace_batch32x4_default.md", "r06_kernel_trace_streams1.md",
                 "r06_pmc_sq_batch8.md", "r06_pmc_sq_batch32.md", "r06_bench_default.json", "r06_bench_variance.md", "r06_streaming.md", "r06_wave_vs_lane.md",
                 "r06_pmc_map_kernels_8m.md", "r06_open_picks.json", "r05_ceilings_trip3.md", "r03_eigen_exposure.md", "r02_relaxed_sums.md", "r02_issue_rates.md"):
        assert os.path.getsize(os.path.join(ROOT, "profiles", name)) > 200, name
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_default.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "verified_timed_region"):
        assert key in rec, key
    assert set(rec["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "whole_frame_kernel_weighted", "binding_roof"}
    assert set(rec["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    # the record's run checked the maps the timed region itself left behind (hundreds of thousands of surfels each) against the oracle
    assert rec["verified"] is True and rec["verified_timed_region"] is True
    assert all(row["equal"] and row["surfels"] > 262144 for row in rec["verification"]["timed_region"]["checked"])
    for leg in ("kitti_like", "tum_like", "sharded_replay", "fullhd_2M", "streamed_input", "single_sequence", "dropin_pcie_inclusive", "multi_gpu"):
        assert leg in rec, leg
    assert rec["multi_gpu"]["backend"] == "nccl" and rec["multi_gpu"]["final_cloud_all_gather_ms"] > 0  # (RCCL ran: a group of one)
    assert 0.9 <= rec["tum_like"]["ratio_to_ideal_sensor"] <= 1.1 and rec["roofline"]["own_bytes"]["frac"] > 0


def test_settled_mean_depth_shortcut(hostemu_lib):
    """dsm_math.h, mean_depth_is_settled: a superpixel with a +inf member depth (the reference's feed has them wherever the
    disparity is 0) keeps the +inf mean it starts from through every Huber-Newton pass, so the kernels skip the passes.
    huber_mean_depth WITH the shortcut against the reference's loop written out, on 20 000 lists: ordinary depths, one or
    many +inf members anywhere in the list, sums that overflow to +inf, single-element lists."""
    import ctypes as C
    lib = C.CDLL(hostemu_lib)
    lib.emu_settled_mean_mismatches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
    rng = np.random.default_rng(7)
    n_lists, stride = 20000, 64
    lists = (rng.uniform(0.11, 80.0, size=(n_lists, stride))).astype(np.float32)
    lens = rng.integers(1, stride + 1, size=n_lists).astype(np.int32)
    kind = rng.integers(0, 4, size=n_lists)
    for i in np.nonzero(kind == 1)[0]:  # one +inf member
        lists[i, rng.integers(0, lens[i])] = np.inf
    for i in np.nonzero(kind == 2)[0]:  # many
        lists[i, :lens[i]][rng.random(lens[i]) < 0.4] = np.inf
    for i in np.nonzero(kind == 3)[0][:2000]:  # finite members whose sum overflows
        lists[i, :lens[i]] = np.float32(3.0e38)
    n_inf_lists = int(sum(np.isinf(lists[i, :lens[i]]).any() or kind[i] == 3 for i in range(n_lists)))
    assert n_inf_lists > 5000
    for huber in (0.4, 0.05):
        assert lib.emu_settled_mean_mismatches(lists.ctypes.data, lens.ctypes.data, n_lists, stride, huber) == 0


def test_shard_subsequences():
    from densesurfelmapping_amd.replay import shard_subsequences
    sh = shard_subsequences(4541, 8)
    assert sh[0] == (0, 568) and sh[-1][1] == 4541 and len(sh) == 8
    assert sorted(b - a for a, b in sh) == [567] * 3 + [568] * 5
    assert shard_subsequences(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]


_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from densesurfelmapping_amd.replay import merge_clouds, shard_subsequences
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = int(sys.argv[2])
a, b = shard_subsequences(N, world)[rank]
n = (b - a) * 3 + (0 if rank else 2)          # ragged, rank-dependent cloud sizes (0 for an empty shard behind non-empty ones)
cloud = (np.arange(n * 44, dtype=np.int64) * (rank + 1) % 251).astype(np.uint8)
merged, counts = merge_clouds(torch.from_numpy(cloud))
want = np.concatenate([(np.arange(c * 44, dtype=np.int64) * (r + 1) % 251).astype(np.uint8) for r, c in enumerate(counts)])
assert counts == [((q - p) * 3 + (0 if r else 2)) for r, (p, q) in enumerate(shard_subsequences(N, world))], counts
assert N >= world or 0 in counts
assert np.array_equal(merged.numpy(), want)
empty, c0 = merge_clouds(torch.zeros(0, dtype=torch.uint8))
assert empty.numel() == 0 and c0 == [0] * world
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    """a TCP port the OS says is free right now (fixed ports collide when two runs share a host)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun(world, script, *args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script), *[str(a) for a in args]],
                          env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world,frames", [(2, 11), (8, 5)])
def test_merge_clouds_gloo(tmp_path, world, frames):
    """merge_clouds over gloo at world size 2 and at the target world size 8 -- there with fewer frames than ranks, so that
    ranks with an empty cloud sit among ranks with surfels (and the all-empty case on every rank)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    r = _torchrun(world, script, ROOT, frames, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == world


_SHARD_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from densesurfelmapping_amd import replay, synth
from oracle.bindings import PortOracle, SURFEL_DTYPE
class OracleEngine:  # TEST stand-in for the HIP engine: the C restatement behind the engine interface of replay.py
    def __init__(self, cam):
        self.orc, self.local = PortOracle(cam), np.zeros(0, SURFEL_DTYPE)
    def fuse(self, image, depth, pose, ref_idx):
        self.local, _ = self.orc.fuse_map(ref_idx, image, depth, pose, self.local)
    def cloud(self):
        return self.local
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
src = replay.SyntheticSource(int(sys.argv[2]), camera="TINY", seed=77)
out = sys.argv[3]
summary = replay.run_rank(src, rank, world, engine_factory=OracleEngine, backend="gloo", save_shards=out, out=os.path.join(out, "merged.npy"))
# this rank's shard against the oracle driven directly over frames [a, b): keyframe indices restart at 0
a, b = replay.shard_subsequences(src.n_frames, world)[rank]
orc, lo = PortOracle(src.cam), np.zeros(0, SURFEL_DTYPE)
for t, img, dep, pose, ref in synth.sequence(src.cam, src.scene, b - a, start=a):
    assert ref == (t - a) // 5
    lo, _ = orc.fuse_map(ref, img, dep, pose, lo)
mine = np.load(os.path.join(out, f"shard_{rank}.npy"))
assert (len(lo) > 0) == (b > a) and mine.tobytes() == lo.tobytes(), (rank, len(mine), len(lo))
dist.barrier()
if rank == 0:
    merged = np.load(os.path.join(out, "merged.npy"))
    parts = [np.load(os.path.join(out, f"shard_{r}.npy")) for r in range(world)]
    assert merged.tobytes() == b"".join(p.tobytes() for p in parts)
    assert summary["counts"] == [len(p) for p in parts] and summary["merged_surfels"] == len(merged)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,frames", [(2, 23), (8, 6)])
def test_sharded_replay_gloo(oracle_built, tmp_path, world, frames):
    """BASELINE configs[2] end to end on the CPU side: densesurfelmapping_amd.replay's driver (frame source ->
    shard_subsequences -> one engine per rank -> merge_clouds) with gloo ranks and the C restatement standing in for
    the HIP engine (injected here, the product has no such engine): rank r's map is the oracle's map of frames
    [a_r, b_r) fused from an empty map with keyframe indices restarting at 0, and the merged cloud is the concatenation
    of the shards in rank order.  World size 2, and the node's 8 with fewer frames than ranks: two
    ranks replay nothing and contribute an empty cloud."""
    script = tmp_path / "worker.py"
    script.write_text(_SHARD_WORKER)
    out = tmp_path / "shards"
    r = _torchrun(world, script, ROOT, frames, out)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == world


def test_replay_cli_refuses_without_gpus():
    """`python -m densesurfelmapping_amd.replay --gpus N` on a box with fewer devices reports that and replays nothing;
    with one rank and no GPU the engine itself refuses (no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, "-m", "densesurfelmapping_amd.replay", "--synthetic", "4", "--camera", "TINY", "--gpus", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "visible GPUs" in r.stderr and "{" not in r.stdout


# ------------------------------------------------------------------ C++ facade
def _build_facade_test(oracle_built):
    from densesurfelmapping_amd import api, build
    build.build_library()
    out = os.path.join(ROOT, "tests", "_build", "facade_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    pkg = os.path.dirname(api.LIB_PATH)
    subprocess.run(["g++", "-std=c++11", "-O1", os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", out,
                    "-L" + pkg, "-ldsm_hip", "-L" + oracle_built, "-loracle_port",
                    "-Wl,-rpath," + pkg, "-Wl,-rpath," + oracle_built, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def test_cpp_facade_links_and_refuses_without_gpu(oracle_built):
    """include/dsm_fusion_functions.hpp compiles as C++11 with stand-in cv::Mat / Eigen types and links
    against the C ABI; without a GPU it must report 'no device' (exit 77), never compute."""
    import torch
    exe = _build_facade_test(oracle_built)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 77, r.stdout + r.stderr


def _build_node_replay_test():
    from densesurfelmapping_amd import api, build
    build.build_library()
    out = os.path.join(ROOT, "tests", "_build", "node_replay_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    pkg = os.path.dirname(api.LIB_PATH)
    subprocess.run(["g++", "-std=c++11", "-O1", os.path.join(ROOT, "tests", "cpp", "node_replay_test.cpp"), "-o", out,
                    "-L" + pkg, "-ldsm_hip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def _write_node_events(path, cam, case, synth):
    """The scenario's message stream as a message log (densesurfelmapping_amd/msglog.py)."""
    import node_state
    from densesurfelmapping_amd import msglog
    scene = node_state.camera_and_scene(case, synth)[1]
    msglog.write_log(path, cam, case["drift_free_poses"], synth.node_messages(cam, scene, case["frames"], **case["kw"]))


def test_message_log_round_trip(synth, tmp_path):
    """write_log -> read_log returns the same messages, bit for bit."""
    import node_state
    from densesurfelmapping_amd import msglog
    case = dict(node_state.SCENARIOS[1], frames=12)
    cam = synth.NODE_CAM
    path = str(tmp_path / "log.bin")
    _write_node_events(path, cam, case, synth)
    cam2, dfp, events = msglog.read_log(path)
    assert (cam2.width, cam2.height, dfp) == (cam.width, cam.height, case["drift_free_poses"])
    want = list(synth.node_messages(cam, synth.Scene(), case["frames"], **case["kw"]))
    got = list(events)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a[0] == b[0] and tuple(a[1]) == tuple(b[1])
        for x, y in zip(a[2:], b[2:]):
            assert np.asarray(x).tobytes() == np.ascontiguousarray(y, dtype=np.asarray(x).dtype).tobytes()


def test_cpp_surfel_map_wrapper_links_and_refuses_without_gpu(synth, tmp_path):
    """include/dsm_surfel_map.hpp compiles as C++11 against plain message structs and links against the C ABI;
    without a GPU the constructor must report 'no device' (exit 77), never compute."""
    import torch
    import node_state
    exe = _build_node_replay_test()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the replay itself is checked by the -m gpu suite")
    case = dict(node_state.SCENARIOS[0], frames=2)
    ev = str(tmp_path / "events.bin")
    _write_node_events(ev, synth.NODE_CAM, case, synth)
    r = subprocess.run([exe, ev, str(tmp_path / "a.PCD"), str(tmp_path / "a.PLY")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


# ----------------------------------------------------------------------------------------------------------
# node level (SurfelMap: message callbacks, pose graph, active / inactive sets, loop-closure warp, exports)

def _node_cases():
    import node_state
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "node_golden.json")))
    assert [c["name"] for c in gold["cases"]] == [c["name"] for c in node_state.SCENARIOS]
    return list(zip(node_state.SCENARIOS, gold["cases"]))


def _node_cases_large():
    import node_state
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "node_golden.json")))
    return list(zip(node_state.SCENARIOS_LARGE, gold["large_cases"]))


def _check_node_run(case, gold, make_node):
    """Run a scenario and compare with the golden record of the reference node, most telling check first."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_node_golden
    import node_state
    briefs, checkpoints, final, files = make_node_golden.run_case(case, make_node)
    for i, (a, b) in enumerate(zip(briefs, gold["briefs"])):
        assert a == b, f"{case['name']}: after pose message {i}: [fused, keyframes, local, inactive] = {a}, reference {b}"
    assert len(briefs) == len(gold["briefs"])
    if gold["final"]:  # (the large scenarios keep digests only)
        ref_final = np.load(os.path.join(ROOT, "tests", "golden", gold["final"]))
        for key in ("attached_counts", "begin", "is_local", "links"):
            assert np.array_equal(final[key], ref_final[key]), (case["name"], key)
        assert final["poses"].tobytes() == ref_final["poses"].tobytes(), "keyframe poses (fp64) differ"
        for key in ("local", "attached"):
            assert fields_equal(final[key], ref_final[key]) == [], (case["name"], key)
        assert np.array_equal(final["cloud"].view("u4"), ref_final["cloud"].view("u4")), "inactive_pointcloud"
    assert checkpoints == gold["checkpoints"]
    assert node_state.digest(final) == gold["final_digest"]
    for kind in ("pcd", "ply"):
        assert files[kind]["head"] == gold["files"][kind]["head"], kind
        assert files[kind]["bytes"] == gold["files"][kind]["bytes"], kind
        assert files[kind]["sha256"] == gold["files"][kind]["sha256"], kind


def test_node_golden_is_the_reference_node(oracle_built, ob, synth):
    """tests/golden/node_* are outputs of the reference's own surfel_map.cpp (compiled in place); regenerate
    and compare wherever the reference is present."""
    if not ob.have_ref("map"):
        pytest.skip("reference sources not present (GPU box): golden fixtures are used as committed")
    for case, gold in _node_cases():
        _check_node_run(case, gold, lambda cam, d: ob.RefSurfelMap(cam, drift_free_poses=d))


def test_node_host_logic_matches_reference_node(node_hostemu_lib, synth):
    """csrc/dsm_surfel_map.cpp (stamp matching, KITTI transform, pose graph, drift-free window, inactive-set
    bookkeeping, warp matrices, PCD / PLY writers) over a CPU stand-in engine == the reference node, bit for bit."""
    from densesurfelmapping_amd import surfel_map
    emu = C.CDLL(node_hostemu_lib)
    for case, gold in _node_cases():
        _check_node_run(case, gold, lambda cam, d: surfel_map.SurfelMap(cam, drift_free_poses=d, _library=emu))


def test_node_host_logic_at_kitti_resolution(node_hostemu_lib, synth):
    """The same at 1226x370: 130 frames, a 50-frame lap, loop closure over ~50 k inactive surfels, re-activation, a
    lagging loop path -- digests of the reference node's state every ten pose messages, at the end, and of its exports."""
    from densesurfelmapping_amd import surfel_map
    emu = C.CDLL(node_hostemu_lib)
    for case, gold in _node_cases_large():
        _check_node_run(case, gold, lambda cam, d: surfel_map.SurfelMap(cam, drift_free_poses=d, _library=emu))


def test_node_host_logic_fuzz_against_reference_node(node_hostemu_lib, ob, synth):
    """Random pose graphs: every frame's reference keyframe drawn among the latest three, random extra loop edges,
    random drift-free ranges -- the product's host logic (CPU stand-in engine) and the reference node side by side,
    compared after every pose message (counts) and at the end (whole state)."""
    if not ob.have_ref("map"):
        pytest.skip("reference sources not present (GPU box)")
    import node_state
    from densesurfelmapping_amd import surfel_map
    emu = C.CDLL(node_hostemu_lib)
    cam, scene = synth.NODE_CAM, synth.Scene()
    frames = {i: synth.render(cam, scene, i)[:2] for i in range(30)}
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        n = 90
        n_kf = n // 5
        extra = {}
        for _ in range(6):
            t = int(rng.integers(20, n))
            a, b = sorted(int(v) for v in rng.integers(0, max(t // 5, 1), size=2))
            if a != b:
                extra.setdefault(t, []).append((b, a))
        d = int(rng.integers(2, 6))
        kw = dict(lap=30, extra_loops=extra, frames=frames, path_lag=int(rng.integers(0, 3)))
        a_node = surfel_map.SurfelMap(cam, drift_free_poses=d, _library=emu)
        b_node = ob.RefSurfelMap(cam, drift_free_poses=d)
        ev_a = synth.node_messages(cam, scene, n, ref_rng=np.random.default_rng(seed), **kw)
        ev_b = synth.node_messages(cam, scene, n, ref_rng=np.random.default_rng(seed), **kw)
        for i, (ea, eb) in enumerate(zip(ev_a, ev_b)):
            a_node.feed(ea)
            b_node.feed(eb)
            if ea[0] == "orb":
                assert node_state.brief(a_node) == node_state.brief(b_node), (seed, i, n_kf)
        assert node_state.digest(node_state.snapshot(a_node)) == node_state.digest(node_state.snapshot(b_node)), seed
        a_node.close()
        b_node.close()


def test_node_refuses_what_the_reference_would_crash_on(node_hostemu_lib, synth):
    from densesurfelmapping_amd import api, surfel_map
    emu = C.CDLL(node_hostemu_lib)
    cam = synth.NODE_CAM
    node = surfel_map.SurfelMap(cam, drift_free_poses=3, _library=emu)
    img, dep, _ = synth.render(cam, synth.Scene(), 0)
    with pytest.raises(api.DsmError):   # cv_bridge conversions are not part of the library
        node.image_input((1, 0), img, encoding="bgr8")
    with pytest.raises(api.DsmError):   # wrong size
        node.image_input((1, 0), img[:-8])
    cov = np.zeros(36)
    cov[1] = 3.0                        # reference keyframe that does not exist: poses_database[3] on an empty database
    with pytest.raises(api.DsmError):
        node.orb_results_input((1, 0), [], np.zeros((1, 7)), synth.pose7(np.eye(4)), cov)
    cov[1] = 0.0
    node.orb_results_input((1, 0), [], np.stack([synth.pose7(np.eye(4))]), synth.pose7(np.eye(4)), cov)
    assert node.pose_count == 1 and node.frames_fused == 0   # pose queued, no image yet
    node.image_input((1, 0), img)
    node.depth_input((1, 0), dep)
    assert node.frames_fused == 1 and len(node.local_surfels()) > 0
    with pytest.raises(api.DsmError):   # empty loop path with keyframes present: SM.cpp:258-262 reads poses[-1]
        node.orb_results_input((1, 100000000), [], np.zeros((0, 7)), synth.pose7(np.eye(4)), cov)
    with pytest.raises(api.DsmError):   # no point passes update_times >= 5 yet and the inactive set is empty
        node.save_cloud(os.path.join(ROOT, "tests", "_build", "empty.PCD"))
    node.close()


def test_node_survives_a_lost_frame_and_bounds_its_buffers(node_hostemu_lib, synth):
    """A pose whose image (or depth) message was lost: the reference's synchronize_msgs spins forever once a newer
    frame waits at the front of the buffer (surfel_map.cpp:114-139).  Here the pose is dropped and counted, later
    poses fuse; frames nobody claims are bounded by max_buffered_frames."""
    from densesurfelmapping_amd import surfel_map
    emu = C.CDLL(node_hostemu_lib)
    cam = synth.NODE_CAM
    node = surfel_map.SurfelMap(cam, drift_free_poses=3, max_buffered_frames=4, _library=emu)
    img, dep, _ = synth.render(cam, synth.Scene(), 0)
    ident = synth.pose7(np.eye(4))
    cov = np.zeros(36)
    cov[0] = 1.0
    node.orb_results_input((1, 0), [], np.stack([ident]), ident, cov)      # keyframe 0, its frame never arrives
    assert node.frames_fused == 0 and node.dropped_poses == 0             # nothing buffered yet: the pose waits
    node.image_input((2, 0), img)                                          # the next frame's image is already there
    assert node.dropped_poses == 1 and node.frames_fused == 0
    cov[0] = 0.0
    node.orb_results_input((2, 0), [], np.stack([ident]), ident, cov)
    node.depth_input((2, 0), dep)
    assert node.frames_fused == 1 and node.dropped_poses == 1
    # a depth message goes missing: image (3,0) alone, then the whole of frame (4,0)
    node.orb_results_input((3, 0), [], np.stack([ident]), ident, cov)
    node.image_input((3, 0), img)
    node.orb_results_input((4, 0), [], np.stack([ident]), ident, cov)
    node.image_input((4, 0), img)
    node.depth_input((4, 0), dep)                                          # depth front (4,0) is newer than pose (3,0)
    assert node.dropped_poses == 2 and node.frames_fused == 2
    for k in range(20):                                                    # frames without poses do not pile up
        node.image_input((10 + k, 0), img)
        node.depth_input((10 + k, 0), dep)
    node.orb_results_input((29, 0), [], np.stack([ident]), ident, cov)     # the newest one is still matchable
    assert node.frames_fused == 3
    node.orb_results_input((12, 0), [], np.stack([ident]), ident, cov)     # this one fell off the bounded buffer
    assert node.frames_fused == 3 and node.dropped_poses == 3
    node.close()


# ------------------------------------------------------------------ KITTI-layout reader
def _write_png_grey(path, img, filters=(0, 1, 2, 3, 4)):
    """Minimal PNG writer for the test (cycles through all five row filters so that the decoder sees each)."""
    import struct
    import zlib
    h, w = img.shape
    rows = bytearray()
    prev = np.zeros(w, np.int32)
    for y in range(h):
        f = filters[y % len(filters)]
        cur = img[y].astype(np.int32)
        left = np.concatenate([[0], cur[:-1]])
        ul = np.concatenate([[0], prev[:-1]])
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - left
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - (left + prev) // 2
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            enc = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        rows += bytes([f]) + (enc % 256).astype(np.uint8).tobytes()
        prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(bytes(rows))) + chunk(b"IEND", b""))


def test_kitti_layout_reader(synth, tmp_path):
    """kitti_publisher's directory layout (image_0/%06d.png, depth_0/%06d.npy with depth = 386.1448 / disparity,
    publisher.py:31-38) + a KITTI-format pose file -> the node's message stream: a synthetic sequence written in that
    layout comes back as the messages synth.node_messages would have produced, and round-trips through a message log."""
    from densesurfelmapping_amd import kitti, msglog
    cam, scene = synth.NODE_CAM, synth.Scene(seed=9)
    seq = tmp_path / "sequences" / "00"
    (seq / "image_0").mkdir(parents=True)
    (seq / "depth_0").mkdir()
    n = 7
    frames = [synth.render(cam, scene, t) for t in range(n)]
    with open(tmp_path / "poses.txt", "w") as f:
        for t, (img, dep, pose) in enumerate(frames):
            _write_png_grey(str(seq / "image_0" / ("%06d.png" % t)), img)
            with np.errstate(divide="ignore"):
                np.save(str(seq / "depth_0" / ("%06d.npy" % t)), (kitti.BF_SEQ_00_02 / dep.astype(np.float64)).astype(np.float32))
            f.write(" ".join("%.17g" % v for v in pose.astype(np.float64)[:3].ravel()) + "\n")
    open(seq / "calib.txt", "w").write(f"P0: {cam.fx} 0 {cam.cx} 0 0 {cam.fy} {cam.cy} 0 0 0 1 0\n")
    # both PNG paths: Pillow (if present) and the built-in decoder
    raw = open(seq / "image_0" / "000003.png", "rb").read()
    assert np.array_equal(kitti.decode_png(raw), frames[3][0]) and np.array_equal(kitti.read_grey(str(seq / "image_0" / "000003.png")), frames[3][0])
    got_cam = kitti.camera_from_calib(str(seq), cam.width, cam.height)
    assert (got_cam.fx, got_cam.cy) == (cam.fx, cam.cy)
    ev = list(kitti.messages(str(seq), kitti.read_poses(str(tmp_path / "poses.txt")), keyframe_every=3))
    assert len(ev) == 3 * n and [e[0] for e in ev[:3]] == ["image", "depth", "orb"]
    for t in range(n):
        img_e, dep_e, orb_e = ev[3 * t:3 * t + 3]
        assert np.array_equal(img_e[2], frames[t][0])
        valid = frames[t][1] > 0
        assert np.allclose(dep_e[2][valid], frames[t][1][valid], rtol=1e-6) and (dep_e[2][~valid] == 0).all()
        n_kf_before = (t + 2) // 3                               # keyframes at t = 0, 3, 6
        assert orb_e[5][0] == (1.0 if t % 3 == 0 else 0.0) and orb_e[5][1] == max(n_kf_before - 1, 0)
        assert np.allclose(orb_e[4], synth.pose7(frames[t][2]), atol=1e-12)
        assert len(orb_e[3]) == t // 3 + 1                       # loop path = keyframes so far
    log = str(tmp_path / "kitti.log")
    msglog.write_log(log, got_cam, 10, iter(ev))
    cam2, dfp, back = msglog.read_log(log)
    back = list(back)
    assert (cam2.width, cam2.height, dfp) == (cam.width, cam.height, 10) and len(back) == len(ev)
    assert np.array_equal(back[4][2], ev[4][2]) and np.array_equal(back[5][3], ev[5][3])
    # the replay driver's frame source over the same directory: serial, and decoded by worker processes -- same frames, in order
    from densesurfelmapping_amd import replay
    serial = replay.KittiSource(str(seq), str(tmp_path / "poses.txt"))
    pooled = replay.KittiSource(str(seq), str(tmp_path / "poses.txt"), decode_workers=2)
    assert serial.n_frames == pooled.n_frames == n
    fa, fb = list(serial.frames(1, n)), list(pooled.frames(1, n))
    pooled.close()
    assert len(fa) == len(fb) == n - 1
    for (ia, da, pa), (ib, db, pb), t in zip(fa, fb, range(1, n)):
        assert np.array_equal(ia, frames[t][0]) and np.array_equal(ia, ib) and np.array_equal(da, db) and np.array_equal(pa, pb)
        assert (da[frames[t][1] == 0] == 0).all()  # (this directory stores depth 0 as an infinite disparity; bf / inf = 0)


# ------------------------------------------------------------------ the reference's own sources on top of the product
def test_reference_ros_node_compiles_unchanged_against_the_product(ros_node_on_product, synth, tmp_path):
    """surfel_fusion/src/ros_node.cpp (main(), the nh.subscribe / message_filters wiring, the save calls) builds against
    include/ros_compat/surfel_map.h without an edit; without a GPU its `SurfelMap surfel_map(nh)` refuses to start."""
    import node_state
    case = node_state.SCENARIOS[0]
    log = str(tmp_path / "events.bin")
    _write_node_events(log, synth.NODE_CAM, dict(case, frames=4), synth)
    env = dict(os.environ, DSM_ROS_SHIM_LOG=log, DSM_ROS_SHIM_SAVE_NAME=str(tmp_path / "out"))
    r = subprocess.run([ros_node_on_product], env=env, capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr[-2000:]
    else:
        assert r.returncode != 0 and "dsm::SurfelMap" in r.stderr, r.stderr[-2000:]


def test_reference_surfel_map_compiles_unchanged_on_the_engine_facade(ref_map_on_product):
    """surfel_map.cpp + surfel_map.h of the reference, not a line changed, with FusionFunctions =
    the HIP engine's facade.  (Run on the GPU by test_gpu_parity.py::test_reference_node_on_the_hip_engine.)"""
    out = subprocess.run(["nm", "-D", "--undefined-only", ref_map_on_product], capture_output=True, text=True, check=True).stdout
    assert " dsm_fuse_initialize_map" in out and " dsm_create" in out  # the reference's node calls into the product's C ABI
    assert "fuse_surfels_kernel" not in subprocess.run(["nm", "-D", "-C", ref_map_on_product], capture_output=True, text=True).stdout


# ------------------------------------------------------------------ parity at scale (vectors: tests/golden/make_golden_long.py)
def _map_sha(a, dtype):
    from node_state import _canon
    return hashlib.sha256(_canon(np.ascontiguousarray(a, dtype))).hexdigest()


def test_port_oracle_matches_long_golden(ob, synth):
    """oracle/dsm_oracle.c over the 200-frame 1226x370 parity sequence and the large-map cases against the digests
    recorded from the reference's own fusion_functions.cpp: the port is what the GPU tests compare with byte for
    byte when a digest differs, so it is pinned at these sizes too."""
    import scale_cases
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "long_golden.json")))
    assert gold["stereo_sequences"][0]["n_nonfinite"] > 0, "the +inf feed leaves no non-finite surfel: the case lost its point"
    for tum in gold["tum_sequences"]:  # BASELINE configs[3]: 640x480, the RGB-D constant set, TUM-style depth
        per = tum["per_frame"]
        assert sum(f["n_holes"] for f in per) > 1000 and any(f["n_holes"] > f["n_new"] for f in per) and any(0 < f["n_holes"] < f["n_new"] for f in per), \
            tum["name"] + ": pruning or one of the compaction branches never occurs"
    # (stereo: the reference's real feed, publisher.py:37-40)
    for case in [gold["sequence"]] + gold["stereo_sequences"] + gold["tum_sequences"]:
        cam, scene = getattr(synth, case["camera"]), synth.Scene(**case["scene"])
        orc = ob.PortOracle(cam)
        local = np.zeros(0, ob.SURFEL_DTYPE)
        for (t, img, dep, pose, ref), want in zip(synth.sequence(cam, scene, case["frames"], keyframe_every=case.get("keyframe_every", 5)), case["per_frame"]):
            local, k = orc.fuse_map(ref, img, dep, pose, local)
            assert (k, len(local)) == (want["n_new"], want["n_local"]), (case["name"], t)
            assert hashlib.sha256(orc.labels().tobytes()).hexdigest() == want["labels_sha256"], (case["name"], t)
            if str(t + 1) in case["map_sha256"]:
                assert _map_sha(local, ob.SURFEL_DTYPE) == case["map_sha256"][str(t + 1)], (case["name"], t)
    for key, sc in (("large_map", scale_cases.LARGE_MAP), ("fullhd_2m", scale_cases.FULLHD_2M)):
        cam = getattr(synth, sc["camera"])
        big, (t, img, dep, pose, ref) = scale_cases.large_map_inputs(ob.PortOracle(cam), synth, ob.SURFEL_DTYPE, sc)
        for trial, want in zip(sc["trials"], gold[key]):
            m = scale_cases.large_map_variant(big, trial, synth)
            assert _map_sha(m, ob.SURFEL_DTYPE) == want["in_sha256"], (key, trial)
            after, k = ob.PortOracle(cam).fuse_map(ref, img, dep, pose, m)
            assert (k, len(after)) == (want["n_new"], want["n_local"]), (key, trial)
            assert _map_sha(after, ob.SURFEL_DTYPE) == want["map_sha256"], (key, trial)


def test_bench_gpus_flag_fails_loudly_without_devices():
    """`python bench.py --gpus 2` where two GPUs are not visible (this container: none) exits non-zero and prints no result
    line -- round 2's bench ran ONE rank and reported n_gpus 1 (missing #1)."""
    import subprocess
    import sys
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
            pytest.skip("two GPUs visible: covered by the gpu tests")
    except ImportError:
        pytest.skip("no torch")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSM_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible GPUs" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_eigen_last_place_exposure_is_bounded(oracle_built):
    """The two Eigen operations of the hot path (Matrix4f::inverse FF.cpp:59, Matrix4d::inverse FF.cpp:176) are restated,
    not pinned (Eigen is absent).  tools/eigen_exposure.py moves every element of either result by one ulp inside the
    reference's own TU and replays a sequence; profiles/r03_eigen_exposure.md holds the table for 100 frames at 1226x370.
    Gate, on a short sequence: the double inverse of the Gauss-Newton step has no effect at all (the update is rounded to
    float), the float inverse of the pose changes no surfel count here and moves floats by < 1e-3 relative."""
    if not os.path.isdir("/root/reference/surfel_fusion/src"):
        pytest.skip("needs the reference sources (the perturbable TU is built from them)")
    subprocess.run(["make", "-s", "-C", oracle_built, "perturb"], check=True)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eigen_exposure
    jobs = [("TINY", 30, name, f, d) for name, f, d in eigen_exposure.patterns(n_random=2)]
    res = [eigen_exposure.replay(j) for j in jobs[:1]]
    keep = [j for j in jobs[1:] if "random" in j[2] or j[2].endswith("[0]+1ulp") or j[2].endswith("[14]-1ulp")]
    res += [eigen_exposure.replay(j) for j in keep]
    base = res[0]
    for name, counts, m in res[1:]:
        row = eigen_exposure.compare(base[1], base[2], counts, m)
        if name.startswith("gn_inverse_f64"):
            assert row["first_frame_with_other_counts"] is None and row["surfels_differing_in_any_bit"] == 0, (name, row)
        else:
            assert abs(row["final_surfels"] - row["final_surfels_baseline"]) <= 2, (name, row)
            if row["first_frame_with_other_counts"] is None:
                assert row["integer_fields_changed"] == 0 and row["max_rel_float_drift"] < 1e-3, (name, row)


def test_inverse_pose_golden_is_the_perturbed_reference_tu(oracle_built):
    """tests/golden/inv_pose_perturbed.npz (what the GPU test of the *_inv entry points replays) regenerated from the
    reference's own TU where its sources exist: the recorded inverses, counts and final map are that TU's, byte for byte;
    everywhere: the fixture is self-consistent (the recorded inverses are the closed-form inverses moved by the recorded
    ulps -- a float-level check that needs no reference)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "inv_pose_perturbed.npz"))
    from densesurfelmapping_amd import synth
    cam, scene, n = getattr(synth, str(g["camera"])), synth.Scene(seed=int(g["scene_seed"])), int(g["frames"])
    assert g["inv_poses_cm"].shape == (n, 16) and len(g["n_new"]) == n and len(g["final_map"]) == int(g["n_local"][-1])
    # pose * recorded inverse ~ identity (the perturbation is a few ulps)
    for t, img, dep, pose, ridx in synth.sequence(cam, scene, n):
        inv = g["inv_poses_cm"][t].reshape(4, 4).T.astype(np.float64)
        assert np.allclose(np.asarray(pose, np.float64) @ inv, np.eye(4), atol=1e-4), t
    if not os.path.isdir("/root/reference/surfel_fusion/src"):
        return
    subprocess.run(["make", "-s", "-C", oracle_built, "perturb"], check=True)
    import ctypes
    from oracle.bindings import SURFEL_DTYPE, RefOracle
    ref = RefOracle(cam, kind="serial_perturb")
    ref.lib.dsmref_set_eigen_perturb.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ref.lib.dsmref_inverse4f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ulps = np.ascontiguousarray(g["ulps"], np.int32)
    ref.lib.dsmref_set_eigen_perturb(ulps.ctypes.data_as(ctypes.c_void_p), None)
    local = np.zeros(0, SURFEL_DTYPE)
    for t, img, dep, pose, ridx in synth.sequence(cam, scene, n):
        pose_cm = np.ascontiguousarray(np.asarray(pose, np.float32).T).ravel()
        out = np.zeros(16, np.float32)
        ref.lib.dsmref_inverse4f(pose_cm.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        assert out.tobytes() == g["inv_poses_cm"][t].tobytes(), t
        local, k = ref.fuse_map(ridx, img, dep, pose, local)
        assert (k, len(local)) == (int(g["n_new"][t]), int(g["n_local"][t])), t
    assert local.tobytes() == g["final_map"].tobytes()
    ref.lib.dsmref_set_eigen_perturb(None, None)


def test_shipped_sources_carry_no_hooks():
    """The product translation units: no environment lookups, no compile-time experiment switches, no CUDA shims, and
    nothing from oracle/ -- experiments live as patches under tools/_exp, the oracle is test infrastructure."""
    import re
    csrc = os.path.join(ROOT, "densesurfelmapping_amd", "csrc")
    banned = [r"\bgetenv\s*\(", r"__HIP_PLATFORM_(AMD|NVIDIA)__", r"cuda_runtime", r"#\s*include\s*[\"<][^\">]*oracle"]
    seen = []
    for name in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, name)).read()
        for pat in banned:
            for m in re.finditer(pat, text):
                seen.append((name, text.count("\n", 0, m.start()) + 1, m.group(0)))
        # ANY preprocessor conditional on a DSM_ macro is a compile-time switch -- #if, #ifdef, #ifndef, #elif, defined(...) --
        # except the one instrumented build the tools make (DSM_WAVE_STAMPS, tools/wave_stamps.py) and include guards
        for m in re.finditer(r"^[ \t]*#[ \t]*(if|ifdef|ifndef|elif)\b[^\n]*\bDSM_\w+", text, re.M):
            macros = set(re.findall(r"\bDSM_\w+", m.group(0)))
            guard = m.group(1) == "ifndef" and any(g.endswith(("_H", "_H_", "_HPP")) for g in macros)
            if not guard and macros - {"DSM_WAVE_STAMPS"}:
                seen.append((name, text.count("\n", 0, m.start()) + 1, m.group(0).strip()))
    assert seen == [], seen
    probe = "#ifndef DSM_BATCH_PARAMS_ON_BATCH\n#  if defined(DSM_EXP_X)\n#ifdef DSM_WAVE_STAMPS\n"
    assert len(re.findall(r"^[ \t]*#[ \t]*(if|ifdef|ifndef|elif)\b[^\n]*\bDSM_\w+", probe, re.M)) == 3  # (the pattern does see such lines)
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):
        path = os.path.join(ROOT, "include", name)
        if os.path.isfile(path):
            assert "oracle" not in open(path).read(), name
