"""Large-map parity cases shared by tests/golden/make_golden_long.py (reference TU, this container) and the GPU tests
(port oracle, GPU box): deterministic construction of maps far above the 65 536 surfels a short replay produces.

A big map is the map of a short replay replicated `reps` times with sub-millimetre jitter (so that the replicas project
into the next frame and take the fusion branch), then a hash-selected fraction is made stale (`last_update` far in
the past, `update_times` < 5: pruned by the next frame, FF.cpp:206-210) and a few slots are deleted outright
(`update_times == 0`, what move_add_surfels leaves behind, SM.cpp:1494).  Everything is a counter-based integer
hash (synth._uniform01), never numpy's RNG streams, so the same bytes come out everywhere; the golden record pins
the SHA-256 of every input map besides the outputs.
"""
import numpy as np

LARGE_MAP = {
    "camera": "KITTI_1226", "scene": {"seed": 12345}, "base_frames": 6, "target": 600_000,
    # stale fraction, fraction of slots already deleted on entry
    "trials": [{"stale": 0.10, "dead": 0.0}, {"stale": 0.50, "dead": 0.02}, {"stale": 0.90, "dead": 0.0}],
}
# BASELINE configs[4]: 1920x1080 against >= 2 M live surfels
FULLHD_2M = {"camera": "FULLHD", "scene": {"seed": 12345, "frames_per_period": 10}, "base_frames": 3, "target": 2_000_000,
             "trials": [{"stale": 0.05, "dead": 0.0}]}


def _u01(synth, n, salt):
    return synth._uniform01(np.arange(n, dtype=np.uint32), salt)


def base_map(case, oracle, synth, dtype):
    """Map after `base_frames` frames of the case's sequence on `oracle`, and the next frame (t, img, dep, pose, ref)."""
    cam = getattr(synth, case["camera"])
    scene = synth.Scene(**case["scene"])
    local = np.zeros(0, dtype)
    frames = list(synth.sequence(cam, scene, case["base_frames"] + 1))
    for t, img, dep, pose, ref in frames[:-1]:
        local, _ = oracle.fuse_map(ref, img, dep, pose, local)
    return local, frames[-1]


def tiled(case, base, synth):
    reps = max(1, -(-case["target"] // max(len(base), 1)))
    big = np.tile(base, reps)
    n = len(big)
    for i, f in enumerate(("px", "py", "pz")):
        big[f] = big[f] + ((_u01(synth, n, 101 + i) - 0.5) * 1e-3).astype(np.float32)
    big["update_times"] = 9
    return big


def variant(big, trial, synth):
    m = big.copy()
    n = len(m)
    stale = _u01(synth, n, 7) < trial["stale"]
    m["last_update"][stale] = -100                      # ref_idx - last_update > 5
    m["update_times"][stale] = 1 + (_u01(synth, n, 8)[stale] * 4).astype(np.int32)  # 1..4 (< 5)
    dead = _u01(synth, n, 9) < trial["dead"]
    m["update_times"][dead] = 0
    return m


def large_map_inputs(oracle, synth, dtype, case=None):
    case = case or LARGE_MAP
    base, frame = base_map(case, oracle, synth, dtype)
    return tiled(case, base, synth), frame


def large_map_variant(big, trial, synth=None):
    if synth is None:
        from densesurfelmapping_amd import synth  # noqa: F811
    return variant(big, trial, synth)
