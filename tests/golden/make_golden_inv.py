"""Golden vectors for the caller-provided pose inverse (include/dsm.h, the *_inv entry points).

The reference inverts the pose with Eigen (`pose.inverse()`, fusion_functions.cpp:59); Eigen is absent here and its
last-place behaviour is unpinned.  oracle/_ref/libdsm_ref_serial_perturb.so is the reference's own
translation unit with every element of that inverse movable by a few ulps -- a stand-in for "some other Eigen build".
This script replays a short sequence through it under a fixed perturbation pattern and records, per frame, the inverse
the TU actually used and the counts, and the final map.  The GPU test feeds the recorded inverses through
dsm_fuse_map_inv / dsm_replay_enqueue_inv and must reproduce the perturbed TU's map byte for byte -- and must NOT without
them (the pattern is chosen so that it matters).

    make -C oracle perturb && python tests/golden/make_golden_inv.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from densesurfelmapping_amd import synth  # noqa: E402
from oracle.bindings import SURFEL_DTYPE, RefOracle  # noqa: E402

CAMERA, SCENE_SEED, FRAMES = "TINY", 12345, 40
ULPS = np.array([+1, -1, +2, 0, -2, +1, +1, -1, 0, +2, -1, +1, -1, +1, -2, 0], np.int32)  # column-major elements of the inverse


def main():
    cam = getattr(synth, CAMERA)
    scene = synth.Scene(seed=SCENE_SEED)
    ref = RefOracle(cam, kind="serial_perturb")
    ref.lib.dsmref_set_eigen_perturb.argtypes = [C.c_void_p, C.c_void_p]
    ref.lib.dsmref_inverse4f.argtypes = [C.c_void_p, C.c_void_p]
    ref.lib.dsmref_set_eigen_perturb(ULPS.ctypes.data_as(C.c_void_p), None)
    local = np.zeros(0, SURFEL_DTYPE)
    inv, n_new, n_local = [], [], []
    for t, img, dep, pose, ridx in synth.sequence(cam, scene, FRAMES):
        pose_cm = np.ascontiguousarray(np.asarray(pose, np.float32).T).ravel()
        out = np.zeros(16, np.float32)
        ref.lib.dsmref_inverse4f(pose_cm.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        inv.append(out)
        local, k = ref.fuse_map(ridx, img, dep, pose, local)
        n_new.append(k)
        n_local.append(len(local))
    # the unperturbed TU on the same frames, for the control ("without the caller's inverse the maps differ")
    base = RefOracle(cam, kind="serial")
    lb = np.zeros(0, SURFEL_DTYPE)
    for t, img, dep, pose, ridx in synth.sequence(cam, scene, FRAMES):
        lb, _ = base.fuse_map(ridx, img, dep, pose, lb)
    differs = len(lb) != len(local) or lb.tobytes() != local.tobytes()
    assert differs, "this perturbation pattern changes nothing: pick another"
    np.savez_compressed(os.path.join(HERE, "inv_pose_perturbed.npz"), camera=CAMERA, scene_seed=SCENE_SEED, frames=FRAMES,
                        ulps=ULPS, inv_poses_cm=np.stack(inv), n_new=np.array(n_new, np.int32), n_local=np.array(n_local, np.int32),
                        final_map=local, unperturbed_final_count=len(lb))
    print(f"{FRAMES} frames at {cam.width}x{cam.height}: perturbed final map {len(local)} surfels, unperturbed {len(lb)}; "
          f"{int((np.frombuffer(lb.tobytes(), np.uint8)[:min(len(lb), len(local)) * 44] != np.frombuffer(local.tobytes(), np.uint8)[:min(len(lb), len(local)) * 44]).sum())} bytes differ")


if __name__ == "__main__":
    main()
