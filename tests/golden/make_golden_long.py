"""Generate tests/golden/long_golden.json from the REFERENCE's own translation unit: the SURVEY.md §8(d)
parity sequence at the headline resolution (200 frames at 1226x370) and a large-map compaction case.

Runs only where /root/reference exists (this container): oracle/_ref/libdsm_ref_serial.so is
surfel_fusion/src/fusion_functions.cpp compiled in place (oracle/Makefile, `make ref`).  Only digests are
committed (the maps themselves are 8 MB and more): per frame the SHA-256 of the label image, the new / total
surfel counts and the number of deleted slots the frame's compaction consumed; every 50 frames the
NaN-canonical SHA-256 of the whole surfel array (tests/node_state.py `_canon`).

    python tests/golden/make_golden_long.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from densesurfelmapping_amd import synth  # noqa: E402
from node_state import _canon  # noqa: E402
from oracle.bindings import SURFEL_DTYPE, RefOracle  # noqa: E402
import scale_cases  # noqa: E402


def map_sha(a):
    return hashlib.sha256(_canon(np.ascontiguousarray(a, SURFEL_DTYPE))).hexdigest()


LONG = {"name": "kitti1226_drive_200", "camera": "KITTI_1226", "scene": {"seed": 12345}, "frames": 200, "checkpoint_every": 50}
# The reference's real feed (kitti_publisher/scripts/publisher.py:37-40): depth = bf / disparity -- disparity-quantised,
# +inf where the disparity is 0 -- and an image with saturated highlights and eight grey levels; once with the
# infinities as the publisher sends them, once with those pixels at depth 0 (a publisher that masks them).
_STEREO = {"seed": 12345, "stereo": True, "saturate_above": 150.0, "intensity_levels": 8}
STEREO = [
    {"name": "kitti1226_stereo_inf_60", "camera": "KITTI_1226", "scene": dict(_STEREO), "frames": 60, "checkpoint_every": 20},
    {"name": "kitti1226_stereo_zero_60", "camera": "KITTI_1226", "scene": dict(_STEREO, zero_disparity_inf=False), "frames": 60, "checkpoint_every": 20},
]


# BASELINE configs[3]: TUM-RGBD-style feed at 640x480 under the RGB-D constant set (fusion_functions.h:17-21; the library
# is libdsm_ref_serial_rgbd.so = the reference TU compiled with that set): a hand-held sweep through a room, depth quantised
# as a Kinect + the dataset's uint16 / 5000 PNGs deliver it, zero in shadows / blobs / out of range, a keyframe every 4
# frames.  `room`: two laps of a 100-frame loop (revisits: fusion into mature surfels, pruning of what the pan left
# behind); `sparse`: a sensor that loses a quarter of its pixels and sees 3.2 m far.
_TUM = {"seed": 7, "tum": True, "frames_per_period": 100, "intensity_noise": 8.0, "checker": 25.0, "n_boxes": 6}
TUM = [
    {"name": "tum_rgbd_room_200", "camera": "VGA_RGBD", "scene": dict(_TUM), "frames": 200, "checkpoint_every": 50, "keyframe_every": 4},
    {"name": "tum_rgbd_sparse_60", "camera": "VGA_RGBD", "scene": dict(_TUM, seed=11, frames_per_period=60, hole_fraction=0.06, blob_fraction=0.08, tum_far=3.2),
     "frames": 60, "checkpoint_every": 20, "keyframe_every": 4},
]


def long_sequence(case=LONG):
    cam, scene = getattr(synth, case["camera"]), synth.Scene(**case["scene"])
    ref = RefOracle(cam)
    local = np.zeros(0, SURFEL_DTYPE)
    per_frame, checkpoints = [], {}
    for t, img, dep, pose, ridx in synth.sequence(cam, scene, case["frames"], keyframe_every=case.get("keyframe_every", 5)):
        before = len(local)
        local, k = ref.fuse_map(ridx, img, dep, pose, local)
        per_frame.append({"n_new": int(k), "n_local": int(len(local)), "n_holes": int(before + k - len(local)),
                          "labels_sha256": hashlib.sha256(ref.labels().tobytes()).hexdigest()})
        if (t + 1) % case["checkpoint_every"] == 0:
            checkpoints[str(t + 1)] = map_sha(local)
            print(case["name"], "frame", t + 1, "surfels", len(local), flush=True)
    return dict(case, per_frame=per_frame, map_sha256=checkpoints,
                n_mature=int((local["update_times"] >= 5).sum()),
                n_nonfinite=int(sum((~np.isfinite(local[f])).sum() for f in SURFEL_DTYPE.names if local[f].dtype.kind == "f")))


def large_map(case):
    """One fuse_map into a 600 k-surfel map with 10-90 % of it stale (pruned by this frame): the compaction's
    multi-round hole scan, the saturated fuse grid and the K < k tail-hole chains at a realistic size.  With the
    FULLHD_2M case: BASELINE configs[4], one 1920x1080 frame fused into a 2 M-surfel map."""
    out = []
    cam = getattr(synth, case["camera"])
    base, frame = scale_cases.large_map_inputs(RefOracle(cam), synth, SURFEL_DTYPE, case)
    t, img, dep, pose, ridx = frame
    for trial in case["trials"]:
        m = scale_cases.large_map_variant(base, trial)
        ref = RefOracle(cam)
        after, k = ref.fuse_map(ridx, img, dep, pose, m)
        out.append({"trial": trial, "n_in": int(len(m)), "n_stale": int((m["last_update"] < 0).sum()), "n_new": int(k),
                    "n_local": int(len(after)), "n_holes": int(len(m) + k - len(after)), "in_sha256": map_sha(m),
                    "map_sha256": map_sha(after), "labels_sha256": hashlib.sha256(ref.labels().tobytes()).hexdigest()})
        print(case["camera"], trial, out[-1]["n_in"], "->", out[-1]["n_local"], "holes", out[-1]["n_holes"], flush=True)
    return out


def main():
    path = os.path.join(HERE, "long_golden.json")
    if "--only-tum" in sys.argv:  # add / refresh the RGB-D sequences, keep the rest of the record
        out = json.load(open(path))
        out["tum_sequences"] = [long_sequence(c) for c in TUM]
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        return
    if "--only-stereo" in sys.argv:  # add / refresh the stereo sequences, keep the rest of the record
        out = json.load(open(path))
    else:
        out = {"generator": "oracle/_ref/libdsm_ref_serial.so (reference fusion_functions.cpp, serial thread schedule)",
               "sequence": long_sequence(), "large_map": large_map(scale_cases.LARGE_MAP),
               "fullhd_2m": large_map(scale_cases.FULLHD_2M)}
    out["stereo_sequences"] = [long_sequence(c) for c in STEREO]
    out["tum_sequences"] = [long_sequence(c) for c in TUM]
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
