"""Generate tests/golden/ from the REFERENCE's own translation unit.

Runs only where /root/reference exists (this container): oracle/_ref/libdsm_ref_serial*.so is
surfel_fusion/src/fusion_functions.cpp compiled in place (oracle/Makefile, `make ref`), driven with
the deterministic synthetic sequences of densesurfelmapping_amd/synth.py.  The fixtures travel to
the GPU box, the reference does not.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from densesurfelmapping_amd import synth  # noqa: E402
from oracle.bindings import SURFEL_DTYPE, RefOracle  # noqa: E402

CASES = [
    {"name": "tiny_drive_48", "camera": "TINY", "scene": {"seed": 12345}, "frames": 48},
    {"name": "kitti1226_drive_5", "camera": "KITTI_1226", "scene": {"seed": 12345}, "frames": 5},
    {"name": "vga_rgbd_4", "camera": "VGA_RGBD", "scene": {"seed": 5, "scale": 0.12, "step": 0.05}, "frames": 4},
]


def main():
    out = {"generator": "oracle/_ref/libdsm_ref_serial*.so (reference fusion_functions.cpp, serial thread schedule)",
           "cases": []}
    for case in CASES:
        cam = getattr(synth, case["camera"])
        scene = synth.Scene(**case["scene"])
        ref = RefOracle(cam)
        local = np.zeros(0, SURFEL_DTYPE)
        per_frame = []
        for t, img, dep, pose, ridx in synth.sequence(cam, scene, case["frames"]):
            local, k = ref.fuse_map(ridx, img, dep, pose, local)
            seeds = ref.seeds()
            per_frame.append({
                "n_new": int(k), "n_local": int(len(local)),
                "labels_sha256": hashlib.sha256(ref.labels().tobytes()).hexdigest(),
                "n_stable": int(seeds["stable"].sum()),
                "n_nan_seeds": int(np.isnan(seeds["norm_x"]).sum()),
            })
        fname = case["name"] + "_final_map.npy"
        np.save(os.path.join(HERE, fname), local)
        out["cases"].append(dict(case, per_frame=per_frame, final_map=fname))
        print(case["name"], "final surfels", len(local))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
