#!/usr/bin/env python
"""What does the reference's summation ORDER buy?  Runs a build of the library (DSM_LIB_PATH; meant for
tools/_exp/libdsm_relaxed.so = -DDSM_RELAXED_SUMS=1, where the plane fit's order-sensitive sums are four interleaved
partial sums) over the 200-frame 1226x370 parity sequence and measures it against the oracle under the north-star
contract -- label image and surfel counts exact, float attributes within 1e-4 relative -- frame by frame, every frame
starting from the ORACLE's map so that one flipped threshold does not hide the frames after it.  Prints one JSON line.

    DSM_LIB_PATH=$PWD/tools/_exp/libdsm_relaxed.so python tools/relaxed_check.py [frames]
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densesurfelmapping_amd import api, synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402

subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cam, scene = synth.KITTI_1226, synth.Scene(seed=12345)
ff = api.FusionFunctions.from_camera(cam, frame_slots=1, surfel_capacity=1 << 20)
orc = ob.PortOracle(cam)
lo = np.zeros(0, ob.SURFEL_DTYPE)
out = {"lib": os.path.basename(api.LIB_PATH), "frames": n_frames, "label_mismatch_frames": 0, "count_mismatch_frames": [],
       "surfel_bytes_equal_frames": 0, "seed_flag_flips": 0, "max_rel_err": 0.0, "max_rel_err_field": None, "nan_mask_mismatch": 0,
       "outside_rtol1e-4_atol1e-6": 0, "max_abs_err": 0.0}


def rel_err(a, b):
    m = ~(np.isnan(a) | np.isnan(b))
    if not m.any():
        return 0.0
    den = np.maximum(np.abs(b[m]), 1e-6)
    return float((np.abs(a[m] - b[m]) / den).max())


for t, img, dep, pose, ref in synth.sequence(cam, scene, n_frames):
    ff.frame_upload(0, img, dep)
    ff.map_upload(lo.astype(api.SURFEL_DTYPE))
    ff.fuse_frame_resident(0, ref, pose)
    k_g = ff.last_new_count()
    got = ff.map_download()
    lo, k_o = orc.fuse_map(ref, img, dep, pose, lo)
    if not np.array_equal(ff.labels(), orc.labels()):
        out["label_mismatch_frames"] += 1
    sg, so = ff.seeds(), orc.seeds()
    out["seed_flag_flips"] += int(((sg["fused"] != so["fused"]) | ((sg["norm_x"] == 0) != (so["norm_x"] == 0)) |
                                   ((sg["view_cos"] < 0.1) != (so["view_cos"] < 0.1))).sum())
    if k_g != k_o or len(got) != len(lo):
        out["count_mismatch_frames"].append({"frame": t, "new": [k_g, k_o], "total": [len(got), len(lo)]})
        continue
    want = lo.astype(api.SURFEL_DTYPE)
    if got.tobytes() == want.tobytes():
        out["surfel_bytes_equal_frames"] += 1
    for f in api.SURFEL_DTYPE.names:
        if got[f].dtype.kind == "f":
            out["nan_mask_mismatch"] += int((np.isnan(got[f]) != np.isnan(want[f])).sum())
            ok = ~(np.isnan(got[f]) | np.isnan(want[f]))
            out["outside_rtol1e-4_atol1e-6"] += int((~np.isclose(got[f][ok], want[f][ok], rtol=1e-4, atol=1e-6)).sum())
            if ok.any():
                out["max_abs_err"] = max(out["max_abs_err"], float(np.abs(got[f][ok] - want[f][ok]).max()))
            e = rel_err(got[f], want[f])
            if e > out["max_rel_err"]:
                out["max_rel_err"], out["max_rel_err_field"] = e, f
        elif not np.array_equal(got[f], want[f]):
            out["count_mismatch_frames"].append({"frame": t, "int_field": f})
out["n_count_mismatch_frames"] = len(out["count_mismatch_frames"])
out["count_mismatch_frames"] = out["count_mismatch_frames"][:5]
# the tests' form of the float contract (conftest.fields_close): |a - b| <= 1e-6 + 1e-4 |b|, identical NaN masks
out["contract_met"] = (out["label_mismatch_frames"] == 0 and out["n_count_mismatch_frames"] == 0 and
                       out["outside_rtol1e-4_atol1e-6"] == 0 and out["nan_mask_mismatch"] == 0)
print(json.dumps(out))
