#!/usr/bin/env python
"""Per-wave phase stamps of k_seed_fit in a launch batched over B handles (DSM_FLAG_WAVE_STAMPS): where does a wave's
life go?  usage (repo root, GPU): python tools/_exp/r04_fit_stamps.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from densesurfelmapping_amd import api, synth  # noqa: E402

cam = synth.KITTI_1226
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
period, total = 6, 6
scenes = [synth.Scene(seed=12345 + 17 * (b % 16), frames_per_period=50) for b in range(B)]
frames = synth.render_many([(cam, scenes[b], i + 25 * (b // 16)) for b in range(B) for i in range(period)])
handles, plans = [], []
for b in range(B):
    ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 20, pipeline_depth=1,
                                         flags=api.DSM_FLAG_WAVE_STAMPS)
    for i in range(period):
        ff.frame_upload(i, *frames[b * period + i][:2])
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    plans.append(api.FusionFunctions.pack_replay(list(range(total)), [t // 5 for t in range(total)],
                                                 np.stack([scenes[b].pose(t + 25 * (b // 16)) for t in range(total)])))
    handles.append(ff)
bt = api.Batch(handles)
s, r, p, n = api.Batch.pack(plans)
bt.replay_enqueue(s, r, p, n)
bt.synchronize()
phase = ["entry -> header", "gather + points", "step 1", "steps 2..5", "store"]
rows = []
for ff in handles:
    a = ff.debug_wave_stamps()[4]
    rows.append(a[a[:, 5] > 0])
a = np.concatenate(rows)
t0 = a[:, 0].min()
print(f"== k_seed_fit, last frame: {len(a)} groups stamped, span {a[:, 5].max() - t0} clk")
st = a[:, 0] - t0
print(f"   start offset: median {np.median(st):.0f} p90 {np.percentile(st, 90):.0f} max {st.max():.0f}")
for ph in range(1, 6):
    ok = (a[:, ph] > 0) & (a[:, ph - 1] > 0)
    d = a[ok, ph] - a[ok, ph - 1]
    print(f"   {phase[ph - 1]:18s}: n={ok.sum():6d} median {np.median(d):8.0f} p90 {np.percentile(d, 90):8.0f} max {d.max():8.0f}")
tot = a[:, 5] - a[:, 0]
print(f"   total: median {np.median(tot):.0f} p90 {np.percentile(tot, 90):.0f} max {tot.max():.0f};  longest list median {np.median(a[:, 7]):.0f} max {a[:, 7].max():.0f}")
busy = a[a[:, 7] > 0]
tot = busy[:, 5] - busy[:, 0]
print(f"   groups with a list: {len(busy)}, total median {np.median(tot):.0f}")
for lo, hi in ((1, 40), (40, 64), (64, 80), (80, 121)):
    sel = (busy[:, 7] >= lo) & (busy[:, 7] < hi)
    if sel.any():
        d = busy[sel]
        print(f"   longest list {lo:3d}..{hi - 1:3d}: {sel.sum():6d} groups; gather {np.median(d[:, 2] - d[:, 1]):7.0f}  step1 {np.median(d[:, 3] - d[:, 2]):7.0f}  "
              f"steps2-5 {np.median(d[:, 4] - d[:, 3]):7.0f}  total {np.median(d[:, 5] - d[:, 0]):7.0f}")
