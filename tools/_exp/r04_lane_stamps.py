#!/usr/bin/env python
"""Per-workgroup phase stamps of the lane-per-seed update kernel in a launch batched over B handles
(DSM_FLAG_WAVE_STAMPS): where does a workgroup's life go?  usage: python tools/lane_stamps.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densesurfelmapping_amd import api, synth  # noqa: E402

cam = synth.KITTI_1226
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
period, total = 12, 12
scenes = [synth.Scene(seed=12345 + 17 * b, frames_per_period=50) for b in range(B)]
frames = synth.render_many([(cam, scenes[b], i) for b in range(B) for i in range(period)])
handles, plans = [], []
for b in range(B):
    ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 20, pipeline_depth=1,
                                         flags=api.DSM_FLAG_WAVE_STAMPS)
    for i in range(period):
        ff.frame_upload(i, *frames[b * period + i])
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    plans.append(api.FusionFunctions.pack_replay(list(range(total)), [t // 5 for t in range(total)],
                                                 np.stack([scenes[b].pose(t) for t in range(total)])))
    handles.append(ff)
bt = api.Batch(handles)
s, r, p, n = api.Batch.pack(plans)
bt.replay_enqueue(s, r, p, n)
bt.synchronize()
names = ["update_seeds_0", "update_seeds_1", "update_seeds_2"]
phase = ["entry->ctx+tmin", "walk", "barrier+store+barrier", "sum", "huber pass 1", "queue+finish"]
for k in range(3):
    rows = []
    for ff in handles:
        a = ff.debug_wave_stamps()[k]
        rows.append(a[a[:, 6] > 0])
    a = np.concatenate(rows)
    t0 = a[:, 0].min()
    print(f"== {names[k]}: {len(a)} workgroups, span {a[:, 6].max() - t0} clk (100 MHz wall clock? see clock64)")
    st = a[:, 0] - t0
    print(f"   start offset: median {np.median(st):.0f} p90 {np.percentile(st, 90):.0f} max {st.max():.0f}")
    for ph in range(1, 7):
        d = a[:, ph] - a[:, ph - 1]
        print(f"   {phase[ph - 1]:24s}: median {np.median(d):8.0f} p90 {np.percentile(d, 90):8.0f} max {d.max():8.0f}")
    tot = a[:, 6] - a[:, 0]
    print(f"   total: median {np.median(tot):.0f} p90 {np.percentile(tot, 90):.0f} max {tot.max():.0f};  longest list median {np.median(a[:, 7]):.0f} max {a[:, 7].max():.0f}")
