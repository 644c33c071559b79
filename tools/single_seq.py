#!/usr/bin/env python
"""One subsequence through one handle with frame groups (bench.py's single_sequence leg alone, for profiling)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densesurfelmapping_amd import api, synth
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 640
cam, period = synth.KITTI_1226, 50
scene = synth.Scene(seed=12345, frames_per_period=period)
frames = synth.render_many([(cam, scene, i) for i in range(period)])
ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 21, pipeline_depth=depth)
for i, (img, dep) in enumerate(frames):
    ff.frame_upload(i, img, dep)
ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
total = 160 + n
s, r, p = api.FusionFunctions.pack_replay([t % period for t in range(total)], [t // 5 for t in range(total)], np.stack([scene.pose(t) for t in range(total)]))
ff.replay_enqueue(s[:160], r[:160], p[:160]); ff.synchronize()
t0 = time.perf_counter()
ff.replay_enqueue(s[160:], r[160:], p[160:])
te = time.perf_counter() - t0
ff.synchronize()
dt = time.perf_counter() - t0
print(f"depth {depth}: {n / dt:.0f} frames/s, enqueue {te * 1e3:.1f} ms of {dt * 1e3:.1f} ms")
