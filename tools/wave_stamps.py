#!/usr/bin/env python
"""Per-wave phase timing of the per-seed kernels (DSM_FLAG_WAVE_STAMPS): where does a wave's life go?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the shipped library holds no stamp code: this tool runs on its own instrumented copy (built here if missing)
STAMPED = os.path.join(ROOT, "tools", "_exp", "ab", "libdsm_hip_stamps.so")
if not os.path.exists(STAMPED):
    from densesurfelmapping_amd import build  # noqa: E402
    build.build_library(force=True, defines=("DSM_WAVE_STAMPS=1",), out=STAMPED)
os.environ["DSM_LIB_PATH"] = STAMPED
from densesurfelmapping_amd import api, synth  # noqa: E402

cam, scene = synth.KITTI_1226, synth.Scene()
n = 12
ff = api.FusionFunctions.from_camera(cam, frame_slots=n, surfel_capacity=1 << 20, flags=api.DSM_FLAG_NO_GRAPH | api.DSM_FLAG_WAVE_STAMPS)
frames = list(synth.sequence(cam, scene, n))
for t, img, dep, pose, ref in frames:
    ff.frame_upload(t, img, dep)
ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
for t, img, dep, pose, ref in frames:
    ff.fuse_frame_resident(t, ref, pose)
ff.synchronize()
st = ff.debug_wave_stamps()
names = ["update_seeds_0", "update_seeds_1", "update_seeds_2", "seed_points", "seed_fit"]
for k in range(5):
    a = st[k]
    live = a[:, 5] > 0
    if not live.any():
        continue
    t0 = a[live, 0].min()
    print(f"== {names[k]}: {live.sum()} waves reached the end, kernel span (first start -> last end) "
          f"{(a[live, 5].max() - t0)} clk")
    start = a[live, 0] - t0
    print(f"   wave start offset: median {np.median(start):.0f} p99 {np.percentile(start, 99):.0f} max {start.max():.0f}")
    for ph in range(1, 6):
        ok = live & (a[:, ph] > 0) & (a[:, ph - 1] > 0)
        if ok.any():
            d = a[ok, ph] - a[ok, ph - 1]
            print(f"   phase {ph-1}->{ph}: n={ok.sum():5d} median {np.median(d):8.0f} p90 {np.percentile(d, 90):8.0f} "
                  f"p99 {np.percentile(d, 99):8.0f} max {d.max():8.0f}")
    tot = a[live, 5] - a[live, 0]
    print(f"   total: median {np.median(tot):.0f} p90 {np.percentile(tot, 90):.0f} p99 {np.percentile(tot, 99):.0f} max {tot.max():.0f}")
    nn = a[live, 7]
    worst = np.argsort(tot)[-5:]
    print("   slowest waves: total clk / list length:", [(int(tot[i]), int(nn[i])) for i in worst])
    print(f"   list length: median {np.median(nn):.0f} p99 {np.percentile(nn, 99):.0f} max {nn.max():.0f}")
