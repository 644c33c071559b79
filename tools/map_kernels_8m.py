#!/usr/bin/env python3
"""The two map-sized kernels on a working set the 256 MB Infinity Cache cannot hold -- the same two measurements as
bench.py's `fuse_8M` and `map_warp_8M` legs, alone (GPU box, repo root):

    python tools/map_kernels_8m.py [n_surfels] [--warp-only]

k_fuse_surfels: a 1920x1080 frame fused into a live map of n surfels (the map of a short 1080p replay replicated with
millimetre jitter, so that its surfels project into the frame and take the fusion branch); HIP events around the kernel
on the handle's stream.  k_warp: 30 back-to-back dsm_map_warp calls.  88 algorithmic bytes per surfel for both."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from densesurfelmapping_amd import api, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_want = int(argv[0]) if argv else 8_000_000
    out = {}
    if "--warp-only" not in sys.argv:
        out["fuse"] = fuse_leg(n_want)
    out["warp"] = warp_leg(n_want)
    print(json.dumps(out))


def fuse_leg(n_want):
    cam, scene = synth.FULLHD, synth.Scene(seed=12345, frames_per_period=10)
    frames = synth.render_many([(cam, scene, i) for i in range(10)], min(10, os.cpu_count() or 1))
    plan = api.FusionFunctions.pack_replay([t % 10 for t in range(60)], [t // 5 for t in range(60)],
                                           np.stack([scene.pose(t % 10) for t in range(60)]))
    ff = api.FusionFunctions.from_camera(cam, frame_slots=10, surfel_capacity=1 << 20, pipeline_depth=1)
    for i, (img, dep) in enumerate(frames):
        ff.frame_upload(i, img, dep)
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    ff.replay_enqueue(plan[0][:10], plan[1][:10], plan[2][:10])
    base = ff.map_download()
    ff.close()
    rng = np.random.default_rng(0)
    big = np.tile(base, max(1, -(-n_want // max(len(base), 1))))
    for f in ("px", "py", "pz"):
        big[f] += rng.normal(scale=1e-3, size=len(big)).astype(np.float32)
    big["update_times"] = 9
    big["last_update"] = 2
    ff = api.FusionFunctions.from_camera(cam, frame_slots=10, surfel_capacity=len(big) + 600_000, pipeline_depth=1)
    for i, (img, dep) in enumerate(frames):
        ff.frame_upload(i, img, dep)
    ff.map_upload(big)
    del big
    ff.replay_enqueue(plan[0][10:14], plan[1][10:14], plan[2][10:14])
    ff.synchronize()
    st, _ = ff.replay_timed(plan[0][40:48], plan[1][40:48], plan[2][40:48])
    ovh = ff.event_overhead_ms * 1e3
    us_f = max(st["fuse_surfels"][0] / max(st["fuse_surfels"][1], 1) * 1e3 - ovh, 1e-3)
    us_t = max(st["frame_tail"][0] / max(st["frame_tail"][1], 1) * 1e3 - ovh, 1e-3)
    m = ff.timed_mean_local
    ff.close()
    return {"live_surfels": round(m), "us": round(us_f, 1), "TBps": round(88 * m / us_f / 1e6, 3),
            "hbm_frac": round(88 * m / us_f / 1e3 / HBM_PEAK_GBS, 4), "frame_tail_us": round(us_t, 1)}


def warp_leg(n_want):
    wm = np.zeros(n_want, api.SURFEL_DTYPE)
    wm["px"] = np.arange(n_want, dtype=np.float32) * 1e-3
    wm["nz"] = 1.0
    wm["update_times"] = 3
    ff = api.FusionFunctions.from_camera(synth.TINY, surfel_capacity=n_want + 64)
    ff.map_upload(wm)
    del wm
    wp = np.eye(4, dtype=np.float32)
    wp[:3, 3] = (0.01, -0.02, 0.005)
    ff.map_warp(wp)
    ff.synchronize()
    stream = torch.cuda.ExternalStream(ff.stream())  # (fetched right before use: include/dsm.h, dsm_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(30):
        ff.map_warp(wp)
    e1.record(stream)
    e1.synchronize()
    us_w = e0.elapsed_time(e1) * 1e3 / 30
    ff.close()
    return {"surfels": n_want, "us": round(us_w, 1), "TBps": round(88 * n_want / us_w / 1e6, 3),
            "hbm_frac": round(88 * n_want / us_w / 1e3 / HBM_PEAK_GBS, 4)}


if __name__ == "__main__":
    main()
