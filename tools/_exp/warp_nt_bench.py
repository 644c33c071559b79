import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from densesurfelmapping_amd import api, synth
n_w = 8_000_000
wm = np.zeros(n_w, api.SURFEL_DTYPE); wm["px"] = np.arange(n_w, dtype=np.float32) * 1e-3; wm["nz"] = 1.0; wm["update_times"] = 3
ff = api.FusionFunctions.from_camera(synth.TINY, surfel_capacity=n_w + 64)
ff.map_upload(wm)
wp = np.eye(4, dtype=np.float32); wp[:3, 3] = (0.01, -0.02, 0.005)
stream = torch.cuda.ExternalStream(ff.stream())
ff.map_warp(wp); ff.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(30): ff.map_warp(wp)
e1.record(stream); e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
print("warp 8M: %.1f us  %.2f TB/s" % (us, 88 * n_w / us / 1e6))
# fuse: surfels that project into a TINY frame?  use timing of fuse kernel via replay_timed on a synthetic scene map is complex: warp only here
