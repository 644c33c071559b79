"""one rank of the sharded replay (replay.HipEngine) at several chunk sizes / pipeline depths, frames already page-locked"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: F401  (first: see __graft_entry__.py)
from densesurfelmapping_amd import replay as rp, synth
cam = synth.KITTI_1226
src = rp.SyntheticSource(4000, camera="KITTI_1226", seed=12345, prerender=True)
out = {}
for depth, chunk in ((24, 24), (24, 48), (24, 96), (24, 192), (32, 64), (16, 48)):
    eng = rp.HipEngine(cam, capacity=1 << 21, pipeline_depth=depth, chunk=chunk)
    eng.replay(src, 0, 480)
    eng.replay(src, 480, 480 + 2880)
    st = eng.stats
    out[f"depth{depth}_chunk{chunk}"] = round(st["frames"] / st["seconds"], 1)
    eng.close()
src.close()
print(json.dumps(out))
