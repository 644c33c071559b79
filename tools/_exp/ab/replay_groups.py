"""one rank of the sharded replay (replay.HipEngine) with two and with three groups of frame slots, one process per run"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: F401  (first: see __graft_entry__.py)
from densesurfelmapping_amd import replay as rp, synth
groups = int(sys.argv[1])
rp.HipEngine.GROUPS = groups
cam = synth.KITTI_1226
src = rp.SyntheticSource(4000, camera="KITTI_1226", seed=12345, prerender=True)
eng = rp.HipEngine(cam, capacity=1 << 21, pipeline_depth=24, chunk=48)
eng.replay(src, 0, 480)
eng.replay(src, 480, 480 + 2880)
st = eng.stats
print(json.dumps({"groups": groups, "frames_per_s": round(st["frames"] / st["seconds"], 1)}))
eng.close(); src.close()
