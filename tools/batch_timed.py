import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from densesurfelmapping_amd import synth, api
cam = synth.KITTI_1226
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
period = 50
scenes = [synth.Scene(seed=12345 + 17 * b, frames_per_period=period) for b in range(B)]
frames = synth.render_many([(cam, scenes[b], i) for b in range(B) for i in range(period)])
handles, plans = [], []
total = 96 + 48
for b in range(B):
    ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 20, pipeline_depth=1)
    for i in range(period):
        ff.frame_upload(i, *frames[b * period + i])
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    plans.append(api.FusionFunctions.pack_replay([t % period for t in range(total)], [t // 5 for t in range(total)], np.stack([scenes[b].pose(t) for t in range(total)])))
    handles.append(ff)
bt = api.Batch(handles)
s, r, p, n = api.Batch.pack([(pl[0][:96], pl[1][:96], pl[2][:96]) for pl in plans])
bt.replay_enqueue(s, r, p, n); bt.synchronize()
s, r, p, n = api.Batch.pack([(pl[0][96:], pl[1][96:], pl[2][96:]) for pl in plans])
st, nfr = bt.replay_timed(s, r, p, n)
ovh = bt.event_overhead_ms * 1e3
per = {k: v[0] / v[1] * 1e3 - ovh for k, v in st.items()}
print(B, "ovh", round(ovh, 2), "sum", round(sum(per.values()), 1), {k: round(v, 1) for k, v in per.items()})
