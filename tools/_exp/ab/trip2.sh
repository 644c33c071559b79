#!/bin/bash
# GPU trip 2 (round 5): GPU tests of the new build, then A/B on one box: in-tree (fit16 + settled means + wide tiers) | the
# same without fit16 | round start (f17bec1) | round 4 -- headline and kitti_like leg, two alternating rounds; then one full
# default bench of the in-tree build
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25 > gpurun_out/t2_pytest.log
cat gpurun_out/t2_pytest.log
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for v in new nofit16 head r04; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$v.so $L
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --legs kitti_like > gpurun_out/t2_$v.$r.json 2> gpurun_out/t2_$v.$r.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t2_$v.$r.json"))
    k=d.get("kitti_like",{})
    b=d.get("batched_kernel_us",{})
    print("$v", "headline", d["value"], "kitti_like", k.get("value"), "fit", b.get("seed_fit"), "upd", b.get("update_seeds_1"), "| kitti fit", k.get("batched_kernel_us",{}).get("seed_fit"), "upd2", k.get("batched_kernel_us",{}).get("update_seeds_2"), "rf", d["roofline"]["frac"])
except Exception as e:
    print("$v", "FAILED", e)
PY
done
done
cp /tmp/new.so $L
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/t2_full.json 2> gpurun_out/t2_full.err
tail -c 600 gpurun_out/t2_full.err
python - <<PY
import json
d=json.load(open("gpurun_out/t2_full.json"))
print("FULL", d["value"], "verified", d.get("verified"), d.get("verified_timed_region"), "fullhd", d.get("fullhd_2M",{}).get("frames_per_s_by_pipeline_depth"), "sharded", d.get("sharded_replay",{}).get("page_locked_source"), d.get("sharded_replay",{}).get("prefetch_thread_copy"), "streamed", d.get("streamed_input",{}).get("value"), "single", d.get("single_sequence",{}).get("value"), "kitti", d.get("kitti_like",{}).get("value"))
PY
