#!/bin/bash
# GPU trip 1 (round 5): GPU tests, then A/B r04 vs new on the headline and on the kitti_like leg
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t1_pytest.log
cat gpurun_out/t1_pytest.log
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for v in new r04; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_r04.so $L
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --legs kitti_like > gpurun_out/t1_$v.$r.json 2> gpurun_out/t1_$v.$r.err
  python - <<PY
import json
d=json.load(open("gpurun_out/t1_$v.$r.json"))
k=d.get("kitti_like",{})
print("$v", "headline", d["value"], "kitti_like", k.get("value"), "assign", k.get("assign_us_per_launch"), "default assign", k.get("assign_us_per_launch_default_scenes"), "rf", d["roofline"]["frac"])
PY
done
done
cp /tmp/new.so $L
