#!/bin/bash
# GPU trip 10 (round 5): the inliers' normals by k_seed_normals (64 lists of like length per wave; no plane, no k_pixel_normals)
# + packed products in the fit's chains (in-tree) | packed products only | the build before: parity subset, then headline +
# kitti_like, two alternating rounds
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 -k "golden_batched or four_batches or stereo or edge_inputs or other_baseline_sizes or fullhd_frame_groups or streamed_input or batched_large or fit_tiers or tiny_sequence" 2>&1 | tail -8
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for v in new pk base; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$v.so $L
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --legs kitti_like > gpurun_out/t10_$v.$r.json 2> gpurun_out/t10_$v.$r.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t10_$v.$r.json"))
    b=d.get("batched_kernel_us",{}); k=d.get("kitti_like",{})
    print("$v", "headline", d["value"], "kitti_like", k.get("value"), "seed_points", b.get("seed_points"), "fit", b.get("seed_fit"), "| kitti seed_points", k.get("batched_kernel_us",{}).get("seed_points"), "sum/frame", d.get("batched_frame_kernel_sum_us_per_frame"))
except Exception as e:
    print("$v", "FAILED", e)
PY
done
done
cp /tmp/new.so $L
