#!/bin/bash
# GPU trip 3 (round 5): ceilings.  What would the two memory-side items of VERDICT r04 buy at most?  in-tree | no normal plane
# (normals free: WRONG results) | k_update_seeds at eight waves per CU (short LDS rows: WRONG results) -- headline + stage
# times, two alternating rounds; then the replay engine at several chunk sizes
mkdir -p gpurun_out
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for v in new nonormals updocc; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$v.so $L
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-dropin > gpurun_out/t3_$v.$r.json 2> gpurun_out/t3_$v.$r.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t3_$v.$r.json"))
    b=d.get("batched_kernel_us",{})
    print("$v", "headline", d["value"], "seed_points", b.get("seed_points"), "upd", b.get("update_seeds_0"), b.get("update_seeds_1"), b.get("update_seeds_2"), "sum/frame", d.get("batched_frame_kernel_sum_us_per_frame"))
except Exception as e:
    print("$v", "FAILED", e)
PY
done
done
cp /tmp/new.so $L
timeout 600 python tools/_exp/ab/replay_chunks.py 2> gpurun_out/t3_chunks.err | tee gpurun_out/t3_chunks.json
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -k "fullhd_frame_groups or replay_engine or wave_stamps or api_errors" 2>&1 | tail -5
