#!/bin/bash
# GPU trip 14 (round 5): a batch's params on the batch stream, handle streams ordered behind their batch lazily (in-tree)
# against the build before (params and markers on the handles' own streams): batch parity subset, then the headline alone,
# four alternating rounds
mkdir -p gpurun_out
timeout 420 python -m pytest tests -m gpu -q -x --timeout 200 -k "golden_batched or four_batches or handles_and_batches or streamed_input or batched_large or map_grows or bench_ranks or async_uploads or quiet_scene" 2>&1 | tail -4
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2 3 4; do
for v in new oldparams; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$v.so $L
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-roofline --no-dropin > gpurun_out/t14_$v.$r.json 2> gpurun_out/t14_$v.$r.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t14_$v.$r.json"))
    print("$v", "headline", d["value"], "enqueue_s", d["config"]["host_enqueue_seconds"])
except Exception as e:
    print("$v", "FAILED", e)
PY
done
done
cp /tmp/new.so $L
