#!/bin/bash
# GPU trip 23 (round 5): a pipelined handle's params on the stream that reads them first (in-tree) against the copy stream:
# parity of the pipelined forms, then single_sequence + sharded_replay, two alternating rounds
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q -x --timeout 200 -k "long_sequence_kitti_golden and drive200 or replay_engine or streamed_input and drive200 or map_grows or async_uploads or rgbd_constant or node_matches" 2>&1 | tail -3
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for v in new copystream; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$v.so $L
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-roofline --legs single_sequence,sharded_replay > gpurun_out/t23_$v.$r.json 2> gpurun_out/t23_$v.$r.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t23_$v.$r.json"))
    print("$v", "headline", d["value"], "single", d["single_sequence"]["frames_per_s_by_pipeline_depth"], "sharded", d["sharded_replay"]["page_locked_source"]["frames_per_s"], d["sharded_replay"]["prefetch_thread_copy"]["frames_per_s"])
except Exception as e:
    print("$v", "FAILED", e)
PY
done
done
cp /tmp/new.so $L
