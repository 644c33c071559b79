#!/bin/bash
# GPU trip 7 (round 5): uploads ordered behind the newest enqueue that reads their slots (three slot groups in replay.py)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 -k "replay_engine or sharded_replay or streamed_input or bench_ranks or handles_and_batches or rgbd_constant" 2>&1 | tail -6
timeout 600 python tools/_exp/ab/replay_chunks.py 2> gpurun_out/t7_chunks.err | tee gpurun_out/t7_chunks.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-roofline --legs sharded_replay,streamed > gpurun_out/t7_bench.json 2> gpurun_out/t7_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/t7_bench.json"))
print("headline", d["value"], "sharded", d.get("sharded_replay"), "streamed", d.get("streamed_input",{}).get("value"), d.get("streamed_input",{}).get("link_alone_GBps"))
PY
