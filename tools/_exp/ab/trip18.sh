#!/bin/bash
# GPU trip 18: every GPU test of the build with decoupled batches, then the default bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 300 gpurun_out/r05_bench_default.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05_bench_default.json"))
print("FULL", d["value"], "verified", d.get("verified"), d.get("verified_timed_region"), "fullhd", d.get("fullhd_2M",{}).get("frames_per_s_by_pipeline_depth"), "sharded", d.get("sharded_replay",{}).get("value"), "streamed", d.get("streamed_input",{}).get("value"), "single", d.get("single_sequence",{}).get("frames_per_s_by_pipeline_depth"), "kitti", d.get("kitti_like",{}).get("value"), "bounded", d.get("bounded_map",{}).get("value"), "rf", d["roofline"]["frac"])
PY
