#!/bin/bash
# GPU trip 6 (round 5): the shipped build -- every GPU test, smoke, then the default bench as the driver runs it
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 > gpurun_out/t6_pytest.log
cat gpurun_out/t6_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 400 gpurun_out/r05_bench_default.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05_bench_default.json"))
print("FULL", d["value"], "verified", d.get("verified"), d.get("verified_timed_region"), "fullhd", d.get("fullhd_2M",{}).get("frames_per_s_by_pipeline_depth"), "sharded", d.get("sharded_replay",{}).get("value"), "streamed", d.get("streamed_input",{}).get("value"), "single", d.get("single_sequence",{}).get("frames_per_s_by_pipeline_depth"), "kitti", d.get("kitti_like",{}).get("value"), "rf", d["roofline"]["frac"])
PY
