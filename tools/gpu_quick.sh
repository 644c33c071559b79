#!/bin/bash
# usage (GPU box, repo root): tools/gpu_quick.sh <tag> [pytest -k expr]
# quick iteration loop: a parity subset, then a short bench with per-kernel event times
tag=$1; kexpr=${2:-"tiny_sequence or resident_replay_kitti or state_level or quiet_scene or golden_fixture"}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$kexpr" 2>&1 | tail -6 > gpurun_out/quick_$tag.test
cat gpurun_out/quick_$tag.test
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > gpurun_out/quick_$tag.json 2> gpurun_out/quick_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/quick_$tag.json"))
print("fps", d["value"], "sum_us", d["frame_kernel_sum_us"])
print({k: v for k, v in d["kernel_us"].items()})
PY
