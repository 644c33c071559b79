#!/bin/bash
# GPU trip 20 (round 5): more than four batches in flight?  batches 5..8 on four more streams at the highest stream priority
# (a second pool of hardware queues) against the shipped four batches
mkdir -p gpurun_out
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
run() { # tag lib args...
  tag=$1; lib=$2; shift 2
  [ $lib = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$lib.so $L
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-roofline --no-dropin "$@" > gpurun_out/t20_$tag.json 2> gpurun_out/t20_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t20_$tag.json"))
    print("$tag", d["value"], d["config"]["launch_mode"][:40], d["config"]["subsequences_per_gpu"])
except Exception as e:
    print("$tag", "FAILED", e)
PY
}
run base4x32.1 new
run prio8x16.1 prio8 --batches 8 --streams 128
run prio8x32.1 prio8 --batches 8 --streams 256
run norm8x16.1 new --batches 8 --streams 128
run base4x32.2 new
run prio8x16.2 prio8 --batches 8 --streams 128
run prio8x32.2 prio8 --batches 8 --streams 256
run base4x64.1 new --batches 4 --streams 256
cp /tmp/new.so $L
