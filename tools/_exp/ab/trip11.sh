#!/bin/bash
# GPU trip 11 (round 5): the streamed-input tests of the shipped build, eight times over (a hang would show as a timeout)
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
  t0=$(date +%s)
  timeout 200 python -m pytest tests -m gpu -q -x --timeout 150 -k "streamed_input or replay_engine or sharded_replay" 2>&1 | tail -1
  echo "run $i: $(( $(date +%s) - t0 )) s, rc ${PIPESTATUS[0]}"
done
