#!/bin/bash
# usage (GPU box, repo root): tools/wave_vs_lane.sh <lib with kLaneBatch = 1> > gpurun_out/wave_vs_lane.md
# The per-seed stages exist in two forms with identical results: a wave per seed (launches of fewer than kLaneBatch = 8 frames)
# and a lane per seed.  This measures what the wave forms are still worth where they are used: batches of
# 1 / 2 / 4 subsequences and frame groups of 4, the shipped library (wave forms there) against a build that always takes the
# lane forms, alternating, one box.
other=$1
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/wl_shipped.so; cp $other /tmp/wl_lanes.so
echo "| handles per launch | form | frames/s (two runs) |"; echo "|---|---|---|"
for n in 1 2 4; do
  for v in shipped lanes; do
    cp /tmp/wl_$v.so $L
    r=""
    for i in 1 2; do
      f=$(python bench.py --mode batched --streams $n --batches 1 --steps 6 --warmup 2 --frames-per-step 32 --no-cpu-baseline --no-dropin --no-roofline --no-verify --no-rccl-world1 2>/dev/null | python -c 'import sys,json; print(round(json.loads(sys.stdin.readline())["value"]))')
      r="$r $f"
    done
    echo "| $n | $([ $v = shipped ] && echo 'wave per seed (shipped)' || echo 'lane per seed') |$r |"
  done
done
for v in shipped lanes; do
  cp /tmp/wl_$v.so $L
  python bench.py --steps 6 --warmup 2 --legs single_sequence --no-cpu-baseline --no-roofline --no-verify --no-rccl-world1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('| one sequence, frame groups: depth 16 = groups of 4, 24 / 32 = groups of 8 (lane forms either way) | $v |', d['single_sequence']['frames_per_s_by_pipeline_depth'], '|')"
done
cp /tmp/wl_shipped.so $L
