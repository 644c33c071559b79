#!/bin/bash
# usage (GPU box, repo root): tools/bench_variance.sh <runs> <out.md>
# the default bench (headline only) <runs> times back to back, separate processes, one box -> a markdown table
n=${1:-10}; out=${2:-gpurun_out/bench_variance.md}
echo "| run | frames/s | shader clock GHz (median, min-max) | device |" > $out
echo "|---|---|---|---|" >> $out
for i in $(seq $n); do
  python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-dropin --no-roofline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
c = d.get('clocks') or {}
v = d.get('valu_issue') or {}
mm = c.get('shader_clock_ghz_min_max') or [None, None]
print(f\"| $i | {d['value']:.0f} | {v.get('shader_clock_ghz')} ({mm[0]}-{mm[1]}) | {c.get('sysfs_device')} |\")" >> $out
done
cat $out
