#!/bin/bash
# usage (GPU box, repo root): tools/ab_bench.sh <rounds> "<bench args>" name=lib.so [name=lib.so ...]
# A/B/.. of several builds of the library on ONE box, alternating; "new" = the in-tree libdsm_hip.so.  Prints frames/s of
# every run.  (Box-to-box and run-to-run levels differ by several per cent: only an alternating comparison on one box
# says anything.)
rounds=$1; bargs=$2; shift; shift
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/ab_new.so
names="new"
for kv in "$@"; do n=${kv%%=*}; cp ${kv#*=} /tmp/ab_$n.so; names="$names $n"; done
for r in $(seq $rounds); do
  for v in $names; do
    cp /tmp/ab_$v.so $L
    fps=$(python bench.py --no-cpu-baseline --no-dropin --no-roofline --no-verify $bargs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"],1))')
    echo "$v $fps"
  done
done
cp /tmp/ab_new.so $L
