#!/bin/bash
# GPU trip 8 (round 5): upload streams at the highest stream priority (a hardware queue pool of their own) against normal
# priority: the replay engine and the streamed legs, three alternating rounds (separate processes: stream -> queue assignment
# is made at creation)
mkdir -p gpurun_out
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for v in new noprio; do
  [ $v = new ] && cp /tmp/new.so $L || cp tools/_exp/ab/libdsm_hip_$v.so $L
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-roofline --legs sharded_replay,streamed > gpurun_out/t8_$v.$r.json 2> gpurun_out/t8_$v.$r.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/t8_$v.$r.json"))
    sr=d.get("sharded_replay",{})
    print("$v", "headline", d["value"], "sharded", sr.get("page_locked_source",{}).get("frames_per_s"), sr.get("prefetch_thread_copy",{}).get("frames_per_s"), "streamed", d.get("streamed_input",{}).get("value"), "link", d.get("streamed_input",{}).get("link_alone_GBps"))
except Exception as e:
    print("$v", "FAILED", e)
PY
done
done
cp /tmp/new.so $L
