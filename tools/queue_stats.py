#!/usr/bin/env python
"""Which hardware queues the batched frame kernels of a traced bench run went to (rocprofv3 --kernel-trace csv):

    python tools/queue_stats.py gpurun_out/prof_<tag>

prints {queue id: (launches, busy ms, stream ids)} for the kernels launched over 8 handles.  Four queues with equal shares
are what the reserved batch streams (dsm_api.hip, BatchStreamPool) are for; two batches on one queue run one after the other."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
q = defaultdict(lambda: [0, 0.0]); st = defaultdict(set)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsm::k_" in r["Kernel_Name"] and "repack" not in r["Kernel_Name"] and "Grid_Size_Z" in r and int(r["Grid_Size_Z"]) == 8:
            q[r["Queue_Id"]][0] += 1; q[r["Queue_Id"]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); st[r["Queue_Id"]].add(r["Stream_Id"])
print({k: (v[0], round(v[1] / 1e6, 1), sorted(st[k])) for k, v in sorted(q.items())})
