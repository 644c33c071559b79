#!/usr/bin/env python
"""Per-DISPATCH counter values of one kernel from a rocprofv3 --pmc run (pmc_stats.py averages over all dispatches, which
mixes the small-map and the large-map launches of tools/map_kernels_8m.py):

    python tools/pmc_dispatches.py <dir> <kernel substring> [last N dispatches, default 8]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, pat = sys.argv[1], sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    per = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-last:]
    names = sorted({n for i in ids for n in per[i]})
    print("| dispatch | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for i in ids:
        print(f"| {i} | " + " | ".join(f"{per[i].get(n, float('nan')):.4g}" for n in names) + " |")
    if ids:
        print("| mean | " + " | ".join(f"{sum(per[i].get(n, 0.0) for i in ids) / len(ids):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main()
