#!/usr/bin/env python
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (csv output)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                agg[r["Kernel_Name"]]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    names = sorted({c for v in agg.values() for c in v})
    lines = ["| kernel | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
    for k, v in sorted(agg.items()):
        lines.append(f"| {k[:48]} | " + " | ".join(f"{sum(v[c])/max(len(v[c]),1):.4g}" if c in v else "-" for c in names) + " |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
