#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (csv or rocpd sqlite output) as a per-kernel table.

    python tools/kernel_stats.py <dir> [out.md]
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def rows_from(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    if rows:
        return rows
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        rows += [(n, e - s) for n, s, e in db.execute("select name, start, end from kernels")]
    return rows


def main():
    d = sys.argv[1]
    rows = rows_from(d)
    agg = defaultdict(list)
    for n, t in rows:
        agg[n].append(t)
    tot = sum(sum(v) for v in agg.values()) or 1
    lines = ["| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| {n[:70]} | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | "
                     f"{sum(v)/1e6:.3f} | {100*sum(v)/tot:.1f} |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
