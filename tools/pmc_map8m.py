#!/usr/bin/env python
"""profiles/rNN_pmc_map_kernels_8m.{json,md} from the three passes of tools/gpu_collect_map8m.sh: per kernel the mean of the
last four dispatches (the launches on the 8 M-surfel map).  Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md
prescribes: FETCH_SIZE / WRITE_SIZE come in KB; on gfx950 FETCH_SIZE tallies a 128-byte read request as 64 bytes, so the
reads of these streaming kernels (16-byte vectors, whole lines) are the reported figure doubled; WRITE_SIZE as reported.

    python tools/pmc_map8m.py gpurun_out r06
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def last_dispatches(d, pat, n=4):
    per = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-n:]
    names = sorted({c for i in ids for c in per[i]})
    return ids, names, per


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    rec, md = {}, [f"# Map-sized kernels at 8 M surfels: memory-side counters per launch ({tag})", "",
                   "`tools/gpu_collect_map8m.sh`: `tools/map_kernels_8m.py` under `rocprofv3 --pmc <counters> --kernel-include-regex 'k_fuse_surfels|k_warp'`, one pass "
                   "per counter group, the last four launches (the ones on the 8 M-surfel map).  FETCH_SIZE / WRITE_SIZE in KB as reported; on gfx950 FETCH_SIZE "
                   "tallies a 128-byte read request as 64 bytes (MI355X_MICROARCH.md, HBM)."]
    for kern in ("k_fuse_surfels", "k_warp"):
        row = {}
        for first in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
            ids, names, per = last_dispatches(os.path.join(out_dir, f"pmc_{tag}_m8_{first}"), kern)
            md += ["", f"## {kern} -- pass {first}", "", "| dispatch | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
            for i in ids:
                md.append(f"| {i} | " + " | ".join(f"{per[i].get(c, float('nan')):.4g}" for c in names) + " |")
            for c in names:
                row[c] = sum(per[i].get(c, 0.0) for i in ids) / max(len(ids), 1)
        hit, miss = row.get("TCC_HIT_sum", 0.0), row.get("TCC_MISS_sum", 0.0)
        rec[kern] = {"FETCH_SIZE_KB_reported": round(row.get("FETCH_SIZE", 0.0)), "fetch_bytes_corrected": int(row.get("FETCH_SIZE", 0.0) * 1024 * 2),
                     "WRITE_SIZE_KB": round(row.get("WRITE_SIZE", 0.0)), "write_bytes": int(row.get("WRITE_SIZE", 0.0) * 1024),
                     "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None}
    rec["note"] = ("rocprofv3 --pmc passes over tools/map_kernels_8m.py (tools/gpu_collect_map8m.sh), mean of the last 4 launches (8 M-surfel map); FETCH_SIZE doubled "
                   "as MI355X_MICROARCH.md prescribes for gfx950 (a 128-byte request is tallied at 64), WRITE_SIZE as reported")
    json.dump(rec, open(os.path.join(out_dir, f"{tag}_pmc_map_kernels_8m.json"), "w"), indent=1)
    open(os.path.join(out_dir, f"{tag}_pmc_map_kernels_8m.md"), "w").write("\n".join(md) + "\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
