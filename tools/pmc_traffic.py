#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass).

    python tools/pmc_traffic.py gpurun_out/pmc_<fetch-tag> gpurun_out/pmc_<write-tag> profiles/r02_pmc_traffic.json "<command>"

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters come out in KB; on gfx950
FETCH_SIZE under-reports reads by a width-dependent factor and has to be calibrated on a kernel whose read bytes are
known in the same access pattern.  Calibration kernel: k_assign<true> reads exactly the image (1 B/pixel) and depth
(4 B/pixel) planes and writes the label plane (2 B/pixel since round 4's 16-bit labels); the factors found are recorded
in the output.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

W, H = 1226, 370


def per_kernel(d, counter):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def short(name):
    n = name.replace("void ", "").replace("dsm::", "")
    n = n.split("(")[0]
    return n


def single_name(k):
    """the frame kernels carry a trailing BATCH template argument; the one-subsequence instantiation (false) is
    recorded under the plain name: k_update_seeds<true, false> -> k_update_seeds<true>, k_seed_fit<false> -> k_seed_fit"""
    if not k.endswith(">") or k.startswith("k_repack_rows"):
        return k
    base, args = k[:-1].split("<", 1)
    if base == "k_seed_fit":  # <BATCH, tier>: the one-subsequence launch has a single tier
        return base
    args = [a.strip() for a in args.split(",")]
    if base == "k_assign":  # <FIRST, BATCH, pixels per thread>
        args = args[:1]
    elif base == "k_update_seeds_wave":  # the one-subsequence form of the update_seeds stage
        base, args = "k_update_seeds", []
    elif args[-1] == "false":
        args = args[:-1]
    return base + ("<" + ", ".join(args) + ">" if args else "")


def main():
    fetch_dir, write_dir, out_path, cmd = sys.argv[1:5]
    # (launches batched over several subsequences: the calibration kernel's known bytes scale with the batch)
    nb = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    name = (lambda k: short(k)) if nb > 1 else (lambda k: single_name(short(k)))
    fetch = {name(k): v for k, v in per_kernel(fetch_dir, "FETCH_SIZE").items()}
    write = {name(k): v for k, v in per_kernel(write_dir, "WRITE_SIZE").items()}
    n = W * H
    cal_name = "k_assign<true, true, 4>" if nb > 1 else "k_assign<true>"
    n *= nb
    cal_f = (5 * n / 1024.0) / fetch[cal_name]
    cal_w = (2 * n / 1024.0) / write[cal_name]
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `{cmd}` (tools/gpu_pmc.sh), 1226x370",
           "units": "FETCH_SIZE / WRITE_SIZE are reported in KB",
           "subsequences_per_launch": nb,
           "calibration": {"kernel": cal_name, "known_read_bytes": 5 * n, "known_write_bytes": 2 * n,
                           "fetch_factor_found": round(cal_f, 3), "write_factor_found": round(cal_w, 3),
                           "applied": "FETCH_SIZE x fetch_factor_found; WRITE_SIZE as reported when its factor is within 5 % of 1"},
           "kernels": {}}
    wf = cal_w if abs(cal_w - 1) > 0.05 else 1.0
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KB": round(f, 2), "WRITE_SIZE_KB": round(w, 2),
                             "hbm_bytes_per_launch": int((f * cal_f + w * wf) * 1024)}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out["calibration"]))
    for k, v in out["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
