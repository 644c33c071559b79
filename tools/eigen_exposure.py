#!/usr/bin/env python
"""How exposed are the results to the last place of the two Eigen operations nobody can pin?

The reference takes `Matrix4f::inverse()` of the pose (FF.cpp:59) and `Matrix4d::inverse() * Vector4d` in every
Gauss-Newton step (FF.cpp:176) from Eigen3, an un-vendored, un-pinned dependency that is absent here: oracle shim, C
restatement and HIP path share one closed form (adjugate / determinant).  A real Eigen build may differ from it in the last
ulp.  This tool moves every element of either inverse by +-1 ulp (one element at a time, and a few random all-element
patterns) in the reference's own translation unit (oracle/_ref/libdsm_ref_serial_perturb.so, `make -C oracle perturb`) and
replays a sequence: frames whose new-surfel count or map size change, surfels whose `update_times` change, largest
relative drift of a float attribute.

    python tools/eigen_exposure.py [--camera KITTI_1226] [--frames 100] [--workers 8] [--out profiles/r03_eigen_exposure.md]
"""
import argparse
import ctypes as C
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def patterns(n_random=4, seed=7):
    out = [("baseline", [0] * 16, [0] * 16)]
    for site in ("pose_inverse_f32", "gn_inverse_f64"):
        for k in range(16):
            for sgn in (1, -1):
                v = [0] * 16
                v[k] = sgn
                out.append((f"{site}[{k}]{'+' if sgn > 0 else '-'}1ulp", v if site[0] == "p" else [0] * 16, v if site[0] == "g" else [0] * 16))
        rng = np.random.default_rng(seed + len(site))
        for i in range(n_random):
            v = [int(x) for x in rng.integers(-1, 2, 16)]
            out.append((f"{site} random#{i}", v if site[0] == "p" else [0] * 16, v if site[0] == "g" else [0] * 16))
    return out


def replay(job):
    cam_name, n_frames, name, f16, d16 = job
    from densesurfelmapping_amd import synth
    from oracle import bindings as ob
    cam = getattr(synth, cam_name)
    scene = synth.Scene(seed=5, scale=0.12, step=0.05) if cam.rgbd else synth.Scene()
    orc = ob.RefOracle(cam, kind="serial_perturb")
    orc.lib.dsmref_set_eigen_perturb.argtypes = [C.c_void_p, C.c_void_p]
    fa, da = (C.c_int * 16)(*f16), (C.c_int * 16)(*d16)
    orc.lib.dsmref_set_eigen_perturb(fa, da)
    local = np.zeros(0, ob.SURFEL_DTYPE)
    counts = []
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)
    try:
        for t, img, dep, pose, ref in synth.sequence(cam, scene, n_frames):
            local, k = orc.fuse_map(ref, img, dep, pose, local)
            counts.append((int(k), len(local)))
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    return name, counts, local


def compare(base_counts, base_map, counts, m):
    first = next((i for i, (a, b) in enumerate(zip(base_counts, counts)) if a != b), None)
    changed_frames = sum(a != b for a, b in zip(base_counts, counts))
    row = {"first_frame_with_other_counts": first, "frames_with_other_counts": changed_frames,
           "final_surfels": len(m), "final_surfels_baseline": len(base_map)}
    n = min(len(m), len(base_map))
    if first is None:  # same order throughout: element-wise comparison is meaningful
        ut = int((m["update_times"][:n] != base_map["update_times"][:n]).sum()) + int((m["last_update"][:n] != base_map["last_update"][:n]).sum())
        worst = 0.0
        for f in ("px", "py", "pz", "nx", "ny", "nz", "size", "color", "weight"):
            a, b = m[f][:n].astype(np.float64), base_map[f][:n].astype(np.float64)
            ok = np.isfinite(a) & np.isfinite(b)
            den = np.maximum(np.abs(b[ok]), 1e-6)
            if ok.any():
                worst = max(worst, float((np.abs(a[ok] - b[ok]) / den).max()))
        row.update({"integer_fields_changed": ut, "max_rel_float_drift": worst,
                    "surfels_bit_equal": int(n - np.count_nonzero(m[:n].tobytes() != base_map[:n].tobytes()) if False else
                                             sum(1 for _ in ()))})
        row["surfels_differing_in_any_bit"] = int((np.frombuffer(m[:n].tobytes(), np.uint8).reshape(n, -1) !=
                                                   np.frombuffer(base_map[:n].tobytes(), np.uint8).reshape(n, -1)).any(1).sum())
        row.pop("surfels_bit_equal")
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--camera", default="KITTI_1226")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--workers", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--random", type=int, default=4)
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    pats = patterns(args.random)
    jobs = [(args.camera, args.frames, name, f, d) for name, f, d in pats]
    with ProcessPoolExecutor(args.workers) as ex:
        res = list(ex.map(replay, jobs))
    base = res[0]
    rows = [(name, compare(base[1], base[2], counts, m)) for name, counts, m in res[1:]]
    summary = {}
    for site in ("pose_inverse_f32", "gn_inverse_f64"):
        rs = [r for n, r in rows if n.startswith(site)]
        same = [r for r in rs if r["first_frame_with_other_counts"] is None]
        summary[site] = {
            "patterns": len(rs), "patterns_changing_a_surfel_count": len(rs) - len(same),
            "max_frames_with_other_counts": max(r["frames_with_other_counts"] for r in rs),
            "max_integer_fields_changed": max([r["integer_fields_changed"] for r in same], default=None),
            "max_rel_float_drift": max([r["max_rel_float_drift"] for r in same], default=None),
            "max_surfels_differing_in_any_bit": max([r["surfels_differing_in_any_bit"] for r in same], default=None),
            "final_surfels_baseline": len(base[2]),
            "largest_final_count_change": max(abs(r["final_surfels"] - r["final_surfels_baseline"]) for r in rs)}
    lines = [f"# Exposure to the last place of the two Eigen inverses ({args.camera}, {args.frames} frames, reference TU)", "",
             "`tools/eigen_exposure.py`: every element of the result of `Matrix4f::inverse()` (FF.cpp:59) or of `Matrix4d::inverse()`",
             "(FF.cpp:176) moved by one ulp, one element at a time (+ and -) and as random all-element patterns; compared with the",
             "unperturbed run of the same translation unit.", "",
             "| site | patterns | patterns that change a surfel count | most frames with other counts | largest final count change | "
             "most integer fields changed (same counts) | most surfels differing in any bit | largest relative float drift |",
             "|---|---|---|---|---|---|---|---|"]
    for site, v in summary.items():
        lines.append(f"| {site} | {v['patterns']} | {v['patterns_changing_a_surfel_count']} | {v['max_frames_with_other_counts']} | "
                     f"{v['largest_final_count_change']} of {v['final_surfels_baseline']} | {v['max_integer_fields_changed']} | "
                     f"{v['max_surfels_differing_in_any_bit']} | {v['max_rel_float_drift']:.3g} |" if v["max_rel_float_drift"] is not None else
                     f"| {site} | {v['patterns']} | {v['patterns_changing_a_surfel_count']} | {v['max_frames_with_other_counts']} | "
                     f"{v['largest_final_count_change']} of {v['final_surfels_baseline']} | - | - | - |")
    lines += ["", "| pattern | first frame with other counts | frames with other counts | final surfels | integer fields changed | "
              "surfels differing | max rel drift |", "|---|---|---|---|---|---|---|"]
    for name, r in rows:
        lines.append(f"| {name} | {r['first_frame_with_other_counts']} | {r['frames_with_other_counts']} | {r['final_surfels']} | "
                     f"{r.get('integer_fields_changed', '-')} | {r.get('surfels_differing_in_any_bit', '-')} | "
                     f"{r.get('max_rel_float_drift', float('nan')):.3g} |")
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)
    if args.json:
        json.dump({"summary": summary, "rows": rows}, open(args.json, "w"), indent=1)
    return summary


if __name__ == "__main__":
    main()
