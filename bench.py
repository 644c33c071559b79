#!/usr/bin/env python
"""Benchmark of the per-frame surfel-fusion hot path (BASELINE.json: depth frames fused/sec @ 1226x370).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE configs[1] -- synthetic KITTI-shaped replay, 1226x370, one frame
= SLIC superpixels + normals/plane fit + projective fuse + new surfels + map compaction
(FusionFunctions::fuse_initialize_map + SurfelMap::fuse_map), inputs resident in HBM.  Each rank owns
one GPU and replays B independent subsequences on B handles (streams); one STEP advances every
subsequence of the rank by one frame, so a step fuses B frames per GPU.  There is no collective on the
data path (weak scaling: work per GPU is fixed); the final clouds are merged once, outside the timed
region, with an RCCL all-gather.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def stage_alg_bytes(stage, cam, n_seed, m_avg, k_avg):
    """Algorithmic bytes one launch of `stage` must move: planes it has to read or
    write once, nothing for re-reads of overlapping windows or intermediates."""
    n = cam.width * cam.height
    base = stage.rstrip("_012")
    if base == "assign":
        first = stage.endswith("_0")
        return n * (1 + 4) + n * 4 + (0 if first else n * 4) + n_seed * 24
    if base == "apply":
        return n * 4 * 3
    if base == "update_seeds":
        return n * (4 + 1 + 4) + n_seed * 32
    if base == "seed_points":  # labels + depth in, per-seed state in (the centred points handed to the fit are an intermediate)
        return n * (4 + 4) + n_seed * 16
    if base == "seed_fit":  # seed table + prepared surfel out
        return n_seed * (60 + 44 + 2)
    if base == "fuse_surfels":
        return m_avg * 88
    if base == "new_surfels":
        return n_seed * 60 + k_avg * 44
    if base == "compact":
        return k_avg * 88
    if base == "hole_scan":
        return m_avg / 8
    return n_seed * 16


def pmc_traffic(stage):
    """HBM-side bytes per launch of `stage` from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs, tools/gpu_pmc.sh; profiles/r01_pmc_traffic.json records
    them with the calibration used).  None when no measurement is on file."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    kernels = json.load(open(path)).get("kernels", {})
    rec = kernels.get(stage) or kernels.get(stage.rstrip("012").rstrip("_"))
    return rec["hbm_bytes_per_launch"] if rec else None


def cpu_baseline(cam, scene, synth, n_warm=20, n_timed=300, budget_s=25.0):
    """The reference's own fusion_functions.cpp (oracle/_ref, real 10-thread schedule) if its prebuilt
    library is present, else our C restatement (1 thread), on a bounded sample of the same workload:
    BASELINE.md §3 -- 20 warm-up frames, then up to 300 timed frames (bounded to ~25 s), whole-sample rate
    plus median / p10 / p90 of the per-frame rates."""
    from oracle import bindings as ob
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    if ob.have_ref("threads"):
        orc, kind, cores = ob.RefOracle(cam, kind="threads"), "reference", min(10, os.cpu_count() or 1)
    else:
        orc, kind, cores = ob.PortOracle(cam), "port", 1
    local = np.zeros(0, ob.SURFEL_DTYPE)
    per_frame = []
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the reference prints timers on every frame
    try:
        t_start = time.perf_counter()
        for t, img, dep, pose, ref in synth.sequence(cam, scene, n_warm + n_timed):
            t0 = time.perf_counter()
            local, _ = orc.fuse_map(ref, img, dep, pose, local)
            if t >= n_warm:
                per_frame.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s and len(per_frame) >= 20:
                break
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    pf = np.array(per_frame)
    rates = 1.0 / pf
    return {"value": round(len(pf) / pf.sum(), 2), "unit": "frames/s", "cores": cores, "kind": kind,
            "host_cpus": os.cpu_count(), "median": round(float(np.median(rates)), 2),
            "p10": round(float(np.percentile(rates, 10)), 2), "p90": round(float(np.percentile(rates, 90)), 2),
            "final_map_surfels": int(len(local)),
            "sample": f"frames {n_warm}..{n_warm + len(pf) - 1} of the same synthetic 1226x370 sequence after {n_warm} warm-up "
                      f"frames, fuse_initialize_map + compaction per frame, "
                      f"{'10 std::threads per stage as in the reference' if kind == 'reference' else 'scalar C restatement'}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSM_BENCH_STREAMS", "8")),
                    help="independent subsequences (handles/streams) per GPU")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("DSM_BENCH_HOST_THREADS", "4")),
                    help="host threads enqueueing graph replays (each drives streams/threads handles)")
    ap.add_argument("--pipeline-depth", type=int, default=int(os.environ.get("DSM_BENCH_PIPELINE_DEPTH", "0")),
                    help="frames of one subsequence whose superpixel stages may be in flight (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    args = ap.parse_args()

    import torch
    from densesurfelmapping_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DSM_BENCH_BACKEND=gloo + DSM_BENCH_ONE_DEVICE=1 run the multi-rank logic on a single GPU (tests only):
    # every rank uses cuda:0 and the collectives run on CPU tensors.
    backend = os.environ.get("DSM_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("DSM_BENCH_ONE_DEVICE", "0") == "1"
    device = 0 if (world == 1 or one_device) else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    coll_dev = f"cuda:{device}" if backend == "nccl" else "cpu"
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    cam = synth.KITTI_1226
    B, K, W = args.streams, args.steps, args.warmup
    period = 50
    n_seed = (cam.width // 8) * (cam.height // 8)

    # one scene period of frames per rank (seed differs per rank); the rank's B subsequences replay
    # the same images into B independent maps
    scene = synth.Scene(seed=12345 + 1000 * rank, frames_per_period=period)
    rendered = [synth.render(cam, scene, i)[:2] for i in range(period)]
    handles, plans = [], []
    for b in range(B):
        ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=period, surfel_capacity=1 << 21,
                                             pipeline_depth=args.pipeline_depth or 1)  # B subsequences already fill the queues
        for i, (img, dep) in enumerate(rendered):
            ff.frame_upload(i, img, dep)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        total = W + K
        slots = [t % period for t in range(total)]
        refs = [t // 5 for t in range(total)]
        poses = np.stack([scene.pose(t) for t in range(total)])
        plans.append(api.FusionFunctions.pack_replay(slots, refs, poses))
        handles.append(ff)

    # Enqueue: every handle has its own stream; T host threads each drive B/T handles (the C ABI is
    # thread-safe per handle and ctypes drops the GIL during the call), chunk by chunk so that all
    # subsequences advance together.
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(args.host_threads, B))
    pool = ThreadPoolExecutor(n_thr)
    enqueue_s = [0.0]

    def drive(group, lo, hi, chunk):
        for c0 in range(lo, hi, chunk):
            c1 = min(hi, c0 + chunk)
            for b in group:
                s, r, p = plans[b]
                handles[b].replay_enqueue(s[c0:c1], r[c0:c1], p[c0:c1])

    def run(lo, hi, chunk=64):
        t_e = time.perf_counter()
        groups = [list(range(t, B, n_thr)) for t in range(n_thr)]
        list(pool.map(lambda g: drive(g, lo, hi, chunk), groups))
        enqueue_s[0] = time.perf_counter() - t_e

    def sync_all():
        for ff in handles:
            ff.synchronize()
        torch.cuda.synchronize()

    run(0, W)
    sync_all()
    m_start = [ff.map_size() for ff in handles]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, W + K)
    sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    m_end = [ff.map_size() for ff in handles]

    # merge of the final clouds (outside the timed region): RCCL all-gather over xGMI
    merged_total = sum(m_end)
    if world > 1:
        from densesurfelmapping_amd.replay import merge_clouds
        clouds = []
        for ff, m in zip(handles, m_end):
            buf = torch.empty(m * 44, dtype=torch.uint8, device=f"cuda:{device}")
            ff.map_copy_to_device(buf.data_ptr(), m)
            clouds.append(buf)
        merged, counts = merge_clouds(torch.cat(clouds).to(coll_dev))
        merged_total = int(sum(counts))

    for ff in handles:  # the extra measurements below run alone on the GPU
        ff.close()
    handles = []
    frames_total = world * B * K
    fps = frames_total / dt
    m_avg = float(np.mean([(a + b) / 2 for a, b in zip(m_start, m_end)]))
    k_avg = 1400.0
    n = cam.width * cam.height
    b_alg_frame = 9 * n + 60 * n_seed + 88 * m_avg + 44 * k_avg

    out = {
        "metric": "depth frames fused/sec @ KITTI 1226x370",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: synthetic KITTI-shaped replay 1226x370, full superpixel+normal+"
                               "fuse+compaction HIP path, frames and map resident in HBM",
                   "subsequences_per_gpu": B, "frames_per_step_per_gpu": B, "host_enqueue_threads": n_thr,
                   "pipeline_depth": args.pipeline_depth or 1,
                   "host_enqueue_seconds": round(enqueue_s[0], 4), "timed_seconds": round(dt, 4), "scene_period_frames": period,
                   "mean_live_surfels": round(m_avg), "final_surfels_all_ranks": merged_total,
                   "parallelism": f"{world} GPU x {B} independent subsequences, all-gather of final cloud only"},
        "e2e_algorithmic_GBps": round(fps * b_alg_frame / 1e9, 2),
        "e2e_hbm_frac": round(fps * b_alg_frame / 1e9 / (HBM_PEAK_GBS * world), 5),
    }

    if rank == 0 and not args.no_roofline:
        # per-kernel durations, measured live with HIP events on the handle's own stream (eager replay
        # of the same workload on a fresh handle; a long delay kernel in front of each frame keeps the
        # host launch latency out of the intervals)
        ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=period, surfel_capacity=1 << 21)
        for i, (img, dep) in enumerate(rendered):
            ff.frame_upload(i, img, dep)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        nt = min(W + K, 120)
        s, r, p = plans[0][0][:nt], plans[0][1][:nt], plans[0][2][:nt]
        ff.replay_enqueue(s[:20], r[:20], p[:20])
        ff.synchronize()
        m0 = ff.map_size()
        stages, nfr = ff.replay_timed(s[20:], r[20:], p[20:])
        m1 = ff.map_size()
        mt = (m0 + m1) / 2
        ovh = ff.event_overhead_ms * 1e3  # an empty event-to-event interval, subtracted from every stage
        per = {k: max(v[0] / max(v[1], 1) * 1e3 - ovh, 0.0) for k, v in stages.items()}  # us per launch
        dom = max(per, key=per.get)
        alg = stage_alg_bytes(dom, cam, n_seed, mt, k_avg)
        achieved = alg / (per[dom] * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom),
                           "event_overhead_us": round(ovh, 2),
                           "alg_bytes_per_launch": int(alg), "avg_launch_us": round(per[dom], 2)}
        out["kernel_us"] = {k: round(v, 2) for k, v in per.items()}
        out["frame_kernel_sum_us"] = round(sum(per.values()), 1)
        ksum = sum(per.values()) * 1e-6
        b_alg_t = 9 * n + 60 * n_seed + 88 * mt + 44 * k_avg
        out["kernel_time_weighted_hbm_frac"] = round(b_alg_t / ksum / 1e9 / HBM_PEAK_GBS, 5)
        ff.close()

    if rank == 0 and world == 1 and not args.no_dropin:
        # ONE sequence (BASELINE configs[1] as the reference would replay it): frames are strictly ordered, but
        # only fuse + tail need the map -- the superpixel stages of up to 8 frames run ahead on their own streams
        ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=period, surfel_capacity=1 << 21, pipeline_depth=8)
        for i, (img, dep) in enumerate(rendered):
            ff.frame_upload(i, img, dep)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        s1, r1, p1 = plans[0]
        ff.replay_enqueue(s1[:W], r1[:W], p1[:W])
        ff.synchronize()
        t_s = time.perf_counter()
        ff.replay_enqueue(s1[W:W + K], r1[W:W + K], p1[W:W + K])
        ff.synchronize()
        out["single_sequence"] = {"value": round(K / (time.perf_counter() - t_s), 1), "unit": "frames/s", "pipeline_depth": 8,
                                  "note": "one subsequence, one handle: superpixel stages of 8 consecutive frames in flight, "
                                          "fuse + compaction strictly in frame order; same results as the serial order"}
        ff.close()

    if rank == 0 and world == 1 and not args.no_dropin:
        # the synchronous drop-in call (host buffers in and out over PCIe every frame), for DESIGN.md;
        # never the headline value
        ff = api.FusionFunctions.from_camera(cam, device=device, surfel_capacity=1 << 21)
        local = np.zeros(0, api.SURFEL_DTYPE)
        for t in range(40):
            local, _ = ff.fuse_map(t // 5, rendered[t % period][0], rendered[t % period][1], scene.pose(t), local)
        t_d = time.perf_counter()
        for t in range(40, 70):
            local, _ = ff.fuse_map(t // 5, rendered[t % period][0], rendered[t % period][1], scene.pose(t), local)
        out["dropin_pcie_inclusive"] = {"value": round(30 / (time.perf_counter() - t_d), 1), "unit": "frames/s",
                                        "map_surfels": int(len(local)),
                                        "note": "dsm_fuse_map with host buffers: frame H2D + map H2D/D2H + sync per frame"}
        ff.close()

    if rank == 0 and world == 1 and not args.no_dropin:
        # SURVEY.md §8(f) row 1, BASELINE configs[4] size: loop-closure deformation of a 2 M-surfel resident map
        # (surfel_map.cpp:750-789), the one purely HBM-bound stage: 88 B per surfel
        n_w = 2_000_000
        wm = np.zeros(n_w, api.SURFEL_DTYPE)
        wm["px"] = np.arange(n_w, dtype=np.float32) * 1e-3
        wm["nz"] = 1.0
        wm["update_times"] = 3
        ff = api.FusionFunctions.from_camera(synth.TINY, device=device, surfel_capacity=n_w + 64)
        ff.map_upload(wm)
        wp = np.eye(4, dtype=np.float32)
        wp[:3, 3] = (0.01, -0.02, 0.005)
        for _ in range(5):
            ff.map_warp(wp)
        ff.synchronize()
        t_w = time.perf_counter()
        for _ in range(100):
            ff.map_warp(wp)
        ff.synchronize()
        dt_w = (time.perf_counter() - t_w) / 100
        out["map_warp_2M"] = {"us_per_call": round(dt_w * 1e6, 1), "achieved_GBps": round(n_w * 88 / dt_w / 1e9, 1),
                              "hbm_frac": round(n_w * 88 / dt_w / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "dsm_map_warp incl. its per-call host sync; 176 MB working set sits in the 256 MB Infinity Cache "
                                      "(kernel alone 26.8 us in profiles/r01_kernel_trace_warp_2M.md); 8 M surfels (704 MB): 5.1 TB/s"}
        ff.close()

    if rank == 0 and world == 1 and not args.no_dropin:
        # BASELINE configs[3]: live callback, 640x480 RGB-D constants, one frame at a time: host frame in
        # (H2D), resident map, one hipGraph replay, wait -- the latency the 30 Hz node would see per frame
        cam_v = synth.VGA_RGBD
        scene_v = synth.Scene(seed=5, scale=0.12, step=0.05, frames_per_period=30)
        frames_v = [synth.render(cam_v, scene_v, i)[:2] for i in range(30)]
        ff = api.FusionFunctions.from_camera(cam_v, device=device, frame_slots=2, surfel_capacity=1 << 20)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        lat = []
        for t in range(150):
            img_v, dep_v = frames_v[t % 30]
            t_l = time.perf_counter()
            ff.frame_upload(t & 1, img_v, dep_v)
            ff.fuse_frame_resident(t & 1, t // 5, scene_v.pose(t))
            ff.synchronize()
            if t >= 30:
                lat.append(time.perf_counter() - t_l)
        lat = np.array(lat) * 1e3
        out["live_callback_640x480"] = {"latency_ms_p50": round(float(np.median(lat)), 3), "latency_ms_p99": round(float(np.percentile(lat, 99)), 3),
                                        "map_surfels": ff.map_size(),
                                        "note": "per frame: pageable host image+depth H2D, fuse (one graph replay), stream sync; RGB-D constant set"}
        ff.close()

    if rank == 0 and world == 1 and not args.no_dropin:
        # the whole node through its message callbacks (stamp matching, pose graph, active /
        # inactive sets in HBM, loop closure at the start of the second lap), host-inclusive: every frame is copied into
        # the node's page-locked pool and uploaded
        from densesurfelmapping_amd import surfel_map
        n_node = 3 * period
        events = list(synth.node_messages(cam, scene, n_node, lap=period, frames={i: f for i, f in enumerate(rendered)}))
        node = surfel_map.SurfelMap(cam, drift_free_poses=10, device=device, surfel_capacity=1 << 21)
        for ev in events[:60]:
            node.feed(ev)
        node.local_surfels()
        t_n = time.perf_counter()
        for ev in events[60:]:
            node.feed(ev)
        n_act = len(node.local_surfels())
        dt_n = time.perf_counter() - t_n
        out["node_callbacks"] = {"value": round((n_node - 20) / dt_n, 1), "unit": "frames/s", "frames": n_node - 20,
                                 "keyframes": node.pose_count, "active_surfels": n_act, "inactive_surfels": len(node.inactive_cloud()),
                                 "note": "image_input + depth_input + orb_results_input per frame (drift_free_poses 10, keyframe every 5, "
                                         "loop closure with warp of active and inactive surfels at frame %d); host-inclusive" % period}
        node.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cam, synth.Scene(seed=12345, frames_per_period=period), synth)

    if rank == 0:
        print(json.dumps(out))
    for ff in handles:
        ff.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
