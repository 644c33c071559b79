#!/usr/bin/env python
"""Benchmark of the per-frame surfel-fusion hot path (BASELINE.json: depth frames fused/sec @ 1226x370).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B] [--frames-per-step F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE configs[1] -- synthetic KITTI-shaped replay, 1226x370, one frame
= SLIC superpixels + normals/plane fit + projective fuse + new surfels + map compaction
(FusionFunctions::fuse_initialize_map + SurfelMap::fuse_map), inputs resident in HBM.  Each rank owns
one GPU and replays B independent subsequences (B handles, B different synthetic scenes); one STEP advances
every subsequence of the rank by F frames, so a step fuses B*F frames per GPU.  There is no collective on the
data path (weak scaling: work per GPU is fixed); the final clouds are merged once, outside the timed
region, with an RCCL all-gather.

The CPU baseline replays subsequence 0 of rank 0 from its first frame and is timed over the SAME frame
indices, against the same map state, as the GPU's timed region (a bounded prefix of it when the region is long).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def stage_alg_bytes(stage, n, n_seed, m_avg, k_avg):
    """Algorithmic bytes one launch of `stage` must move: planes it has to read or
    write once, nothing for re-reads of overlapping windows or intermediates handed between kernels."""
    base = stage.rstrip("_012")
    if base == "assign":
        first = stage.endswith("_0")
        return n * (1 + 4) + n * 4 + (0 if first else n * 4) + n_seed * 24
    if base == "update_seeds":
        return n * (4 + 1 + 4) + n_seed * 32
    if base == "seed_points":  # labels + depth in, per-seed state in (the centred points handed to the fit are an intermediate)
        return n * (4 + 4) + n_seed * 16
    if base == "seed_fit":  # seed table + prepared surfel out
        return n_seed * (60 + 44 + 2)
    if base == "fuse_surfels":
        return m_avg * 88
    if base == "frame_tail":
        return n_seed * 2 + k_avg * (4 + 88) + m_avg / 8
    return n_seed * 16


# stage (position in the frame's launch sequence) -> kernel function(s), as rocprofv3 names them.  Launches batched over
# eight or more subsequences use the lane-per-seed forms of the per-seed stages (two or three kernels per stage), a single
# subsequence the wave-per-seed forms.
KERNEL_OF_STAGE = {
    "init_seeds": "k_init_seeds", "assign_0": "k_assign<true>", "assign_1": "k_assign<false>", "assign_2": "k_assign<false>",
    "resolve_1": "k_resolve", "resolve_2": "k_resolve",
    "update_seeds_0": "k_update_seeds", "update_seeds_1": "k_update_seeds", "update_seeds_2": "k_update_seeds",
    "commit_seeds_0": "k_commit_seeds", "commit_seeds_1": "k_commit_seeds", "commit_seeds_2": "k_commit_seeds",
    "seed_points": "k_seed_points", "seed_fit": "k_seed_fit", "fuse_surfels": "k_fuse_surfels", "frame_tail": "k_frame_tail",
}
BATCHED_KERNELS_OF_STAGE = {
    "assign_0": ["k_assign<true, true, 4>"], "assign_1": ["k_assign<false, true, 4>"], "assign_2": ["k_assign<false, true, 4>"],
    "update_seeds_0": ["k_update_seeds<true>", "k_update_seeds_rest<true>"],
    "update_seeds_1": ["k_update_seeds<true>", "k_update_seeds_rest<true>"],
    "update_seeds_2": ["k_update_seeds<true>", "k_update_seeds_rest<true>"],
    "resolve_1": ["k_resolve<true>", "k_apply_labels<true>"], "resolve_2": ["k_resolve<true>", "k_apply_labels<true>"],
    "seed_points": ["k_pixel_normals<true>", "k_seed_stats<true>"],
    "seed_fit": ["k_seed_fit<true, 1>", "k_seed_fit<true, 2>", "k_seed_finish<true>"],
}
PMC_TRAFFIC_SINGLE, PMC_TRAFFIC_BATCHED, PMC_SQ_BATCHED = "r06_pmc_traffic.json", "r06_pmc_traffic_batched.json", "r06_pmc_sq_batch32.md"
PMC_SQ_LAUNCH = 32  # subsequences per launch of the SQ pass (the launches the timed region makes)
DEFAULT_SUBSEQUENCES = 128  # batched mode: 4 batches of 32 (round 4; 32 in 4 batches of 8 until then)


def pmc_traffic(kernels, name, per_launch=None):
    """HBM-side bytes per launch of the given kernel(s) (summed) from the committed rocprofv3 --pmc passes of this round
    (FETCH_SIZE and WRITE_SIZE collected in separate runs, tools/gpu_pmc.sh; the JSON records them with the calibration
    used).  Counters cannot be read from inside the run; None when no measurement is on file (per_launch: only a record of
    launches batched over that many subsequences counts)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    if per_launch is not None and rec.get("subsequences_per_launch") != per_launch:
        return None, None
    table = rec.get("kernels", {})
    if isinstance(kernels, str):
        kernels = [kernels]
    recs = [table.get(k) for k in kernels]
    if not recs or any(r is None for r in recs):
        return None, None
    return sum(r["hbm_bytes_per_launch"] for r in recs), name


def valu_issue(fps_per_gpu, clock_ghz=None):
    """The roof that binds the batched superpixel stages: VALU instruction issue.  Wave-instructions per frame from the
    committed rocprofv3 --pmc SQ_INSTS_VALU pass over launches batched over 32 subsequences -- the launches the timed region
    makes (profiles/r06_pmc_sq_batch32.md; the counts are per launch, a frame launches the sweep kernels two or three times)
    -- against what 1 024 SIMDs issue at one wave64 instruction per 4 cycles.  clock_ghz: the shader clock sampled during the timed region (2.4 GHz assumed when it
    could not be read).  None when no pass is on file."""
    path = os.path.join(ROOT, "profiles", PMC_SQ_BATCHED)
    if not os.path.exists(path):
        return None
    per_launch = {}
    col = None
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if col is None:
            if "SQ_INSTS_VALU" in cells:
                col = cells.index("SQ_INSTS_VALU")
            continue
        if "dsm::k_" in cells[0] and col < len(cells):
            name = cells[0].split("dsm::", 1)[1].split("(")[0]
            try:
                per_launch[name] = float(cells[col])
            except ValueError:
                pass
    if not per_launch:
        return None
    twice = ("k_assign<false, true, 4>", "k_resolve<true>", "k_apply_labels<true>")
    launches = {k: 2 for k in twice}
    for k in ("k_commit_seeds<true>", "k_update_seeds<true>", "k_update_seeds_rest<true>"):
        launches[k] = 3
    per_frame = sum(v * launches.get(k, 1) for k, v in per_launch.items() if not k.startswith("k_repack")) / float(PMC_SQ_LAUNCH)
    ghz = clock_ghz or 2.4
    peak = 256 * 4 * ghz * 1e9 / 4.0
    return {"valu_wave_insts_per_frame": round(per_frame), "peak_wave_insts_per_s": peak, "shader_clock_ghz": ghz,
            "shader_clock_source": "sampled during the timed region" if clock_ghz else "assumed (no clock sample available)",
            "frames_per_s_at_peak": round(peak / per_frame, 1), "frac": round(fps_per_gpu * per_frame / peak, 4),
            "source": f"profiles/{PMC_SQ_BATCHED} (rocprofv3 --pmc SQ_INSTS_VALU, launches batched over {PMC_SQ_LAUNCH} subsequences)",
            "note": "every instruction priced at the fp32 rate; float<->double conversions issue at a quarter of it, and a wave "
                    "alone on its SIMD (the lane-per-seed kernels) issues one instruction per ~5.5 cycles, not 4"}


class ClockSampler:
    """Shader clock during the timed region: a thread reads the current sclk from sysfs (pp_dpm_sclk marks the active level
    with '*') every 20 ms.  None if the file is not there (no amdgpu sysfs in the container)."""

    def __init__(self, pci_bdf=None):
        import glob
        # the node's sysfs lists every GPU of the machine, whichever one this process was given: go by PCI address
        base = f"/sys/bus/pci/devices/{pci_bdf}" if pci_bdf and os.path.isdir(f"/sys/bus/pci/devices/{pci_bdf}") else None
        self.device = pci_bdf if base else None
        pat = base if base else "/sys/class/drm/card*/device"
        self.files = sorted(glob.glob(pat + "/pp_dpm_sclk"))
        self.mfiles = sorted(glob.glob(pat + "/pp_dpm_mclk"))
        self.pfiles = sorted(glob.glob(pat + "/hwmon/hwmon*/power1_average"))
        self.samples, self.msamples, self.psamples, self._stop, self._thr = [], [], [], False, None

    def _read(self, files=None):
        for f in (self.files if files is None else files)[:1]:
            try:
                for line in open(f):
                    if "*" in line:
                        return float(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
            except (OSError, ValueError, IndexError):
                return None
        return None

    def __enter__(self):
        if self.files:
            import threading

            def loop():
                while not self._stop:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    m = self._read(self.mfiles)
                    if m:
                        self.msamples.append(m)
                    for f in self.pfiles[:1]:
                        try:
                            self.psamples.append(float(open(f).read()) / 1e6)
                        except (OSError, ValueError):
                            pass
                    time.sleep(0.02)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._thr:
            self._thr.join()

    def ghz(self):
        return round(float(np.median(self.samples)) / 1e3, 3) if self.samples else None

    def other(self):
        """memory clock (GHz) and board power (W) over the same samples, where sysfs has them"""
        return {"sysfs_device": self.device or "first card of the node (PCI address of the HIP device not found: the clocks may be another GPU's)",
                "memory_clock_ghz": round(float(np.median(self.msamples)) / 1e3, 3) if self.msamples else None,
                "power_w_median": round(float(np.median(self.psamples)), 1) if self.psamples else None,
                "shader_clock_ghz_min_max": [round(min(self.samples) / 1e3, 3), round(max(self.samples) / 1e3, 3)] if self.samples else None}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def canon_bytes(a: np.ndarray) -> bytes:
    """the bytes of a surfel array with every NaN replaced by one canonical NaN (the reference's arithmetic does not
    define a NaN's sign or payload; everything else is compared bit for bit)"""
    a = np.ascontiguousarray(a)
    words = a.view(np.uint32).reshape(len(a), -1).copy()
    for i, name in enumerate(a.dtype.names):
        if a.dtype[name].kind == "f":
            words[np.isnan(a[name]), i] = 0x7FC00000
    return words.tobytes()


def oracle_replay_worker(spec_json):
    """`python bench.py --oracle-replay-worker '<json>'`: the CPU oracle's replay of one subsequence of the headline from an
    empty map through frame `frames` - 1 -- the CHECKER of the timed region (started beside the GPU run, never inside a timed
    interval); prints the NaN-canonical SHA-256 of the whole map."""
    import hashlib
    from densesurfelmapping_amd import synth
    from oracle.bindings import SURFEL_DTYPE as O_DTYPE, PortOracle
    spec = json.loads(spec_json)
    cam = getattr(synth, spec["camera"])
    scene = synth.Scene(seed=spec["seed"], frames_per_period=spec["period"])
    per = spec["period"]
    frames = synth.render_many([(cam, scene, i) for i in range(per)], workers=1)  # (rendered and cached by the parent already)
    orc, lo = PortOracle(cam), np.zeros(0, O_DTYPE)
    for t in range(spec["frames"]):  # (keyframe indices count from the subsequence's first frame; `phase` = where in the scene it starts)
        img, dep = frames[(t + spec["phase"]) % per]
        lo, _ = orc.fuse_map(t // 5, img, dep, scene.pose(t + spec["phase"]), lo)
    print(json.dumps({"subsequence": spec["subsequence"], "frames": spec["frames"], "surfels": int(len(lo)),
                      "sha256": hashlib.sha256(canon_bytes(lo)).hexdigest()}), flush=True)


def start_oracle_replays(specs):
    """one fresh interpreter per spec (clean environment: nothing inherited from rocprofv3 / torchrun), all at once"""
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER", "LD_PRELOAD", "OMP_", "MKL_")) or k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))}
    env["OMP_NUM_THREADS"] = "1"
    return [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-replay-worker", json.dumps(sp)], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, env=env, cwd=ROOT, text=True) for sp in specs]


def cpu_baseline(cam, scene, rendered, period, lo, hi, budget_s=25.0):
    """The reference's own fusion_functions.cpp (oracle/_ref, real 10-thread schedule) if its prebuilt library is
    present, else our C restatement (1 thread).  Replays the subsequence from frame 0 (untimed up to `lo`: that builds
    the map state the GPU's timed region starts from) and times frames [lo, hi) -- the GPU's timed frame indices --
    stopping early once `budget_s` of timed work is spent.  Also returns the measured K and M of those frames."""
    from oracle import bindings as ob
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    if ob.have_ref("threads"):
        orc, kind, cores = ob.RefOracle(cam, kind="threads"), "reference", min(10, os.cpu_count() or 1)
    else:
        orc, kind, cores = ob.PortOracle(cam), "port", 1
    local = np.zeros(0, ob.SURFEL_DTYPE)
    per_frame, news, sizes = [], [], []
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the reference prints timers on every frame
    try:
        spent = 0.0
        for t in range(hi):
            img, dep = rendered[t % period]
            t0 = time.perf_counter()
            local, k = orc.fuse_map(t // 5, img, dep, scene.pose(t), local)
            dt = time.perf_counter() - t0
            if t >= lo:
                per_frame.append(dt)
                news.append(k)
                sizes.append(len(local))
                spent += dt
                if spent > budget_s and len(per_frame) >= 20:
                    break
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    pf = np.array(per_frame)
    rates = 1.0 / pf
    return {"value": round(len(pf) / pf.sum(), 2), "unit": "frames/s", "cores": cores, "kind": kind,
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "median": round(float(np.median(rates)), 2),
            "p10": round(float(np.percentile(rates, 10)), 2), "p90": round(float(np.percentile(rates, 90)), 2),
            "mean_live_surfels": round(float(np.mean(sizes))), "mean_new_surfels": round(float(np.mean(news)), 1),
            "sample": f"frames {lo}..{lo + len(pf) - 1} of subsequence 0 (the GPU's timed region is frames {lo}..{hi - 1} of every "
                      f"subsequence), same map state: the {lo} frames before them replayed untimed; fuse_initialize_map + compaction "
                      f"per frame, {'10 std::threads per stage as in the reference' if kind == 'reference' else 'scalar C restatement'}"}


PMC_MAP_8M = os.path.join(ROOT, "profiles", "r06_pmc_map_kernels_8m.json")


def pmc_8m(kernel, us):
    """Memory-side bytes per launch of a map-sized kernel at 8 M surfels (committed rocprofv3 --pmc passes), next to the
    live duration: what the memory system moved, gathers included, as opposed to the 88 algorithmic bytes per surfel."""
    try:
        row = json.load(open(PMC_MAP_8M))[kernel]
    except (OSError, KeyError, ValueError):
        return {"traffic": None}
    moved = row["fetch_bytes_corrected"] + row["write_bytes"]
    return {"traffic": {"read_bytes": row["fetch_bytes_corrected"], "write_bytes": row["write_bytes"], "l2_hit_rate": row["l2_hit_rate"],
                        "memory_side_GBps": round(moved / us / 1e3, 1), "of_achievable_6300_GBps": round(moved / us / 1e3 / 6300.0, 3),
                        "source": "profiles/r06_pmc_map_kernels_8m.json (rocprofv3 --pmc on this round's build, FETCH_SIZE corrected x2 for gfx950)"}}


def event_timer(torch, ff):
    """HIP events on the handle's own stream (torch.cuda.Event only sees torch's current stream by default)."""
    def timed(fn, reps):
        # fetched per measurement: dsm_stream is also the call that puts the handle's stream behind a batch it advanced with
        # (include/dsm.h, ABI 4) -- a stream cached before batch calls would not be ordered behind them
        stream = torch.cuda.ExternalStream(ff.stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps  # us per call
    return timed


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: check that N devices are visible and re-run this command under
    torch.distributed.run, one rank per GPU (RCCL rendezvous on 127.0.0.1).  Returns the launcher's exit code."""
    import socket
    import subprocess
    one_device = os.environ.get("DSM_BENCH_ONE_DEVICE", "0") == "1"
    import torch
    seen = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not one_device and seen < n_gpus:
        print(f"bench.py: --gpus {n_gpus} needs {n_gpus} visible GPUs, this box shows {seen}; nothing was measured", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.run(cmd, env=env).returncode


class stdout_to_stderr:
    """fd-level: what native libraries print to stdout inside the block goes to stderr (RCCL prints a version banner when a
    communicator is created or destroyed; the contract is ONE JSON line on stdout)"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        sys.stdout.flush()
        # (C stdio is fully buffered when stdout is a file or a pipe: without this the banner would sit in libc's buffer
        # and come out on fd 1 at exit, after the JSON line)
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def close_group(dist, world):
    if dist is None:
        return
    with stdout_to_stderr():
        if world > 1:
            dist.barrier()  # (rank 0 may still have been checking its maps against the oracle)
        dist.destroy_process_group()


def open_group(args, world, rank, local_rank):
    """Device of this rank and the process group of the final merge -> (torch, device, dist or None, device of the collectives'
    tensors, why there is no group).  DSM_BENCH_BACKEND=gloo + DSM_BENCH_ONE_DEVICE=1 run the multi-rank logic on a single GPU
    (tests only): every rank uses cuda:0 and the collectives run on CPU tensors."""
    import torch
    backend = os.environ.get("DSM_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("DSM_BENCH_ONE_DEVICE", "0") == "1"
    device = 0 if (world == 1 or one_device) else local_rank
    torch.cuda.set_device(device)
    # At --gpus 1 too (a group of one on a loopback port): the merge of the final clouds then goes through RCCL on the one GPU
    # -- ncclCommInitRank, the all-gather of the counts, the all-gather of the padded cloud -- exactly as it does on a node;
    # nothing of it is inside the timed region.
    dist, group_error = None, None
    if world > 1 or not args.no_rccl_world1:
        from densesurfelmapping_amd.replay import init_collective
        try:
            with stdout_to_stderr():
                dist = init_collective(backend, world, rank, device)
        except Exception as e:  # noqa: BLE001 -- a group of one is an extra; a world of several cannot do without
            if world > 1:
                raise
            group_error = repr(e)[:300]
    coll_dev = f"cuda:{device}" if backend == "nccl" else "cpu"
    if world > 1:
        if not one_device and torch.cuda.device_count() < world:
            sys.exit(f"bench.py: rank {rank} sees {torch.cuda.device_count()} GPUs for a world of {world}")
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
    return torch, device, dist, coll_dev, group_error


def sharded_workload(args, world, rank, device, dist, coll_dev, group_error):
    """`--workload sharded`: BASELINE configs[2] as the north star states it.  One synthetic KITTI-shaped sequence of
    world x (W + K) x F frames is cut into `world` contiguous subsequences (replay.shard_subsequences); rank r streams ITS
    subsequence from page-locked host memory through ONE handle (replay.HipEngine: pipeline depth 24, every group of eight frames
    uploaded on the stream that runs its superpixel stages, right in front of them: dsm_replay_enqueue_host), keyframe indices
    restarting at its first frame, the map resident; the final clouds are merged by one all-gather of the counts and one of the
    padded clouds (RCCL).  A step = F frames of every rank's subsequence; the warm-up steps are replayed first (same engine,
    same map), the K timed steps between two barriers as ONE streamed replay.  Every rank's final map is checked against the CPU
    oracle's replay of its own subsequence, started beside the GPU work."""
    import hashlib
    import torch
    from densesurfelmapping_amd import api, synth, replay as rp
    K, W, F = args.steps, args.warmup, args.frames_per_step
    per_rank = (W + K) * F
    shards = rp.shard_subsequences(world * per_rank, world)
    a, b = shards[rank]
    oracle_job, spec = None, None
    if not args.no_verify and per_rank <= 1600:  # every rank checks its own shard (the others start behind rank 0's build: the barrier below)
        if rank == 0:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
        spec = {"subsequence": 0, "camera": "KITTI_1226", "seed": 12345, "period": 50, "phase": a, "frames": per_rank}
    if world > 1 and rank != 0:
        dist.barrier()  # rank 0 renders the scene's period (cached in /tmp), the others read it
    t_r = time.perf_counter()
    src = rp.SyntheticSource(world * per_rank, camera="KITTI_1226", seed=12345, prerender=True)
    src.prepare(a, b)  # (poses up front, as a log's are)
    render_s = time.perf_counter() - t_r
    if world > 1 and rank == 0:
        dist.barrier()
    if spec:
        oracle_job = start_oracle_replays([spec])[0]
    cam = src.cam
    if per_rank >= 96:  # the runtime's one-time costs (profiles/r06_streaming.md): a throw-away engine over the shard's first frames
        pre = rp.HipEngine(cam, device=device, capacity=1 << 21, pipeline_depth=args.pipeline_depth or 24, chunk=48)
        pre.replay(src, a, a + min(480, per_rank), origin=a)
        pre.close()
    eng = rp.HipEngine(cam, device=device, capacity=1 << 21, pipeline_depth=args.pipeline_depth or 24, chunk=48)
    if W:
        eng.replay(src, a, a + W * F, origin=a)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.replay(src, a + W * F, b, origin=a)  # (returns when the device has finished)
    torch.cuda.synchronize()
    dt_mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    st = dict(eng.stats)
    rank_fps = [K * F / dt_mine]
    if world > 1:
        mine = torch.tensor([dt_mine, dt], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_fps = [K * F / float(t_[0].item()) for t_ in every]
        dt = max(float(t_[1].item()) for t_ in every)
    n_final = eng.ff.map_size()
    got = eng.cloud() if spec else None
    merge_s, counts = None, [n_final]
    if dist is not None:
        cloud = eng.cloud_tensor(torch, f"cuda:{device}").to(coll_dev)
        try:
            with stdout_to_stderr():
                rp.merge_clouds(cloud[:44])  # communicator warm-up
            torch.cuda.synchronize()
            t_m = time.perf_counter()
            merged, counts = rp.merge_clouds(cloud)
            torch.cuda.synchronize()
            merge_s = time.perf_counter() - t_m
            assert counts[rank] == n_final and merged.numel() == 44 * sum(counts)
        except Exception as e:  # noqa: BLE001
            if world > 1:
                raise
            group_error, dist = "merge failed: " + repr(e)[:300], None
    n_pix, n_seed = cam.width * cam.height, (cam.width // 8) * (cam.height // 8)
    fps = world * K * F / dt
    out = {"metric": "depth frames fused/sec @ KITTI 1226x370", "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
           "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2]: one KITTI-shaped 1226x370 sequence split into one contiguous subsequence per GPU, every rank's frames "
                                  "streamed from page-locked host memory through one handle (frame groups), map resident, RCCL all-gather of the final clouds",
                      "frames_per_rank_per_step": F, "frames_per_rank": per_rank, "timed_frames_per_rank": K * F, "shards": [list(s_) for s_ in shards],
                      "pipeline_depth": args.pipeline_depth or 24, "chunk_frames": 48, "host_blocks": rp.HipEngine.BLOCKS,
                      "host_to_device_GBps_per_rank": round(st["frames"] * st["bytes_per_frame"] / st["seconds"] / 1e9, 2),
                      "frames_already_page_locked": st["zero_copy"], "timed_seconds": round(dt, 4), "host_render_seconds": round(render_s, 1),
                      "final_surfels_all_ranks": int(sum(counts)),
                      "parallelism": f"{world} GPU x 1 streamed subsequence, all-gather of final cloud only",
                      "pcie_note": "inputs are NOT resident when the clock starts: every timed frame crosses the host link inside the timed region "
                                   "(the resident-input headline is --workload headline)"}}
    if dist is None:
        out["multi_gpu"] = {"world_size_seen_by_backend": None, "backend": None, "note": "no process group: " + (group_error or "--no-rccl-world1")}
    else:
        out["multi_gpu"] = {"world_size_seen_by_backend": dist.get_world_size(), "backend": dist.get_backend(),
                            "per_rank_frames_per_s": [round(v, 1) for v in rank_fps], "min_rank_frames_per_s": round(min(rank_fps), 1),
                            "max_rank_frames_per_s": round(max(rank_fps), 1), "final_cloud_all_gather_ms": round(merge_s * 1e3, 3),
                            "final_cloud_bytes_all_ranks": int(sum(counts)) * 44, "per_rank_surfels": [int(c) for c in counts]}
    if rank == 0 and not args.no_roofline:
        # the dominant stage launched for ONE frame (HIP events on the handle's stream, eager replay): the frame groups of the
        # timed region launch it for eight frames at a time, for which the library has no event-timed form
        ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=50, surfel_capacity=1 << 21)
        for i, (img, dep) in enumerate(src._period):
            ff.frame_upload(i, img, dep)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        n_ev = min(48, per_rank)
        plan = api.FusionFunctions.pack_replay([t % 50 for t in range(a, a + n_ev)], [t // 5 for t in range(n_ev)], np.stack([src.pose(t) for t in range(a, a + n_ev)]))
        stages, nfr = ff.replay_timed(*plan)
        ovh = ff.event_overhead_ms * 1e3
        per = {k: max(v[0] / max(v[1], 1) * 1e3 - ovh, 0.0) for k, v in stages.items()}
        dom = [k for k in per if k.startswith("update_seeds")]
        dom_us = float(np.mean([per[k] for k in dom]))
        alg = stage_alg_bytes("update_seeds_0", n_pix, n_seed, ff.timed_mean_local, ff.timed_mean_new)
        traffic, tsrc = pmc_traffic("k_update_seeds", PMC_TRAFFIC_SINGLE)
        out["roofline"] = {"bound": "hbm", "kernel": "k_update_seeds_wave", "launches_per_frame": len(dom), "achieved": round(alg / dom_us / 1e3, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / dom_us / 1e3 / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "traffic_source": f"profiles/{tsrc}" if tsrc else None, "alg_bytes_per_launch": int(alg), "avg_launch_us": round(dom_us, 2),
                           "frames_timed": int(nfr),
                           "note": "the dominant stage launched for one frame; the timed region launches the superpixel stages for eight consecutive "
                                   "frames at a time (frame groups); e2e_hbm_frac is the whole streamed replay's"}
        out["kernel_us"] = {k: round(v, 2) for k, v in per.items()}
        ff.close()
    if spec:
        try:
            so, _ = oracle_job.communicate(timeout=300)
            rec = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
        except (subprocess.TimeoutExpired, IndexError, ValueError):
            oracle_job.kill()
            rec = None
        same = bool(rec) and rec["surfels"] == len(got) and rec["sha256"] == hashlib.sha256(canon_bytes(got)).hexdigest()
        rows = [[int(same) if rec else -1, int(len(got)), rec["surfels"] if rec else 0]]
        if world > 1:
            mine = torch.tensor(rows[0], dtype=torch.int64, device=coll_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rows = [[int(v) for v in t_.tolist()] for t_ in every]
        out["verified"] = out["verified_timed_region"] = all(v[0] == 1 for v in rows) if all(v[0] >= 0 for v in rows) else None
        out["verification"] = {"what": "every rank's final map (its shard of the sequence from an empty map, warm-up and timed steps) against the "
                                       "CPU oracle's replay of the same subsequence, NaN-canonical SHA-256 of the whole map",
                               "surfels": rows[0][1], "oracle_surfels": rows[0][2] if rows[0][0] >= 0 else None, "equal": bool(rows[0][0]) if rows[0][0] >= 0 else None,
                               "ranks": [{"rank": r_, "frames": [shards[r_][0], shards[r_][1] - 1], "surfels": v[1], "oracle_surfels": v[2],
                                          "equal": None if v[0] < 0 else bool(v[0])} for r_, v in enumerate(rows)]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cam, src.scene, src._period, 50, W * F, per_rank)
    m_mean = n_final / 2.0
    out["e2e_hbm_frac"] = round(fps * (9 * n_pix + 60 * n_seed + 88 * m_mean + 44 * 1400.0) / 1e9 / (HBM_PEAK_GBS * world), 5)
    eng.close()
    src.close()
    close_group(dist, world)
    if rank == 0:
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--streamed-rows", choices=("tight", "pitch"), default="pitch",
                    help="the `streamed_input` leg's page-locked frames: rows at the frame slots' pitch (default: one transfer per plane straight into "
                         "the slots), or `width` elements apart (4.4 %% fewer bytes over the link, a repack kernel behind every transfer: measured slower)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSM_BENCH_STREAMS", "0")),
                    help=f"independent subsequences (handles) per GPU; 0 = {DEFAULT_SUBSEQUENCES} in batched mode, 8 in streams mode")
    ap.add_argument("--frames-per-step", type=int, default=int(os.environ.get("DSM_BENCH_FRAMES_PER_STEP", "0")),
                    help="frames every subsequence advances per step (0 = 32; 48, one upload chunk, with --workload sharded)")
    ap.add_argument("--mode", choices=("batched", "streams"), default=os.environ.get("DSM_BENCH_MODE", "batched"),
                    help="batched: the B subsequences advance in lockstep, one launch per kernel for all of them (dsm_batch_*); "
                         "streams: B handles on B streams, the hardware queues overlap their kernels (round 1's mode)")
    ap.add_argument("--batches", type=int, default=int(os.environ.get("DSM_BENCH_BATCHES", "4")),
                    help="batched mode: split the B subsequences into this many batches, each on its own stream")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("DSM_BENCH_HOST_THREADS", "4")),
                    help="host threads enqueueing graph replays (each drives streams/threads handles)")
    ap.add_argument("--pipeline-depth", type=int, default=int(os.environ.get("DSM_BENCH_PIPELINE_DEPTH", "0")),
                    help="frames of one subsequence whose superpixel stages may be in flight (0 = 1: the B subsequences already fill the queues)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the legs beside the headline (drop-in, configs 4 and 5, node)")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the timed form against the CPU oracle")
    ap.add_argument("--legs", default=os.environ.get("DSM_BENCH_LEGS", "all"),
                    help="comma-separated legs beside the headline to run (single_sequence, dropin, fullhd, live, node, kitti_like, "
                         "streamed, sharded_replay, bounded_map, tum_like); default all")
    ap.add_argument("--workload", choices=("headline", "sharded"), default=os.environ.get("DSM_BENCH_WORKLOAD", "headline"),
                    help="headline: BASELINE configs[1], batched resident subsequences (weak scaling); sharded: BASELINE configs[2] itself -- ONE "
                         "subsequence per GPU streamed from page-locked host memory through replay.HipEngine, all-gather of the final clouds")
    ap.add_argument("--no-rccl-world1", action="store_true",
                    help="--gpus 1: do not create the process group of one through which the final cloud merge runs on RCCL")
    ap.add_argument("--oracle-replay-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.frames_per_step <= 0:
        args.frames_per_step = 48 if args.workload == "sharded" else 32

    if args.oracle_replay_worker:
        oracle_replay_worker(args.oracle_replay_worker)
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # one rank per GPU under torch.distributed.run; this process only waits
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}), or let "
                 f"`python bench.py --gpus {args.gpus}` launch them")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "sharded":
        _, device, dist, coll_dev, group_error = open_group(args, world, rank, local_rank)
        sharded_workload(args, world, rank, device, dist, coll_dev, group_error)
        return
    B, K, W, F = args.streams or (DEFAULT_SUBSEQUENCES if args.mode == "batched" else 8), args.steps, args.warmup, args.frames_per_step
    period = 50
    extras = rank == 0 and world == 1 and not args.no_dropin
    legs = set(args.legs.split(","))
    leg_on = lambda name: extras and ("all" in legs or name in legs)  # noqa: E731

    # ---- all synthetic frames first, on worker processes (fresh interpreters, clean environment; cached in /tmp)
    from densesurfelmapping_amd import synth
    cam = synth.KITTI_1226
    n_seed = (cam.width // 8) * (cam.height // 8)
    n_pix = cam.width * cam.height
    # up to 16 different scenes per rank; subsequences beyond that replay a scene from half a period further on (other
    # images at any time, another map)
    n_scene = min(B, 16)
    scenes = [synth.Scene(seed=12345 + 1000 * rank + 17 * b, frames_per_period=period) for b in range(n_scene)]
    scene_of = [b % n_scene for b in range(B)]
    phase_of = [(b // n_scene) * (period // 2) % period for b in range(B)]
    jobs = [(cam, scenes[b], i) for b in range(n_scene) for i in range(period)]
    cam_v, scene_v = synth.VGA_RGBD, synth.Scene(seed=5, scale=0.12, step=0.05, frames_per_period=30)
    cam_h, scene_h = synth.FULLHD, synth.Scene(seed=12345, frames_per_period=10)
    # the reference's own kind of input (kitti_publisher/scripts/publisher.py:37-40): depth = bf / quantised disparity, +inf where
    # the disparity is 0, an image with eight grey levels and saturated highlights -- the `kitti_like` leg
    n_scene_k = min(B, 4)
    scenes_k = [synth.Scene(seed=12345 + 1000 * rank + 17 * b, frames_per_period=period, stereo=True, saturate_above=150.0, intensity_levels=8)
                for b in range(n_scene_k)]
    # BASELINE configs[3]'s kind of input (the `tum_like` leg): 640x480 under the RGB-D constant set, a hand-held loop through a
    # room, depth as a Kinect + the TUM dataset's uint16 / 5000 PNGs deliver it; and the same room through an ideal sensor
    period_t = 100
    scenes_t = [synth.Scene(seed=7 + 1000 * rank + 17 * b, tum=True, frames_per_period=period_t, intensity_noise=8.0, checker=25.0, n_boxes=6)
                for b in range(n_scene_k)]
    scenes_ts = [synth.Scene(**dict(synth.dataclasses.asdict(sc_), tum_sensor=False)) for sc_ in scenes_t]
    tum_on = extras and ("all" in legs or "tum_like" in legs)
    if extras:
        jobs += [(cam_v, scene_v, i) for i in range(30)] + [(cam_h, scene_h, i) for i in range(10)]
        jobs += [(cam, scenes_k[b], i) for b in range(n_scene_k) for i in range(period)]
    n_before_t = len(jobs)
    if tum_on:
        jobs += [(cam_v, sc_, i) for sc_ in scenes_t + scenes_ts for i in range(period_t)]
    workers = max(1, min(48, (os.cpu_count() or 2) // (2 * max(world, 1))))
    t_r = time.perf_counter()
    frames_all = synth.render_many(jobs, workers)
    render_s = time.perf_counter() - t_r
    rendered = [frames_all[b * period:(b + 1) * period] for b in range(n_scene)]
    frames_v = frames_all[n_scene * period:n_scene * period + 30] if extras else []
    frames_h = frames_all[n_scene * period + 30:n_scene * period + 40] if extras else []
    rendered_k = [frames_all[n_scene * period + 40 + b * period:n_scene * period + 40 + (b + 1) * period] for b in range(n_scene_k)] if extras else []
    rendered_t = [frames_all[n_before_t + b * period_t:n_before_t + (b + 1) * period_t] for b in range(2 * n_scene_k)] if tum_on else []

    # The checker of the timed region, started now so that it runs beside everything below: the CPU oracle replays one
    # subsequence per batch from its empty map through the LAST timed frame (a single-threaded replay of (W + K) * F frames
    # at ~10-15 frames/s: a minute or two, on cores the GPU legs do not use); its maps are compared with the maps the timed
    # run itself leaves behind -- maps of 300-450 k surfels with the driver's W and K, beyond k_frame_tail's one-workgroup
    # path -- see "verified" below.  Nothing of it runs inside a timed interval's critical path.
    total_frames = (W + K) * F
    n_bat_plan = max(1, min(args.batches, B)) if args.mode == "batched" else 0
    oracle_jobs, oracle_specs = [], []
    if rank == 0 and n_bat_plan and not args.no_verify and total_frames <= 1600 and args.workload == "headline":  # (multi-rank runs too: rank 0's subsequences)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
        oracle_specs = [{"subsequence": g, "camera": "KITTI_1226", "seed": 12345 + 1000 * rank + 17 * scene_of[g], "period": period,
                         "phase": phase_of[g], "frames": total_frames} for g in range(n_bat_plan)]  # subsequence g = the first of batch g
        oracle_jobs = start_oracle_replays(oracle_specs)

    import torch
    from densesurfelmapping_amd import api

    torch, device, dist, coll_dev, group_error = open_group(args, world, rank, local_rank)
    # Every OTHER rank checks a subsequence of its own too (the first of its first batch): started behind the rendezvous, by
    # which time rank 0 has built the oracle; the verdicts are gathered at the end (`verification.ranks`).
    verify_ranks = world > 1 and n_bat_plan and not args.no_verify and total_frames <= 1600 and args.workload == "headline"
    if verify_ranks and rank > 0:
        oracle_specs = [{"subsequence": 0, "camera": "KITTI_1226", "seed": 12345 + 1000 * rank + 17 * scene_of[0], "period": period,
                         "phase": phase_of[0], "frames": total_frames}]
        oracle_jobs = start_oracle_replays(oracle_specs)

    total = (W + K) * F
    lo_t, hi_t = W * F, total  # frame indices of the timed region, per subsequence
    capacity = 1 << 21
    def make_handle(b, **kw):
        """Subsequence b: its scene's period of frames resident in HBM, an empty map, and its replay plan."""
        ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=period, surfel_capacity=capacity, **kw)
        for i, (img, dep) in enumerate(rendered[scene_of[b]]):
            ff.frame_upload(i, img, dep)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        return ff

    def make_plan(b):
        slots = [(t + phase_of[b]) % period for t in range(total)]
        refs = [t // 5 for t in range(total)]
        poses = np.stack([scenes[scene_of[b]].pose(t + phase_of[b]) for t in range(total)])
        return api.FusionFunctions.pack_replay(slots, refs, poses)

    handles = [make_handle(b, pipeline_depth=args.pipeline_depth or 1) for b in range(B)]  # B subsequences already fill the queues
    plans = [make_plan(b) for b in range(B)]

    # Enqueue: every handle has its own stream; T host threads each drive B/T handles (the C ABI is
    # thread-safe per handle and ctypes drops the GIL during the call), chunk by chunk so that all
    # subsequences advance together.
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(args.host_threads, B))
    pool = ThreadPoolExecutor(max(n_thr, args.batches))
    enqueue_s = [0.0]

    def drive(group, lo, hi, chunk):
        for c0 in range(lo, hi, chunk):
            c1 = min(hi, c0 + chunk)
            for b in group:
                s, r, p = plans[b]
                handles[b].replay_enqueue(s[c0:c1], r[c0:c1], p[c0:c1])

    n_bat = max(1, min(args.batches, B)) if args.mode == "batched" else 0
    groups_b = [list(range(g, B, n_bat)) for g in range(n_bat)]
    batches = [api.Batch([handles[b] for b in grp]) for grp in groups_b]

    def drive_batch(g, lo, hi, chunk):
        for c0 in range(lo, hi, chunk):
            c1 = min(hi, c0 + chunk)
            s, r, p, n = api.Batch.pack([(plans[b][0][c0:c1], plans[b][1][c0:c1], plans[b][2][c0:c1]) for b in groups_b[g]])
            batches[g].replay_enqueue(s, r, p, n)

    def run(lo, hi, chunk=64):
        t_e = time.perf_counter()
        if batches:
            list(pool.map(lambda g: drive_batch(g, lo, hi, chunk), range(n_bat)))
        else:
            groups = [list(range(t, B, n_thr)) for t in range(n_thr)]
            list(pool.map(lambda g: drive(g, lo, hi, chunk), groups))
        enqueue_s[0] = time.perf_counter() - t_e

    def sync_all():
        for ff in handles:
            ff.synchronize()
        torch.cuda.synchronize()

    run(0, lo_t)
    sync_all()
    m_start = [ff.map_size() for ff in handles]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    try:
        pr = torch.cuda.get_device_properties(device)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except (AttributeError, RuntimeError):
        bdf = None
    clock = ClockSampler(bdf)
    t0 = time.perf_counter()
    with clock:
        run(lo_t, hi_t)
        sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rank_fps = [B * K * F / dt]
    if world > 1:
        mine = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_fps = [B * K * F / float(t.item()) for t in every]  # each rank's own rate between the two barriers
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    m_end = [ff.map_size() for ff in handles]
    # the maps the timed run leaves behind, for the oracle's verdict further down
    import hashlib
    end_maps = {}
    for sp in oracle_specs:
        got = handles[sp["subsequence"]].map_download()
        end_maps[sp["subsequence"]] = (int(len(got)), hashlib.sha256(canon_bytes(got)).hexdigest())

    # merge of the final clouds (outside the timed region): RCCL all-gather over xGMI
    merged_total = sum(m_end)
    merge_s = None
    if dist is not None:
        from densesurfelmapping_amd.replay import merge_clouds
        clouds = []
        for ff, m in zip(handles, m_end):
            buf = torch.empty(m * 44, dtype=torch.uint8, device=f"cuda:{device}")
            ff.map_copy_to_device(buf.data_ptr(), m)
            clouds.append(buf)
        mine = torch.cat(clouds).to(coll_dev)
        del clouds
        try:
            with stdout_to_stderr():
                merge_clouds(mine[:44])  # communicator warm-up (first-collective setup is not the merge)
            torch.cuda.synchronize()
            t_m = time.perf_counter()
            merged, counts = merge_clouds(mine)
            torch.cuda.synchronize()
            merge_s = time.perf_counter() - t_m
            assert world > 1 or (counts == [sum(m_end)] and torch.equal(merged, mine)), "a group of one must hand the cloud back unchanged"
            merged_total = int(sum(counts))
            del merged
        except Exception as e:  # noqa: BLE001 -- at --gpus 1 the merge is an extra: report, do not lose the measurement
            if world > 1:
                raise
            group_error, dist = "merge failed: " + repr(e)[:300], None
        del mine

    for bt in batches:
        bt.close()
    for ff in handles:  # the extra measurements below run alone on the GPU
        ff.close()
    handles = []
    frames_total = world * B * K * F
    fps = frames_total / dt
    m_avg = float(np.mean([(a + b) / 2 for a, b in zip(m_start, m_end)]))

    out = {
        "metric": "depth frames fused/sec @ KITTI 1226x370",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: synthetic KITTI-shaped replay 1226x370, full superpixel+normal+"
                               "fuse+compaction HIP path, frames and map resident in HBM",
                   "launch_mode": (f"batched: {n_bat} batch(es) of subsequences advancing in lockstep, every kernel launched once per batch"
                                   if args.mode == "batched" else "streams: one handle and stream per subsequence"),
                   "subsequences_per_gpu": B, "frames_per_subsequence_per_step": F, "frames_per_step_per_gpu": B * F,
                   "timed_frame_indices": [lo_t, hi_t - 1], "host_enqueue_threads": n_thr,
                   "pipeline_depth": args.pipeline_depth or 1,
                   "host_enqueue_seconds": round(enqueue_s[0], 4), "timed_seconds": round(dt, 4), "scene_period_frames": period,
                   "scenes": f"{n_scene} synthetic scenes per rank (seed 12345 + 1000*rank + 17*b); subsequences beyond {n_scene} replay a "
                             "scene from half a period further on",
                   "host_render_seconds": round(render_s, 1),
                   "mean_live_surfels": round(m_avg), "final_surfels_all_ranks": merged_total,
                   "parallelism": f"{world} GPU x {B} independent subsequences, all-gather of final cloud only"},
    }
    if dist is None:
        out["multi_gpu"] = {"world_size_seen_by_backend": None, "backend": None,
                            "note": "no process group: " + (group_error or "--no-rccl-world1")}
    else:
        out["multi_gpu"] = {"world_size_seen_by_backend": dist.get_world_size(), "backend": dist.get_backend(),
                            "per_rank_frames_per_s": [round(v, 1) for v in rank_fps],
                            "min_rank_frames_per_s": round(min(rank_fps), 1), "max_rank_frames_per_s": round(max(rank_fps), 1),
                            "final_cloud_all_gather_ms": round(merge_s * 1e3, 3),
                            "final_cloud_bytes_all_ranks": int(merged_total) * 44,
                            "note": "value = frames of all ranks / the slowest rank's time between the barriers; the all-gather of "
                                    "the final clouds (counts, then the padded clouds) is outside the timed region"
                                    + ("; a group of ONE: the merge ran through RCCL on this GPU (library path exercised, no link crossed)" if world == 1 else "")}

    k_avg = None
    if rank == 0 and not args.no_roofline:
        # per-kernel durations, measured live with HIP events on the handle's own stream (eager replay of frames of
        # the timed region of subsequence 0 on a fresh handle; a long delay kernel in front of each frame keeps the
        # host launch latency out of the intervals).  K and M of B_alg are measured on the same frames.
        ff = make_handle(0)
        s, r, p = plans[0]
        ff.replay_enqueue(s[:lo_t], r[:lo_t], p[:lo_t])
        ff.synchronize()
        n_ev = min(hi_t - lo_t, 96)
        stages, nfr = ff.replay_timed(s[lo_t:lo_t + n_ev], r[lo_t:lo_t + n_ev], p[lo_t:lo_t + n_ev])
        mt, k_avg = ff.timed_mean_local, ff.timed_mean_new
        ovh = ff.event_overhead_ms * 1e3  # an empty event-to-event interval, subtracted from every stage
        per = {k: max(v[0] / max(v[1], 1) * 1e3 - ovh, 0.0) for k, v in stages.items()}  # us per launch
        # dominant kernel = the kernel FUNCTION with the largest share of a frame's GPU time, as `rocprofv3 --stats`
        # ranks them (k_update_seeds<true> runs twice per frame, k_assign<false> twice, ...); achieved = algorithmic
        # bytes of one launch / its average launch duration.  Every stage's own fraction is in kernel_hbm_frac.
        groups = {}
        for st_name, us in per.items():
            groups.setdefault(KERNEL_OF_STAGE.get(st_name, st_name), []).append(st_name)
        dom_fn = max(groups, key=lambda g: sum(per[x] for x in groups[g]))
        dom_stages = groups[dom_fn]
        dom_us = float(np.mean([per[x] for x in dom_stages]))
        alg = float(np.mean([stage_alg_bytes(x, n_pix, n_seed, mt, k_avg) for x in dom_stages]))
        achieved = alg / (dom_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(dom_fn, PMC_TRAFFIC_SINGLE)
        out["roofline"] = {"bound": "hbm", "kernel": dom_fn, "launches_per_frame": len(dom_stages),
                           "share_of_frame_kernel_time": round(sum(per[x] for x in dom_stages) / sum(per.values()), 3),
                           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "traffic_source": f"profiles/{traffic_src} (rocprofv3 --pmc, separate passes)" if traffic_src else None,
                           "event_overhead_us": round(ovh, 2),
                           "alg_bytes_per_launch": int(alg), "avg_launch_us": round(dom_us, 2),
                           "frames_timed": int(nfr), "mean_live_surfels": round(mt), "mean_new_surfels": round(k_avg, 1),
                           "longest_single_launch": {"stage": max(per, key=per.get), "us": round(max(per.values()), 2),
                                                     "hbm_frac": round(stage_alg_bytes(max(per, key=per.get), n_pix, n_seed, mt, k_avg)
                                                                       / (max(per.values()) * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)}}
        out["kernel_us"] = {k: round(v, 2) for k, v in per.items()}
        out["kernel_hbm_frac"] = {k: round(stage_alg_bytes(k, n_pix, n_seed, mt, k_avg) / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                                  for k, v in per.items() if v > 0}
        out["frame_kernel_sum_us"] = round(sum(per.values()), 1)
        ksum = sum(per.values()) * 1e-6
        b_alg_t = 9 * n_pix + 60 * n_seed + 88 * mt + 44 * k_avg
        out["kernel_time_weighted_hbm_frac"] = round(b_alg_t / ksum / 1e9 / HBM_PEAK_GBS, 5)
        ff.close()
        if args.mode == "batched":
            # ... and of the launches the timed region actually makes: one batch of the subsequences, every kernel
            # launched once for all of them.  The roofline is that of these launches (bytes of all subsequences of the
            # batch / launch duration).
            nb = len(groups_b[0])
            hs = [make_handle(b, pipeline_depth=1) for b in groups_b[0]]
            bt = api.Batch(hs)
            sb, rb, pb, nn = api.Batch.pack([(plans[b][0][:lo_t], plans[b][1][:lo_t], plans[b][2][:lo_t]) for b in groups_b[0]])
            bt.replay_enqueue(sb, rb, pb, nn)
            bt.synchronize()
            n_evb = min(hi_t - lo_t, 48)
            sb, rb, pb, nn = api.Batch.pack([(plans[b][0][lo_t:lo_t + n_evb], plans[b][1][lo_t:lo_t + n_evb], plans[b][2][lo_t:lo_t + n_evb])
                                             for b in groups_b[0]])
            stb, nfb = bt.replay_timed(sb, rb, pb, nn)
            mtb, kb = bt.timed_mean_local, bt.timed_mean_new
            ovb = bt.event_overhead_ms * 1e3
            perb = {k: max(v[0] / max(v[1], 1) * 1e3 - ovb, 0.0) for k, v in stb.items()}  # us per batched launch
            dom_usb = float(np.mean([perb[x] for x in dom_stages]))
            algb = nb * float(np.mean([stage_alg_bytes(x, n_pix, n_seed, mtb, kb) for x in dom_stages]))
            achb = algb / (dom_usb * 1e-6) / 1e9
            # (a batched stage may be two or three kernels: the lane-per-seed forms)
            fn_b = BATCHED_KERNELS_OF_STAGE.get(dom_stages[-1]) or [dom_fn[:-1] + ", true>" if dom_fn.endswith(">") else dom_fn + "<true>"]
            traffic_b, traffic_src_b = pmc_traffic(fn_b, PMC_TRAFFIC_BATCHED, nb)
            out["roofline_single_launch"] = out["roofline"]
            out["roofline"] = {"bound": "hbm", "kernel": " + ".join(fn_b), "stage": dom_fn, "launches_per_frame": len(dom_stages), "subsequences_per_launch": nb,
                               "achieved": round(achb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achb / HBM_PEAK_GBS, 5),
                               "traffic": traffic_b,
                               "traffic_source": (f"profiles/{traffic_src_b} (rocprofv3 --pmc of launches batched over {nb} subsequences, separate passes)"
                                                  if traffic_src_b else None),
                               "event_overhead_us": round(ovb, 2), "alg_bytes_per_launch": int(algb), "avg_launch_us": round(dom_usb, 2),
                               "frames_timed": int(nfb), "mean_live_surfels": round(mtb), "mean_new_surfels": round(kb, 1),
                               "timing": "HIP events between the kernels of an eager replay of ONE batch alone on the GPU, the empty-interval "
                                         "overhead (event_overhead_us) subtracted; `rocprofv3 --kernel-trace` of the same launches: "
                                         "profiles/r06_kernel_trace_batch32x1.md; in the timed region four batches share the machine and a "
                                         "launch takes longer (profiles/r06_kernel_trace_batch32x4_default.md)",
                               "note": "the timed region launches every kernel once per batch of subsequences; roofline_single_launch is the "
                                       "same kernel launched for one subsequence"}
            # the stage priced by ITS OWN compulsory bytes: the label planes here are 2 bytes per pixel (SURVEY.md's 9N counts the
            # reference's 4-byte labels), so update_seeds must move 7N + 32S
            if dom_fn == "k_update_seeds":
                own = nb * (7 * n_pix + 32 * n_seed)
                out["roofline"]["own_bytes"] = {"alg_bytes_per_launch": int(own), "achieved": round(own / (dom_usb * 1e-6) / 1e9, 1),
                                                "frac": round(own / (dom_usb * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                                "traffic_over_own_bytes": round(traffic_b / own, 3) if traffic_b else None,
                                                "note": "7N + 32S per frame: labels as this build stores them (2 B/pixel); `frac` above prices SURVEY.md's 9N + 32S"}
            # the WHOLE frame's algorithmic bytes over the whole frame's kernel time (the launch set
            # of one batch alone on the GPU), beside the dominant stage's own number -- and which roof actually binds
            b_alg_b = 9 * n_pix + 60 * n_seed + 88 * mtb + 44 * kb
            t_frame_b = sum(perb.values()) / nb * 1e-6
            out["roofline"]["whole_frame_kernel_weighted"] = {
                "alg_bytes_per_frame": int(b_alg_b), "kernel_us_per_frame": round(t_frame_b * 1e6, 2),
                "achieved": round(b_alg_b / t_frame_b / 1e9, 1), "unit": "GB/s", "frac": round(b_alg_b / t_frame_b / 1e9 / HBM_PEAK_GBS, 5),
                "note": "B_alg = 9N + 60S + 88M + 44K of one frame / the sum of its sixteen stages' launch durations per frame"}
            out["roofline"]["binding_roof"] = "valu_issue"
            out["roofline"]["binding_roof_note"] = ("the superpixel stages are bound by VALU instruction issue (see `valu_issue`: wave-instructions per "
                                                    "frame x frames/s against 1 024 SIMDs x clock / 4), not by HBM; `bound`: \"hbm\" names the roof "
                                                    "`frac` is measured against, as BASELINE.json's metric asks")
            out["batched_kernel_us"] = {k: round(v, 2) for k, v in perb.items()}
            out["batched_kernel_hbm_frac"] = {k: round(nb * stage_alg_bytes(k, n_pix, n_seed, mtb, kb) / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                                              for k, v in perb.items() if v > 0}
            out["batched_frame_kernel_sum_us_per_frame"] = round(sum(perb.values()) / nb, 1)
            bt.close()
            for h_ in hs:
                h_.close()

    if leg_on("single_sequence"):
        # ONE sequence (BASELINE configs[1] as the reference would replay it): frames are strictly ordered, but
        # only fuse + tail need the map -- the superpixel stages of up to 8 frames run ahead on their own streams
        s1, r1, p1 = plans[0]
        by_depth = {}
        for depth in (16, 24, 32):
            ff = make_handle(0, pipeline_depth=depth)
            ff.replay_enqueue(s1[:lo_t], r1[:lo_t], p1[:lo_t])
            ff.synchronize()
            t_s = time.perf_counter()
            ff.replay_enqueue(s1[lo_t:hi_t], r1[lo_t:hi_t], p1[lo_t:hi_t])
            t_enq = time.perf_counter() - t_s
            ff.synchronize()
            by_depth[depth] = (round((hi_t - lo_t) / (time.perf_counter() - t_s), 1), round(t_enq, 4))
            ff.close()
        best = max(by_depth, key=lambda d_: by_depth[d_][0])
        out["single_sequence"] = {"value": by_depth[best][0], "unit": "frames/s", "pipeline_depth": best,
                                  "host_enqueue_seconds": by_depth[best][1],
                                  "frames_per_s_by_pipeline_depth": {str(d_): v[0] for d_, v in by_depth.items()},
                                  "note": "one subsequence, one handle: the superpixel stages of depth/4 consecutive frames as one batched "
                                          "launch per kernel (three or four groups of pipelines in turn; three leave the map stream a hardware queue of its own), fuse + compaction strictly in "
                                          "frame order on the map stream; same results as the serial order"}
    if leg_on("dropin"):
        # the synchronous drop-in call (host buffers in and out over PCIe every frame), for DESIGN.md;
        # never the headline value
        ff = api.FusionFunctions.from_camera(cam, device=device, surfel_capacity=capacity)
        buf = np.zeros(400_000, api.SURFEL_DTYPE)  # the caller's std::vector<SurfelElement> storage (pageable)
        n_loc, n_d0, n_d1 = 0, 200, 500
        for t in range(n_d0):
            n_loc, _ = ff.fuse_map_inplace(t // 5, rendered[0][t % period][0], rendered[0][t % period][1], scenes[0].pose(t), buf, n_loc)
        m_d0 = n_loc
        t_d = time.perf_counter()
        for t in range(n_d0, n_d1):
            n_loc, _ = ff.fuse_map_inplace(t // 5, rendered[0][t % period][0], rendered[0][t % period][1], scenes[0].pose(t), buf, n_loc)
        out["dropin_pcie_inclusive"] = {"value": round((n_d1 - n_d0) / (time.perf_counter() - t_d), 1), "unit": "frames/s",
                                        "map_surfels": [int(m_d0), int(n_loc)],
                                        "note": "dsm_fuse_map as SurfelMap::fuse_map calls it: pageable host image, depth and surfel vector in, "
                                                "updated vector out, synchronous; per frame the frame goes up through page-locked staging, the "
                                                "vector is compared with what the previous call returned (and uploaded only if the caller "
                                                "changed it), of the map only the 64-record groups the frame changed come back (delta download)",
                                        "delta_download": ff.debug_dropin_stats()}
        ff.close()
    if leg_on("fullhd"):
        # BASELINE configs[4]: 1920x1080 depth stream against >= 2 M live surfels, and the loop-closure deformation
        # (surfel_map.cpp:750-789).  The big map is the map of a short 1080p replay replicated
        # with millimetre jitter, so that its surfels project into the frames and take the fusion branch.
        n_h = (cam_h.width // 8) * (cam_h.height // 8)
        plan_h = api.FusionFunctions.pack_replay([t % 10 for t in range(100)], [t // 5 for t in range(100)],
                                                 np.stack([scene_h.pose(t % 10) for t in range(100)]))

        def handle_h(depth, cap=3_200_000):
            ff_ = api.FusionFunctions.from_camera(cam_h, device=device, frame_slots=10, surfel_capacity=cap, pipeline_depth=depth)
            for i, (img, dep) in enumerate(frames_h):
                ff_.frame_upload(i, img, dep)
            return ff_
        ff = handle_h(1)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        ff.replay_enqueue(plan_h[0][:10], plan_h[1][:10], plan_h[2][:10])
        base = ff.map_download()
        rng = np.random.default_rng(0)
        big = np.tile(base, max(1, -(-2_000_000 // max(len(base), 1))))
        for f in ("px", "py", "pz"):
            big[f] += rng.normal(scale=1e-3, size=len(big)).astype(np.float32)
        big["update_times"] = 9
        big["last_update"] = 2
        # one 1080p sequence through one handle: strictly serial (depth 1), and with the superpixel stages of G consecutive
        # frames as one batched launch per kernel while fuse + compaction stay in frame order on the map stream (frame
        # groups, DESIGN.md section 4: the map is the only thing frame t + 1 needs from frame t, SM.cpp:161)
        by_depth_h, final_h = {}, {}
        for depth in (1, 12, 24):
            fd = ff if depth == 1 else handle_h(depth)
            fd.map_upload(big)
            fd.replay_enqueue(plan_h[0][10:34], plan_h[1][10:34], plan_h[2][10:34])
            fd.synchronize()
            t_h = time.perf_counter()
            fd.replay_enqueue(plan_h[0][34:82], plan_h[1][34:82], plan_h[2][34:82])
            fd.synchronize()
            by_depth_h[depth] = (time.perf_counter() - t_h) / 48
            final_h[depth] = fd.map_size()
            if depth > 1:
                fd.close()
        assert len(set(final_h.values())) == 1, final_h  # (same frames, same map: the depth changes nothing in the result)
        best_h = min(by_depth_h, key=by_depth_h.get)
        dt_h = by_depth_h[best_h]
        # per-kernel times: the depth-1 handle, eager replay of the next frames
        st_h, _ = ff.replay_timed(plan_h[0][82:94], plan_h[1][82:94], plan_h[2][82:94])
        ovh_h = ff.event_overhead_ms * 1e3
        per_h = {k: max(v[0] / max(v[1], 1) * 1e3 - ovh_h, 0.0) for k, v in st_h.items()}
        m_h = ff.timed_mean_local
        b_h = 9 * cam_h.width * cam_h.height + 60 * n_h + 88 * m_h + 44 * ff.timed_mean_new
        timed = event_timer(torch, ff)
        wp = np.eye(4, dtype=np.float32)
        wp[:3, 3] = (0.01, -0.02, 0.005)
        ff.map_warp(wp)
        us_w2 = timed(lambda: ff.map_warp(wp), 50)
        m_w2 = ff.map_size()
        out["fullhd_2M"] = {
            "workload": "BASELINE configs[4]: 1920x1080 depth stream against a live map of >= 2 M surfels, one sequence through one handle",
            "frames_per_s": round(1.0 / dt_h, 1), "ms_per_frame": round(dt_h * 1e3, 3), "live_surfels": round(m_h),
            "pipeline_depth": best_h,
            "frames_per_s_by_pipeline_depth": {str(d_): round(1.0 / v, 1) for d_, v in by_depth_h.items()},
            "pipeline_note": "depth 1 = strictly serial on one stream; 12 / 24 = the superpixel stages of 4 / 8 consecutive frames as one "
                             "batched launch per kernel (frame groups), fuse + compaction in frame order on the map stream; identical "
                             "maps (tests/test_gpu_scale.py::test_fullhd_frame_groups)",
            "alg_bytes_per_frame": int(b_h), "e2e_hbm_frac": round(b_h / dt_h / 1e9 / HBM_PEAK_GBS, 4),
            "kernel_us": {k: round(v, 1) for k, v in per_h.items() if k in ("seed_points", "seed_fit", "fuse_surfels", "frame_tail",
                                                                            "update_seeds_0", "assign_0")},
            "frame_kernel_sum_us": round(sum(per_h.values()), 1),
            "fuse_surfels": {"us": round(per_h["fuse_surfels"], 1), "alg_bytes": int(88 * m_h),
                             "achieved_GBps": round(88 * m_h / per_h["fuse_surfels"] / 1e3, 1),
                             "hbm_frac": round(88 * m_h / per_h["fuse_surfels"] / 1e3 / HBM_PEAK_GBS, 4),
                             "timing": "HIP events around the kernel on the handle's stream (eager replay, 12 frames)"},
            "map_warp": {"surfels": m_w2, "us": round(us_w2, 1), "alg_bytes": 88 * m_w2,
                         "achieved_GBps": round(88 * m_w2 / us_w2 / 1e3, 1), "hbm_frac": round(88 * m_w2 / us_w2 / 1e3 / HBM_PEAK_GBS, 4),
                         "timing": "HIP events around 50 back-to-back dsm_map_warp calls on the handle's stream; the 172 MB "
                                   "array fits the 256 MB Infinity Cache, see map_warp_8M for the HBM-bound size"},
        }
        # ... and the per-frame fuse against a map the Infinity Cache cannot hold (8 M live surfels: 352 MB in, <= 352 MB out)
        big8 = np.tile(base, max(1, -(-8_000_000 // max(len(base), 1))))
        for f in ("px", "py", "pz"):
            big8[f] += rng.normal(scale=1e-3, size=len(big8)).astype(np.float32)
        big8["update_times"] = 9
        big8["last_update"] = 2
        ff.close()
        ff = api.FusionFunctions.from_camera(cam_h, device=device, frame_slots=10, surfel_capacity=len(big8) + 600_000, pipeline_depth=1)
        for i, (img, dep) in enumerate(frames_h):
            ff.frame_upload(i, img, dep)
        ff.map_upload(big8)
        del big8
        ff.replay_enqueue(plan_h[0][10:14], plan_h[1][10:14], plan_h[2][10:14])
        ff.synchronize()
        st_8, _ = ff.replay_timed(plan_h[0][40:48], plan_h[1][40:48], plan_h[2][40:48])
        ovh_8 = ff.event_overhead_ms * 1e3
        us_f8 = max(st_8["fuse_surfels"][0] / max(st_8["fuse_surfels"][1], 1) * 1e3 - ovh_8, 1e-3)
        us_t8 = max(st_8["frame_tail"][0] / max(st_8["frame_tail"][1], 1) * 1e3 - ovh_8, 1e-3)
        m_8 = ff.timed_mean_local
        out["fuse_8M"] = {"workload": "1920x1080 frame fused into a live map of >= 8 M surfels (beyond the 256 MB Infinity Cache)",
                          "live_surfels": round(m_8), "fuse_surfels_us": round(us_f8, 1), "alg_bytes": int(88 * m_8),
                          "achieved_GBps": round(88 * m_8 / us_f8 / 1e3, 1), "hbm_frac": round(88 * m_8 / us_f8 / 1e3 / HBM_PEAK_GBS, 4),
                          "frame_tail_us": round(us_t8, 1),
                          "timing": "HIP events around the kernel on the handle's stream (eager replay, 8 frames)"}
        out["fuse_8M"].update(pmc_8m("k_fuse_surfels", us_f8))
        ff.close()
        # the warp kernel on a working set the Infinity Cache cannot hold: 8 M surfels = 352 MB read + 352 MB written
        n_w = 8_000_000
        wm = np.zeros(n_w, api.SURFEL_DTYPE)
        wm["px"] = np.arange(n_w, dtype=np.float32) * 1e-3
        wm["nz"] = 1.0
        wm["update_times"] = 3
        ff = api.FusionFunctions.from_camera(synth.TINY, device=device, surfel_capacity=n_w + 64)
        ff.map_upload(wm)
        del wm
        timed = event_timer(torch, ff)
        ff.map_warp(wp)
        us_w8 = timed(lambda: ff.map_warp(wp), 30)
        out["map_warp_8M"] = {"surfels": n_w, "us": round(us_w8, 1), "alg_bytes": 88 * n_w, "achieved_GBps": round(88 * n_w / us_w8 / 1e3, 1),
                              "hbm_frac": round(88 * n_w / us_w8 / 1e3 / HBM_PEAK_GBS, 4),
                              "timing": "HIP events around 30 back-to-back dsm_map_warp calls (704 MB moved per call)"}
        out["map_warp_8M"].update(pmc_8m("k_warp", us_w8))
        ff.close()
    if leg_on("live"):
        # BASELINE configs[3]: live callback, 640x480 RGB-D constants, one frame at a time: host frame in
        # (H2D), resident map, one hipGraph replay, wait -- the latency the 30 Hz node would see per frame
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
    if leg_on("node"):
        # the whole node through its message callbacks (stamp matching, pose graph, active /
        # inactive sets in HBM, loop closure at the start of the second lap), host-inclusive: every frame is copied into
        # the node's page-locked pool and uploaded
        from densesurfelmapping_amd import surfel_map
        n_node = 3 * period
        events = list(synth.node_messages(cam, scenes[0], n_node, lap=period, frames={i: f for i, f in enumerate(rendered[0])}))
        node = surfel_map.SurfelMap(cam, drift_free_poses=10, device=device, surfel_capacity=capacity)
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

    if rank == 0 and args.mode == "batched" and not args.no_verify:  # (multi-rank runs too: a SCALE line is verified like a BENCH line)
        # What was timed, checked: fresh handles, the timed region's own form -- every batch of subsequences enqueued by
        # its own host thread on its own stream, all batches in flight at once -- over the first frames of every plan, and
        # the maps of one handle per batch against the CPU oracle's replay of the same frames (untimed; the oracle is the
        # checker here, nothing of it runs inside any timed region).
        from oracle.bindings import SURFEL_DTYPE as O_DTYPE, PortOracle
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
        n_v = 12
        hs = [make_handle(b, pipeline_depth=1) for b in range(B)]
        bts = [api.Batch([hs[b] for b in grp]) for grp in groups_b]

        def one_v(g):
            for c0 in range(0, n_v, 5):  # several enqueue calls per batch, like the timed region's chunks
                c1 = min(n_v, c0 + 5)
                sb, rb, pb, nn = api.Batch.pack([(plans[b][0][c0:c1], plans[b][1][c0:c1], plans[b][2][c0:c1]) for b in groups_b[g]])
                bts[g].replay_enqueue(sb, rb, pb, nn)
        list(pool.map(one_v, range(n_bat)))
        for bt_ in bts:
            bt_.synchronize()
        checked, bad = [], []
        for g in range(n_bat):
            b = groups_b[g][g % len(groups_b[g])]
            got = hs[b].map_download()
            orc, lo_ = PortOracle(cam), np.zeros(0, O_DTYPE)
            for t in range(n_v):
                img, dep = rendered[scene_of[b]][(t + phase_of[b]) % period]
                lo_, _ = orc.fuse_map(t // 5, img, dep, scenes[scene_of[b]].pose(t + phase_of[b]), lo_)
            same = len(got) == len(lo_) and canon_bytes(got) == canon_bytes(lo_)
            checked.append({"subsequence": b, "batch": g, "surfels": int(len(got)), "oracle_surfels": int(len(lo_)), "equal": bool(same)})
            if not same:
                bad.append(b)
        # ... and the timed run ITSELF: the maps it left at the end of the timed region against the oracle replays started
        # before it (one subsequence per batch, from the empty map through the last timed frame)
        timed_rows, timed_ok = [], None
        if oracle_jobs:
            timed_ok = True
            t_wait = time.perf_counter()
            for sp, job in zip(oracle_specs, oracle_jobs):
                try:
                    so, _ = job.communicate(timeout=max(5.0, 240.0 - (time.perf_counter() - t_wait)))
                    rec = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
                except (subprocess.TimeoutExpired, IndexError, ValueError):
                    job.kill()
                    rec = None
                n_gpu, sha_gpu = end_maps[sp["subsequence"]]
                same = bool(rec) and rec["surfels"] == n_gpu and rec["sha256"] == sha_gpu
                timed_rows.append({"subsequence": sp["subsequence"], "frames": sp["frames"], "surfels": n_gpu,
                                   "oracle_surfels": rec["surfels"] if rec else None, "equal": same if rec else None})
                timed_ok = timed_ok and same
            oracle_jobs = []
        out["verified"] = not bad and timed_ok is not False
        out["verified_timed_region"] = timed_ok
        out["verification"] = {"timed_region": {"checked": timed_rows,
                                                "what": "the maps the TIMED run itself left after its last frame (one subsequence per batch, "
                                                        f"frames 0..{total_frames - 1} from an empty map) against the CPU oracle's replay of the same frames, "
                                                        "run on host cores beside the GPU legs; NaN-canonical SHA-256 of the whole map"
                                                        if timed_rows else "skipped (more than 1600 frames per subsequence, or --no-verify)"},
                               "form": f"{n_bat} batches of {len(groups_b[0])} subsequences in flight at once, one host thread and stream per batch (the timed region's form)",
                               "frames_per_subsequence": n_v, "checked": checked,
                               "against": "oracle/liboracle_port.so (C restatement of fusion_functions.cpp + surfel_map.cpp:1077-1109), NaN-canonical bytes of the whole map"}
        for bt_ in bts:
            bt_.close()
        for h_ in hs:
            h_.close()

    if leg_on("streamed") and args.mode == "batched":
        # Frames arriving from the HOST (what a KITTI replay does: the reference receives every frame through image_input /
        # depth_input, surfel_map.cpp:83-101) instead of a scene period resident in HBM: the same batched replay with 2 x C
        # frame slots per subsequence, chunk k + 1 sent up from page-locked memory (dsm_frame_upload_async, one transfer
        # per plane) while chunk k is being fused; maps resident.  The resident headline needs 29 k x 2.37 MB = 69 GB/s of
        # input -- more than the PCIe link carries -- so this leg is what a real log can sustain.
        C_s, k_s, w_s = 16, min(K, 10), 3
        hs = [api.FusionFunctions.from_camera(cam, device=device, frame_slots=2 * C_s, surfel_capacity=capacity, pipeline_depth=1) for _ in range(B)]
        pins = []
        for sc in range(n_scene):
            pf = api.PinnedFrames(hs[0], period, tight=args.streamed_rows == "tight")
            for i, (img, dep) in enumerate(rendered[sc]):
                pf.set(i, img, dep)
            pins.append(pf)
        for h_ in hs:
            h_.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        bts = [api.Batch([hs[b] for b in grp]) for grp in groups_b]
        n_chunks = (w_s + k_s) * F // C_s

        def send(g, k):  # chunk k of every subsequence of batch g -> slot half k & 1, as runs of consecutive frames of the period
            for b in groups_b[g]:
                pf, i = pins[scene_of[b]], 0
                while i < C_s:
                    t = (k * C_s + i + phase_of[b]) % period
                    run_n = min(C_s - i, period - t)
                    hs[b].frames_upload_async((k & 1) * C_s + i, pf, t, run_n)
                    i += run_n

        def chunk_plan(b, k):
            lo, hi = k * C_s, (k + 1) * C_s
            return (np.ascontiguousarray([(k & 1) * C_s + i for i in range(C_s)], np.int32), plans[b][1][lo:hi], plans[b][2][lo:hi])

        def stream_batch(g, k0, k1):
            for k in range(k0, k1):
                if k + 1 < n_chunks:
                    send(g, k + 1)  # BEFORE chunk k is enqueued: ordered behind chunk k - 1, whose slots it overwrites
                sb, rb, pb, nn = api.Batch.pack([chunk_plan(b, k) for b in groups_b[g]])
                bts[g].replay_enqueue(sb, rb, pb, nn)

        # what the link alone delivers with these transfers (no kernels running): one subsequence's chunk, 20 times
        for h_ in hs[:2]:
            h_.frames_upload_async(0, pins[0], 0, C_s)
            h_.frame_uploads_wait()
        t_l = time.perf_counter()
        for _ in range(20):
            for h_ in hs[:2]:  # two subsequences: one per upload stream of the device
                h_.frames_upload_async(0, pins[0], 0, C_s)
        for h_ in hs[:2]:
            h_.frame_uploads_wait()
        link_GBps = 40 * C_s * pins[0].pitch * cam.height * 5 / (time.perf_counter() - t_l) / 1e9
        for g in range(n_bat):
            send(g, 0)
        k_w = w_s * F // C_s
        list(pool.map(lambda g: stream_batch(g, 0, k_w), range(n_bat)))
        for bt_ in bts:
            bt_.synchronize()
        t_s = time.perf_counter()
        list(pool.map(lambda g: stream_batch(g, k_w, n_chunks), range(n_bat)))
        for bt_ in bts:
            bt_.synchronize()
        dt_s = time.perf_counter() - t_s
        fps_s = B * (n_chunks - k_w) * C_s / dt_s
        frame_bytes = pins[0].pitch * cam.height * 5  # what one frame moves over the link: image + depth rows, tight or at the slots' pitch
        out["streamed_input"] = {"value": round(fps_s, 1), "unit": "frames/s", "pcie_GBps": round(fps_s * frame_bytes / 1e9, 2),
                                 "link_alone_GBps": round(link_GBps, 2), "link_alone_frames_per_s": round(link_GBps * 1e9 / frame_bytes, 1),
                                 "bytes_per_frame": int(frame_bytes), "fraction_of_resident_rate": round(fps_s / fps, 3),
                                 "host_rows": args.streamed_rows + (" (width elements apart: no pad bytes cross the link; the upload sets them to the slots' pitch on the device)"
                                                                      if args.streamed_rows == "tight" else " (the slots' own pitch: one transfer per plane straight into the slots)"),
                                 "frame_slots_per_subsequence": 2 * C_s, "chunk_frames": C_s, "subsequences": B, "steps": k_s,
                                 "mean_live_surfels": round(float(np.mean([h_.map_size() for h_ in hs]))),
                                 "note": "the headline's batched replay with the frames streamed from page-locked host memory "
                                         "(dsm_frame_upload_async on the device's upload stream, double-buffered in chunks) instead of "
                                         "resident in HBM; maps resident; parity: tests/test_gpu_scale.py::test_long_sequence_streamed_input"}
        for bt_ in bts:
            bt_.close()
        for h_ in hs:
            h_.frame_uploads_wait()
            h_.close()
        for pf in pins:
            pf.close()

    if leg_on("sharded_replay"):
        # BASELINE configs[2]'s own driver on one rank (densesurfelmapping_amd/replay.py: what every rank of
        # `python -m densesurfelmapping_amd.replay --gpus G` runs on its shard): ONE sequence through one handle at pipeline
        # depth 24, frames streamed from page-locked host memory in chunks of 48 beside the kernels of the chunk before.
        # Pre-rendered synthetic frames, so that the engine is measured and not numpy: once with the source's frames
        # already page-locked (no host copy at all), once through the prefetch thread that copies every frame into the
        # engine's page-locked blocks (what a decoded KITTI log goes through).
        from densesurfelmapping_amd import replay as rp
        n_w, n_t = 480, 2880
        res = {}
        for label in ("page_locked_source_first_engine", "page_locked_source", "prefetch_thread_copy"):
            src = rp.SyntheticSource(n_w + n_t, camera="KITTI_1226", seed=12345, prerender=True)
            if label == "prefetch_thread_copy":
                src.pinned_run = None  # (the generic path: frames() only)
            src.prepare(0, n_w + n_t)  # (poses up front, as a log's are)
            eng = rp.HipEngine(cam, device=device, capacity=capacity, pipeline_depth=24, chunk=48)
            eng.replay(src, 0, n_w)
            eng.replay(src, n_w, n_w + n_t, origin=0)
            st = eng.stats
            res[label] = {"frames_per_s": round(st["frames"] / st["seconds"], 1),
                          "host_to_device_GBps": round(st["frames"] * st["bytes_per_frame"] / st["seconds"] / 1e9, 2),
                          "final_surfels": eng.ff.map_size()}
            eng.close()
            src.close()
        out["sharded_replay"] = {"value": res["page_locked_source"]["frames_per_s"], "unit": "frames/s per rank", "frames": n_t,
                                 "pipeline_depth": 24, "chunk_frames": 48, **res,
                                 "note": "one rank of BASELINE configs[2] (replay.HipEngine.replay): dsm_replay_enqueue_host per chunk of 48 frames -- every group "
                                         "of eight frames uploaded on the stream that runs its superpixel stages, in front of them; the FIRST streaming engine "
                                         "of a process runs below the ones after it for a few thousand frames (one-time costs of the runtime: "
                                         "profiles/r06_streaming.md), so the same replay is measured on a first engine and on a second, `value` = the second; "
                                         "parity per shard: tests/test_gpu_parity.py::test_sharded_replay_*, test_replay_engine_streams_in_chunks"}

    if leg_on("kitti_like") and args.mode == "batched":
        # The reference's REAL input distribution through the timed form: the same batched replay on frames
        # as kitti_publisher makes them -- depth = 386.1448 / disparity with the disparity in steps of 1/16 and +inf where
        # it is 0, eight grey levels, saturated highlights.  A tenth of the first sweep's pixels then lie exactly between
        # two seeds that the fp32 cost filter cannot tell apart (k_assign's list of open picks), most seeds of the sky end
        # with an infinite mean depth, and the maps fill with non-finite surfels as the reference's do.
        def make_handle_k(b):
            ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=period, surfel_capacity=capacity, pipeline_depth=1)
            for i, (img, dep) in enumerate(rendered_k[b % n_scene_k]):
                ff.frame_upload(i, img, dep)
            ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
            return ff

        def plan_k(b, n):
            ph = (b // n_scene_k) * 7 % period
            return api.FusionFunctions.pack_replay([(t + ph) % period for t in range(n)], [t // 5 for t in range(n)],
                                                   np.stack([scenes_k[b % n_scene_k].pose(t + ph) for t in range(n)]))
        k_k, w_k = min(K, 10), 3
        n_k = (w_k + k_k) * F
        hs = [make_handle_k(b) for b in range(B)]
        pl_k = [plan_k(b, n_k) for b in range(B)]
        bts = [api.Batch([hs[b] for b in grp]) for grp in groups_b]

        def run_k(lo, hi):
            def one(g):
                for c0 in range(lo, hi, 64):
                    c1 = min(hi, c0 + 64)
                    sb, rb, pb, nn = api.Batch.pack([(pl_k[b][0][c0:c1], pl_k[b][1][c0:c1], pl_k[b][2][c0:c1]) for b in groups_b[g]])
                    bts[g].replay_enqueue(sb, rb, pb, nn)
            list(pool.map(one, range(n_bat)))
            for bt_ in bts:
                bt_.synchronize()
        run_k(0, w_k * F)
        t_k = time.perf_counter()
        run_k(w_k * F, n_k)
        dt_k = time.perf_counter() - t_k
        fps_k = B * k_k * F / dt_k
        sizes_k = [h_.map_size() for h_ in hs]
        nonfinite = int(sum((~np.isfinite(m_[f])).sum() for m_ in [hs[0].map_download()] for f in m_.dtype.names if m_[f].dtype.kind == "f"))
        for bt_ in bts:
            bt_.close()
        # the sweep kernels of ONE batch alone on the GPU, by HIP events (as `roofline` does for the default scenes)
        nb = len(groups_b[0])
        bt = api.Batch(hs[:nb])  # (any nb handles: all are in step)
        sb, rb, pb, nn = api.Batch.pack([(pl_k[b][0][:24], pl_k[b][1][:24], pl_k[b][2][:24]) for b in range(nb)])
        stk, _ = bt.replay_timed(sb, rb, pb, nn)
        ovk = bt.event_overhead_ms * 1e3
        perk = {k: max(v[0] / max(v[1], 1) * 1e3 - ovk, 0.0) for k, v in stk.items()}
        bt.close()
        for h_ in hs:
            h_.close()
        open_picks = None
        try:
            open_picks = json.load(open(os.path.join(ROOT, "profiles", "r06_open_picks.json")))
        except (OSError, ValueError):
            pass
        out["kitti_like"] = {"value": round(fps_k, 1), "unit": "frames/s", "fraction_of_headline": round(fps_k / (fps / world), 3),
                             "subsequences": B, "steps": k_k, "mean_live_surfels": round(float(np.mean(sizes_k))),
                             "nonfinite_fields_in_map_0": nonfinite,
                             "batched_kernel_us": {k: round(v, 2) for k, v in perk.items()},
                             "assign_us_per_launch": [round(perk.get(k, 0.0), 2) for k in ("assign_0", "assign_1", "assign_2")],
                             "assign_us_per_launch_default_scenes": ([out["batched_kernel_us"].get(k) for k in ("assign_0", "assign_1", "assign_2")]
                                                                     if "batched_kernel_us" in out else None),
                             "open_picks": open_picks,
                             "input": "depth = float32(386.1448) / disparity, disparity = round(16 bf / z) / 16, 0 (-> +inf) on holes and sky "
                                      "(kitti_publisher/scripts/publisher.py:37-40); image quantised to 8 grey levels, 255 above 150",
                             "note": f"the headline's form ({n_bat} batches of {nb} in flight) on {n_scene_k} stereo scenes; parity of this input "
                                     "family: tests/test_gpu_scale.py [stereo_inf, stereo_zero] against reference-TU vectors"}

    if tum_on and args.mode == "batched":
        # BASELINE configs[3] through the timed form: the headline's batched replay -- 128 subsequences, four
        # batches of 32 in flight -- at 640x480 under the RGB-D constant set (fusion_functions.h:17-21) on TUM-style frames
        # (Kinect-quantised uint16 / 5000 depth: a hundred-odd distinct values per frame; zero in the projector's shadows,
        # in blobs, beyond 0.4-5 m and at the border; a hand-held closed loop, a keyframe every 4 frames), beside the SAME
        # room and trajectory through an ideal sensor; then the live callback (host frame in, one graph replay, wait) on
        # the same frames.
        n_seed_v = (cam_v.width // 8) * (cam_v.height // 8)

        def family(rendered_f, scenes_f, label):
            def mk(b):
                ff = api.FusionFunctions.from_camera(cam_v, device=device, frame_slots=period_t, surfel_capacity=1 << 20, pipeline_depth=1)
                for i, (img, dep) in enumerate(rendered_f[b % n_scene_k]):
                    ff.frame_upload(i, img, dep)
                ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
                return ff

            def plan_f(b, n):
                ph = (b // n_scene_k) * 13 % period_t
                return api.FusionFunctions.pack_replay([(t + ph) % period_t for t in range(n)], [t // 4 for t in range(n)],
                                                       np.stack([scenes_f[b % n_scene_k].pose(t + ph) for t in range(n)]))
            k_f, w_f = min(K, 10), 3
            n_f = (w_f + k_f) * F
            hs = [mk(b) for b in range(B)]
            pl = [plan_f(b, n_f) for b in range(B)]
            bts = [api.Batch([hs[b] for b in grp]) for grp in groups_b]

            def run_f(lo, hi):
                def one(g):
                    for c0 in range(lo, hi, 64):
                        c1 = min(hi, c0 + 64)
                        sb, rb, pb, nn = api.Batch.pack([(pl[b][0][c0:c1], pl[b][1][c0:c1], pl[b][2][c0:c1]) for b in groups_b[g]])
                        bts[g].replay_enqueue(sb, rb, pb, nn)
                list(pool.map(one, range(n_bat)))
                for bt_ in bts:
                    bt_.synchronize()
            run_f(0, w_f * F)
            t_f = time.perf_counter()
            run_f(w_f * F, n_f)
            dt_f = time.perf_counter() - t_f
            tiers = [hs[b].debug_tier_counts() for b in range(0, B, max(1, B // 8))]  # the last frame of eight of the subsequences
            sizes_f = [h_.map_size() for h_ in hs]
            for bt_ in bts:
                bt_.close()
            nb = len(groups_b[0])
            bt = api.Batch(hs[:nb])
            sb, rb, pb, nn = api.Batch.pack([(pl[b][0][:24], pl[b][1][:24], pl[b][2][:24]) for b in range(nb)])
            st_f, _ = bt.replay_timed(sb, rb, pb, nn)
            ov_f = bt.event_overhead_ms * 1e3
            per_f = {k: max(v[0] / max(v[1], 1) * 1e3 - ov_f, 0.0) for k, v in st_f.items()}
            bt.close()
            for h_ in hs:
                h_.close()
            mean = lambda key, i=None: round(float(np.mean([(t_[key] if i is None else t_[key][i]) for t_ in tiers])), 1)  # noqa: E731
            return {"value": round(B * k_f * F / dt_f, 1), "unit": "frames/s", "subsequences": B, "steps": k_f, "input": label,
                    "mean_live_surfels": round(float(np.mean(sizes_f))),
                    "batched_kernel_us": {k: round(v, 2) for k, v in per_f.items()},
                    "batched_frame_kernel_sum_us_per_frame": round(sum(per_f.values()) / nb, 1),
                    "second_tier_seeds_per_frame": {"of_seeds": n_seed_v,
                                                    "huber_rest_by_sweep": [mean("huber_rest_by_sweep", i) for i in range(3)],
                                                    "long_list_by_sweep": [mean("long_list_by_sweep", i) for i in range(3)],
                                                    "fit_long_groups_of_4": mean("fit_long_groups")}}
        res_t = family(rendered_t[:n_scene_k], scenes_t, "Kinect-quantised uint16 / 5000 depth, shadows, blobs, 0.4-5 m")
        res_s = family(rendered_t[n_scene_k:], scenes_ts, "the same room and trajectory through an ideal sensor (float depth, 2 % holes)")
        # the live callback on the TUM frames: pageable host frame in, one graph replay, wait (30 Hz budget: 33 ms)
        ff = api.FusionFunctions.from_camera(cam_v, device=device, frame_slots=2, surfel_capacity=1 << 20)
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        lat = []
        for t in range(300):
            img_v, dep_v = rendered_t[0][t % period_t]
            t_l = time.perf_counter()
            ff.frame_upload(t & 1, img_v, dep_v)
            ff.fuse_frame_resident(t & 1, t // 4, scenes_t[0].pose(t))
            ff.synchronize()
            if t >= 50:
                lat.append(time.perf_counter() - t_l)
        lat = np.array(lat) * 1e3
        live_map = ff.map_size()
        ff.close()
        open_picks = None
        try:
            open_picks = json.load(open(os.path.join(ROOT, "profiles", "r06_open_picks.json")))["families"].get("tum_like (640x480, RGB-D constants)")
        except (OSError, ValueError, KeyError):
            pass
        out["tum_like"] = dict(res_t, ratio_to_ideal_sensor=round(res_t["value"] / res_s["value"], 3), ideal_sensor=res_s,
                               live_callback={"latency_ms_p50": round(float(np.median(lat)), 3), "latency_ms_p99": round(float(np.percentile(lat, 99)), 3),
                                              "frames": int(len(lat)), "map_surfels": live_map},
                               open_picks=open_picks,
                               note=f"BASELINE configs[3]'s input family in the headline's form ({n_bat} batches of {len(groups_b[0])} in flight, 640x480, RGB-D "
                                    "constants, keyframe every 4); parity: tests/test_gpu_scale.py [tum_room, tum_sparse] against reference-TU vectors")

    if leg_on("bounded_map") and args.mode == "batched":
        # The headline replay never lets a keyframe leave the window, so its maps grow without bound and 83 % of B_alg is
        # the 88 B/surfel map term.  The node keeps the surfels of the ~10 drift-free keyframes active (SM.cpp:154,
        # 1456-1595: move_add_surfels): the same batched replay with every handle's keyframes older than 10 moved to its
        # inactive store between steps (untimed: the node does it on the host's schedule) keeps M near 70 k.
        hs = [make_handle(b, pipeline_depth=1) for b in range(B)]
        bts = [api.Batch([hs[b] for b in grp]) for grp in groups_b]
        next_key = [0] * B
        k_b, w_b = min(K, 10), 3

        def trim(t_first):
            for b in range(B):
                last = t_first // 5 - 10
                while next_key[b] <= last:
                    hs[b].store_deactivate(next_key[b])
                    next_key[b] += 1

        def step_b(i):
            lo, hi = i * F, (i + 1) * F

            def one(g):
                sb, rb, pb, nn = api.Batch.pack([(plans[b][0][lo:hi], plans[b][1][lo:hi], plans[b][2][lo:hi]) for b in groups_b[g]])
                bts[g].replay_enqueue(sb, rb, pb, nn)
            list(pool.map(one, range(n_bat)))
            for bt_ in bts:
                bt_.synchronize()

        spent, sizes = 0.0, []
        for i in range(w_b + k_b):
            trim(i * F)
            t_b = time.perf_counter()
            step_b(i)
            if i >= w_b:
                spent += time.perf_counter() - t_b
                sizes.append(float(np.mean([h_.map_size() for h_ in hs])))
        fps_b = B * F * k_b / spent
        m_b = float(np.mean(sizes))
        b_alg_b = 9 * n_pix + 60 * n_seed + 88 * m_b + 44 * (k_avg or 1400.0)
        out["bounded_map"] = {"value": round(fps_b, 1), "unit": "frames/s", "mean_live_surfels": round(m_b), "steps": k_b,
                              "e2e_algorithmic_GBps": round(fps_b * b_alg_b / 1e9, 2),
                              "e2e_hbm_frac_bounded_map": round(fps_b * b_alg_b / 1e9 / HBM_PEAK_GBS, 5),
                              "note": "same batched replay; between steps every handle moves the keyframes older than 10 to its "
                                      "inactive store (dsm_store_deactivate, untimed), as the node's drift-free window does"}
        for bt_ in bts:
            bt_.close()
        for h_ in hs:
            h_.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cam, scenes[0], rendered[0], period, lo_t, hi_t)
        if k_avg is None:
            k_avg = out["cpu_baseline"]["mean_new_surfels"]

    if k_avg is None:
        k_avg = 1400.0  # no measurement in this run (ranks > 0 / --no-roofline --no-cpu-baseline): a typical value
        out["config"]["mean_new_surfels_assumed"] = k_avg
    b_alg_frame = 9 * n_pix + 60 * n_seed + 88 * m_avg + 44 * k_avg
    out["e2e_algorithmic_GBps"] = round(fps * b_alg_frame / 1e9, 2)
    out["e2e_hbm_frac"] = round(fps * b_alg_frame / 1e9 / (HBM_PEAK_GBS * world), 5)
    # what the end-to-end fraction is made of: 88 B per live surfel of a map this replay lets grow
    # without bound (no keyframe ever leaves the window), and the superpixel stages' own 9N + 60S
    out["e2e_hbm_frac_note"] = (f"growing map: mean {round(m_avg)} live surfels, {round(100 * 88 * m_avg / b_alg_frame)} % of B_alg is the 88 B/surfel "
                                "map term; see bounded_map for the window the node keeps")
    out["superpixel_stage_hbm_frac"] = round(fps * (9 * n_pix + 60 * n_seed) / 1e9 / (HBM_PEAK_GBS * world), 5)
    if args.mode == "batched":
        out["valu_issue"] = valu_issue(fps / world, clock.ghz())
        out["clocks"] = clock.other()

    if verify_ranks:
        # one row per rank: [verdict (1 equal, 0 different, -1 no verdict), surfels of the GPU's map, surfels of the oracle's]
        row = [-1, 0, 0]
        if rank == 0:
            mine_rows = (out.get("verification") or {}).get("timed_region", {}).get("checked") or []
            if mine_rows:
                row = [int(all(r["equal"] for r in mine_rows)), mine_rows[0]["surfels"], mine_rows[0]["oracle_surfels"] or 0]
        elif oracle_jobs:
            try:
                so, _ = oracle_jobs[0].communicate(timeout=300.0)
                rec = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
                n_gpu, sha_gpu = end_maps[0]
                row = [int(rec["surfels"] == n_gpu and rec["sha256"] == sha_gpu), n_gpu, rec["surfels"]]
            except (subprocess.TimeoutExpired, IndexError, ValueError):
                oracle_jobs[0].kill()
            oracle_jobs = []
        mine = torch.tensor(row, dtype=torch.int64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        if rank == 0:
            rows = [[int(v) for v in t.tolist()] for t in every]
            out["verification"]["ranks"] = [{"rank": r, "subsequence": 0, "frames": total_frames, "surfels": v[1], "oracle_surfels": v[2],
                                             "equal": None if v[0] < 0 else bool(v[0])} for r, v in enumerate(rows)]
            out["verified"] = bool(out.get("verified")) and all(v[0] == 1 for v in rows)
            out["verified_timed_region"] = bool(out.get("verified_timed_region")) and all(v[0] == 1 for v in rows)
    for job in oracle_jobs:  # (the verification block did not run: e.g. --mode streams)
        job.kill()
    for ff in handles:
        ff.close()
    close_group(dist, world)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
