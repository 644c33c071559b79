"""Offline replay of an image/depth/pose sequence, sharded one subsequence per GPU (BASELINE configs[2]).

The per-frame path has a strict temporal dependency (frame t+1 fuses into the map frame t produced,
surfel_map.cpp:161), so a single sequence does not shard; independent subsequences do (SURVEY.md
§8(e)).  Each rank owns one GPU and one handle, replays its contiguous subsequence with keyframe
indices restarting at 0 -- frames arrive from page-locked host memory in chunks whose transfers run beside the
kernels of the chunk before, through the library's one-sequence fast path (frame groups), the map never leaves HBM --
and the final clouds are merged with one all-gather of the counts and one
all-gather of the padded clouds (RCCL over xGMI when the backend is "nccl"; the same code runs on
"gloo" with CPU tensors in the tests).  There is no collective on the per-frame path.

    python -m densesurfelmapping_amd.replay --synthetic 4541 --gpus 8 [--out merged.npy]
    python -m densesurfelmapping_amd.replay --kitti <seq_dir> --poses poses.txt --gpus 8

`--kitti` reads the layout kitti_publisher reads (image_0/%06d.png, depth_0/%06d.npy with depth = bf / disparity,
kitti_publisher/scripts/publisher.py:31-41) and a KITTI-format pose file (densesurfelmapping_amd/kitti.py);
`--synthetic N` renders N frames of the synthetic drive of SURVEY.md §8(d).  Without a launcher, `--gpus G` starts G
ranks itself (torch.distributed.run on 127.0.0.1) after checking that G devices are visible.  Parity of a sharded run
is per subsequence: rank r's map is the reference's map of frames [a_r, b_r) fused from an empty map with keyframe
indices restarting at 0 -- NOT a slice of the one-sequence map (surfels seen on both sides of a cut are not fused
across it); tests/test_cpu.py::test_sharded_replay_gloo and tests/test_gpu_parity.py::test_sharded_replay_*
check exactly that, and that the merged cloud is the concatenation of the shards in rank order.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

SURFEL_BYTES = 44
KEYFRAME_EVERY = 5  # every 5th frame is a keyframe, reference index = latest keyframe


def shard_subsequences(n_frames: int, world_size: int):
    """Contiguous split of [0, n_frames) into world_size subsequences whose lengths differ by at
    most one (4541 KITTI frames over 8 ranks -> 5 x 568 + 3 x 567)."""
    if world_size <= 0 or n_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_frames, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def merge_clouds(local_cloud, group=None):
    """All-gather the per-rank surfel clouds.

    local_cloud: torch.uint8 tensor [n_r * 44] (the rank's SurfelElement array as bytes), on the
    device the process group communicates on.  Returns (merged uint8 tensor [sum n_r * 44] in rank
    order, list of per-rank counts)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = local_cloud.numel() // SURFEL_BYTES
    assert local_cloud.dtype == torch.uint8 and local_cloud.numel() == n_local * SURFEL_BYTES
    counts = torch.zeros(world, dtype=torch.int64, device=local_cloud.device)
    mine = torch.tensor([n_local], dtype=torch.int64, device=local_cloud.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_l = [int(c) for c in counts.tolist()]
    n_max = max(counts_l) if counts_l else 0
    if n_max == 0:
        return local_cloud.new_zeros(0), counts_l
    padded = local_cloud.new_zeros(n_max * SURFEL_BYTES)
    padded[: local_cloud.numel()] = local_cloud
    gathered = local_cloud.new_zeros(world * n_max * SURFEL_BYTES)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * n_max * SURFEL_BYTES: r * n_max * SURFEL_BYTES + counts_l[r] * SURFEL_BYTES]
             for r in range(world)]
    return torch.cat(parts), counts_l


# ---------------------------------------------------------------------------------------------- frame sources
class SyntheticSource:
    """The synthetic drive of SURVEY.md §8(d) (densesurfelmapping_amd/synth.py): frame t of the one long sequence.
    prerender=True renders the scene's period once, on worker processes (synth.render_many); the engine then keeps it in
    page-locked memory laid out like its frame slots (`pinned_run`) and a replay moves no frame on the host."""

    def __init__(self, n_frames, camera="KITTI_1226", seed=12345, prerender=False):
        from . import synth
        self.cam = getattr(synth, camera)
        self.scene = synth.Scene(seed=seed)
        self.n_frames = n_frames
        self._period = None
        self._pinned = None
        self._poses = {}
        if prerender:
            per = self.scene.frames_per_period
            self._period = synth.render_many([(self.cam, self.scene, i) for i in range(min(per, n_frames))])

    def frames(self, a, b):
        """(image uint8 [H,W], depth float32 [H,W], pose 4x4 cam->world) of frames a .. b-1"""
        from . import synth
        if self._period is not None:
            for t in range(a, b):
                image, depth = self._period[t % len(self._period)]
                yield image, depth, self.pose(t)
            return
        for t, image, depth, pose, _ in synth.sequence(self.cam, self.scene, b - a, start=a):
            yield image, depth, pose

    def pose(self, t):
        p = self._poses.get(t)
        return p if p is not None else self.scene.pose(t)

    def prepare(self, a, b):
        """the poses of frames a .. b-1, worked out once (a log's poses are an array read up front: kitti.read_poses; here they
        are trigonometry per frame, which a replay at twenty thousand frames a second would spend a third of its time on)"""
        for t in range(a, b):
            if t not in self._poses:
                self._poses[t] = self.scene.pose(t)

    def pinned_run(self, api, t, n_max):
        """frames t .. t+n-1 (n <= n_max) as a run of a page-locked block in slot layout: (block, first, n), or None when
        the source holds no pre-rendered frames.  The block is two periods back to back, so that a run may cross the end
        of a period; a run never exceeds one period."""
        if self._period is None:
            return None
        per = len(self._period)
        if self._pinned is None:
            self._pinned = api.PinnedFrames((self.cam.height, self.cam.width), 2 * per)
            for i in range(2 * per):
                self._pinned.set(i, *self._period[i % per])
        return self._pinned, t % per, min(n_max, per)

    def close(self):
        if self._pinned is not None:
            self._pinned.close()
            self._pinned = None


class KittiSource:
    """A KITTI odometry sequence directory in the layout kitti_publisher reads + a KITTI-format pose file.
    decode_workers > 0: PNG + .npy decoding on that many worker PROCESSES (fresh interpreters: a process that holds a GPU
    runtime is not forked), frames handed back in order -- one core decodes a few hundred 1226x370 frames per second at
    best, the engine fuses thirteen thousand."""

    def __init__(self, seq_dir, poses, bf=None, n_frames=None, decode_workers=0):
        from . import kitti
        self.seq_dir, self.bf = seq_dir, bf if bf else kitti.BF_SEQ_00_02
        self.poses = kitti.read_poses(poses)
        n = 0
        while n < len(self.poses) and all(os.path.isfile(p) for p in kitti.frame_paths(seq_dir, n)):
            n += 1  # (publisher.py:34 stops at the first missing file)
        self.n_frames = min(n, n_frames) if n_frames else n
        if self.n_frames == 0:
            raise FileNotFoundError(f"{seq_dir}: no image_0/000000.png + depth_0/000000.npy with a pose")
        first = kitti.read_grey(kitti.frame_paths(seq_dir, 0)[0])
        self.cam = kitti.camera_from_calib(seq_dir, first.shape[1], first.shape[0])
        self.decode_workers = int(decode_workers)
        self._pool = None

    def frames(self, a, b):
        from . import kitti
        if self.decode_workers > 0 and b - a > 1:
            import functools
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            if self._pool is None:
                self._pool = ProcessPoolExecutor(self.decode_workers, mp_context=mp.get_context("spawn"))
            load = functools.partial(kitti.load_frame, self.seq_dir, bf=self.bf)
            # a bounded window of decodes in flight (Executor.map would submit the whole range at once and buffer every result:
            # a shard of decoded frames in RAM when the consumer is slower than the workers)
            import collections
            ahead = max(2, 4 * self.decode_workers)
            pending, nxt = collections.deque(), a
            while pending or nxt < b:
                while nxt < b and len(pending) < ahead:
                    pending.append((nxt, self._pool.submit(load, nxt)))
                    nxt += 1
                i, fut = pending.popleft()
                image, depth = fut.result()
                yield image, depth, self.poses[i].astype(np.float32)
            return
        for i in range(a, b):
            image, depth = kitti.load_frame(self.seq_dir, i, self.bf)  # publisher.py:35-40
            yield image, depth, self.poses[i].astype(np.float32)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True, cancel_futures=True)  # (at most `ahead` decodes are in flight: the wait is short)
            self._pool = None


# ---------------------------------------------------------------------------------------------- the engine
class HipEngine:
    """One handle of the HIP engine (include/dsm.h) with a resident map, driven the way the library is fastest for ONE
    sequence: frame groups (pipeline depth 24: the superpixel stages of eight consecutive frames per batched launch, fuse
    + compaction in frame order on the map stream) and the frames coming WITH the enqueue call from page-locked host memory
    (dsm_replay_enqueue_host, round 6): every group of eight frames goes up on the stream that runs its superpixel stages,
    right in front of them -- no upload stream, no event between a transfer and its consumer; the transfer of one group runs
    beside the kernels of the two before it, each on a hardware queue of its own.  (Until round 6 the frames went ahead on
    the device's upload stream, ordered by events against the pipelines that share its hardware queue: 11-14 k frames/s,
    +- 20 % with the queue the stream happened to be given.)  Frames reach page-locked blocks on a prefetch thread (decode /
    render / copy; FOUR blocks of `chunk` frames in turn: one being filled, three in flight) unless the source already keeps
    them there (`pinned_run`).  `fuse` is the frame-at-a-time form of the same thing (SurfelMap::fuse_map,
    surfel_map.cpp:1060-1113).  There is no other engine in this package: without a gfx950 device the constructor raises
    (DSM_E_NO_DEVICE)."""

    BLOCKS = 4  # page-locked blocks of `chunk` frames the prefetch thread fills in turn (one being filled, up to three in flight)

    def __init__(self, cam, device=0, capacity=0, pipeline_depth=24, chunk=48):
        from . import api
        self._api = api
        self.chunk = max(1, int(chunk))
        self.depth = int(pipeline_depth) if pipeline_depth else 4
        self.ff = api.FusionFunctions.from_camera(cam, device=device, frame_slots=max(self.depth, 2), surfel_capacity=capacity,
                                                  pipeline_depth=pipeline_depth)
        self.ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        self.n = 0
        self._pins = None
        self.stats = {}

    def fuse(self, image, depth, pose, ref_idx):  # one frame: blocking upload into a slot, one enqueue
        slot = self.n % max(self.depth, 2)
        self.ff.frame_upload(slot, image, depth)
        self.ff.fuse_frame_resident(slot, ref_idx, pose)
        self.n += 1

    def replay(self, source, a, b, keyframe_every=KEYFRAME_EVERY, origin=None):
        """frames [a, b) of `source`, streamed in chunks (see the class comment).  Keyframe indices count from frame `origin`
        (default: a -- they restart at 0 with every call; a caller that replays one subsequence in several calls passes the
        subsequence's first frame)"""
        origin = a if origin is None else origin
        import queue
        import threading
        api, ff, C = self._api, self.ff, self.chunk
        n_total = b - a
        chunks = [(a + c0, min(C, n_total - c0)) for c0 in range(0, n_total, C)]  # (first frame, frames)
        zero_copy = getattr(source, "pinned_run", None) is not None and source.pinned_run(api, a, 1) is not None
        ready = queue.Queue()
        free = threading.Semaphore(self.BLOCKS)
        stop = threading.Event()
        if not zero_copy and self._pins is None:
            self._pins = [api.PinnedFrames(ff, C) for _ in range(self.BLOCKS)]

        def produce():  # decode / render / copy chunk k into page-locked block k mod BLOCKS, as soon as that block is free
            try:
                for k, (t0, n) in enumerate(chunks):
                    free.acquire()
                    if stop.is_set():
                        return
                    images, depths, poses = [], [], []
                    for image, depth, pose in source.frames(t0, t0 + n):
                        images.append(image)
                        depths.append(depth)
                        poses.append(pose)
                    # one call for the chunk: the library's host threads copy it (dsm_host_pack_frames), no GIL held
                    self._pins[k % self.BLOCKS].set_many(0, images, depths)
                    ready.put((k, poses, None))
                ready.put(None)
            except BaseException as e:  # noqa: BLE001 -- handed to the consumer
                ready.put((None, None, e))

        th = None
        if not zero_copy:
            th = threading.Thread(target=produce, daemon=True)
            th.start()
        t_start = time.perf_counter()
        try:
            for k, (t0, n) in enumerate(chunks):
                refs = np.array([(t0 - origin + i) // keyframe_every for i in range(n)], np.int32)
                if zero_copy:
                    i = 0
                    while i < n:  # runs of the source's page-locked block (a run ends where its period does)
                        pf, first, m = source.pinned_run(api, t0 + i, n - i)
                        poses = np.stack([api.pose_to_colmajor(source.pose(t)) for t in range(t0 + i, t0 + i + m)])
                        ff.replay_enqueue_host(pf, first, refs[i:i + m], poses)
                        i += m
                else:
                    item = ready.get()
                    if item is None or item[2] is not None:
                        raise item[2] if item else RuntimeError("frame source ended early")
                    assert item[0] == k
                    poses = np.stack([api.pose_to_colmajor(p) for p in item[1]])
                    ff.replay_enqueue_host(self._pins[k % self.BLOCKS], 0, refs, poses)
                    if k >= self.BLOCKS - 1:
                        # once chunk k - (BLOCKS - 1) has been fused its block may be refilled -- with chunk k + 1; the chunks
                        # in between stay in flight
                        ff.replay_wait(self.BLOCKS - 1)
                        free.release()
            ff.synchronize()
        finally:
            stop.set()
            for _ in range(self.BLOCKS):
                free.release()
            if th is not None:
                th.join(timeout=60)
        self.n += n_total
        dt = time.perf_counter() - t_start
        self.stats = {"frames": n_total, "seconds": dt, "chunk_frames": C, "zero_copy": bool(zero_copy),
                      "bytes_per_frame": int(ff.frame_pitch() * ff.height * 5)}
        return n_total

    def cloud(self):
        """the final map as a numpy array of 44-byte records (synchronises)"""
        return self.ff.map_download()

    def cloud_tensor(self, torch, device):
        """the final map as bytes in device memory (no host trip), for the RCCL merge"""
        n = self.ff.map_size()
        t = torch.empty(max(n, 1) * SURFEL_BYTES, dtype=torch.uint8, device=device)
        got = self.ff.map_copy_to_device(t.data_ptr(), max(n, 1))
        return t[: got * SURFEL_BYTES]

    def close(self):
        self.ff.frame_uploads_wait()
        self.ff.close()
        for pf in self._pins or []:
            pf.close()
        self._pins = None


def init_collective(backend, world, rank, device):
    """The process group of the final merge.  world > 1: the launcher's rendezvous (MASTER_ADDR / MASTER_PORT).  world == 1:
    a group of one on a loopback port of its own -- the merge then runs through the same library calls (RCCL's
    ncclCommInitRank + ncclAllGather when the backend is "nccl") as on a node, on the one GPU there is."""
    import torch
    import torch.distributed as dist
    kw = {}
    if world == 1 and "MASTER_ADDR" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            kw = {"init_method": f"tcp://127.0.0.1:{sk.getsockname()[1]}", "rank": 0, "world_size": 1}
    if backend == "nccl":  # RCCL over xGMI
        dist.init_process_group("nccl", device_id=torch.device("cuda", device), **kw)
    else:
        dist.init_process_group(backend, **kw)
    return dist


def replay_shard(engine, source, a, b, keyframe_every=KEYFRAME_EVERY):
    """Frames [a, b) of `source` through `engine`, keyframe indices restarting at 0."""
    if hasattr(engine, "replay"):  # the HIP engine streams the shard in chunks
        return engine.replay(source, a, b, keyframe_every)
    for k, (image, depth, pose) in enumerate(source.frames(a, b)):
        engine.fuse(image, depth, pose, k // keyframe_every)
    return b - a


def run_rank(source, rank, world, *, engine_factory=None, backend="nccl", device=0, save_shards=None, out=None, group=None,
             engine_options=None):
    """What one rank of a sharded replay does; returns the summary (rank 0's carries the merged cloud's digest).
    engine_factory(cam) -> engine: the tests' CPU stand-in goes in here; the default is the HIP engine."""
    import torch
    shards = shard_subsequences(source.n_frames, world)
    a, b = shards[rank]
    engine = engine_factory(source.cam) if engine_factory else HipEngine(source.cam, device=device, **(engine_options or {}))
    t0 = time.perf_counter()
    n = replay_shard(engine, source, a, b)
    on_device = backend == "nccl" and hasattr(engine, "cloud_tensor")
    if on_device:
        cloud = engine.cloud_tensor(torch, f"cuda:{device}")  # (synchronises)
    else:
        mine = np.ascontiguousarray(engine.cloud())
        cloud = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
    t_replay = time.perf_counter() - t0
    if save_shards:
        os.makedirs(save_shards, exist_ok=True)
        np.save(os.path.join(save_shards, f"shard_{rank}.npy"), engine.cloud())
    t1 = time.perf_counter()
    import torch.distributed as dist
    if world > 1 or (dist.is_available() and dist.is_initialized()):  # (a group of one: --merge-at-world1)
        merged, counts = merge_clouds(cloud, group)
    else:
        merged, counts = cloud, [cloud.numel() // SURFEL_BYTES]
    merged_np = merged.cpu().numpy()
    t_merge = time.perf_counter() - t1
    stream_stats = dict(getattr(engine, "stats", None) or {})
    if hasattr(engine, "close"):
        engine.close()
    if hasattr(source, "close"):
        source.close()
    summary = {"rank": rank, "world": world, "frames": [a, b], "replayed": n, "surfels": counts[rank], "replay_s": round(t_replay, 3),
               "frames_per_s": round(n / t_replay, 1) if t_replay > 0 else None, "merge_s": round(t_merge, 4),
               "backend": (dist.get_backend(group) if dist.is_available() and dist.is_initialized() else None)}
    if stream_stats.get("seconds"):  # the streamed replay alone (without the final download of the cloud)
        summary["streamed"] = {"frames_per_s": round(stream_stats["frames"] / stream_stats["seconds"], 1),
                               "host_to_device_GBps": round(stream_stats["frames"] * stream_stats["bytes_per_frame"] / stream_stats["seconds"] / 1e9, 2),
                               "chunk_frames": stream_stats["chunk_frames"], "frames_already_page_locked": stream_stats["zero_copy"]}
    if rank == 0:
        summary.update({"shards": [list(s) for s in shards], "counts": counts, "merged_surfels": int(merged_np.size // SURFEL_BYTES),
                        "merged_sha256": hashlib.sha256(merged_np.tobytes()).hexdigest()})
        if out:
            from . import api
            np.save(out, merged_np.view(api.SURFEL_DTYPE))
    return summary


def _self_launch(n_gpus, one_device):
    import socket
    import subprocess
    import torch
    seen = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not one_device and seen < n_gpus:
        print(f"replay: --gpus {n_gpus} needs {n_gpus} visible GPUs, this box shows {seen}; nothing was replayed", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "densesurfelmapping_amd.replay"] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    ap = argparse.ArgumentParser(description="sharded offline replay: one contiguous subsequence per GPU, clouds merged by all-gather")
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--kitti", metavar="SEQ_DIR", help="KITTI odometry sequence directory in kitti_publisher's layout")
    src.add_argument("--synthetic", type=int, metavar="N", help="N frames of the synthetic drive")
    ap.add_argument("--poses", help="KITTI-format pose file (12 numbers per line), with --kitti")
    ap.add_argument("--bf", type=float, help="baseline x focal: depth = bf / disparity (default 386.1448; 379.8145 for sequences 04-12)")
    ap.add_argument("--frames", type=int, help="use only the first N frames of the KITTI sequence")
    ap.add_argument("--decode-workers", type=int, default=0, help="KITTI: decode PNG / .npy frames on this many worker processes per rank (0 = in the prefetch thread)")
    ap.add_argument("--camera", default="KITTI_1226", help="synthetic camera (densesurfelmapping_amd.synth)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--prerender", action="store_true", help="synthetic: render the scene's period up front and keep it page-locked (a replay then measures the engine, not numpy)")
    ap.add_argument("--pipeline-depth", type=int, default=24, help="frames of the sequence in flight (frame groups; 1 = strictly serial)")
    ap.add_argument("--chunk", type=int, default=48, help="frames per host-to-device transfer")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="nccl = RCCL over xGMI; gloo moves the clouds through host memory")
    ap.add_argument("--one-device", action="store_true", help="all ranks on cuda:0 (a one-GPU box; needs --backend gloo)")
    ap.add_argument("--merge-at-world1", action="store_true", help="--gpus 1: run the final merge through a process group of one (RCCL on the one GPU) instead of skipping it")
    ap.add_argument("--out", help="rank 0 writes the merged cloud here (.npy of 44-byte SurfelElement records)")
    ap.add_argument("--save-shards", metavar="DIR", help="every rank writes its own map to DIR/shard_<rank>.npy")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        sys.exit("replay: --gpus must be >= 1")
    if args.kitti and not args.poses:
        sys.exit("replay: --kitti needs --poses")
    if args.one_device and args.gpus > 1 and args.backend != "gloo":
        sys.exit("replay: --one-device puts every rank on one GPU, which RCCL cannot do: use --backend gloo")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus, args.one_device))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit(f"replay: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    device = 0 if (world == 1 or args.one_device) else int(os.environ.get("LOCAL_RANK", "0"))
    source = KittiSource(args.kitti, args.poses, args.bf, args.frames, args.decode_workers) if args.kitti else SyntheticSource(args.synthetic, args.camera, args.seed, prerender=args.prerender)
    import torch
    torch.cuda.set_device(device)
    grouped = world > 1 or args.merge_at_world1
    if grouped:
        dist = init_collective(args.backend, world, rank, device)
    summary = run_rank(source, rank, world, backend=args.backend, device=device, save_shards=args.save_shards, out=args.out,
                       engine_options={"pipeline_depth": args.pipeline_depth, "chunk": args.chunk})
    if grouped:
        rows = [None] * world
        dist.all_gather_object(rows, summary)
        dist.destroy_process_group()
    else:
        rows = [summary]
    if rank == 0:
        head = dict(rows[0])
        head["per_rank"] = [{k: r.get(k) for k in ("rank", "frames", "surfels", "replay_s", "frames_per_s", "streamed")} for r in rows]
        head["frames_per_s_all_ranks"] = round(sum(r["replayed"] for r in rows) / max(r["replay_s"] for r in rows), 1)
        print(json.dumps(head))


if __name__ == "__main__":
    main()
