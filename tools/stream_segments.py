#!/usr/bin/env python3
"""One streamed sequence through replay.HipEngine (BASELINE configs[2]'s per-rank path), timed segment by segment (GPU box):

    python tools/stream_segments.py [--segment 480] [--segments 8] [--engines 2] [--whole] [--copy]

Every engine replays the same frames from an empty map; a segment is one HipEngine.replay call (it returns when the device
has finished).  --whole: after the segments, a fresh engine replays warm-up + everything else as ONE call (what bench.py's
`sharded_replay` leg times).  --copy: frames through the prefetch thread's copies instead of the source's page-locked block."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densesurfelmapping_amd import replay as rp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segment", type=int, default=480)
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--engines", type=int, default=2)
    ap.add_argument("--whole", action="store_true")
    ap.add_argument("--copy", action="store_true")
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--chunk", type=int, default=48)
    ap.add_argument("--capacity", type=int, default=1 << 22)
    args = ap.parse_args()
    n = args.segment * args.segments
    src = rp.SyntheticSource(n, camera="KITTI_1226", seed=12345, prerender=True)
    if args.copy:
        src.pinned_run = None
    src.prepare(0, n)
    out = {"segment_frames": args.segment, "engines": []}
    for e in range(args.engines):
        eng = rp.HipEngine(src.cam, device=0, capacity=args.capacity, pipeline_depth=args.depth, chunk=args.chunk)
        rates, sizes = [], []
        for s in range(args.segments):
            eng.replay(src, s * args.segment, (s + 1) * args.segment, origin=0)
            rates.append(round(eng.stats["frames"] / eng.stats["seconds"]))
            sizes.append(eng.ff.map_size())
        out["engines"].append({"frames_per_s": rates, "map_surfels": sizes})
        eng.close()
    if args.whole:
        eng = rp.HipEngine(src.cam, device=0, capacity=args.capacity, pipeline_depth=args.depth, chunk=args.chunk)
        eng.replay(src, 0, args.segment)
        t0 = time.perf_counter()
        eng.replay(src, args.segment, n, origin=0)
        out["whole"] = {"frames": n - args.segment, "frames_per_s": round((n - args.segment) / (time.perf_counter() - t0)), "map_surfels": eng.ff.map_size()}
        eng.close()
    src.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
