"""Reader for the reference's on-disk input layout -> the node's message stream (the ROS-free replacement for
kitti_publisher + the ORB-SLAM2 pose feed).

Layout, as ``kitti_publisher/scripts/publisher.py:23-41`` reads it from a KITTI odometry sequence directory:

    <seq>/image_0/%06d.png     left grey image, published as mono8           (publisher.py:31,35,46)
    <seq>/depth_0/%06d.npy     disparity of the stereo network               (publisher.py:33,37)
    depth = 386.1448 / disparity   (sequences 00-02; 379.8145 for 04-12)     (publisher.py:38-39)

Poses: the reference receives them from its modified ORB-SLAM2 as three synchronised topics (this_pose / loop_path /
loop_stamps, ros_node.cpp:27-31).  Offline they come from a text file in the KITTI odometry / ORB-SLAM2
``SaveTrajectoryKITTI`` convention: one line per frame, the 12 row-major entries of the 3x4 cam0 -> world matrix.
Every ``keyframe_every``-th frame is announced as a keyframe (``pose.covariance[0] > 0``, surfel_map.cpp:320) with
the latest keyframe as its reference (``covariance[1]``, :337,356); the loop path is the keyframe poses so far; loop
edges can be given as ``{frame: [(kf_a, kf_b), ...]}`` together with a corrected pose file to replay a loop closure
(surfel_map.cpp:235-314).  Stamps advance 0.1 s per frame (KITTI's 10 Hz).

    python -m densesurfelmapping_amd.kitti <seq_dir> <poses.txt> run.log [--frames N] [--bf 386.1448]
    python -m densesurfelmapping_amd.msglog run.log --save-cloud map.PCD --save-mesh map_mesh.PLY

PNG decoding: Pillow if importable, else the small pure-numpy decoder below (8-bit grey / RGB / RGBA / palette-free,
non-interlaced -- what KITTI ships).
"""
from __future__ import annotations

import argparse
import os
import struct
import zlib

import numpy as np

from . import synth

BF_SEQ_00_02 = 386.1448   # publisher.py:38
BF_SEQ_04_12 = 379.8145   # publisher.py:39


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def decode_png(data: bytes) -> np.ndarray:
    """8-bit non-interlaced PNG -> uint8 [H,W] (grey) or [H,W,C]."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, ihdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.uint8)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:]
        if f == 0:
            cur = line.copy()
        elif f == 2:
            cur = line + prev
        else:  # sub / average / paeth run along the row: pixel by pixel over `ch`-strided columns
            cur = np.zeros_like(line)
            for x in range(0, w * ch, ch):
                left = cur[x - ch:x] if x else np.zeros(ch, np.uint8)
                up, ul = prev[x:x + ch], (prev[x - ch:x] if x else np.zeros(ch, np.uint8))
                if f == 1:
                    cur[x:x + ch] = line[x:x + ch] + left
                elif f == 3:
                    cur[x:x + ch] = line[x:x + ch] + ((left.astype(np.int32) + up) // 2).astype(np.uint8)
                elif f == 4:
                    cur[x:x + ch] = line[x:x + ch] + _paeth(left, up, ul)
                else:
                    raise ValueError("bad PNG filter")
        out[y] = cur
        prev = cur
    return out.reshape(h, w) if ch == 1 else out.reshape(h, w, ch)


def read_grey(path: str) -> np.ndarray:
    """cv2.imread(path, 0): 8-bit grey."""
    try:
        from PIL import Image
        return np.asarray(Image.open(path).convert("L"), dtype=np.uint8)
    except ImportError:
        img = decode_png(open(path, "rb").read())
        if img.ndim == 3:  # OpenCV's BGR2GRAY weights on RGB input
            rgb = img[..., :3].astype(np.float32)
            img = np.clip(np.floor(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2] + 0.5), 0, 255).astype(np.uint8)
        return img


def read_poses(path: str) -> np.ndarray:
    """KITTI odometry / ORB-SLAM2 SaveTrajectoryKITTI: 12 numbers per line -> [n,4,4] cam -> world."""
    rows = np.loadtxt(path, dtype=np.float64, ndmin=2)
    if rows.shape[1] != 12:
        raise ValueError(f"{path}: expected 12 numbers per line, got {rows.shape[1]}")
    out = np.tile(np.eye(4), (len(rows), 1, 1))
    out[:, :3, :] = rows.reshape(-1, 3, 4)
    return out


def frame_paths(seq_dir: str, i: int):
    return os.path.join(seq_dir, "image_0", "%06d.png" % i), os.path.join(seq_dir, "depth_0", "%06d.npy" % i)


def load_frame(seq_dir: str, i: int, bf: float = BF_SEQ_00_02):
    """frame i as kitti_publisher hands it over: (grey uint8 [H,W], depth float32 [H,W] = bf / disparity, +inf where the
    disparity is 0 -- publisher.py:35-40).  A module-level function, so that decode workers (fresh interpreters) can run it."""
    img_path, dep_path = frame_paths(seq_dir, i)
    image = read_grey(img_path)
    with np.errstate(divide="ignore"):
        depth = (bf / np.load(dep_path)).astype(np.float32)
    return image, depth


def camera_from_calib(seq_dir: str, width: int, height: int, far=30.0, near=0.5) -> synth.Camera:
    """Intrinsics from the sequence's calib.txt (P0: fx 0 cx 0 0 fy cy 0 ...), else KITTI04-12.yaml:8-11's values."""
    calib = os.path.join(seq_dir, "calib.txt")
    if os.path.exists(calib):
        for line in open(calib):
            if line.startswith("P0:"):
                p = [float(v) for v in line.split()[1:]]
                return synth.Camera(width, height, p[0], p[5], p[2], p[6], far=far, near=near)
    return synth.Camera(width, height, 707.0912, 707.0912, 601.8873, 183.1104, far=far, near=near)


def messages(seq_dir: str, poses: np.ndarray, n_frames=None, start: int = 0, bf: float = BF_SEQ_00_02, keyframe_every: int = 5,
             loop_poses: np.ndarray = None, loops=None):
    """Yield ("image", stamp, uint8[H,W]) / ("depth", stamp, float32[H,W]) / ("orb", stamp, loop_values, loop_path,
    this_pose, covariance) for the frames present in seq_dir (stops at the first missing file, as publisher.py:34 does).
    loop_poses: the same trajectory after loop closure; it replaces the loop path from the first frame in `loops` on."""
    loops = loops or {}
    kf_idx, edges, closed = [], [], False
    t = 0
    while n_frames is None or t < n_frames:
        i = start + t
        img_path, dep_path = frame_paths(seq_dir, i)
        if not (os.path.isfile(img_path) and os.path.isfile(dep_path)) or i >= len(poses):
            break
        image = read_grey(img_path)
        with np.errstate(divide="ignore"):
            depth = (bf / np.load(dep_path)).astype(np.float32)  # publisher.py:37-38; a zero disparity is an infinite depth
        stamp = (1000 + t // 10, (t % 10) * 100000000)
        is_kf = t % keyframe_every == 0
        ref_kf = max(len(kf_idx) - 1, 0)
        if is_kf:
            kf_idx.append(i)
        if t in loops:
            edges.extend(loops[t])
            closed = closed or loop_poses is not None
        src = loop_poses if closed else poses
        path = np.stack([synth.pose7(src[k]) for k in kf_idx]) if kf_idx else np.stack([synth.pose7(src[i])])
        cov = np.zeros(36)
        cov[0] = 1.0 if is_kf else 0.0
        cov[1] = float(ref_kf)
        yield ("image", stamp, image)
        yield ("depth", stamp, depth)
        yield ("orb", stamp, np.array([v for ab in edges for v in ab], dtype=np.float32), path, synth.pose7(src[i]), cov)
        t += 1


def main():
    ap = argparse.ArgumentParser(description="KITTI-layout sequence directory + pose file -> message log for densesurfelmapping_amd.msglog")
    ap.add_argument("seq_dir")
    ap.add_argument("poses")
    ap.add_argument("log")
    ap.add_argument("--frames", type=int)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--bf", type=float, default=BF_SEQ_00_02, help="baseline x focal: depth = bf / disparity (379.8145 for sequences 04-12)")
    ap.add_argument("--drift-free-poses", type=int, default=10)
    args = ap.parse_args()
    from . import msglog
    first = read_grey(frame_paths(args.seq_dir, args.start)[0])
    cam = camera_from_calib(args.seq_dir, first.shape[1], first.shape[0])
    msglog.write_log(args.log, cam, args.drift_free_poses,
                     messages(args.seq_dir, read_poses(args.poses), args.frames, args.start, args.bf))
    print(f"wrote {args.log}: {cam.width}x{cam.height}, fx {cam.fx}")


if __name__ == "__main__":
    main()
