"""Deterministic synthetic driving scene: grey image + metric depth + camera pose.

Replaces the reference's KITTI feed (``kitti_publisher/scripts/publisher.py:15-71``
publishes mono8 ``/left_image`` and 32FC1 ``/depth_image``; poses come from
ORB-SLAM2).  There is no dataset in this environment, so every test and the
benchmark draw frames from this generator.

Scene (camera convention of KITTI: x right, y down, z forward):
  * ground plane ``y = +1.65`` m, side walls ``x = ±6`` m (4.65 m tall),
  * axis-aligned boxes standing on the ground, laid out periodically along z,
  * the camera drives along +z at ``step`` m/frame with a small periodic yaw and
    lateral sway whose period equals the scene period, so frame ``t`` and frame
    ``t + frames_per_period`` see the same image from a pose shifted by exactly
    one period: an arbitrarily long replay needs only one period of images.

All randomness is a counter-based 32-bit integer hash of (pixel, frame, seed),
so the same bytes come out on any machine regardless of numpy's RNG streams.
"""
from __future__ import annotations

import dataclasses
import numpy as np


@dataclasses.dataclass(frozen=True)
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    far: float = 30.0
    near: float = 0.5
    rgbd: bool = False  # selects the RGB-D constant set of fusion_functions.h:17-21


# Intrinsics named in SURVEY.md §8(d).
KITTI_1226 = Camera(1226, 370, 707.0912, 707.0912, 601.8873, 183.1104)   # KITTI04-12.yaml:8-11
KITTI_1241 = Camera(1241, 376, 718.856, 718.856, 607.1928, 185.2157)     # kitti_orb.launch:5-16
VGA_RGBD = Camera(640, 480, 525.0, 525.0, 319.5, 239.5, far=6.0, near=0.3, rgbd=True)
VGA_DRIVE = Camera(640, 480, 525.0, 525.0, 319.5, 239.5)
FULLHD = Camera(1920, 1080, 1400.0, 1400.0, 959.5, 539.5)
TINY = Camera(160, 96, 120.0, 120.0, 79.5, 47.5)                          # unit-test size
# (size mod 8) > 4: the rightmost / bottom pixels have no candidate superpixel (label -1, dsm_math.h has_candidate_cell)
KITTI_1242 = Camera(1242, 375, 721.5377, 721.5377, 609.5593, 172.854)    # KITTI raw, rectified (calib_cam_to_cam P_rect_00)
TINY_RAGGED = Camera(166, 103, 120.0, 120.0, 82.5, 51.0)


def _hash32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer hash, vectorised on uint32 (wrapping arithmetic)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _uniform01(idx: np.ndarray, salt: int) -> np.ndarray:
    h = _hash32(idx.astype(np.uint32) ^ np.uint32(salt & 0xFFFFFFFF))
    return (h >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


@dataclasses.dataclass
class Scene:
    seed: int = 12345
    frames_per_period: int = 50
    step: float = 0.8                # metres per frame (KITTI-like at 10 Hz)
    yaw_amp_deg: float = 1.5
    sway_amp: float = 0.25
    depth_noise: float = 0.002       # multiplicative sigma
    hole_fraction: float = 0.02
    intensity_noise: float = 20.0
    checker: float = 40.0            # +- amplitude of the 1 m world-space checker
    max_depth: float = 80.0
    n_boxes: int = 10
    scale: float = 1.0               # shrink the whole scene (RGB-D range tests)
    # ---- the reference's real feed (kitti_publisher/scripts/publisher.py:37-40): depth = bf / disparity of a stereo
    # matcher -- disparity-quantised, and `x / 0` = +inf wherever the matcher left disparity 0 (sky, occlusions) --
    # and a camera image with saturated highlights and few grey levels
    stereo: bool = False
    stereo_bf: float = 386.1448      # baseline x focal of sequences 00-02, publisher.py:38
    disparity_step: float = 0.0625   # sub-pixel resolution of the disparity map
    zero_disparity_inf: bool = True  # False: an unmatched pixel is depth 0 (a publisher that masks them)
    saturate_above: float = 0.0      # > 0: every grey value above it reads 255
    intensity_levels: int = 0        # > 0: the grey values below saturation quantised to this many levels
    # ---- a TUM-RGBD-style feed (BASELINE configs[3]; ORB_SLAM2/Examples/ROS/ORB_SLAM2/src/ros_rgbd.cc feeds the node from a
    # Kinect): a hand-held camera in a closed room, depth as the sensor + the dataset's 16-bit PNGs deliver it -- quantised
    # in disparity (1/8 pixel of a 7.5 cm x 580 px projector baseline), stored as uint16 = metres x 5000, ZERO outside
    # 0.4-5 m, in the projector's occlusion shadows beside every foreground edge, in blobs the sensor lost and in a
    # band at the right border.  The trajectory is a closed loop: pose(t) == pose(t mod frames_per_period).
    tum: bool = False
    tum_depth_scale: float = 5000.0
    tum_near: float = 0.4
    tum_far: float = 5.0
    kinect_bf: float = 43.5          # baseline x focal of the depth sensor, metre x pixel
    kinect_subpixel: float = 0.125
    blob_fraction: float = 0.012     # fraction of 5x7-pixel blocks the sensor returns nothing for
    tum_border: int = 8              # columns at the right border without depth (registration to the colour image)
    tum_sensor: bool = True          # False: same room and trajectory through an ideal sensor (float depth + noise, 2 % holes): the bench's comparison run

    @property
    def period(self) -> float:
        return self.frames_per_period * self.step

    def boxes(self) -> np.ndarray:
        """[n,7]: xmin,xmax,ytop,zmin,zmax,albedo,_ within one period."""
        out = np.zeros((self.n_boxes, 6), dtype=np.float64)
        idx = np.arange(self.n_boxes, dtype=np.uint32)
        r = lambda k: _uniform01(idx * np.uint32(7) + np.uint32(k), self.seed * 31 + 17)
        cx = -4.5 + 9.0 * r(0)
        # keep the driving corridor |x| < 1 free so the camera never enters a box
        cx = np.where(np.abs(cx) < 1.8, np.sign(cx + 1e-9) * (1.8 + np.abs(cx)), cx)
        wx = 0.6 + 1.2 * r(1)
        hz = 0.8 + 2.0 * r(2)
        hy = 0.6 + 1.8 * r(3)
        cz = self.period * (idx + r(4)) / self.n_boxes
        out[:, 0] = cx - wx / 2
        out[:, 1] = cx + wx / 2
        out[:, 2] = 1.65 - hy
        out[:, 3] = cz
        out[:, 4] = cz + hz
        out[:, 5] = 60.0 + 140.0 * r(5)
        out[:, :5] *= self.scale
        return out

    def pose(self, t: int) -> np.ndarray:
        """cam->world 4x4 float32 for frame ``t`` (any non-negative integer)."""
        if self.tum:
            return _tum_pose(self, t)
        ph = 2.0 * np.pi * (t % self.frames_per_period) / self.frames_per_period
        yaw = np.deg2rad(self.yaw_amp_deg) * np.sin(ph)
        c, s = np.cos(yaw), np.sin(yaw)
        m = np.eye(4, dtype=np.float64)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
        m[0, 3] = self.sway_amp * np.sin(ph) * self.scale
        m[2, 3] = self.step * t * self.scale
        return m.astype(np.float32)


def _rot(axis: int, a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    m = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


def _tum_pose(scene: Scene, t: int) -> np.ndarray:
    """Hand-held sweep through a room, closed after `frames_per_period` frames: a +-40 degree pan with a nodding pitch, a
    little roll, a 0.3 m wander of the camera centre and a tremor (hash of the frame index) on all six."""
    P = scene.frames_per_period
    tl = t % P
    ph = 2.0 * np.pi * tl / P
    j = (_uniform01(np.arange(6, dtype=np.uint32) + np.uint32(6 * tl), scene.seed * 977 + 5) - 0.5) * 2.0
    yaw = np.deg2rad(40.0) * np.sin(ph) + np.deg2rad(0.15) * j[0]
    pitch = np.deg2rad(6.0) + np.deg2rad(8.0) * np.sin(2.0 * ph + 0.7) + np.deg2rad(0.15) * j[1]
    roll = np.deg2rad(3.0) * np.sin(3.0 * ph + 0.3) + np.deg2rad(0.15) * j[2]
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = _rot(1, yaw) @ _rot(0, pitch) @ _rot(2, roll)
    m[:3, 3] = (0.30 * np.sin(ph + 0.4) + 0.003 * j[3], 0.06 * np.sin(2.0 * ph) + 0.003 * j[4], 0.25 * np.sin(ph) + 0.003 * j[5])
    m[:3, 3] *= scene.scale
    return m.astype(np.float32)


_TUM_ROOM = ((-2.4, 2.6), (-1.3, 1.2), (-2.2, 4.6))   # x, y (down: 1.2 is the floor), z extents in metres
_TUM_WALL_ALBEDO = ((150.0, 95.0), (210.0, 70.0), (125.0, 165.0))  # (low, high) face of every axis


def _tum_boxes(scene: Scene) -> np.ndarray:
    """[n, 7]: xmin, xmax, ymin, ymax, zmin, zmax, albedo -- a desk with a monitor, a cabinet, a chair, a box close to the
    camera (inside the sensor's 0.4 m blind range for part of the sweep), and `n_boxes` seeded ones standing on the floor."""
    fixed = [(-1.5, 0.3, 0.45, 1.2, 1.6, 2.5, 120.0), (-0.9, -0.3, 0.0, 0.45, 2.1, 2.2, 40.0), (1.8, 2.6, -0.6, 1.2, 0.5, 1.6, 180.0),
             (0.55, 1.0, 0.4, 1.2, 1.0, 1.45, 90.0), (-0.45, -0.15, 0.55, 1.2, 0.62, 0.9, 200.0)]
    idx = np.arange(scene.n_boxes, dtype=np.uint32)
    r = lambda k: _uniform01(idx * np.uint32(7) + np.uint32(k), scene.seed * 31 + 19)
    ang = 2.0 * np.pi * (idx + r(0)) / max(scene.n_boxes, 1)
    rad = 1.3 + 0.8 * r(1)           # a ring around the camera's wander: the camera never enters a box
    cx, cz = rad * np.sin(ang), 0.4 + rad * np.cos(ang)
    w, d, hgt = 0.15 + 0.35 * r(2), 0.15 + 0.35 * r(3), 0.3 + 1.2 * r(4)
    rnd = np.stack([cx - w / 2, cx + w / 2, 1.2 - hgt, np.full(len(idx), 1.2), cz - d / 2, cz + d / 2, 50.0 + 170.0 * r(5)], axis=1)
    out = np.concatenate([np.array(fixed, dtype=np.float64), rnd]) if len(idx) else np.array(fixed, dtype=np.float64)
    out[:, :6] *= scene.scale
    return out


def _render_tum(cam: Camera, scene: Scene, t: int):
    W, H = cam.width, cam.height
    pose = scene.pose(t)
    tl = t % scene.frames_per_period
    P = pose.astype(np.float64)
    org, R, sc = P[:3, 3], P[:3, :3], scene.scale
    u = (np.arange(W, dtype=np.float64) - cam.cx) / cam.fx
    v = (np.arange(H, dtype=np.float64) - cam.cy) / cam.fy
    dc = np.stack([np.broadcast_to(u[None, :], (H, W)), np.broadcast_to(v[:, None], (H, W)), np.ones((H, W))])
    d = np.einsum("ij,jhw->ihw", R, dc)
    best = np.full((H, W), np.inf)
    albedo = np.zeros((H, W))
    with np.errstate(divide="ignore", invalid="ignore"):
        for a in range(3):  # the room from inside: the face every ray leaves through
            lo, hi = _TUM_ROOM[a][0] * sc, _TUM_ROOM[a][1] * sc
            ta = np.where(d[a] > 0, (hi - org[a]) / d[a], (lo - org[a]) / d[a])
            ta = np.where(np.abs(d[a]) > 1e-12, ta, np.inf)
            nearer = ta < best
            best = np.where(nearer, ta, best)
            albedo = np.where(nearer, np.where(d[a] > 0, _TUM_WALL_ALBEDO[a][1], _TUM_WALL_ALBEDO[a][0]), albedo)
        for b in _tum_boxes(scene):
            t0 = [(b[2 * a] - org[a]) / d[a] for a in range(3)]
            t1 = [(b[2 * a + 1] - org[a]) / d[a] for a in range(3)]
            tn = np.maximum(np.maximum(np.minimum(t0[0], t1[0]), np.minimum(t0[1], t1[1])), np.minimum(t0[2], t1[2]))
            tf = np.minimum(np.minimum(np.maximum(t0[0], t1[0]), np.maximum(t0[1], t1[1])), np.maximum(t0[2], t1[2]))
            ok = (tn <= tf) & (tn > 0) & (tn < best)
            best = np.where(ok, tn, best)
            albedo = np.where(ok, b[6], albedo)
    tt = best  # camera-frame z == ray parameter (dir_c.z = 1); a closed room: every ray hits
    wpt = org[:, None, None] + tt[None] * d
    cell = (np.floor(wpt[0] / (0.25 * sc) + 1000.0) + np.floor(wpt[1] / (0.25 * sc) + 1000.0) + np.floor(wpt[2] / (0.25 * sc) + 1000.0)).astype(np.int64)
    chk = np.where(cell & 1, scene.checker, -scene.checker)
    pix = (np.arange(H * W, dtype=np.uint32)).reshape(H, W)
    salt = scene.seed * 2654435761 + tl * 40503
    img = albedo + chk + _uniform01(pix, salt + 1) * scene.intensity_noise
    image = np.clip(np.floor(img), 0, 255).astype(np.uint8)
    # ---- the sensor
    z = tt * (1.0 + scene.depth_noise * (_uniform01(pix, salt + 2) - 0.5))
    if not scene.tum_sensor:
        return image, np.where(_uniform01(pix, salt + 3) < scene.hole_fraction, 0.0, z).astype(np.float32), pose
    with np.errstate(divide="ignore", invalid="ignore"):
        disp_true = scene.kinect_bf * sc / tt                      # pixels, of the clean geometry: the shadows' extent
        disp = np.round(scene.kinect_bf * sc / z / scene.kinect_subpixel) * scene.kinect_subpixel
        zq = scene.kinect_bf * sc / disp
    valid = np.isfinite(zq) & (zq >= scene.tum_near * sc) & (zq <= scene.tum_far * sc)
    # projector shadow: a point is lit unless something `k` pixels to its left is at least `k` pixels of disparity nearer
    shadow = np.zeros((H, W), dtype=bool)
    for k in range(1, min(W, 96)):
        shadow[:, k:] |= (disp_true[:, :-k] - disp_true[:, k:]) >= k
    valid &= ~shadow
    valid &= ~(_uniform01(pix, salt + 3) < scene.hole_fraction)
    ox, oy = int(_hash32(np.array([(salt + 4) & 0xFFFFFFFF], dtype=np.uint32))[0] % 5), int(_hash32(np.array([(salt + 5) & 0xFFFFFFFF], dtype=np.uint32))[0] % 7)
    yy, xx = np.mgrid[0:H, 0:W]
    block = (((yy + oy) // 7) * 4099 + (xx + ox) // 5).astype(np.uint32)
    valid &= ~(_uniform01(block, salt + 6) < scene.blob_fraction)
    if scene.tum_border > 0:
        valid[:, W - scene.tum_border:] = False
    u16 = np.clip(np.round(np.where(valid, zq, 0.0) * scene.tum_depth_scale), 0, 65535).astype(np.uint16)
    depth = (u16.astype(np.float32) / np.float32(scene.tum_depth_scale)).astype(np.float32)  # what the depth PNG decodes to
    return image, depth, pose


def render(cam: Camera, scene: Scene, t: int):
    """Return (image uint8 [H,W], depth float32 [H,W] metres, 0 = invalid, pose float32 4x4)."""
    if scene.tum:
        return _render_tum(cam, scene, t)
    W, H = cam.width, cam.height
    pose = scene.pose(t)
    tl = t % scene.frames_per_period
    P = pose.astype(np.float64)
    # rendering happens in the first period's coordinates (scene is periodic in z)
    org = np.array([P[0, 3], P[1, 3], scene.step * tl * scene.scale])
    R = P[:3, :3]
    u = (np.arange(W, dtype=np.float64) - cam.cx) / cam.fx
    v = (np.arange(H, dtype=np.float64) - cam.cy) / cam.fy
    dx_c = np.broadcast_to(u[None, :], (H, W))
    dy_c = np.broadcast_to(v[:, None], (H, W))
    dx = R[0, 0] * dx_c + R[0, 1] * dy_c + R[0, 2]
    dy = R[1, 0] * dx_c + R[1, 1] * dy_c + R[1, 2]
    dz = R[2, 0] * dx_c + R[2, 1] * dy_c + R[2, 2]
    sc = scene.scale
    best = np.full((H, W), np.inf)
    albedo = np.zeros((H, W))
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground
        tg = (1.65 * sc - org[1]) / dy
        ok = (dy > 1e-9) & (tg > 0)
        best = np.where(ok, tg, best)
        albedo = np.where(ok, 110.0, albedo)
        # walls
        for sign, alb in ((-1.0, 150.0), (1.0, 90.0)):
            tw = (sign * 6.0 * sc - org[0]) / dx
            yh = org[1] + tw * dy
            ok = (tw > 0) & (yh > -3.0 * sc) & (yh < 1.65 * sc) & (tw < best)
            best = np.where(ok, tw, best)
            albedo = np.where(ok, alb, albedo)
        # boxes: this period and the next two (covers max_depth)
        bx = scene.boxes()
        for k in range(3):
            for b in bx:
                lo = np.array([b[0], b[2], b[3] + k * scene.period * sc])
                hi = np.array([b[1], 1.65 * sc, b[4] + k * scene.period * sc])
                t0x, t1x = (lo[0] - org[0]) / dx, (hi[0] - org[0]) / dx
                t0y, t1y = (lo[1] - org[1]) / dy, (hi[1] - org[1]) / dy
                t0z, t1z = (lo[2] - org[2]) / dz, (hi[2] - org[2]) / dz
                tn = np.maximum(np.maximum(np.minimum(t0x, t1x), np.minimum(t0y, t1y)), np.minimum(t0z, t1z))
                tf = np.minimum(np.minimum(np.maximum(t0x, t1x), np.maximum(t0y, t1y)), np.maximum(t0z, t1z))
                ok = (tn <= tf) & (tn > 0) & (tn < best)
                best = np.where(ok, tn, best)
                albedo = np.where(ok, b[5], albedo)
    hit = np.isfinite(best) & (best * 1.0 < scene.max_depth * sc)
    tt = np.where(hit, best, 0.0)
    # world-space checker (fixed to the world so consecutive frames agree)
    wx = org[0] + tt * dx
    wy = org[1] + tt * dy
    wz = org[2] + tt * dz
    cell = (np.floor(wx / sc + 1000.0) + np.floor(wy / sc + 1000.0) + np.floor(wz / sc + 1000.0)).astype(np.int64)
    chk = np.where(cell & 1, scene.checker, -scene.checker)
    pix = (np.arange(H * W, dtype=np.uint32)).reshape(H, W)
    salt = scene.seed * 2654435761 + tl * 40503
    n_int = _uniform01(pix, salt + 1) * scene.intensity_noise
    img = np.where(hit, albedo + chk + n_int, 200.0 + n_int)  # sky is bright
    img = np.clip(np.floor(img), 0, 255)
    if scene.intensity_levels > 0:
        q = 256.0 / scene.intensity_levels
        img = np.floor(img / q) * q
    if scene.saturate_above > 0:
        img = np.where(img > scene.saturate_above, 255.0, img)
    image = img.astype(np.uint8)
    n_d = _uniform01(pix, salt + 2) - 0.5
    depth = tt * (1.0 + scene.depth_noise * n_d)  # camera-frame z == ray parameter (dir_c.z = 1)
    holes = _uniform01(pix, salt + 3) < scene.hole_fraction
    if scene.stereo:
        # what a stereo matcher hands to publisher.py: a float32 disparity map in steps of `disparity_step`, 0 where it
        # found no match; depth = float32(bf) / disparity, so +inf there (numpy's x / 0), exactly representable
        # disparities elsewhere: neighbouring pixels of a far surface share their depth bit for bit
        with np.errstate(divide="ignore", invalid="ignore"):
            disp = np.where(hit & ~holes, (scene.stereo_bf * sc) / depth, 0.0)
            disp = (np.round(disp / scene.disparity_step) * scene.disparity_step).astype(np.float32)
            depth = (np.float32(scene.stereo_bf * sc) / disp).astype(np.float32)
        if not scene.zero_disparity_inf:
            depth = np.where(np.isfinite(depth), depth, np.float32(0.0)).astype(np.float32)
        return image, depth, pose
    depth = np.where(hit & ~holes, depth, 0.0).astype(np.float32)
    return image, depth, pose


def _job_key(cam: Camera, scene: Scene, t: int) -> str:
    import hashlib
    return hashlib.sha256(repr((dataclasses.astuple(cam), dataclasses.astuple(scene), int(t), "v2")).encode()).hexdigest()[:32]


def render_many(jobs, workers: int = 0, cache_dir: str = "/tmp/dsm_synth_cache"):
    """[(cam, scene, t), ...] -> [(image, depth), ...].  A frame takes 0.4 s at 1226x370 and 3.5 s at 1920x1080 in
    numpy, so batches are rendered by worker PROCESSES started as fresh interpreters (`python -m
    densesurfelmapping_amd.synth --worker`) with profiler / preload variables stripped from their environment: no
    fork of a process that holds a GPU runtime, nothing inherited from rocprofv3 or torchrun.  Frames are exchanged
    through (and kept in) `cache_dir`, so a second run on the same box renders nothing."""
    import json
    import os
    import subprocess
    import sys
    jobs = list(jobs)
    os.makedirs(cache_dir, exist_ok=True)
    paths = [os.path.join(cache_dir, _job_key(*j) + ".npz") for j in jobs]
    todo = [i for i, p in enumerate(paths) if not os.path.exists(p)]
    if workers <= 0:
        workers = max(1, min(48, (os.cpu_count() or 2) // 2))
    workers = min(workers, len(todo))
    if workers > 1:
        env = {k: v for k, v in os.environ.items()
               if not (k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER", "LD_PRELOAD", "OMP_", "MKL_")) or k in ("RANK", "WORLD_SIZE"))}
        env["OMP_NUM_THREADS"] = "1"
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        procs = []
        for w in range(workers):
            mine = [{"cam": dataclasses.asdict(jobs[i][0]), "scene": dataclasses.asdict(jobs[i][1]), "t": int(jobs[i][2]), "out": paths[i]}
                    for i in todo[w::workers]]
            p = subprocess.Popen([sys.executable, "-m", "densesurfelmapping_amd.synth", "--worker"], stdin=subprocess.PIPE,
                                 stdout=subprocess.DEVNULL, env=env, cwd=root)
            p.stdin.write(json.dumps(mine).encode())
            p.stdin.close()
            procs.append(p)
        for p in procs:
            if p.wait(timeout=900) != 0:
                raise RuntimeError("a render worker failed")
    else:
        for i in todo:
            _render_to(jobs[i][0], jobs[i][1], jobs[i][2], paths[i])
    out = []
    for p in paths:
        with np.load(p) as z:
            out.append((z["image"], z["depth"]))
    return out


def _render_to(cam, scene, t, path):
    import os
    image, depth, _ = render(cam, scene, t)
    tmp = f"{path}.{os.getpid()}.tmp.npz"
    np.savez(tmp, image=image, depth=depth)
    os.replace(tmp, path)


def _worker_main():
    import json
    import sys
    for job in json.loads(sys.stdin.read()):
        _render_to(Camera(**job["cam"]), Scene(**job["scene"]), job["t"], job["out"])


def sequence(cam: Camera, scene: Scene, n_frames: int, start: int = 0, keyframe_every: int = 5):
    """Yield (t, image, depth, pose, ref_idx): every 5th (``keyframe_every``) frame is a keyframe and
    ``ref_idx`` is the index of the latest keyframe."""
    cache = {}
    for t in range(start, start + n_frames):
        tl = t % scene.frames_per_period
        if tl not in cache:
            cache[tl] = render(cam, scene, tl)[:2]
        image, depth = cache[tl]
        yield t, image, depth, scene.pose(t), (t - start) // keyframe_every


# ----------------------------------------------------------------------------------------------------------
# Node-level message stream: what kitti_publisher (image, depth) and the modified ORB-SLAM2 (loop_stamps,
# loop_path, this_pose) send to the surfel_fusion node (ros_node.cpp:24-32), as plain tuples.

NODE_CAM = Camera(320, 104, 185.0, 185.0, 159.5, 51.5)   # small frames for node-level tests (8 | W, 8 | H)
NODE_CAM_RGBD = Camera(320, 240, 262.5, 262.5, 159.5, 119.5, far=6.0, near=0.3, rgbd=True)


def _rot_to_quat(R: np.ndarray) -> np.ndarray:
    """x, y, z, w of a proper rotation matrix (any correct conversion will do: both sides get the same bytes)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0) * 2.0
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2.0
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def pose7(m: np.ndarray) -> np.ndarray:
    """4x4 cam->world -> geometry_msgs/Pose as [px, py, pz, qx, qy, qz, qw] (float64)."""
    m = np.asarray(m, dtype=np.float64)
    return np.concatenate([m[:3, 3], _rot_to_quat(m[:3, :3])])


def _drift(t: int, rate: float) -> np.ndarray:
    """Accumulated odometry drift after t frames: a small yaw about the vertical axis plus a lateral offset."""
    a = np.deg2rad(0.02) * rate * t
    d = np.eye(4)
    d[0, 0], d[0, 2], d[2, 0], d[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    d[0, 3] = 0.004 * rate * t
    return d


def node_messages(cam: Camera, scene: Scene, n_frames: int, lap: int = 40, keyframe_every: int = 5, drift_rate: float = 1.0,
                  path_lag: int = 0, pose_first=(), drop_pose=(), extra_loops=None, frames=None, ref_rng=None):
    """Yield the node's input messages for ``n_frames`` frames of a circuit of ``lap`` frames.

    The camera drives the scene's trajectory and jumps back to the start after every ``lap`` frames (a closed
    circuit).  During the first lap the SLAM poses carry a slowly growing drift; when the camera is back at the
    start, the frame is a keyframe with a loop edge to keyframe 0 and from then on the loop path holds the
    drift-free poses, so every keyframe of the first lap is corrected (surfel_map.cpp:235-280).

    Events, in publication order:
      ("image", (sec, nsec), uint8[H,W]) / ("depth", (sec, nsec), float32[H,W])
      ("orb", (sec, nsec), loop_values float32[2k], loop_path float64[n,7], this_pose float64[7], covariance float64[36])
    ``pose_first``: frames whose orb message precedes their images; ``drop_pose``: frames without orb message
    (never a keyframe); ``path_lag``: the loop path misses the newest ``path_lag`` keyframes (:254-270);
    ``extra_loops``: {frame: [(kf_a, kf_b), ...]} additional loop edges announced at that frame;
    ``frames``: already rendered {frame index within the lap: (image, depth)}; ``ref_rng``: a numpy Generator that
    picks each frame's reference keyframe among the latest three instead of the latest (fuzzing the pose graph).
    """
    extra_loops = extra_loops or {}
    frames = dict(frames or {})
    true_kf, est_kf = [], []       # per keyframe: drift-free and estimated cam->world
    loops = []
    closed = False
    for t in range(n_frames):
        tl = t % lap
        if tl not in frames:
            frames[tl] = render(cam, scene, tl)[:2]
        image, depth = frames[tl]
        stamp = (1000 + t // 10, (t % 10) * 100000000)
        true_pose = scene.pose(tl).astype(np.float64)
        is_kf = t % keyframe_every == 0
        if t >= lap and not closed and is_kf:
            closed = True
            loops.append((len(true_kf), 0))
        est_pose = true_pose if closed else _drift(t, drift_rate) @ true_pose
        ref_kf = max(len(true_kf) - 1, 0)
        if ref_rng is not None and ref_kf > 0:
            ref_kf -= int(ref_rng.integers(0, min(3, ref_kf + 1)))
        if is_kf:
            true_kf.append(true_pose)
            est_kf.append(est_pose)
        loops.extend(extra_loops.get(t, []))
        path_src = true_kf if closed else est_kf
        n_path = max(1, len(path_src) - path_lag)
        path = np.stack([pose7(p) for p in path_src[:n_path]])
        cov = np.zeros(36)
        cov[0] = 1.0 if is_kf else 0.0
        cov[1] = float(ref_kf)
        orb = ("orb", stamp, np.array([v for ab in loops for v in ab], dtype=np.float32), path, pose7(est_pose), cov)
        img = [("image", stamp, image), ("depth", stamp, depth)]
        if t in drop_pose and not is_kf:
            yield from img
        elif t in pose_first:
            yield orb
            yield from img
        else:
            yield from img
            yield orb


if __name__ == "__main__":
    import sys
    if "--worker" in sys.argv:
        _worker_main()
