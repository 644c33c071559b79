"""Python mirror of the reference's node class ``SurfelMap`` (surfel_fusion/src/surfel_map.h:48-147) over
the C ABI of include/dsm_surfel_map.h: the three subscriber callbacks, ``save_cloud`` / ``save_mesh`` /
``save_map``, and read-only taps for what the publish_* methods would send.

Messages are plain values instead of ROS types: stamps are ``(sec, nsec)``, poses are 7 doubles
``[px, py, pz, qx, qy, qz, qw]`` (geometry_msgs/Pose).  All state lives in the library: host logic in
csrc/dsm_surfel_map.cpp, surfels in HBM.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api

ABI_SYMBOLS = (
    "dsm_surfel_map_create", "dsm_surfel_map_destroy", "dsm_surfel_map_last_error",
    "dsm_surfel_map_image_input", "dsm_surfel_map_depth_input", "dsm_surfel_map_orb_results_input",
    "dsm_surfel_map_save_cloud", "dsm_surfel_map_save_mesh", "dsm_surfel_map_save_map",
    "dsm_surfel_map_engine", "dsm_surfel_map_frames_fused", "dsm_surfel_map_dropped_poses", "dsm_surfel_map_pose_count",
    "dsm_surfel_map_get_pose", "dsm_surfel_map_get_links", "dsm_surfel_map_get_attached",
    "dsm_surfel_map_get_inactive_cloud",
)

_vp = C.c_void_p


class _Stamp(C.Structure):
    _fields_ = [("sec", C.c_uint32), ("nsec", C.c_uint32)]


class _MapConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("cam_width", C.c_int32), ("cam_height", C.c_int32),
                ("cam_fx", C.c_float), ("cam_fy", C.c_float), ("cam_cx", C.c_float), ("cam_cy", C.c_float),
                ("fuse_far_distence", C.c_float), ("fuse_near_distence", C.c_float),
                ("drift_free_poses", C.c_int32), ("rgbd", C.c_int32), ("device", C.c_int32),
                ("surfel_capacity", C.c_int32), ("max_buffered_frames", C.c_int32)]


def _bind(lib):
    if not getattr(lib, "_dsm_surfel_map_bound", False):
        lib.dsm_surfel_map_create.argtypes = [C.POINTER(_MapConfig), C.POINTER(_vp)]
        lib.dsm_surfel_map_destroy.argtypes = [_vp]
        lib.dsm_surfel_map_destroy.restype = None
        lib.dsm_surfel_map_last_error.argtypes = [_vp]
        lib.dsm_surfel_map_last_error.restype = C.c_char_p
        lib.dsm_surfel_map_image_input.argtypes = [_vp, _Stamp, C.c_int32, C.c_int32, C.c_size_t, C.c_char_p, _vp]
        lib.dsm_surfel_map_depth_input.argtypes = [_vp, _Stamp, C.c_int32, C.c_int32, C.c_size_t, C.c_char_p, _vp]
        lib.dsm_surfel_map_orb_results_input.argtypes = [_vp, _Stamp, _vp, C.c_int32, _vp, C.c_int32, _Stamp, _vp, _vp]
        for name in ("save_cloud", "save_mesh", "save_map"):
            getattr(lib, "dsm_surfel_map_" + name).argtypes = [_vp, C.c_char_p]
        lib.dsm_surfel_map_engine.argtypes = [_vp]
        lib.dsm_surfel_map_engine.restype = _vp
        lib.dsm_surfel_map_frames_fused.argtypes = [_vp]
        lib.dsm_surfel_map_frames_fused.restype = C.c_int64
        lib.dsm_surfel_map_dropped_poses.argtypes = [_vp]
        lib.dsm_surfel_map_dropped_poses.restype = C.c_int64
        lib.dsm_surfel_map_pose_count.argtypes = [_vp]
        lib.dsm_surfel_map_get_pose.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp]
        lib.dsm_surfel_map_get_links.argtypes = [_vp, C.c_int32, _vp, C.c_int32]
        lib.dsm_surfel_map_get_attached.argtypes = [_vp, C.c_int32, _vp, C.c_int32, _vp]
        lib.dsm_surfel_map_get_inactive_cloud.argtypes = [_vp, _vp, C.c_int32, _vp]
        lib.dsm_map_size.argtypes = [_vp, _vp]
        lib.dsm_map_download.argtypes = [_vp, _vp, C.c_int32, _vp]
        lib.dsm_last_error.argtypes = [_vp]
        lib.dsm_last_error.restype = C.c_char_p
        lib._dsm_surfel_map_bound = True
    return lib


def _ptr(a):
    return a.ctypes.data_as(_vp)


class SurfelMap:
    """``SurfelMap(nh)`` of the reference with the node's ROS parameters as keyword arguments
    (surfel_map.cpp:13-28; launch defaults of kitti_orb.launch: drift_free_poses = 10)."""

    def __init__(self, cam, drift_free_poses: int = 10, device: int = 0, surfel_capacity: int = 0,
                 max_buffered_frames: int = 0, _library=None):
        # _library: tests bind the same class to their CPU stand-in build of the host logic (tests/node_hostemu.cpp)
        self._lib = _bind(_library if _library is not None else api.load_library())
        self.cam = cam
        cfg = _MapConfig(C.sizeof(_MapConfig), cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.far, cam.near, drift_free_poses,
                         1 if cam.rgbd else 0, device, surfel_capacity, max_buffered_frames)
        h = _vp()
        rc = self._lib.dsm_surfel_map_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise api.DsmError(rc, "dsm_surfel_map_create failed (no gfx950 device? this package has no CPU fallback)")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dsm_surfel_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise api.DsmError(rc, self._lib.dsm_surfel_map_last_error(self._h).decode())

    # ---- subscriber callbacks (ros_node.cpp:24-32)
    def image_input(self, stamp, image, encoding: str = "mono8"):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        self._check(self._lib.dsm_surfel_map_image_input(self._h, _Stamp(*stamp), img.shape[1], img.shape[0], img.strides[0],
                                                         encoding.encode(), _ptr(img)))

    def depth_input(self, stamp, depth, encoding: str = "32FC1"):
        d = np.ascontiguousarray(depth, dtype=np.float32)
        self._check(self._lib.dsm_surfel_map_depth_input(self._h, _Stamp(*stamp), d.shape[1], d.shape[0], d.strides[0],
                                                         encoding.encode(), _ptr(d)))

    def orb_results_input(self, stamp, loop_values, loop_path, this_pose, covariance, this_stamp=None):
        lv = np.ascontiguousarray(loop_values, dtype=np.float32)
        lp = np.ascontiguousarray(loop_path, dtype=np.float64).reshape(-1, 7)
        tp = np.ascontiguousarray(this_pose, dtype=np.float64)
        cov = np.ascontiguousarray(covariance, dtype=np.float64)
        assert tp.shape == (7,) and cov.shape == (36,)
        self._check(self._lib.dsm_surfel_map_orb_results_input(
            self._h, _Stamp(*stamp), _ptr(lv), lv.size, _ptr(lp), lp.shape[0], _Stamp(*(this_stamp or stamp)), _ptr(tp), _ptr(cov)))

    def feed(self, event):
        """One event of ``synth.node_messages``."""
        if event[0] == "image":
            self.image_input(event[1], event[2])
        elif event[0] == "depth":
            self.depth_input(event[1], event[2])
        else:
            self.orb_results_input(event[1], event[2], event[3], event[4], event[5])

    def save_cloud(self, path: str):
        self._check(self._lib.dsm_surfel_map_save_cloud(self._h, path.encode()))

    def save_mesh(self, path: str):
        self._check(self._lib.dsm_surfel_map_save_mesh(self._h, path.encode()))

    save_map = save_mesh  # surfel_map.cpp:75-81

    # ---- taps
    @property
    def frames_fused(self) -> int:
        return int(self._lib.dsm_surfel_map_frames_fused(self._h))

    @property
    def dropped_poses(self) -> int:
        return int(self._lib.dsm_surfel_map_dropped_poses(self._h))

    @property
    def pose_count(self) -> int:
        return int(self._lib.dsm_surfel_map_pose_count(self._h))

    def local_surfels(self) -> np.ndarray:
        eng = self._lib.dsm_surfel_map_engine(self._h)
        n = C.c_int32()
        rc = self._lib.dsm_map_size(eng, C.byref(n))
        if rc:
            raise api.DsmError(rc, self._lib.dsm_last_error(eng).decode())
        out = np.zeros(max(n.value, 1), dtype=api.SURFEL_DTYPE)
        rc = self._lib.dsm_map_download(eng, _ptr(out), n.value, C.byref(n))
        if rc:
            raise api.DsmError(rc, self._lib.dsm_last_error(eng).decode())
        return out[: n.value]

    def pose(self, i: int):
        cam, loop = np.zeros(7), np.zeros(7)
        n_att, begin, is_local = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.dsm_surfel_map_get_pose(self._h, i, _ptr(cam), _ptr(loop), C.byref(n_att), C.byref(begin), C.byref(is_local)))
        return {"cam_pose": cam, "loop_pose": loop, "n_attached": n_att.value, "points_begin_index": begin.value,
                "is_local": bool(is_local.value), "links": self.links(i)}

    def links(self, i: int):
        out = np.zeros(4096, dtype=np.int32)
        n = self._lib.dsm_surfel_map_get_links(self._h, i, _ptr(out), out.size)
        if n < 0:
            raise api.DsmError(n, "get_links")
        return out[:n].tolist()

    def attached_surfels(self, i: int) -> np.ndarray:
        n = C.c_int32()
        self._lib.dsm_surfel_map_get_attached(self._h, i, None, 0, C.byref(n))
        out = np.zeros(max(n.value, 1), dtype=api.SURFEL_DTYPE)
        self._check(self._lib.dsm_surfel_map_get_attached(self._h, i, _ptr(out), n.value, C.byref(n)))
        return out[: n.value]

    def inactive_cloud(self) -> np.ndarray:
        n = C.c_int32()
        self._lib.dsm_surfel_map_get_inactive_cloud(self._h, None, 0, C.byref(n))
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self._check(self._lib.dsm_surfel_map_get_inactive_cloud(self._h, _ptr(out), n.value, C.byref(n)))
        return out[: n.value]
