"""ORACLE / TEST INFRASTRUCTURE: ctypes bindings for the two CPU checkers.

  * ``RefOracle``  -- oracle/_ref/libdsm_ref_*.so: the reference's own
    fusion_functions.cpp compiled in place (oracle/ref_driver.cpp).
  * ``PortOracle`` -- oracle/liboracle_port.so: our C restatement
    (oracle/dsm_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SURFEL_DTYPE = np.dtype(
    [("px", "<f4"), ("py", "<f4"), ("pz", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
     ("size", "<f4"), ("color", "<f4"), ("weight", "<f4"), ("update_times", "<i4"), ("last_update", "<i4")]
)  # elements.h:22-31, 44 bytes
SEED_DTYPE = np.dtype(
    {"names": ["x", "y", "size", "norm_x", "norm_y", "norm_z", "posi_x", "posi_y", "posi_z", "view_cos",
               "mean_depth", "mean_intensity", "fused", "stable", "min_eigen_value", "max_eigen_value"],
     "formats": ["<f4"] * 12 + ["u1", "u1", "<f4", "<f4"],
     "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 49, 52, 56],
     "itemsize": 60}
)  # elements.h:5-20, 60 bytes
assert SURFEL_DTYPE.itemsize == 44 and SEED_DTYPE.itemsize == 60

_vp = C.c_void_p


def ref_lib_path(kind: str = "serial") -> str:
    return os.path.join(HERE, "_ref", f"libdsm_ref_{kind}.so")


def have_ref(kind: str = "serial") -> bool:
    return os.path.exists(ref_lib_path(kind))


def _ptr(a):
    return a.ctypes.data_as(_vp)


class _Base:
    prefix = ""

    def _bind(self, lib):
        p = self.prefix
        f = getattr(lib, p + "create")
        f.restype = _vp
        f.argtypes = [C.c_int, C.c_int] + [C.c_float] * 6
        getattr(lib, p + "destroy").argtypes = [_vp]
        f = getattr(lib, p + "fuse_initialize_map")
        f.restype = C.c_int
        f.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_int, _vp, C.c_int, _vp]
        f = getattr(lib, p + "fuse_map")
        f.restype = C.c_int
        f.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, C.c_int, _vp]
        for name in ("get_labels", "set_labels", "get_seeds", "set_seeds", "get_norm_map"):
            getattr(lib, p + name).argtypes = [_vp, _vp]
        getattr(lib, p + "set_frame").argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t]
        for name in ("generate_super_pixels", "initialize_seeds", "update_pixels", "update_seeds", "calculate_norms"):
            getattr(lib, p + name).argtypes = [_vp]

    def __init__(self, lib, cam):
        self.lib = lib
        self._bind(lib)
        self.cam = cam
        self.w, self.h = cam.width, cam.height
        self.S = (self.w // 8) * (self.h // 8)
        self.h_ = getattr(lib, self.prefix + "create")(
            cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.far, cam.near)
        self._keep = None

    def close(self):
        if self.h_:
            getattr(self.lib, self.prefix + "destroy")(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *a):
        return getattr(self.lib, self.prefix + name)(self.h_, *a)

    # FusionFunctions::fuse_initialize_map (FF.cpp:30-83)
    def fuse_initialize_map(self, ref_idx, image, depth, pose, local):
        image = np.ascontiguousarray(image, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        pose_cm = np.ascontiguousarray(np.asarray(pose, np.float32).T).ravel()  # column-major
        local = np.ascontiguousarray(local, SURFEL_DTYPE).copy()
        fresh = np.zeros(self.S, SURFEL_DTYPE)
        n_new = C.c_int(0)
        rc = self._call("fuse_initialize_map", ref_idx, _ptr(image), image.strides[0], _ptr(depth), depth.strides[0],
                        _ptr(pose_cm), _ptr(local), len(local), _ptr(fresh), self.S, C.byref(n_new))
        assert rc == 0
        return local, fresh[: n_new.value].copy()

    # SurfelMap::fuse_map (SM.cpp:1060-1113)
    def fuse_map(self, ref_idx, image, depth, pose, local):
        image = np.ascontiguousarray(image, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        pose_cm = np.ascontiguousarray(np.asarray(pose, np.float32).T).ravel()
        cap = len(local) + self.S
        buf = np.zeros(cap, SURFEL_DTYPE)
        buf[: len(local)] = local
        n_local = C.c_int(len(local))
        n_new = C.c_int(0)
        rc = self._call("fuse_map", ref_idx, _ptr(image), image.strides[0], _ptr(depth), depth.strides[0],
                        _ptr(pose_cm), _ptr(buf), C.byref(n_local), cap, C.byref(n_new))
        assert rc == 0
        return buf[: n_local.value].copy(), n_new.value

    def labels(self):
        out = np.zeros((self.h, self.w), np.int32)
        self._call("get_labels", _ptr(out))
        return out

    def seeds(self):
        out = np.zeros(self.S, SEED_DTYPE)
        self._call("get_seeds", _ptr(out))
        return out

    def norm_map(self):
        out = np.zeros((self.h, self.w, 3), np.float32)
        self._call("get_norm_map", _ptr(out))
        return out

    def set_labels(self, labels):
        a = np.ascontiguousarray(labels, np.int32)
        self._call("set_labels", _ptr(a))

    def set_seeds(self, seeds):
        a = np.ascontiguousarray(seeds, SEED_DTYPE)
        self._call("set_seeds", _ptr(a))

    def set_frame(self, image, depth):
        image = np.ascontiguousarray(image, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        self._keep = (image, depth)
        self._call("set_frame", _ptr(image), image.strides[0], _ptr(depth), depth.strides[0])

    def stage(self, name):
        self._call(name)


def port_warp(surfels, warp):
    """dsmo_warp on a copy of `surfels` (SURFEL_DTYPE); warp is a 4x4 row-major numpy matrix."""
    lib = C.CDLL(os.path.join(HERE, "liboracle_port.so"))
    lib.dsmo_warp.argtypes = [_vp, C.c_int, _vp]
    a = np.ascontiguousarray(surfels, SURFEL_DTYPE).copy()
    m = np.ascontiguousarray(np.asarray(warp, np.float32).T).ravel()
    lib.dsmo_warp(_ptr(a), len(a), _ptr(m))
    return a


def port_extract_key(local, key):
    """dsmo_extract_key: returns (local with the slots deleted, extracted surfels)."""
    lib = C.CDLL(os.path.join(HERE, "liboracle_port.so"))
    lib.dsmo_extract_key.argtypes = [_vp, C.c_int, C.c_int, _vp]
    lib.dsmo_extract_key.restype = C.c_int
    a = np.ascontiguousarray(local, SURFEL_DTYPE).copy()
    out = np.zeros(max(len(a), 1), SURFEL_DTYPE)
    k = lib.dsmo_extract_key(_ptr(a), len(a), key, _ptr(out))
    return a, out[:k].copy()


class RefOracle(_Base):
    prefix = "dsmref_"

    def __init__(self, cam, kind=None):
        if kind is None:
            kind = "serial_rgbd" if cam.rgbd else "serial"
        super().__init__(C.CDLL(ref_lib_path(kind)), cam)


class PortOracle(_Base):
    prefix = "dsmo_"

    def __init__(self, cam, threads=1):
        lib = C.CDLL(os.path.join(HERE, "liboracle_port.so"))
        lib.dsmo_set_constants.argtypes = [_vp, C.c_double, C.c_double, C.c_double, C.c_double]
        super().__init__(lib, cam)
        if cam.rgbd:
            lib.dsmo_set_constants(self.h_, 0.05, 0.08, 1.0, 0.05)  # fusion_functions.h:17-21


class RefSurfelMap:
    """oracle/_ref/libdsm_ref_map.so: the reference's node class (surfel_map.cpp + fusion_functions.cpp compiled
    in place, oracle/ref_map_driver.cpp) behind the same message-level interface as
    densesurfelmapping_amd.surfel_map.SurfelMap."""

    def __init__(self, cam, drift_free_poses=10, kind=None, lib_path=None):
        # kind "map_threads": real std::threads, for timing only (the reference's warp_surfels races, SM.cpp:791-824)
        # lib_path: another build of the same driver (tests/_build/libdsm_ref_map_on_product.so: the reference's
        # surfel_map.cpp on top of the product's engine facade)
        lib = C.CDLL(lib_path or ref_lib_path(kind or ("map_rgbd" if cam.rgbd else "map")))
        lib.refmap_create.restype = _vp
        lib.refmap_create.argtypes = [C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_int]
        lib.refmap_destroy.argtypes = [_vp]
        lib.refmap_image_input.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_size_t, _vp]
        lib.refmap_depth_input.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_size_t, _vp]
        lib.refmap_orb_results_input.argtypes = [_vp, C.c_uint32, C.c_uint32, _vp, C.c_int, _vp, C.c_int, C.c_uint32, C.c_uint32, _vp, _vp]
        for name in ("pending_poses", "local_count", "pose_count", "cloud_count"):
            getattr(lib, "refmap_" + name).argtypes = [_vp]
        lib.refmap_get_local.argtypes = [_vp, _vp]
        lib.refmap_get_pose.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
        lib.refmap_get_links.argtypes = [_vp, C.c_int, _vp, C.c_int]
        lib.refmap_get_attached.argtypes = [_vp, C.c_int, _vp]
        lib.refmap_get_cloud.argtypes = [_vp, _vp]
        lib.refmap_save_cloud.argtypes = [_vp, C.c_char_p]
        lib.refmap_save_mesh.argtypes = [_vp, C.c_char_p]
        self._lib = lib
        self.cam = cam
        self._h = lib.refmap_create(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.far, cam.near, drift_free_poses)
        self.frames_fused = 0

    def close(self):
        if self._h:
            self._lib.refmap_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, event):
        before = self._lib.refmap_pending_poses(self._h)
        if event[0] == "image":
            img = np.ascontiguousarray(event[2], dtype=np.uint8)
            self._lib.refmap_image_input(self._h, event[1][0], event[1][1], img.shape[1], img.shape[0], img.strides[0], _ptr(img))
        elif event[0] == "depth":
            d = np.ascontiguousarray(event[2], dtype=np.float32)
            self._lib.refmap_depth_input(self._h, event[1][0], event[1][1], d.shape[1], d.shape[0], d.strides[0], _ptr(d))
        else:
            lv = np.ascontiguousarray(event[2], dtype=np.float32)
            lp = np.ascontiguousarray(event[3], dtype=np.float64).reshape(-1, 7)
            tp = np.ascontiguousarray(event[4], dtype=np.float64)
            cov = np.ascontiguousarray(event[5], dtype=np.float64)
            self._lib.refmap_orb_results_input(self._h, event[1][0], event[1][1], _ptr(lv), lv.size, _ptr(lp), lp.shape[0],
                                               event[1][0], event[1][1], _ptr(tp), _ptr(cov))
            before += 1
        self.frames_fused += before - self._lib.refmap_pending_poses(self._h)  # a pose leaves the buffer when it is fused

    @property
    def pose_count(self):
        return self._lib.refmap_pose_count(self._h)

    def local_surfels(self):
        n = self._lib.refmap_local_count(self._h)
        out = np.zeros(max(n, 1), dtype=SURFEL_DTYPE)
        self._lib.refmap_get_local(self._h, _ptr(out))
        return out[:n]

    def pose(self, i):
        cam, loop = np.zeros(7), np.zeros(7)
        n_att, begin, is_local = C.c_int(), C.c_int(), C.c_int()
        self._lib.refmap_get_pose(self._h, i, _ptr(cam), _ptr(loop), C.byref(n_att), C.byref(begin), C.byref(is_local))
        links = np.zeros(4096, dtype=np.int32)
        n = self._lib.refmap_get_links(self._h, i, _ptr(links), links.size)
        return {"cam_pose": cam, "loop_pose": loop, "n_attached": n_att.value, "points_begin_index": begin.value,
                "is_local": bool(is_local.value), "links": links[:n].tolist()}

    def attached_surfels(self, i):
        n = self.pose(i)["n_attached"]
        out = np.zeros(max(n, 1), dtype=SURFEL_DTYPE)
        self._lib.refmap_get_attached(self._h, i, _ptr(out))
        return out[:n]

    def inactive_cloud(self):
        n = self._lib.refmap_cloud_count(self._h)
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        self._lib.refmap_get_cloud(self._h, _ptr(out))
        return out[:n]

    def save_cloud(self, path):
        if self._lib.refmap_save_cloud(self._h, path.encode()) != 0:
            raise RuntimeError("pcl::PCDWriter: Input point cloud has no data!")

    def save_mesh(self, path):
        self._lib.refmap_save_mesh(self._h, path.encode())
