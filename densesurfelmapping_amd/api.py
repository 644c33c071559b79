"""Host-side mirror of the reference's per-frame fusion interface, over the C ABI of include/dsm.h.

Reference (C++) interface mirrored here, same names and argument meaning:

  * ``FusionFunctions::initialize(w, h, fx, fy, cx, cy, far, near)``
    -- surfel_fusion/src/fusion_functions.h:84-87
  * ``FusionFunctions::fuse_initialize_map(reference_frame_index, image, depth, pose,
    local_surfels, new_surfels)`` -- fusion_functions.h:88-94, fusion_functions.cpp:30-83
  * ``SurfelMap::fuse_map(image, depth, pose, reference_index)`` -- surfel_map.cpp:1060-1113

``SurfelElement`` / ``Superpixel_seed`` arrays are numpy structured arrays with the reference's
byte layout (elements.h:5-31).  All compute happens in the HIP library; if it cannot be loaded, or
no gfx950 device is present, construction raises -- there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSM_LIB_PATH") or os.path.join(HERE, "libdsm_hip.so")  # override: experiments only

SURFEL_DTYPE = np.dtype(
    [("px", "<f4"), ("py", "<f4"), ("pz", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
     ("size", "<f4"), ("color", "<f4"), ("weight", "<f4"), ("update_times", "<i4"), ("last_update", "<i4")]
)  # elements.h:22-31
SEED_DTYPE = np.dtype(
    {"names": ["x", "y", "size", "norm_x", "norm_y", "norm_z", "posi_x", "posi_y", "posi_z", "view_cos",
               "mean_depth", "mean_intensity", "fused", "stable", "min_eigen_value", "max_eigen_value"],
     "formats": ["<f4"] * 12 + ["u1", "u1", "<f4", "<f4"],
     "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 49, 52, 56],
     "itemsize": 60}
)  # elements.h:5-20
assert SURFEL_DTYPE.itemsize == 44 and SEED_DTYPE.itemsize == 60

DSM_FLAG_NO_GRAPH = 1
DSM_FLAG_UPLOAD_STREAM = 2
DSM_FLAG_WAVE_STAMPS = 4
DSM_MAX_STAGES = 32

# every symbol include/dsm.h declares
ABI_SYMBOLS = (
    "dsm_abi_version", "dsm_config_init", "dsm_create", "dsm_destroy", "dsm_last_error",
    "dsm_host_alloc", "dsm_host_free", "dsm_host_pack_frames",
    "dsm_fuse_initialize_map", "dsm_fuse_map", "dsm_fuse_initialize_map_inv", "dsm_fuse_map_inv",
    "dsm_fuse_frame_resident_inv", "dsm_replay_enqueue_inv", "dsm_batch_replay_enqueue_inv",
    "dsm_map_upload", "dsm_map_size", "dsm_map_capacity", "dsm_map_download", "dsm_map_copy_to_device",
    "dsm_map_warp", "dsm_warp_grouped_device", "dsm_map_extract", "dsm_map_append",
    "dsm_store_deactivate", "dsm_store_activate", "dsm_store_erase", "dsm_store_warp", "dsm_store_size",
    "dsm_store_download",
    "dsm_frame_upload", "dsm_frame_upload_device", "dsm_frame_pitch", "dsm_frame_upload_async", "dsm_frames_upload_async", "dsm_frame_uploads_wait", "dsm_fuse_frame_resident", "dsm_replay_enqueue", "dsm_replay_enqueue_host", "dsm_replay_wait",
    "dsm_synchronize", "dsm_last_new_count", "dsm_stream",
    "dsm_batch_create", "dsm_batch_destroy", "dsm_batch_last_error", "dsm_batch_replay_enqueue", "dsm_batch_synchronize",
    "dsm_batch_replay_timed",
    "dsm_get_labels", "dsm_get_seeds", "dsm_seed_count", "dsm_replay_timed", "dsm_debug_wave_stamps", "dsm_debug_set_fit_small_cap", "dsm_debug_tier_counts", "dsm_debug_dropin_stats",
    "dsm_debug_run_stages", "dsm_debug_get_label_buffer", "dsm_debug_set_label_buffer", "dsm_debug_get_seed_state",
    "dsm_debug_set_seed_state",
)


class DsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dsm error {code}: {msg}")
        self.code = code


class _Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("far_dist", C.c_float), ("near_dist", C.c_float),
                ("huber_range", C.c_double), ("baseline", C.c_double),
                ("disparity_error", C.c_double), ("min_tolerate_diff", C.c_double),
                ("device", C.c_int32), ("surfel_capacity", C.c_int32), ("frame_slots", C.c_int32),
                ("flags", C.c_uint32), ("pipeline_depth", C.c_int32)]


class _StageTimes(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("name", C.c_char_p * DSM_MAX_STAGES),
                ("ms", C.c_double * DSM_MAX_STAGES), ("launches", C.c_int64 * DSM_MAX_STAGES),
                ("frames", C.c_int64), ("event_overhead_ms", C.c_double), ("sum_new", C.c_int64), ("sum_local", C.c_int64)]


_vp = C.c_void_p
_lib = None


def load_library():
    """dlopen the in-tree HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m densesurfelmapping_amd.build` "
            "(hipcc, gfx950). This package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.dsm_last_error.restype = C.c_char_p
    lib.dsm_last_error.argtypes = [_vp]
    lib.dsm_config_init.argtypes = [C.POINTER(_Config), C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_int]
    lib.dsm_create.argtypes = [C.POINTER(_Config), C.POINTER(_vp)]
    lib.dsm_destroy.argtypes = [_vp]
    lib.dsm_destroy.restype = None
    lib.dsm_host_alloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    lib.dsm_host_free.argtypes = [_vp]
    lib.dsm_host_free.restype = None
    lib.dsm_host_pack_frames.argtypes = [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t]
    lib.dsm_fuse_initialize_map.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_int32,
                                            _vp, C.c_int32, _vp]
    lib.dsm_fuse_map.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, C.c_int32, _vp]
    lib.dsm_fuse_initialize_map_inv.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, C.c_int32,
                                                _vp, C.c_int32, _vp]
    lib.dsm_fuse_map_inv.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp, C.c_int32, _vp]
    lib.dsm_fuse_frame_resident_inv.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp]
    lib.dsm_replay_enqueue_inv.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp]
    lib.dsm_replay_enqueue_host.argtypes = [_vp, C.c_int32, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp]
    lib.dsm_replay_wait.argtypes = [_vp, C.c_int32]
    lib.dsm_batch_replay_enqueue_inv.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp]
    lib.dsm_map_upload.argtypes = [_vp, _vp, C.c_int32]
    lib.dsm_map_size.argtypes = [_vp, _vp]
    lib.dsm_map_capacity.argtypes = [_vp, _vp]
    lib.dsm_map_download.argtypes = [_vp, _vp, C.c_int32, _vp]
    lib.dsm_map_copy_to_device.argtypes = [_vp, _vp, C.c_int32, _vp]
    lib.dsm_map_warp.argtypes = [_vp, _vp]
    lib.dsm_warp_grouped_device.argtypes = [_vp, _vp, C.c_int32, _vp, _vp]
    lib.dsm_map_extract.argtypes = [_vp, C.c_int32, _vp, C.c_int32, _vp]
    lib.dsm_map_append.argtypes = [_vp, _vp, C.c_int32]
    lib.dsm_store_deactivate.argtypes = [_vp, C.c_int32, _vp, _vp]
    lib.dsm_store_activate.argtypes = [_vp, C.c_int32, C.c_int32]
    lib.dsm_store_erase.argtypes = [_vp, C.c_int32, C.c_int32]
    lib.dsm_store_warp.argtypes = [_vp, C.c_int32, _vp, _vp, _vp]
    lib.dsm_store_size.argtypes = [_vp, _vp]
    lib.dsm_store_download.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp]
    lib.dsm_frame_upload.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]
    lib.dsm_frame_upload_device.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]
    lib.dsm_frame_pitch.argtypes = [_vp, _vp]
    lib.dsm_frame_upload_async.argtypes = [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]
    lib.dsm_frames_upload_async.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t]
    lib.dsm_frame_uploads_wait.argtypes = [_vp]
    lib.dsm_fuse_frame_resident.argtypes = [_vp, C.c_int, C.c_int, _vp]
    lib.dsm_replay_enqueue.argtypes = [_vp, C.c_int32, _vp, _vp, _vp]
    lib.dsm_synchronize.argtypes = [_vp]
    lib.dsm_last_new_count.argtypes = [_vp, _vp]
    lib.dsm_stream.argtypes = [_vp, C.POINTER(_vp)]
    lib.dsm_get_labels.argtypes = [_vp, _vp]
    lib.dsm_get_seeds.argtypes = [_vp, _vp]
    lib.dsm_seed_count.argtypes = [_vp]
    lib.dsm_debug_wave_stamps.argtypes = [_vp, _vp]
    lib.dsm_debug_set_fit_small_cap.argtypes = [_vp, C.c_int32]
    lib.dsm_debug_tier_counts.argtypes = [_vp, _vp]
    lib.dsm_debug_dropin_stats.argtypes = [_vp, _vp]
    lib.dsm_debug_run_stages.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int]
    lib.dsm_debug_get_label_buffer.argtypes = [_vp, C.c_int, _vp]
    lib.dsm_debug_set_label_buffer.argtypes = [_vp, C.c_int, _vp]
    lib.dsm_debug_get_seed_state.argtypes = [_vp, _vp, _vp]
    lib.dsm_debug_set_seed_state.argtypes = [_vp, _vp, _vp]
    lib.dsm_replay_timed.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, C.POINTER(_StageTimes)]
    lib.dsm_batch_create.argtypes = [C.POINTER(_vp), C.c_int32, C.POINTER(_vp)]
    lib.dsm_batch_destroy.argtypes = [_vp]
    lib.dsm_batch_destroy.restype = None
    lib.dsm_batch_last_error.argtypes = [_vp]
    lib.dsm_batch_last_error.restype = C.c_char_p
    lib.dsm_batch_replay_enqueue.argtypes = [_vp, C.c_int32, _vp, _vp, _vp]
    lib.dsm_batch_synchronize.argtypes = [_vp]
    lib.dsm_batch_replay_timed.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, C.POINTER(_StageTimes)]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(_vp)


def _inv_ptr(inv_pose):
    """the caller's own world->cam matrix (4x4 row-major numpy, or 16 column-major floats) as a pointer, or NULL"""
    if inv_pose is None:
        return None, None
    a = np.asarray(inv_pose, np.float32)
    a = pose_to_colmajor(a) if a.shape == (4, 4) else np.ascontiguousarray(a.reshape(16))
    return a, _ptr(a)


def pose_to_colmajor(pose) -> np.ndarray:
    """4x4 cam->world matrix (row-major numpy) -> the 16 floats of an Eigen::Matrix4f."""
    p = np.asarray(pose, np.float32)
    if p.shape != (4, 4):
        raise ValueError("pose must be 4x4")
    return np.ascontiguousarray(p.T).ravel()


class FusionFunctions:
    """Drop-in for the reference's ``FusionFunctions`` (one instance = one handle = one stream)."""

    def __init__(self):
        self._lib = load_library()
        self._h = None

    # fusion_functions.h:84-87; the keyword arguments are what an HBM-resident engine adds
    def initialize(self, width, height, fx, fy, cx, cy, far_dist, near_dist, *, rgbd=False, device=0,
                   surfel_capacity=0, frame_slots=0, flags=0, pipeline_depth=0):
        self.close()
        cfg = _Config()
        rc = self._lib.dsm_config_init(C.byref(cfg), width, height, fx, fy, cx, cy, far_dist, near_dist,
                                       1 if rgbd else 0)
        if rc:
            raise DsmError(rc, "dsm_config_init")
        cfg.device, cfg.surfel_capacity, cfg.frame_slots, cfg.flags = device, surfel_capacity, frame_slots, flags
        cfg.pipeline_depth = pipeline_depth
        h = _vp()
        rc = self._lib.dsm_create(C.byref(cfg), C.byref(h))
        if rc:
            raise DsmError(rc, self._lib.dsm_last_error(None).decode())
        self._h = h
        self.width, self.height = width, height
        self.n_seed = self._lib.dsm_seed_count(h)
        return self

    @classmethod
    def from_camera(cls, cam, **kw):
        return cls().initialize(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.far, cam.near,
                                rgbd=getattr(cam, "rgbd", False), **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dsm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise DsmError(rc, self._lib.dsm_last_error(self._h).decode())

    def _frame_args(self, image, depth):
        image = np.asarray(image)
        depth = np.asarray(depth)
        if image.dtype != np.uint8 or depth.dtype != np.float32:
            raise TypeError("image must be uint8 (CV_8UC1) and depth float32 (CV_32FC1)")
        if image.shape != (self.height, self.width) or depth.shape != (self.height, self.width):
            raise ValueError("image/depth shape does not match initialize()")
        # the C side takes a pointer and a positive row step >= one row (a cv::Mat): anything else -- element stride
        # other than the item size, flipped (negative step) or broadcast (step smaller than a row) views -- is copied
        if image.strides[1] != 1 or image.strides[0] < self.width:
            image = np.ascontiguousarray(image)
        if depth.strides[1] != 4 or depth.strides[0] < self.width * 4:
            depth = np.ascontiguousarray(depth)
        return image, depth

    # fusion_functions.h:88-94: returns (local_surfels updated, new_surfels)
    # inv_pose: the caller's own pose.inverse() (FF.cpp:59), see dsm_fuse_map_inv in include/dsm.h; None = the library's closed form
    def fuse_initialize_map(self, reference_frame_index, image, depth, pose, local_surfels, inv_pose=None):
        image, depth = self._frame_args(image, depth)
        pose_cm = pose_to_colmajor(pose)
        _keep, inv = _inv_ptr(inv_pose)
        local = np.ascontiguousarray(local_surfels, SURFEL_DTYPE).copy()
        fresh = np.zeros(self.n_seed, SURFEL_DTYPE)
        n_new = C.c_int32(0)
        self._check(self._lib.dsm_fuse_initialize_map_inv(
            self._h, reference_frame_index, _ptr(image), image.strides[0], _ptr(depth), depth.strides[0],
            _ptr(pose_cm), inv, _ptr(local), len(local), _ptr(fresh), len(fresh), C.byref(n_new)))
        return local, fresh[: n_new.value].copy()

    # SurfelMap::fuse_map (surfel_map.cpp:1060-1113): returns (local_surfels after compaction, n_new)
    def fuse_map(self, reference_frame_index, image, depth, pose, local_surfels, inv_pose=None):
        image, depth = self._frame_args(image, depth)
        pose_cm = pose_to_colmajor(pose)
        _keep, inv = _inv_ptr(inv_pose)
        cap = len(local_surfels) + self.n_seed
        buf = np.zeros(cap, SURFEL_DTYPE)
        buf[: len(local_surfels)] = local_surfels
        n_local = C.c_int32(len(local_surfels))
        n_new = C.c_int32(0)
        self._check(self._lib.dsm_fuse_map_inv(
            self._h, reference_frame_index, _ptr(image), image.strides[0], _ptr(depth), depth.strides[0],
            _ptr(pose_cm), inv, _ptr(buf), C.byref(n_local), cap, C.byref(n_new)))
        return buf[: n_local.value].copy(), n_new.value

    def fuse_map_inplace(self, reference_frame_index, image, depth, pose, buf, n_local):
        """dsm_fuse_map on the caller's own array, as the C++ caller uses it (`buf` = std::vector storage with
        capacity len(buf), the first n_local records live): no copies on the Python side.  Returns (n_local, n_new)."""
        image, depth = self._frame_args(image, depth)
        pose_cm = pose_to_colmajor(pose)
        assert buf.dtype == SURFEL_DTYPE and buf.flags.c_contiguous
        n = C.c_int32(n_local)
        n_new = C.c_int32(0)
        self._check(self._lib.dsm_fuse_map(
            self._h, reference_frame_index, _ptr(image), image.strides[0], _ptr(depth), depth.strides[0],
            _ptr(pose_cm), _ptr(buf), C.byref(n), len(buf), C.byref(n_new)))
        return n.value, n_new.value

    # ---- resident path -------------------------------------------------------------------
    def map_upload(self, surfels):
        a = np.ascontiguousarray(surfels, SURFEL_DTYPE)
        self._check(self._lib.dsm_map_upload(self._h, _ptr(a), len(a)))

    def map_size(self) -> int:
        n = C.c_int32(0)
        self._check(self._lib.dsm_map_size(self._h, C.byref(n)))
        return n.value

    def map_capacity(self) -> int:
        n = C.c_int32(0)
        self._check(self._lib.dsm_map_capacity(self._h, C.byref(n)))
        return n.value

    def map_download(self) -> np.ndarray:
        n = self.map_size()
        out = np.zeros(max(n, 1), SURFEL_DTYPE)
        m = C.c_int32(0)
        self._check(self._lib.dsm_map_download(self._h, _ptr(out), len(out), C.byref(m)))
        return out[: m.value].copy()

    def map_copy_to_device(self, dst_ptr: int, cap: int) -> int:
        n = C.c_int32(0)
        self._check(self._lib.dsm_map_copy_to_device(self._h, _vp(dst_ptr), cap, C.byref(n)))
        return n.value

    # ---- map maintenance between frames (surfel_map.cpp:681-824, 1456-1595) -----------------
    def map_warp(self, warp):
        """SurfelMap::warp_active_surfels_cpu_kernel: warp = (loop_pose * cam_pose^-1) as 4x4 float."""
        w = pose_to_colmajor(warp)
        self._check(self._lib.dsm_map_warp(self._h, _ptr(w)))

    def warp_grouped_device(self, surfels_ptr, offsets, mats):
        """SurfelMap::warp_inactive_surfels_cpu_kernel on device memory: mats [g,4,4], offsets [g+1]."""
        offsets = np.ascontiguousarray(offsets, np.int32)
        mats_cm = np.ascontiguousarray(np.asarray(mats, np.float32).transpose(0, 2, 1)).reshape(-1, 16)
        self._check(self._lib.dsm_warp_grouped_device(self._h, _vp(surfels_ptr), len(offsets) - 1, _ptr(offsets), _ptr(mats_cm)))

    def map_extract(self, key) -> np.ndarray:
        """move_add_surfels removal: live surfels with last_update == key, in order; their slots are deleted."""
        out = np.zeros(max(self.map_size(), 1), SURFEL_DTYPE)
        n = C.c_int32(0)
        self._check(self._lib.dsm_map_extract(self._h, key, _ptr(out), len(out), C.byref(n)))
        return out[: n.value].copy()

    def map_append(self, surfels):
        a = np.ascontiguousarray(surfels, SURFEL_DTYPE)
        self._check(self._lib.dsm_map_append(self._h, _ptr(a), len(a)))

    # ---- inactive store (device-side attached_surfels + inactive_pointcloud of surfel_map.cpp)
    def store_deactivate(self, key):
        """move_add_surfels removal into the device store; returns the segment (begin, n)."""
        b, n = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.dsm_store_deactivate(self._h, key, C.byref(b), C.byref(n)))
        return b.value, n.value

    def store_activate(self, begin, n):
        self._check(self._lib.dsm_store_activate(self._h, begin, n))

    def store_erase(self, begin, n):
        self._check(self._lib.dsm_store_erase(self._h, begin, n))

    def store_warp(self, offsets, mats, changed):
        """offsets [g+1] tile the store; mats [g,4,4] row-major numpy; changed [g] bool."""
        offsets = np.ascontiguousarray(offsets, np.int32)
        mats_cm = np.ascontiguousarray(np.asarray(mats, np.float32).transpose(0, 2, 1)).reshape(-1, 16)
        ch = np.ascontiguousarray(changed, np.uint8)
        self._check(self._lib.dsm_store_warp(self._h, len(offsets) - 1, _ptr(offsets), _ptr(mats_cm), _ptr(ch)))

    def store_size(self) -> int:
        n = C.c_int32(0)
        self._check(self._lib.dsm_store_size(self._h, C.byref(n)))
        return n.value

    def store_download(self, begin=0, n=None):
        """(surfels, xyzi) of store[begin, begin+n)."""
        if n is None:
            n = self.store_size() - begin
        s = np.zeros(max(n, 1), SURFEL_DTYPE)
        c = np.zeros((max(n, 1), 4), np.float32)
        self._check(self._lib.dsm_store_download(self._h, begin, n, _ptr(s), _ptr(c)))
        return s[:n].copy(), c[:n].copy()

    def frame_upload(self, slot, image, depth):
        image, depth = self._frame_args(image, depth)
        self._check(self._lib.dsm_frame_upload(self._h, slot, _ptr(image), image.strides[0], _ptr(depth),
                                               depth.strides[0]))

    def frame_pitch(self) -> int:
        n = C.c_int32(0)
        self._check(self._lib.dsm_frame_pitch(self._h, C.byref(n)))
        return n.value

    def frame_upload_async(self, slot, image, depth):
        """dsm_frame_upload_async: `image` / `depth` are views of PAGE-LOCKED memory (PinnedFrames below) that stay
        untouched until frame_uploads_wait(); rows with the slot pitch go up as one transfer per plane."""
        if image.dtype != np.uint8 or depth.dtype != np.float32 or image.shape != (self.height, self.width) or depth.shape != image.shape:
            raise TypeError("image must be uint8 [H,W], depth float32 [H,W]")
        if image.strides[1] != 1 or depth.strides[1] != 4:
            raise ValueError("rows must be contiguous")
        self._check(self._lib.dsm_frame_upload_async(self._h, slot, _ptr(image), image.strides[0], _ptr(depth), depth.strides[0]))

    def frames_upload_async(self, slot0, pinned, first, n):
        """frames first .. first+n-1 of a PinnedFrames block (slot layout, back to back) into slots slot0 .. slot0+n-1:
        one transfer per plane for all of them -- of a block with tight rows too (PinnedFrames(..., tight=True))"""
        assert 0 <= first and first + n <= pinned.n and (pinned.h, pinned.w) == (self.height, self.width) and pinned.pitch in (self.frame_pitch(), self.width)
        img, dep = pinned.image(first), pinned.depth(first)
        self._check(self._lib.dsm_frames_upload_async(self._h, slot0, n, _ptr(img), img.strides[0], pinned.pitch * pinned.h,
                                                      _ptr(dep), dep.strides[0], pinned.pitch * pinned.h * 4))

    def frame_uploads_wait(self):
        self._check(self._lib.dsm_frame_uploads_wait(self._h))

    def frame_upload_device(self, slot, image_ptr, img_step, depth_ptr, depth_step):
        self._check(self._lib.dsm_frame_upload_device(self._h, slot, _vp(image_ptr), img_step, _vp(depth_ptr),
                                                      depth_step))

    def fuse_frame_resident(self, slot, reference_frame_index, pose, inv_pose=None):
        pose_cm = pose_to_colmajor(pose)
        _keep, inv = _inv_ptr(inv_pose)
        self._check(self._lib.dsm_fuse_frame_resident_inv(self._h, slot, reference_frame_index, _ptr(pose_cm), inv))

    @staticmethod
    def pack_replay(slots, ref_idx, poses):
        slots = np.ascontiguousarray(slots, np.int32)
        ref_idx = np.ascontiguousarray(ref_idx, np.int32)
        poses_cm = np.ascontiguousarray(np.asarray(poses, np.float32).transpose(0, 2, 1)).reshape(len(slots), 16)
        return slots, ref_idx, poses_cm

    def replay_enqueue(self, slots, ref_idx, poses_cm, inv_poses_cm=None):
        """slots/ref_idx int32 [n], poses_cm float32 [n,16] column-major (see pack_replay); inv_poses_cm: the caller's own
        inverses in the same layout, or None."""
        inv = None
        if inv_poses_cm is not None:
            inv_poses_cm = np.ascontiguousarray(inv_poses_cm, np.float32).reshape(len(slots), 16)
            inv = _ptr(inv_poses_cm)
        self._check(self._lib.dsm_replay_enqueue_inv(self._h, len(slots), _ptr(slots), _ptr(ref_idx), _ptr(poses_cm), inv))

    def replay_enqueue_host(self, pinned, first, ref_idx, poses_cm, inv_poses_cm=None):
        """frames first .. first+n-1 of a PinnedFrames block (n = len(ref_idx)) fused in order, each group of frames uploaded on
        the stream that runs its superpixel stages, right in front of them (include/dsm.h, dsm_replay_enqueue_host); the block's
        frames may be rewritten once replay_wait says the call is done"""
        n = len(ref_idx)
        assert 0 <= first and first + n <= pinned.n and (pinned.h, pinned.w, pinned.pitch) == (self.height, self.width, self.frame_pitch())
        if n == 0:
            return
        img, dep = pinned.image(first), pinned.depth(first)
        inv = None
        if inv_poses_cm is not None:
            inv_poses_cm = np.ascontiguousarray(inv_poses_cm, np.float32).reshape(n, 16)
            inv = _ptr(inv_poses_cm)
        ref_idx = np.ascontiguousarray(ref_idx, np.int32)
        poses_cm = np.ascontiguousarray(poses_cm, np.float32).reshape(n, 16)
        self._check(self._lib.dsm_replay_enqueue_host(self._h, n, _ptr(img), img.strides[0], pinned.pitch * pinned.h, _ptr(dep), dep.strides[0],
                                                      pinned.pitch * pinned.h * 4, _ptr(ref_idx), _ptr(poses_cm), inv))

    def replay_wait(self, calls_back=0):
        """host wait until the frames of the replay_enqueue_host call `calls_back` calls ago have been fused"""
        self._check(self._lib.dsm_replay_wait(self._h, int(calls_back)))

    def synchronize(self):
        self._check(self._lib.dsm_synchronize(self._h))

    def last_new_count(self) -> int:
        n = C.c_int32(0)
        self._check(self._lib.dsm_last_new_count(self._h, C.byref(n)))
        return n.value

    def stream(self) -> int:
        s = _vp()
        self._check(self._lib.dsm_stream(self._h, C.byref(s)))
        return s.value or 0

    # ---- parity taps -----------------------------------------------------------------------
    def labels(self) -> np.ndarray:  # FusionFunctions::superpixel_index
        out = np.zeros((self.height, self.width), np.int32)
        self._check(self._lib.dsm_get_labels(self._h, _ptr(out)))
        return out

    def seeds(self) -> np.ndarray:  # FusionFunctions::superpixel_seeds
        out = np.zeros(self.n_seed, SEED_DTYPE)
        self._check(self._lib.dsm_get_seeds(self._h, _ptr(out)))
        return out

    # ---- state-level test taps ---------------------------------------------------------------
    STAGES = ("init_seeds", "assign_0", "update_seeds_0", "commit_seeds_0", "assign_1", "resolve_1", "update_seeds_1",
              "commit_seeds_1", "assign_2", "resolve_2", "update_seeds_2", "commit_seeds_2", "seed_points", "seed_fit",
              "fuse_surfels", "frame_tail")

    def debug_run_stages(self, slot, reference_frame_index, pose, first, last):
        pose_cm = pose_to_colmajor(pose)
        self._check(self._lib.dsm_debug_run_stages(self._h, slot, reference_frame_index, _ptr(pose_cm),
                                                   self.STAGES.index(first), self.STAGES.index(last)))

    def debug_get_labels(self, which) -> np.ndarray:
        out = np.zeros((self.height, self.width), np.int32)
        self._check(self._lib.dsm_debug_get_label_buffer(self._h, which, _ptr(out)))
        return out

    def debug_set_labels(self, which, labels):
        a = np.ascontiguousarray(labels, np.int32)
        self._check(self._lib.dsm_debug_set_label_buffer(self._h, which, _ptr(a)))

    def debug_get_seed_state(self):
        core = np.zeros((self.n_seed, 4), np.float32)
        stable = np.zeros(self.n_seed, np.int32)
        self._check(self._lib.dsm_debug_get_seed_state(self._h, _ptr(core), _ptr(stable)))
        return core, stable

    def debug_set_seed_state(self, core, stable):
        core = np.ascontiguousarray(core, np.float32)
        stable = np.ascontiguousarray(stable, np.int32)
        self._check(self._lib.dsm_debug_set_seed_state(self._h, _ptr(core), _ptr(stable)))

    def debug_tier_counts(self):
        """second-tier occupancy of the latest frame's lane-per-seed kernels (include/dsm.h, dsm_debug_tier_counts)"""
        out = np.zeros(8, np.int32)
        self._check(self._lib.dsm_debug_tier_counts(self._h, out.ctypes.data_as(_vp)))
        return {"huber_rest_by_sweep": [int(out[0]), int(out[2]), int(out[4])], "long_list_by_sweep": [int(out[1]), int(out[3]), int(out[5])],
                "fit_long_groups": int(out[6])}

    def debug_dropin_stats(self):
        """the drop-in calls' delta downloads so far (include/dsm.h, dsm_debug_dropin_stats)"""
        out = np.zeros(8, np.int64)
        self._check(self._lib.dsm_debug_dropin_stats(self._h, out.ctypes.data_as(_vp)))
        return {"calls": int(out[0]), "delta_calls": int(out[1]), "delta_groups": int(out[2]), "last_groups": int(out[3]),
                "host_us": {"frame_staging": int(out[4]), "map_compare_or_upload": int(out[5]), "gpu_wait": int(out[6]), "fetch_and_patch": int(out[7])}}

    def debug_set_fit_small_cap(self, cap):
        self._check(self._lib.dsm_debug_set_fit_small_cap(self._h, int(cap)))

    def debug_wave_stamps(self) -> np.ndarray:
        out = np.zeros((5, self.n_seed, 8), np.int64)
        self._check(self._lib.dsm_debug_wave_stamps(self._h, _ptr(out)))
        return out

    def replay_timed(self, slots, ref_idx, poses_cm):
        """Eager replay with a HIP event pair around every kernel; returns {stage: (ms_total, launches)}."""
        st = _StageTimes()
        self._check(self._lib.dsm_replay_timed(self._h, len(slots), _ptr(slots), _ptr(ref_idx), _ptr(poses_cm),
                                               C.byref(st)))
        self.event_overhead_ms = st.event_overhead_ms / max(st.frames, 1)
        self.timed_mean_new = st.sum_new / max(st.frames, 1)      # K: surfels created per frame
        self.timed_mean_local = st.sum_local / max(st.frames, 1)  # M: live surfels after a frame
        return {st.name[i].decode(): (st.ms[i], st.launches[i]) for i in range(st.n_stages)}, st.frames


class PinnedFrames:
    """n frames in page-locked host memory (dsm_host_alloc), rows laid out with a handle's slot pitch, pad columns zero:
    image(i) / depth(i) are [H,W] views that dsm_frame_upload_async moves in one transfer per plane."""

    def __init__(self, ff, n: int, tight: bool = False):
        """ff: a FusionFunctions (its slot layout), or a (height, width) pair -- the pitch is then the library's rule,
        ceil(width / 64) * 64 elements per row, and frames_upload_async checks it against the handle's.  tight=True: rows
        `width` elements apart, frames back to back (`pitch` = width) -- the asynchronous uploads then move no pad bytes over
        the link and set the rows to the slots' pitch on the device."""
        self._lib = load_library()
        if isinstance(ff, tuple):
            self.n, self.h, self.w = n, int(ff[0]), int(ff[1])
            self.pitch = (self.w + 63) // 64 * 64
        else:
            self.n, self.h, self.w, self.pitch = n, ff.height, ff.width, ff.frame_pitch()
        if tight:
            self.pitch = self.w
        self._bytes_img, self._bytes_dep = self.pitch * self.h, self.pitch * self.h * 4
        p = _vp()
        rc = self._lib.dsm_host_alloc(C.byref(p), n * (self._bytes_img + self._bytes_dep))
        if rc:
            raise DsmError(rc, "dsm_host_alloc")
        self._p = p
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * (self._bytes_img + self._bytes_dep),))
        raw[:] = 0
        self._img = raw[: n * self._bytes_img].reshape(n, self.h, self.pitch)
        self._dep = raw[n * self._bytes_img:].view(np.float32).reshape(n, self.h, self.pitch)

    def image(self, i):
        return self._img[i, :, : self.w]

    def depth(self, i):
        return self._dep[i, :, : self.w]

    def set(self, i, image, depth):
        self.image(i)[...] = image
        self.depth(i)[...] = depth

    def set_many(self, first, images, depths):
        """frames first .. first+n-1 from n (image uint8 [H,W], depth float32 [H,W]) pairs, copied by the library's host
        threads (dsm_host_pack_frames; the GIL is released for the call)"""
        n = len(images)
        if n == 0:
            return
        if first < 0 or first + n > self.n or len(depths) != n:
            raise ValueError("frames out of range")
        keep = []
        for im, dp in zip(images, depths):
            im = im if (im.dtype == np.uint8 and im.strides[1] == 1) else np.ascontiguousarray(im, np.uint8)
            dp = dp if (dp.dtype == np.float32 and dp.strides[1] == 4) else np.ascontiguousarray(dp, np.float32)
            if im.shape != (self.h, self.w) or dp.shape != (self.h, self.w):
                raise ValueError("frame size")
            keep.append((im, dp))
        ip = (C.c_void_p * n)(*[k[0].ctypes.data for k in keep])
        dp_ = (C.c_void_p * n)(*[k[1].ctypes.data for k in keep])
        ist = (C.c_size_t * n)(*[k[0].strides[0] for k in keep])
        dst = (C.c_size_t * n)(*[k[1].strides[0] for k in keep])
        rc = self._lib.dsm_host_pack_frames(n, self.w, self.h, ip, ist, dp_, dst,
                                            C.c_void_p(self._img[first].ctypes.data), C.c_size_t(self.pitch), C.c_size_t(self._bytes_img),
                                            C.c_void_p(self._dep[first].ctypes.data), C.c_size_t(self.pitch * 4), C.c_size_t(self._bytes_dep))
        if rc:
            raise DsmError(rc, self._lib.dsm_last_error(None).decode())

    def close(self):
        if getattr(self, "_p", None):
            self._img = self._dep = None
            self._lib.dsm_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Handles of equal image size on one device advancing in lockstep: every kernel of a frame is launched once for all
    of them (include/dsm.h, dsm_batch_*).  The handles keep their own maps and frame slots and stay usable on their own
    (map_download, labels, ...) between batch calls."""

    def __init__(self, handles):
        self._lib = load_library()
        self.handles = list(handles)
        arr = (_vp * len(self.handles))(*[h._h for h in self.handles])
        b = _vp()
        rc = self._lib.dsm_batch_create(arr, len(self.handles), C.byref(b))
        if rc:
            raise DsmError(rc, self._lib.dsm_batch_last_error(None).decode())
        self._b = b

    def close(self):
        if getattr(self, "_b", None):
            self._lib.dsm_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise DsmError(rc, self._lib.dsm_batch_last_error(self._b).decode())

    @staticmethod
    def pack(plans):
        """[(slots[n], ref_idx[n], poses_cm[n,16]) per handle] (FusionFunctions.pack_replay) -> handle-major arrays."""
        n = len(plans[0][0])
        assert all(len(p[0]) == n for p in plans), "every handle of a batch advances by the same number of frames"
        return (np.ascontiguousarray(np.concatenate([p[0] for p in plans]), np.int32),
                np.ascontiguousarray(np.concatenate([p[1] for p in plans]), np.int32),
                np.ascontiguousarray(np.concatenate([p[2] for p in plans]), np.float32), n)

    def replay_enqueue(self, slots, ref_idx, poses_cm, n_frames, inv_poses_cm=None):
        inv = None
        if inv_poses_cm is not None:
            inv_poses_cm = np.ascontiguousarray(inv_poses_cm, np.float32).reshape(len(slots), 16)
            inv = _ptr(inv_poses_cm)
        self._check(self._lib.dsm_batch_replay_enqueue_inv(self._b, n_frames, _ptr(slots), _ptr(ref_idx), _ptr(poses_cm), inv))

    def synchronize(self):
        self._check(self._lib.dsm_batch_synchronize(self._b))

    def replay_timed(self, slots, ref_idx, poses_cm, n_frames):
        """Eager batched replay with a HIP event pair around every (batched) kernel; returns {stage: (ms_total, launches)}
        and the number of handle-frames."""
        st = _StageTimes()
        self._check(self._lib.dsm_batch_replay_timed(self._b, n_frames, _ptr(slots), _ptr(ref_idx), _ptr(poses_cm), C.byref(st)))
        self.event_overhead_ms = st.event_overhead_ms / max(st.frames, 1)
        self.timed_mean_new = st.sum_new / max(st.frames, 1)
        self.timed_mean_local = st.sum_local / max(st.frames, 1)
        return {st.name[i].decode(): (st.ms[i], st.launches[i]) for i in range(st.n_stages)}, st.frames
