"""Shared by tests/golden/make_node_golden.py and the node-level tests: snapshots of a node object
(oracle.bindings.RefSurfelMap or densesurfelmapping_amd.surfel_map.SurfelMap -- same tap methods),
digests of them, and the scenarios."""
import hashlib

import numpy as np

SCENARIOS = [
    # one lap with drifting poses, loop closure at the start of the second lap: warp of active + inactive
    # surfels, re-activation of the first keyframes, their second deactivation
    {"name": "circuit_60", "frames": 60, "drift_free_poses": 3, "kw": {"lap": 40}},
    # loop path lagging two keyframes behind (extrapolated corrections), poses arriving before their images,
    # frames without a pose, a second loop edge that re-activates a separate run of the inactive set
    {"name": "circuit_lag_75", "frames": 75, "drift_free_poses": 4,
     "kw": {"lap": 40, "path_lag": 2, "pose_first": (3, 17, 42), "drop_pose": (13, 26), "extra_loops": {52: [(10, 3)]}}},
    # hand-held RGB-D camera in a room-sized scene (BASELINE configs[3] shape scaled down): the RGB-D constant set of
    # fusion_functions.h:17-21, a keyframe every 4 frames, loop closure after 32 frames
    {"name": "rgbd_room_48", "frames": 48, "drift_free_poses": 3, "camera": "NODE_CAM_RGBD",
     "scene": {"seed": 5, "scale": 0.12, "step": 0.05, "frames_per_period": 40},
     "kw": {"lap": 32, "keyframe_every": 4, "drift_rate": 0.1}},
]

# At the headline resolution (BASELINE configs[1]/[2] through the node): a lap of 50 frames at 1226x370, loop closure at
# the start of the second lap with ~50 k inactive surfels on 10 keyframes to warp, re-activation, a lagging loop path.
# Digests only (the final state is tens of MB).
SCENARIOS_LARGE = [
    {"name": "kitti_circuit_130", "frames": 130, "drift_free_poses": 4, "camera": "KITTI_1226", "scene": {"seed": 12345},
     "kw": {"lap": 50, "path_lag": 1, "extra_loops": {90: [(14, 3)]}}},
]


def camera_and_scene(case, synth):
    return getattr(synth, case.get("camera", "NODE_CAM")), synth.Scene(**case.get("scene", {}))


def _canon(a: np.ndarray) -> bytes:
    """bytes of an array with every NaN replaced by one canonical quiet NaN (sign / payload of a NaN is not
    defined by the reference's arithmetic)."""
    a = np.ascontiguousarray(a)
    if a.dtype.names:
        a = a.copy()
        for f in a.dtype.names:
            if a[f].dtype.kind == "f":
                a[f][np.isnan(a[f])] = np.nan
        return a.tobytes()
    if a.dtype.kind == "f":
        a = a.copy()
        a[np.isnan(a)] = np.nan
    return a.tobytes()


def snapshot(node) -> dict:
    n = node.pose_count
    poses = [node.pose(i) for i in range(n)]
    att = [node.attached_surfels(i) for i in range(n)]
    local = node.local_surfels()
    return {
        "local": local,
        "attached": np.concatenate(att) if att else local[:0],
        "attached_counts": np.array([len(a) for a in att], dtype=np.int32),
        "cloud": node.inactive_cloud(),
        "poses": np.array([np.concatenate([p["cam_pose"], p["loop_pose"]]) for p in poses], dtype=np.float64).reshape(n, 14),
        "begin": np.array([p["points_begin_index"] for p in poses], dtype=np.int32),
        "is_local": np.array([p["is_local"] for p in poses], dtype=np.uint8),
        "links": np.array([v for p in poses for v in (p["links"] + [-1])], dtype=np.int32),
    }


def digest(snap: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(snap):
        h.update(k.encode())
        h.update(_canon(snap[k]))
    return h.hexdigest()


def brief(node) -> list:
    return [int(node.frames_fused), int(node.pose_count), int(len(node.local_surfels())), int(len(node.inactive_cloud()))]


def file_digest(path: str):
    """sha256 of a saved map file; "-nan" reads as "nan" (iostreams print the sign bit of a NaN, which the
    reference's arithmetic does not define)."""
    data = open(path, "rb").read().replace(b"-nan", b"nan")
    head = data[:600].decode("ascii", "replace").split("\n")[:12]
    return {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data), "head": head}
