"""Recorded message streams for the node (the ROS-free replay driver of SURVEY.md §8(f) rank 3).

A log holds, in publication order, exactly what the node's three subscribers would receive
(ros_node.cpp:24-32): mono8 images, 32FC1 depth images and the synchronised ORB-SLAM triple
(loop_stamps, loop_path, this_pose).  Little-endian binary, one header, then records:

    header : int32 width, height, drift_free_poses;  float32 fx, fy, cx, cy, far, near
    record : int32 kind (0 image, 1 depth, 2 orb, -1 end of log), uint32 sec, uint32 nsec, payload
       image : width*height uint8          depth : width*height float32
       orb   : int32 n_values, float32 values[n_values]      (channels[0].values: pairs of keyframe indices)
               int32 n_path,   float64 path[n_path][7]       (px py pz qx qy qz qw)
               float64 pose[7], float64 covariance[36]       ([0] > 0: new keyframe, [1]: reference keyframe)

tests/cpp/node_replay_test.cpp reads the same format through include/dsm_surfel_map.hpp.

    python -m densesurfelmapping_amd.msglog LOG --save-cloud map.PCD --save-mesh map_mesh.PLY
"""
from __future__ import annotations

import argparse
import json

import numpy as np

from . import synth

_KIND = {"image": 0, "depth": 1, "orb": 2}


def write_log(path, cam, drift_free_poses, events):
    """events: the tuples of synth.node_messages."""
    with open(path, "wb") as f:
        f.write(np.array([cam.width, cam.height, drift_free_poses], "<i4").tobytes())
        f.write(np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.far, cam.near], "<f4").tobytes())
        for ev in events:
            kind = _KIND[ev[0]]
            f.write(np.array([kind], "<i4").tobytes() + np.array(ev[1], "<u4").tobytes())
            if kind == 0:
                f.write(np.ascontiguousarray(ev[2], "u1").tobytes())
            elif kind == 1:
                f.write(np.ascontiguousarray(ev[2], "<f4").tobytes())
            else:
                values = np.ascontiguousarray(ev[2], "<f4")
                path_poses = np.ascontiguousarray(ev[3], "<f8").reshape(-1, 7)
                f.write(np.array([values.size], "<i4").tobytes() + values.tobytes())
                f.write(np.array([len(path_poses)], "<i4").tobytes() + path_poses.tobytes())
                f.write(np.ascontiguousarray(ev[4], "<f8").tobytes() + np.ascontiguousarray(ev[5], "<f8").tobytes())
        f.write(np.array([-1], "<i4").tobytes())


def read_log(path):
    """Returns (camera, drift_free_poses, iterator over event tuples)."""
    f = open(path, "rb")

    def take(dtype, n=1):
        a = np.frombuffer(f.read(np.dtype(dtype).itemsize * n), dtype=dtype)
        if a.size != n:
            raise EOFError(f"{path}: truncated log")
        return a

    w, h, dfp = (int(v) for v in take("<i4", 3))
    fx, fy, cx, cy, far, near = (float(v) for v in take("<f4", 6))
    cam = synth.Camera(w, h, fx, fy, cx, cy, far=far, near=near)

    def events():
        while True:
            kind = int(take("<i4")[0])
            if kind < 0:
                f.close()
                return
            stamp = tuple(int(v) for v in take("<u4", 2))
            if kind == 0:
                yield ("image", stamp, take("u1", w * h).reshape(h, w))
            elif kind == 1:
                yield ("depth", stamp, take("<f4", w * h).reshape(h, w))
            else:
                values = take("<f4", int(take("<i4")[0]))
                path_poses = take("<f8", 7 * int(take("<i4")[0])).reshape(-1, 7)
                yield ("orb", stamp, values, path_poses, take("<f8", 7), take("<f8", 36))

    return cam, dfp, events()


def replay(path, save_cloud=None, save_mesh=None, device=0, surfel_capacity=0):
    """Feed a log to a SurfelMap on the GPU; returns a summary dict."""
    from . import surfel_map
    cam, dfp, events = read_log(path)
    node = surfel_map.SurfelMap(cam, drift_free_poses=dfp, device=device, surfel_capacity=surfel_capacity)
    n = 0
    for ev in events:
        node.feed(ev)
        n += 1
    out = {"messages": n, "frames_fused": node.frames_fused, "keyframes": node.pose_count,
           "active_surfels": int(len(node.local_surfels())), "inactive_surfels": int(len(node.inactive_cloud()))}
    if save_cloud:
        node.save_cloud(save_cloud)
    if save_mesh:
        node.save_mesh(save_mesh)
    node.close()
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("log")
    ap.add_argument("--save-cloud", help="ASCII PCD (SurfelMap::save_cloud)")
    ap.add_argument("--save-mesh", help="ASCII PLY hexagon mesh (SurfelMap::save_mesh)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--synth", type=int, metavar="N", help="first write LOG: N frames of the synthetic circuit at 1226x370")
    args = ap.parse_args()
    if args.synth:
        cam = synth.KITTI_1226
        write_log(args.log, cam, 10, synth.node_messages(cam, synth.Scene(), args.synth, lap=120))
    print(json.dumps(replay(args.log, args.save_cloud, args.save_mesh, device=args.device)))


if __name__ == "__main__":
    main()
