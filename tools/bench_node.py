#!/usr/bin/env python
"""Node-level rate: frames/s through the message callbacks of SurfelMap (image_input, depth_input,
orb_results_input -> stamp matching, pose graph, active / inactive sets, per-frame engine, loop-closure warp) at
1226x370 with the launch file's drift_free_poses = 10.  Host-inclusive: every frame is copied into the node's
buffers and uploaded over PCIe, as the ROS callbacks imply.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densesurfelmapping_amd import surfel_map, synth  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
cam = synth.KITTI_1226
events = list(synth.node_messages(cam, synth.Scene(), n_frames, lap=120))
node = surfel_map.SurfelMap(cam, drift_free_poses=10, surfel_capacity=1 << 21)
warm = 3 * 20
for ev in events[:warm]:
    node.feed(ev)
node.local_surfels()  # synchronise
t0 = time.perf_counter()
per_frame = []
t_prev = t0
for i, ev in enumerate(events[warm:]):
    node.feed(ev)
    if ev[0] == "orb":
        now = time.perf_counter()
        per_frame.append(now - t_prev)
        t_prev = now
n_local = len(node.local_surfels())
dt = time.perf_counter() - t0
fused = node.frames_fused - 20
per_frame.sort()
cpu = None
if "--cpu-reference" in sys.argv:
    # the reference's own node class (surfel_map.cpp + fusion_functions.cpp compiled in place, real std::threads) on
    # this box's host cores, same message stream -- a reported baseline, test infrastructure (oracle/)
    from oracle import bindings as ob  # noqa: E402
    ref = ob.RefSurfelMap(cam, drift_free_poses=10, kind="map_threads")
    n_cpu = 3 * 120
    for ev in events[:60]:
        ref.feed(ev)
    t_c = time.perf_counter()
    for ev in events[60:n_cpu]:
        ref.feed(ev)
    dt_c = time.perf_counter() - t_c
    cpu = {"value": round((n_cpu // 3 - 20) / dt_c, 1), "unit": "frames/s", "frames": n_cpu // 3 - 20, "kind": "reference",
           "cores": 10, "host_cpus": os.cpu_count(), "local_surfels": len(ref.local_surfels())}
    ref.close()
print(json.dumps({"cpu_reference_node": cpu, "metric": "frames/s through the node's message callbacks", "workload": "1226x370 circuit of 120 frames, keyframe every 5, drift_free_poses 10, loop closure at frame 120",
                  "frames": fused, "value": round(fused / dt, 1), "unit": "frames/s",
                  "ms_per_frame_p50": round(per_frame[len(per_frame) // 2] * 1e3, 3), "ms_per_frame_max": round(per_frame[-1] * 1e3, 3),
                  "keyframes": node.pose_count, "local_surfels": n_local, "inactive_surfels": len(node.inactive_cloud()),
                  "note": "host-inclusive (buffer copies, PCIe upload of 2.27 MB per frame, Python ctypes calls)"}))
