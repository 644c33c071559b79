"""Generate tests/golden/node_*.{json,npz} from the REFERENCE's own node class.

Runs only where /root/reference exists (this container): oracle/_ref/libdsm_ref_map.so is
surfel_fusion/src/surfel_map.cpp + fusion_functions.cpp compiled in place (oracle/Makefile, `make ref`;
oracle/ref_map_driver.cpp), fed with the message streams of densesurfelmapping_amd/synth.node_messages.
The fixtures travel to the GPU box, the reference does not.

    python tests/golden/make_node_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from densesurfelmapping_amd import synth  # noqa: E402
from oracle.bindings import RefSurfelMap  # noqa: E402
import node_state  # noqa: E402


def run_case(case, make_node, checkpoint_every=10):
    """Feed the scenario to a node; returns (per-orb-event briefs, checkpoint digests, final snapshot, file digests)."""
    cam, scene = node_state.camera_and_scene(case, synth)
    node = make_node(cam, case["drift_free_poses"])
    briefs, checkpoints = [], {}
    n_orb = 0
    kw = dict(case["kw"])
    if cam.width * cam.height > 200_000:  # big frames: render the lap once, on worker processes, cached in /tmp
        lap = kw.get("lap", 40)
        kw["frames"] = dict(enumerate(synth.render_many([(cam, scene, t) for t in range(lap)])))
    for ev in synth.node_messages(cam, scene, case["frames"], **kw):
        node.feed(ev)
        if ev[0] == "orb":
            briefs.append(node_state.brief(node))
            n_orb += 1
            if n_orb % checkpoint_every == 0:
                checkpoints[str(n_orb)] = node_state.digest(node_state.snapshot(node))
    final = node_state.snapshot(node)
    files = {}
    with tempfile.TemporaryDirectory() as td:
        pcd, ply = os.path.join(td, "map.PCD"), os.path.join(td, "map_mesh.PLY")
        node.save_cloud(pcd)
        node.save_mesh(ply)
        files = {"pcd": node_state.file_digest(pcd), "ply": node_state.file_digest(ply)}
    node.close()
    return briefs, checkpoints, final, files


def main():
    out = {"generator": "oracle/_ref/libdsm_ref_map.so (reference surfel_map.cpp + fusion_functions.cpp, workers run in index order at join)",
           "cases": []}
    for case in node_state.SCENARIOS:
        briefs, checkpoints, final, files = run_case(case, lambda cam, d: RefSurfelMap(cam, drift_free_poses=d))
        fname = "node_" + case["name"] + "_final.npz"
        np.savez_compressed(os.path.join(HERE, fname), **final)
        out["cases"].append({"name": case["name"], "briefs": briefs, "checkpoints": checkpoints, "final": fname,
                             "final_digest": node_state.digest(final), "files": files})
        print(case["name"], "fused/poses/local/inactive:", briefs[-1], "pcd bytes", files["pcd"]["bytes"], "ply bytes", files["ply"]["bytes"])
    out["large_cases"] = []
    for case in node_state.SCENARIOS_LARGE:
        briefs, checkpoints, final, files = run_case(case, lambda cam, d: RefSurfelMap(cam, drift_free_poses=d))
        out["large_cases"].append({"name": case["name"], "briefs": briefs, "checkpoints": checkpoints, "final": None,
                                   "final_digest": node_state.digest(final), "files": files,
                                   "final_counts": {k: int(len(v)) for k, v in final.items()}})
        print(case["name"], "fused/poses/local/inactive:", briefs[-1], "pcd bytes", files["pcd"]["bytes"], "ply bytes", files["ply"]["bytes"])
    with open(os.path.join(HERE, "node_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
