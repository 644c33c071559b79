"""How often k_assign's fp32 cost filter (dsm_math.h, pick_seed_fast) leaves a pick open, by input family -- counted on the
host by tests/hostemu.cpp, which runs the filter beside the reference's typed pick on every pixel of every sweep (CPU
only; test infrastructure).  Writes profiles/r06_open_picks.json, which bench.py's `kitti_like` and `tum_like` legs quote.

    python tools/open_picks.py [frames]
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from densesurfelmapping_amd import synth  # noqa: E402
from oracle import bindings as ob  # noqa: E402
from test_cpu import Emu, STEREO_SCENE  # noqa: E402

TUM_SCENE = dict(seed=7, tum=True, frames_per_period=100, intensity_noise=8.0, checker=25.0, n_boxes=6)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    lib = os.path.join(ROOT, "tests", "_build", "libhostemu.so")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(ROOT, "tests", "hostemu.cpp"), "-o", lib], check=True)
    out = {"camera": "KITTI_1226 unless the family says otherwise", "frames": frames,
           "source": "tests/hostemu.cpp (pick_seed_fast beside pick_seed on every pixel of every sweep)", "families": {}}
    for name, cam, scene, kf in (("smooth_noise (bench default)", synth.KITTI_1226, synth.Scene(), 5),
                                 ("kitti_like, +inf kept", synth.KITTI_1226, synth.Scene(**STEREO_SCENE), 5),
                                 ("kitti_like, inf -> 0", synth.KITTI_1226, synth.Scene(zero_disparity_inf=False, **STEREO_SCENE), 5),
                                 ("tum_like (640x480, RGB-D constants)", synth.VGA_RGBD, synth.Scene(**TUM_SCENE), 4),
                                 ("tum_like room through an ideal sensor", synth.VGA_RGBD, synth.Scene(tum_sensor=False, **TUM_SCENE), 4)):
        emu = Emu(lib, cam)
        le = np.zeros(0, ob.SURFEL_DTYPE)
        for t, img, dep, pose, ref in synth.sequence(cam, scene, frames, keyframe_every=kf):
            le, _ = emu.fuse_map(ob.SURFEL_DTYPE, ref, img, dep, pose, le)
        st = (C.c_longlong * 24)()
        emu.lib.emu_fast_pick_stats.argtypes = [C.c_void_p, C.c_void_p]
        emu.lib.emu_fast_pick_stats(emu.h, st)
        rows = st[6] * cam.height * ((cam.width + 63) // 64)
        out["families"][name] = {"pixel_picks": st[0], "open": st[1], "open_frac": round(st[1] / st[0], 5), "wrong_answers": st[2],
                                 "open_by_sweep": [st[12], st[13], st[14]],
                                 "open_frac_of_first_sweep": round(st[12] / (st[0] / 3), 5),
                                 "wave_rows_with_an_open_pixel": st[5], "wave_rows": rows, "wave_row_frac": round(st[5] / rows, 4)}
        print(name, out["families"][name], flush=True)
    with open(os.path.join(ROOT, "profiles", "r06_open_picks.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
