"""GPU parity at scale: the regimes the short sequences of test_gpu_parity.py never reach.

  * the SURVEY.md §8(d) parity sequence at the headline resolution: 200 frames at 1226x370 against vectors from
    the reference's own translation unit (tests/golden/make_golden_long.py) -- pruning (FF.cpp:206-210) and the
    hole refill / swap-with-last compaction (SM.cpp:1077-1109) fire at that size;
  * maps far above 65 536 surfels: the multi-round hole scan, the saturated fuse grid (grid-stride loop) and the
    K < k tail-hole chains, against the reference-TU digests AND the port oracle byte for byte;
  * BASELINE configs[4]: one 1920x1080 frame fused into a 2 M-surfel map, then the loop-closure deformation of the
    active map (dsm_map_warp) and of 200 keyframes of an inactive store of the same size (dsm_store_warp).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, fields_equal
import scale_cases
from node_state import _canon

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(oracle_built):
    import torch
    torch.cuda.init()
    from densesurfelmapping_amd import api, synth
    from oracle import bindings
    return api, synth, bindings


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "long_golden.json")))


def map_sha(a, dtype):
    return hashlib.sha256(_canon(np.ascontiguousarray(a, dtype))).hexdigest()


# The sequences of long_golden.json (reference-TU vectors): the 200-frame smooth-noise drive, and the reference's REAL kind
# of input as kitti_publisher/scripts/publisher.py:37-40 makes it -- depth = bf / disparity, disparity-quantised, +inf
# where the disparity is 0 (`stereo_inf`; `stereo_zero`: those pixels at depth 0), an image with eight grey levels and
# saturated highlights: seeds with infinite and NaN mean depths, non-finite surfels, and a tenth of the first sweep's
# pixels exactly between two seeds (k_assign's list of open picks is the common case there, not the exception).
# BASELINE configs[3] (round 6): `tum_room` / `tum_sparse` -- 640x480 under the RGB-D constant set (fusion_functions.h:17-21),
# a hand-held sweep through a room with depth as a Kinect + the TUM dataset's uint16 / 5000 PNGs deliver it (a hundred-odd
# distinct depth values per frame, zero in shadows, blobs, beyond the range and at the border), a keyframe every 4 frames,
# two laps of a closed loop: revisits, pruning (FF.cpp:207-211) and both compaction branches (SM.cpp:1087-1109) all occur.
SEQUENCES = ["drive200", "stereo_inf", "stereo_zero", "tum_room", "tum_sparse"]


@pytest.fixture(params=SEQUENCES)
def seq_case(request, gold):
    if request.param == "drive200":
        return gold["sequence"]
    if request.param.startswith("tum"):
        return gold["tum_sequences"][SEQUENCES.index(request.param) - 3]
    return gold["stereo_sequences"][SEQUENCES.index(request.param) - 1]


def _sequence(synth, case, extra=0):
    cam, scene = getattr(synth, case["camera"]), synth.Scene(**case["scene"])
    return cam, scene, list(synth.sequence(cam, scene, case["frames"] + extra, keyframe_every=case.get("keyframe_every", 5)))


def test_long_sequence_kitti_golden(mods, seq_case):
    """200 frames at 1226x370: per frame the label image (SHA-256), new and total surfel counts; every 50 frames the
    whole map; first frame by frame (one graph replay each), then again as four 50-frame batches with the default
    frame pipelining.  Vectors: the reference TU."""
    api, synth, ob = mods
    case = seq_case
    cam, scene, frames = _sequence(synth, case)
    period = scene.frames_per_period
    per = case["per_frame"]
    # the regimes this test exists for do occur in the reference's run
    assert sum(f["n_holes"] for f in per) > 1000, "no pruning / deletion in the golden run"
    assert any(f["n_holes"] > f["n_new"] for f in per), "the K < k branch (swap-with-last) never fires"
    assert any(0 < f["n_holes"] < f["n_new"] for f in per), "the K > k branch (refill + append) never fires"
    if case["name"] == "kitti1226_drive_200":
        assert case["n_mature"] > 1000
    if case["name"] == "kitti1226_stereo_inf_60":
        assert case["n_nonfinite"] > 10000, "the +inf feed leaves no non-finite surfels: the case lost its point"
    if case["name"].startswith("tum"):
        assert cam.rgbd and (cam.width, cam.height) == (640, 480) and case["keyframe_every"] == 4
        assert np.mean([np.unique(f[2]).size for f in frames[:10]]) < 1000, "the depth is not quantised as a Kinect's"

    ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 20)
    for t in range(period):
        ff.frame_upload(t, frames[t][1], frames[t][2])
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    for (t, img, dep, pose, ref), want in zip(frames, per):
        ff.fuse_frame_resident(t % period, ref, pose)
        assert ff.last_new_count() == want["n_new"], f"frame {t}: new surfels"
        assert ff.map_size() == want["n_local"], f"frame {t}: map size"
        assert hashlib.sha256(ff.labels().tobytes()).hexdigest() == want["labels_sha256"], f"frame {t}: label image"
        if str(t + 1) in case["map_sha256"]:
            assert map_sha(ff.map_download(), api.SURFEL_DTYPE) == case["map_sha256"][str(t + 1)], f"map after frame {t}"
    ff.close()

    # default pipeline depth (4: the superpixel stages of two consecutive frames per batched launch), then 16 and 32 (four
    # and eight per launch; the 50-frame chunks leave ragged ends that go frame by frame)
    for depth in (0, 16, 24, 32):  # 24, 32: eight frames per batched launch -- the lane-per-seed forms of the per-seed stages (24: three groups)
        ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 20, pipeline_depth=depth)
        for t in range(period):
            ff.frame_upload(t, frames[t][1], frames[t][2])
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        step = case["checkpoint_every"]
        for base in range(0, case["frames"], step):
            chunk = frames[base:base + step]
            ff.replay_enqueue(*ff.pack_replay([f[0] % period for f in chunk], [f[4] for f in chunk], [f[3] for f in chunk]))
            got = ff.map_download()
            assert len(got) == per[base + step - 1]["n_local"]
            assert map_sha(got, api.SURFEL_DTYPE) == case["map_sha256"][str(base + step)], \
                f"pipelined replay (depth {depth}), frames {base}..{base + step - 1}"
        assert hashlib.sha256(ff.labels().tobytes()).hexdigest() == per[-1]["labels_sha256"]
        ff.close()


def test_long_sequence_kitti_golden_batched(mods, seq_case):
    """The headline configuration of bench.py against the same vectors: EIGHT handles advancing in lockstep through the
    200 frames at 1226x370, every kernel launched once for all of them (one handle per XCD, the plane fit in its two
    tiers).  Handle b is 7 - b frames ahead of handle 7, so no two ever work on the same frame; every handle's map
    after its own 50th / 100th / 150th / 200th frame is the reference TU's, and so are its labels after its 200th."""
    api, synth, ob = mods
    case = seq_case
    n, B = case["frames"], 8
    cam, scene, frames = _sequence(synth, case, extra=B)  # the leaders run a few frames past the end while the others finish
    period = scene.frames_per_period
    step = case["checkpoint_every"]
    plan = api.FusionFunctions.pack_replay([f[0] % period for f in frames], [f[4] for f in frames], [f[3] for f in frames])
    lead = [B - 1 - b for b in range(B)]
    handles = []
    for b in range(B):
        ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 20, pipeline_depth=1)
        for t in range(period):
            ff.frame_upload(t, frames[t][1], frames[t][2])
        # a batch needs its handles' parameter rings in step: every handle has B - 1 frames behind it when the batch
        # starts -- throw-away frames first (their map is discarded), then the handle's head start on the sequence
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        waste = B - 1 - lead[b]
        if waste:
            ff.replay_enqueue(plan[0][:waste], plan[1][:waste], plan[2][:waste])
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        if lead[b]:
            ff.replay_enqueue(plan[0][:lead[b]], plan[1][:lead[b]], plan[2][:lead[b]])
        ff.synchronize()
        handles.append(ff)
    batch = api.Batch(handles)
    done = list(lead)
    checked = [[] for _ in range(B)]
    while min(done) < n:
        m = min(step - d % step for d in done)  # up to the next checkpoint of whichever handle is closest to one
        s_, r_, p_, _ = api.Batch.pack([(plan[0][done[b]:done[b] + m], plan[1][done[b]:done[b] + m], plan[2][done[b]:done[b] + m])
                                        for b in range(B)])
        batch.replay_enqueue(s_, r_, p_, m)
        batch.synchronize()
        for b in range(B):
            done[b] += m
            if done[b] % step == 0 and done[b] <= n:
                got = handles[b].map_download()
                assert len(got) == case["per_frame"][done[b] - 1]["n_local"], (b, done[b])
                assert map_sha(got, api.SURFEL_DTYPE) == case["map_sha256"][str(done[b])], f"handle {b}: map after {done[b]} frames"
                if done[b] == n:
                    assert hashlib.sha256(handles[b].labels().tobytes()).hexdigest() == case["per_frame"][-1]["labels_sha256"]
                checked[b].append(done[b])
    assert all(c == [step * (i + 1) for i in range(n // step)] for c in checked), checked
    batch.close()
    for ff in handles:
        ff.close()


def test_four_batches_in_flight_against_the_golden(mods, seq_case):
    """The exact form bench.py times: FOUR batches of 32 handles, each batch enqueued by its own host thread on its own
    stream, all in flight at once (128 subsequences sharing the machine, four of them per XCD in every launch, graphs
    captured concurrently at the first frame) -- here every one of the 128 replays the 200-frame 1226x370 golden
    sequence; each batch is checked at the 50-frame checkpoints and every handle's final map and label image are the
    reference TU's."""
    import threading
    api, synth, ob = mods
    case = seq_case
    cam, scene, frames = _sequence(synth, case)
    period, n, step = scene.frames_per_period, case["frames"], case["checkpoint_every"]
    plan = api.FusionFunctions.pack_replay([f[0] % period for f in frames], [f[4] for f in frames], [f[3] for f in frames])
    n_bat, per = 4, 32
    handles = []
    for _ in range(n_bat * per):
        ff = api.FusionFunctions.from_camera(cam, frame_slots=period, surfel_capacity=1 << 19, pipeline_depth=1)
        for t in range(period):
            ff.frame_upload(t, frames[t][1], frames[t][2])
        ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        handles.append(ff)
    batches = [api.Batch(handles[g * per:(g + 1) * per]) for g in range(n_bat)]
    errors = []

    def drive(g):
        try:
            for base in range(0, n, step):
                for c0 in range(base, base + step, 16):  # ragged chunks, several enqueue calls between checkpoints
                    c1 = min(base + step, c0 + 16)
                    s_, r_, p_, m = api.Batch.pack([(plan[0][c0:c1], plan[1][c0:c1], plan[2][c0:c1])] * per)
                    batches[g].replay_enqueue(s_, r_, p_, m)
                batches[g].synchronize()
                got = handles[g * per + (base // step) % per].map_download()
                if map_sha(got, api.SURFEL_DTYPE) != case["map_sha256"][str(base + step)]:
                    errors.append(f"batch {g}: map after {base + step} frames")
        except Exception as e:  # noqa: BLE001 -- reported in the main thread
            errors.append(repr(e))

    th = [threading.Thread(target=drive, args=(g,)) for g in range(n_bat)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    final = case["map_sha256"][str(n)]
    for i, ff in enumerate(handles):
        assert map_sha(ff.map_download(), api.SURFEL_DTYPE) == final, f"handle {i}: final map"
        assert hashlib.sha256(ff.labels().tobytes()).hexdigest() == case["per_frame"][-1]["labels_sha256"], f"handle {i}: labels"
    for bt in batches:
        bt.close()
    for ff in handles:
        ff.close()


def test_long_sequence_streamed_input(mods, seq_case):
    """Frames arriving from the host instead of sitting in HBM (the reference receives every frame through image_input /
    depth_input, surfel_map.cpp:83-101): the 200-frame 1226x370 golden sequence again, every handle with only 2 x 10
    frame slots, chunks of ten frames sent up with dsm_frame_upload_async from page-locked memory while the previous
    chunk is being fused.  ONE pipelined handle (frame groups: the uploads are waited for on the groups' lead streams)
    and the bench's form, EIGHT handles in one batch (handle b starts the sequence b frames late ... all pass the same
    checkpoints).  Maps and labels are the reference TU's: streamed == resident, byte for byte."""
    api, synth, ob = mods
    case = seq_case
    cam, scene, frames = _sequence(synth, case)
    n, step, C = case["frames"], case["checkpoint_every"], 10
    period = scene.frames_per_period

    def host_frames(ff, tight=False):
        pin = api.PinnedFrames(ff, period, tight=tight)  # the scene's period of frames in page-locked memory: slot pitch, or tight rows
        for t in range(period):
            pin.set(t, frames[t][1], frames[t][2])
        return pin

    def plan(lo, hi, base_slot):
        fr = frames[lo:hi]
        return api.FusionFunctions.pack_replay([base_slot + i for i in range(len(fr))], [f[4] for f in fr], [f[3] for f in fr])

    # ---- one handle, frame groups
    ff = api.FusionFunctions.from_camera(cam, frame_slots=2 * C, surfel_capacity=1 << 20, pipeline_depth=8)
    pin = host_frames(ff)
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    def send(k):  # chunk k -> slot half k & 1
        for i, t in enumerate(range(k * C, min(n, (k + 1) * C))):
            ff.frame_upload_async((k & 1) * C + i, pin.image(t % period), pin.depth(t % period))
    send(0)
    for k in range(n // C):
        if (k + 1) * C < n:
            send(k + 1)  # BEFORE chunk k is enqueued: ordered behind chunk k - 1, which used these slots
        ff.replay_enqueue(*plan(k * C, (k + 1) * C, (k & 1) * C))
        if ((k + 1) * C) % step == 0:
            assert map_sha(ff.map_download(), api.SURFEL_DTYPE) == case["map_sha256"][str((k + 1) * C)], f"streamed, one handle: after {(k + 1) * C} frames"
    assert hashlib.sha256(ff.labels().tobytes()).hexdigest() == case["per_frame"][-1]["labels_sha256"]
    ff.frame_uploads_wait()
    ff.close()
    pin.close()

    # ---- eight handles in one batch, all streaming the same sequence from one page-locked copy -- with TIGHT rows (width
    # elements apart: no pad bytes over the link; the upload sets the rows to the slots' pitch on the device)
    B = 8
    handles = [api.FusionFunctions.from_camera(cam, frame_slots=2 * C, surfel_capacity=1 << 20, pipeline_depth=1) for _ in range(B)]
    pin = host_frames(handles[0], tight=True)
    assert pin.pitch == cam.width <= handles[0].frame_pitch()  # (640 pixels: tight IS the pitch)
    for h in handles:
        h.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    batch = api.Batch(handles)
    def send_all(k):  # (period = 5 chunks of ten: a chunk is ten consecutive frames of the page-locked block -> one transfer per plane)
        for h in handles:
            h.frames_upload_async((k & 1) * C, pin, (k * C) % period, C)
    send_all(0)
    for k in range(n // C):
        if (k + 1) * C < n:
            send_all(k + 1)
        s_, r_, p_, m = api.Batch.pack([plan(k * C, (k + 1) * C, (k & 1) * C)] * B)
        batch.replay_enqueue(s_, r_, p_, m)
        if ((k + 1) * C) % step == 0:
            batch.synchronize()
            for b in (0, B - 1):
                assert map_sha(handles[b].map_download(), api.SURFEL_DTYPE) == case["map_sha256"][str((k + 1) * C)], f"streamed, batch handle {b}: after {(k + 1) * C} frames"
    batch.synchronize()
    for h in handles:
        assert hashlib.sha256(h.labels().tobytes()).hexdigest() == case["per_frame"][-1]["labels_sha256"]
        h.frame_uploads_wait()
    batch.close()
    for h in handles:
        h.close()
    pin.close()


def test_tum_live_callback_form(mods, gold):
    """BASELINE configs[3] the way the live node runs it (TUM-RGBD-style 640x480 at 30 Hz, one hipGraph per frame): every frame
    arrives from pageable host memory (dsm_frame_upload into one of two slots in turn), is fused by one graph replay
    (dsm_fuse_frame_resident) and waited for -- 200 frames under the RGB-D constant set, per frame the label image, the new and
    total surfel counts, every 50 frames the whole map, against the reference TU's vectors.  Then the drop-in call
    (dsm_fuse_map: host vector in and out) over the first 50 frames, and the second-tier tap on a batch of eight."""
    api, synth, ob = mods
    case = gold["tum_sequences"][0]
    cam, scene, frames = _sequence(synth, case)
    per = case["per_frame"]
    ff = api.FusionFunctions.from_camera(cam, frame_slots=2, surfel_capacity=1 << 18)
    ff.map_upload(np.zeros(0, api.SURFEL_DTYPE))
    for (t, img, dep, pose, ref), want in zip(frames, per):
        ff.frame_upload(t & 1, img, dep)
        ff.fuse_frame_resident(t & 1, ref, pose)
        ff.synchronize()
        assert (ff.last_new_count(), ff.map_size()) == (want["n_new"], want["n_local"]), f"frame {t}"
        assert hashlib.sha256(ff.labels().tobytes()).hexdigest() == want["labels_sha256"], f"frame {t}: label image"
        if str(t + 1) in case["map_sha256"]:
            assert map_sha(ff.map_download(), api.SURFEL_DTYPE) == case["map_sha256"][str(t + 1)], f"map after frame {t}"
    ff.close()
    # the drop-in call, frame after frame on the caller's own array
    ff = api.FusionFunctions.from_camera(cam, surfel_capacity=1 << 18)
    local = np.zeros(0, api.SURFEL_DTYPE)
    for (t, img, dep, pose, ref), want in zip(frames[:50], per):
        local, k = ff.fuse_map(ref, img, dep, pose, local)
        assert (k, len(local)) == (want["n_new"], want["n_local"]), f"drop-in, frame {t}"
    assert map_sha(local, api.SURFEL_DTYPE) == case["map_sha256"]["50"], "drop-in calls: map after 50 frames"
    ff.close()
    # a batch of eight (lane-per-seed kernels): the tap that says how many seeds went on to the second tiers
    hs = []
    for _ in range(8):
        h = api.FusionFunctions.from_camera(cam, frame_slots=4, surfel_capacity=1 << 18, pipeline_depth=1)
        for i in range(4):
            h.frame_upload(i, frames[i][1], frames[i][2])
        h.map_upload(np.zeros(0, api.SURFEL_DTYPE))
        hs.append(h)
    bt = api.Batch(hs)
    pl = api.FusionFunctions.pack_replay(list(range(4)), [f[4] for f in frames[:4]], [f[3] for f in frames[:4]])
    s_, r_, p_, m = api.Batch.pack([pl] * 8)
    bt.replay_enqueue(s_, r_, p_, m)
    bt.synchronize()
    S = (cam.width // 8) * (cam.height // 8)
    for h in hs:
        tc = h.debug_tier_counts()
        assert all(0 <= v <= S for v in tc["huber_rest_by_sweep"] + tc["long_list_by_sweep"]) and 0 <= tc["fit_long_groups"] <= S // 4, tc
        assert sum(tc["huber_rest_by_sweep"]) > 0, tc  # (some superpixel always needs a second Huber pass on this input)
        assert tc == hs[0].debug_tier_counts()
        assert h.map_size() == per[3]["n_local"]
    bt.close()
    for h in hs:
        h.close()


def _large_case(mods, gold_rows, case, dropin_trial=None):
    api, synth, ob = mods
    cam = getattr(synth, case["camera"])
    big, (t, img, dep, pose, ref) = scale_cases.large_map_inputs(ob.PortOracle(cam), synth, ob.SURFEL_DTYPE, case)
    ff = api.FusionFunctions.from_camera(cam, frame_slots=1, surfel_capacity=len(big) + 65536)
    ff.frame_upload(0, img, dep)
    out = []
    for i, (trial, want) in enumerate(zip(case["trials"], gold_rows)):
        m = scale_cases.large_map_variant(big, trial, synth)
        assert want["trial"] == trial and len(m) == want["n_in"]
        assert map_sha(m, ob.SURFEL_DTYPE) == want["in_sha256"], "the input map is not the one the golden record was made from"
        if i == dropin_trial:  # SurfelMap::fuse_map drop-in call, host buffers in and out
            got, k = ff.fuse_map(ref, img, dep, pose, m.astype(api.SURFEL_DTYPE))
        else:
            ff.map_upload(m.astype(api.SURFEL_DTYPE))
            ff.fuse_frame_resident(0, ref, pose)
            k = ff.last_new_count()
            got = ff.map_download()
        assert k == want["n_new"] and len(got) == want["n_local"], (trial, k, len(got), want["n_new"], want["n_local"])
        assert hashlib.sha256(ff.labels().tobytes()).hexdigest() == want["labels_sha256"]
        assert (got["update_times"] != 0).all()
        if map_sha(got, api.SURFEL_DTYPE) != want["map_sha256"]:  # say where: rerun on the port oracle
            o, _ = ob.PortOracle(cam).fuse_map(ref, img, dep, pose, m)
            raise AssertionError(f"{trial}: map differs from the reference TU's; vs port oracle: {fields_equal(got, o.astype(api.SURFEL_DTYPE))}")
        out.append(got)
    return ff, out, (ref, pose)


def test_large_map_compaction_golden(mods, gold):
    """600 k surfels, 10 / 50 / 90 % stale: tail_hole_scan needs 10 rounds (9 375 bitmap words), the fuse grid is
    saturated (2 048 blocks x 256 < 600 k), and with 90 % stale K << k leaves ~60 k tail holes to chain through."""
    case = scale_cases.LARGE_MAP
    rows = gold["large_map"]
    assert rows[0]["n_in"] > 2048 * 256 and rows[0]["n_in"] // 64 > 1024
    assert rows[2]["n_holes"] > 50 * rows[2]["n_new"] and rows[0]["n_holes"] > rows[0]["n_new"]
    ff, _, _ = _large_case(mods, rows, case, dropin_trial=1)
    ff.close()


def test_batched_large_maps_headline_regime(mods, gold):
    """The regime bench.py's headline runs in, oracle-checked: EIGHT handles in one batch, every one of them starting from
    a map above 262 144 surfels -- k_frame_tail<BATCH> leaves its single-trip path (kTailFastWords) and k_fuse_surfels<BATCH>
    runs its grid-stride loop -- with different stale fractions per handle, so that the K < k chains (SM.cpp:1104-1109),
    K > k appends and the plain refill all occur inside ONE launch.  Frame 1: the 10 / 50 / 90 % stale maps against the
    reference-TU digests, a nothing-stale map (K >= k) against the port oracle.  Then eight more frames, every handle
    against the port oracle byte for byte.  Last, the same start through one handle's frame groups (pipeline depth 16)."""
    api, synth, ob = mods
    case = scale_cases.LARGE_MAP
    rows = gold["large_map"]
    cam = getattr(synth, case["camera"])
    scene = synth.Scene(**case["scene"])
    more = 8
    frames = list(synth.sequence(cam, scene, case["base_frames"] + 1 + more))[case["base_frames"]:]
    big, first = scale_cases.large_map_inputs(ob.PortOracle(cam), synth, ob.SURFEL_DTYPE, case)
    assert first[0] == frames[0][0] and len(big) > 262144 + 8 * 7038
    trials = list(case["trials"]) + [{"stale": 0.0, "dead": 0.0}]
    which = [0, 1, 2, 3, 2, 1, 0, 3]
    starts = [scale_cases.large_map_variant(big, t, synth) for t in trials]
    slots, refs, poses = api.FusionFunctions.pack_replay(list(range(len(frames))), [f[4] for f in frames], [f[3] for f in frames])

    def start_handle(depth):
        ff = api.FusionFunctions.from_camera(cam, frame_slots=len(frames), surfel_capacity=len(big) + 16 * 7038 + 65536, pipeline_depth=depth)
        for i, f in enumerate(frames):
            ff.frame_upload(i, f[1], f[2])
        return ff

    handles = [start_handle(1) for _ in which]
    for ff, w in zip(handles, which):
        ff.map_upload(starts[w].astype(api.SURFEL_DTYPE))
    batch = api.Batch(handles)
    one = api.Batch.pack([(slots[:1], refs[:1], poses[:1])] * len(which))
    batch.replay_enqueue(one[0], one[1], one[2], 1)
    batch.synchronize()
    # what the port oracle makes of every start map, frame by frame (it is what a failing digest is diffed against)
    models = []
    for w in range(len(trials)):
        o, k = ob.PortOracle(cam).fuse_map(frames[0][4], frames[0][1], frames[0][2], frames[0][3], starts[w])
        models.append(o)
        if w < len(rows):
            assert len(o) == rows[w]["n_local"] and map_sha(o, ob.SURFEL_DTYPE) == rows[w]["map_sha256"], "port oracle vs reference TU"
    seen_less, seen_more = False, False
    for b, (ff, w) in enumerate(zip(handles, which)):
        got = ff.map_download()
        assert len(starts[w]) > 262144  # the map the frame was fused into: general tail path, grid-stride fuse
        if w < len(rows):
            assert ff.last_new_count() == rows[w]["n_new"] and len(got) == rows[w]["n_local"], (b, w)
            assert map_sha(got, api.SURFEL_DTYPE) == rows[w]["map_sha256"], f"handle {b} (trial {trials[w]}): map differs from the reference TU's"
            seen_less |= rows[w]["n_holes"] > rows[w]["n_new"]
        else:
            assert len(got) > len(starts[w]), "the nothing-stale start must take the K >= k branch"
            seen_more = True
        assert fields_equal(got, models[w].astype(api.SURFEL_DTYPE)) == [], (b, w)
    assert seen_less and seen_more
    # eight more frames in lockstep, checked after the 4th and the 8th
    oracles = [ob.PortOracle(cam) for _ in trials]
    done = 1
    for chunk in (4, 4):
        pk = api.Batch.pack([(slots[done:done + chunk], refs[done:done + chunk], poses[done:done + chunk])] * len(which))
        batch.replay_enqueue(pk[0], pk[1], pk[2], chunk)
        batch.synchronize()
        for w in range(len(trials)):
            for f in frames[done:done + chunk]:
                models[w], _ = oracles[w].fuse_map(f[4], f[1], f[2], f[3], models[w])
        done += chunk
        for b, (ff, w) in enumerate(zip(handles, which)):
            got = ff.map_download()
            assert fields_equal(got, models[w].astype(api.SURFEL_DTYPE)) == [], f"handle {b} after {done} frames"
        assert sum(len(m) > 262144 for m in models) >= 3  # (the 90 % stale start shrinks below the fast-path limit)
    for b, (ff, w) in enumerate(zip(handles, which)):
        assert np.array_equal(ff.labels(), oracles[w].labels()), b
    batch.close()
    for ff in handles:
        ff.close()
    # frame groups: one handle, depth 16 (four frames per batched superpixel launch), the 50 % stale start
    ff = start_handle(16)
    ff.map_upload(starts[1].astype(api.SURFEL_DTYPE))
    ff.replay_enqueue(slots[:1 + more], refs[:1 + more], poses[:1 + more])
    got = ff.map_download()
    assert len(got) > 262144
    assert fields_equal(got, models[1].astype(api.SURFEL_DTYPE)) == [], "frame groups on a large map"
    ff.close()


def test_map_grows_past_the_tail_fast_path(mods):
    """A map that starts just below 262 144 surfels and grows past it while frames are being replayed: the graphs captured
    for the small map (k_frame_tail with one workgroup) are dropped and come back with the workgroups that list a large
    map's holes (dsm_api.hip, map_grows); k_fuse_surfels starts counting holes per chunk.  One handle with one graph per
    frame, one with frame groups, and a batch of eight, each against the port oracle after every chunk of frames."""
    api, synth, ob = mods
    case = scale_cases.LARGE_MAP
    cam = getattr(synth, case["camera"])
    scene = synth.Scene(**case["scene"])
    n = 9
    frames = list(synth.sequence(cam, scene, case["base_frames"] + n))[case["base_frames"]:]
    big, _ = scale_cases.large_map_inputs(ob.PortOracle(cam), synth, ob.SURFEL_DTYPE, case)
    start = scale_cases.large_map_variant(big[:259000], {"stale": 0.004, "dead": 0.002}, synth)
    slots, refs, poses = api.FusionFunctions.pack_replay(list(range(n)), [f[4] for f in frames], [f[3] for f in frames])
    oracle = ob.PortOracle(cam)
    models, m = [], start
    for f in frames:
        m, _ = oracle.fuse_map(f[4], f[1], f[2], f[3], m)
        models.append(m)
    sizes = [len(start)] + [len(x) for x in models]
    assert sizes[0] < 262144 and sizes[3] < 262144 < sizes[-3], sizes  # crosses in the middle of the replay

    def start_handle(depth):
        ff = api.FusionFunctions.from_camera(cam, frame_slots=n, surfel_capacity=600_000, pipeline_depth=depth)
        for i, f in enumerate(frames):
            ff.frame_upload(i, f[1], f[2])
        ff.map_upload(start.astype(api.SURFEL_DTYPE))
        return ff

    for depth in (1, 16):
        ff = start_handle(depth)
        done = 0
        for chunk in (2, 4, 3):
            ff.replay_enqueue(slots[done:done + chunk], refs[done:done + chunk], poses[done:done + chunk])
            done += chunk
            got = ff.map_download()
            assert fields_equal(got, models[done - 1].astype(api.SURFEL_DTYPE)) == [], (depth, done)
        ff.close()
    handles = [start_handle(1) for _ in range(8)]
    batch = api.Batch(handles)
    done = 0
    for chunk in (3, 3, 3):
        pk = api.Batch.pack([(slots[done:done + chunk], refs[done:done + chunk], poses[done:done + chunk])] * len(handles))
        batch.replay_enqueue(pk[0], pk[1], pk[2], chunk)
        batch.synchronize()
        done += chunk
        for b, ff in enumerate(handles):
            assert fields_equal(ff.map_download(), models[done - 1].astype(api.SURFEL_DTYPE)) == [], (b, done)
    batch.close()
    for ff in handles:
        ff.close()


def _random_rigid(rng, scale=0.05):
    a = rng.normal(size=3) * scale
    th = np.linalg.norm(a)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = rng.normal(size=3) * 0.3
    return m.astype(np.float32)


def test_fullhd_2m_fuse_and_warp(mods, gold):
    """BASELINE configs[4]: a 1920x1080 frame fused into a 2 M-surfel live map (reference-TU digests), then the
    loop-closure deformation kernels at that size against the oracle's warp: the active map by one matrix
    (SM.cpp:750-789), and the same 2 M surfels as the inactive store of 200 keyframes with per-keyframe matrices,
    untouched keyframes and the stale last point of every warped patch (SM.cpp:681-748)."""
    api, synth, ob = mods
    case = scale_cases.FULLHD_2M
    rows = gold["fullhd_2m"]
    assert rows[0]["n_in"] >= 2_000_000
    ff, maps, (ref, pose) = _large_case(mods, rows, case)
    local = maps[0]
    rng = np.random.default_rng(777)
    # active map: ONE matrix for all (the resident map still holds maps[0])
    warp = _random_rigid(rng)
    ff.map_warp(warp)
    got = ff.map_download()
    want = ob.port_warp(local.astype(ob.SURFEL_DTYPE), warp).astype(api.SURFEL_DTYPE)
    assert fields_equal(got, want) == []
    # inactive store: give the surfels 200 keyframes, deactivate them all (order of keys shuffled), warp
    n_key = 200
    lu = (np.arange(len(got)) * 2654435761 % (1 << 32) >> 8) % n_key
    got["last_update"] = lu.astype(np.int32)
    ff.map_upload(got)
    model = got.astype(ob.SURFEL_DTYPE)
    store = []
    offsets = [0]
    keys = rng.permutation(n_key)
    for key in keys:
        b, n = ff.store_deactivate(int(key))
        seg = model[model["last_update"] == key]
        assert (b, n) == (offsets[-1], len(seg))
        store.append(seg)
        offsets.append(b + n)
    assert ff.map_size() == len(model)  # slots are marked deleted, the array keeps its length until the next fuse
    store = np.concatenate(store)
    cloud = np.stack([store["px"], store["py"], store["pz"], store["color"]], axis=1).astype(np.float32)
    offsets = np.array(offsets, np.int32)
    mats = np.stack([_random_rigid(rng) for _ in keys])
    changed = (rng.random(n_key) < 0.7).astype(np.uint8)
    ff.store_warp(offsets, mats, changed)
    for g in range(n_key):
        if not changed[g]:
            continue
        b, e = offsets[g], offsets[g + 1]
        store[b:e] = ob.port_warp(store[b:e], mats[g])
        if e - b > 1:
            cloud[b:e - 1] = np.stack([store["px"][b:e - 1], store["py"][b:e - 1], store["pz"][b:e - 1], store["color"][b:e - 1]], axis=1)
    s, c = ff.store_download()
    assert fields_equal(s, store.astype(api.SURFEL_DTYPE)) == []
    assert np.array_equal(c.view("u4"), cloud.view("u4"))
    ff.close()


def test_fullhd_frame_groups(mods, gold):
    """BASELINE configs[4] as one sequence through one handle with frames in flight: fourteen 1920x1080 frames fused into the
    2 M-surfel map of test_fullhd_2m_fuse_and_warp, strictly serial (pipeline depth 1) and through frame groups (depth 12:
    the superpixel stages of four frames per batched launch, wave per seed, a ragged end of two frame by frame; depth 24: eight
    frames, lane per seed, a ragged end of six), each
    against the port oracle's replay byte for byte -- the first frame also against the reference-TU digest."""
    api, synth, ob = mods
    case = scale_cases.FULLHD_2M
    row = gold["fullhd_2m"][0]
    cam = getattr(synth, case["camera"])
    scene = synth.Scene(**case["scene"])
    n = 14
    frames = list(synth.sequence(cam, scene, case["base_frames"] + n))[case["base_frames"]:]
    big, first = scale_cases.large_map_inputs(ob.PortOracle(cam), synth, ob.SURFEL_DTYPE, case)
    assert first[0] == frames[0][0]
    start = scale_cases.large_map_variant(big, case["trials"][0], synth)
    assert map_sha(start, ob.SURFEL_DTYPE) == row["in_sha256"]
    oracle, m, models = ob.PortOracle(cam), start, []
    for f in frames:
        m, _ = oracle.fuse_map(f[4], f[1], f[2], f[3], m)
        models.append(m)
    assert map_sha(models[0], ob.SURFEL_DTYPE) == row["map_sha256"], "port oracle vs reference TU"
    slots, refs, poses = api.FusionFunctions.pack_replay(list(range(n)), [f[4] for f in frames], [f[3] for f in frames])
    for depth in (1, 12, 24):
        ff = api.FusionFunctions.from_camera(cam, frame_slots=n, surfel_capacity=len(start) + 400_000, pipeline_depth=depth)
        for i, f in enumerate(frames):
            ff.frame_upload(i, f[1], f[2])
        ff.map_upload(start.astype(api.SURFEL_DTYPE))
        ff.replay_enqueue(slots, refs, poses)
        got = ff.map_download()
        assert len(got) == len(models[-1]) and len(got) > 1_900_000, (depth, len(got), len(models[-1]))
        assert fields_equal(got, models[-1].astype(api.SURFEL_DTYPE)) == [], f"pipeline depth {depth}"
        assert np.array_equal(ff.labels(), oracle.labels()), depth
        ff.close()


def test_node_at_kitti_resolution(mods):
    """The whole node on the GPU at the headline resolution (130 frames at 1226x370 through the message callbacks: stamp
    matching, pose graph, active / inactive sets in HBM, loop closure with the warp of ~50 k inactive surfels on ten
    keyframes and of the active map, re-activation) against digests recorded from the reference's own surfel_map.cpp:
    counts after every pose message, the whole state every ten, the final state and the saved PCD / PLY."""
    import test_cpu
    from densesurfelmapping_amd import surfel_map
    for case, gold in test_cpu._node_cases_large():
        assert gold["briefs"][-1][3] > 20000, "the scenario no longer builds a sizeable inactive set"
        test_cpu._check_node_run(case, gold, lambda cam, d: surfel_map.SurfelMap(cam, drift_free_poses=d, surfel_capacity=1 << 20))
