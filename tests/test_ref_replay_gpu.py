"""GPU parity against the REFERENCE ITSELF: every trackNewCoarse and every optimize call the reference's own FullSystem made on a synthetic
sequence (tests/golden/reference_run_256x192.npz, recorded from oracle/_ref/libref.so by tests/golden/make_reference_run.py) is replayed
through libdmvio_hip.so from the recorded inputs, and the results are compared with what the reference produced.
Tolerances are the north-star's: 1e-3 m on poses (measured: < 1e-5), 1e-4 relative on the photometric energy."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import replay  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_run_256x192.npz")
IDENT = np.array([[0, 0, 0, 0, 0, 0, 1.0]])


@pytest.fixture(scope="module")
def golden(synth):
    g = replay.load_golden(GOLDEN)
    K4, imgs, _ = replay.make_sequence(synth, g["w"], g["h"], g["n_frames"], g["step"])
    g["imgs"] = imgs
    return g


def test_hip_replays_every_recorded_track_new_coarse(pkg, golden, gpu_required):
    w, h = golden["w"], golden["h"]
    ctx = pkg.Context(w, h, n_slots=golden["n_frames"])
    for k, img in enumerate(golden["imgs"]):
        ctx.frame_upload(k, img)
    trk = pkg.CoarseTrackerHip(ctx)
    tracks, _ = replay.pair_events(golden["events"])
    cur = None
    worst = np.zeros(3)
    for sr, ti, to in tracks:
        if cur is not sr:
            trk.makeK(sr["K4"])
            trk.setCoarseTrackingRef(int(sr["ref_id"]), sr["u"], sr["v"], sr["idepth"], sr["hdiF"], ref_exposure=float(sr["exposure"]), ref_aff=sr["aff"]); cur = sr
        if ti["n_history"] == 2 or not ti["poses_valid"]:
            tries = IDENT
        else:
            tries = pkg.make_track_hypotheses(ti["slast_c2w"], ti["sprelast_c2w"], ti["lastF_c2w"])
        r = trk.trackNewCoarse(int(ti["frame_id"]), tries, aff_last=ti["aff_last"], lastCoarseRMSE=ti["lastCoarseRMSE"], reTrackThreshold=float(ti["reTrackThreshold"]))
        assert r["good"] == bool(to["good"])
        dp = np.abs(r["pose7"] - to["refToNew"]).max(); da = np.abs(r["aff"] - to["aff"]) / np.array([1.0, 100.0])
        # what north_star bounds is the ENERGY: per-level photometric energy (achieved residual squared) within 1e-4 relative of the reference's own
        dr = np.nanmax(np.abs(r["achievedRes"] ** 2 - to["lastCoarseRMSE"] ** 2) / to["lastCoarseRMSE"] ** 2)
        worst = np.maximum(worst, [dp, da.max(), dr])
        # affine brightness is only sanity-checked: gain a and offset b trade against each other along a flat valley of the photometric energy (the energy is equal to ~6e-6
        # relative either way — asserted above); the fp32 sums of the two implementations stop at slightly different points of it — seen up to 1.7e-3 in a, 0.23 grey values
        # in b on early frames with few points
        assert dp < 1e-4 and dr < 1e-4 and da[0] < 2e-2 and da[1] < 2e-2, (ti["frame_id"], dp, da, dr)
    print("worst over %d recorded calls: pose %.2e, affine %.2e, residual %.2e (relative)" % (len(tracks), worst[0], worst[1], worst[2]))
    assert len(tracks) >= 50 and worst[0] < 1e-4      # bar: 1e-3 m; most calls agree to < 1e-6, early frames right after the initialiser (few points, weak geometry) to ~5e-5
    trk.close(); ctx.close()


@pytest.mark.parametrize("accumulators", [None, 1], ids=["default-4-partials", "single-threaded-order"])
def test_hip_replays_every_recorded_optimize(pkg, golden, gpu_required, accumulators):
    """Both accumulation orders of the library: its default (4 partial accumulators per bucket — the structure of the reference's multi-threaded mode) and the reference's
    single-threaded order (what the recording was made with)."""
    w, h = golden["w"], golden["h"]
    _, opts = replay.pair_events(golden["events"])
    ctx = pkg.Context(w, h, n_slots=8)
    seen = []
    for a, b in opts:
        case = replay.window_case(a, golden["imgs"], w, h)
        for k in range(a["F"]):
            ctx.frame_upload(k, case["imgs"][k])
        ba = pkg.BundleAdjusterHip(ctx, accumulators=accumulators)
        ba.set_case(case, list(range(a["F"])))
        replay.apply_window_state(ba, a)
        r = ba.optimize(6)
        dpose = max(np.abs(ba.frame_pose(k)[0] - b["frames"][k]["w2c"]).max() for k in range(a["F"]))
        daff = max(np.abs(ba.frame_pose(k)[1] - np.array([b["frames"][k]["state"][6] * 10.0, b["frames"][k]["state"][7] * 1000.0])).max() for k in range(a["F"]))
        seen.append((a["F"], a["R"], r["rmse"], b["rmse"], dpose, daff))
        assert abs(r["rmse"] - b["rmse"]) < 1e-4 * b["rmse"] and dpose < 1e-3 and daff < 1e-2, seen[-1]
        ba.close()
    print("\n".join("F=%d R=%d rmse %.6f (reference %.6f) dpose %.2e daff %.2e" % s for s in seen))
    fs = [s[0] for s in seen]
    assert fs[:7] == [2, 3, 4, 5, 6, 7, 8] and all(f == 8 for f in fs[7:])
    ctx.close()
