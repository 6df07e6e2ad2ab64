"""Every trackNewCoarse and every optimize call that the REFERENCE'S OWN FullSystem made on a synthetic sequence, replayed through the CPU
oracle from the recorded inputs and compared with what the reference produced:
  * the committed fixture tests/golden/reference_run_256x192.npz (made by tests/golden/make_reference_run.py from oracle/_ref/libref.so),
  * a fresh run of the reference, where libref.so is available (this container)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import replay  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_run_256x192.npz")
IDENT = np.array([[0, 0, 0, 0, 0, 0, 1.0]])


def replay_tracks_oracle(O, events, imgs, w, h):
    """-> list of (pose diff, aff diff, residual diff, good agrees) per recorded call"""
    tracks, _ = replay.pair_events(events)
    T = O.Tracker(w, h)
    cur = None
    dI = {}
    out = []
    for sr, ti, to in tracks:
        assert sr is not None
        if cur is not sr:
            T.make_k(sr["K4"])
            T.set_ref(O.make_images(imgs[sr["ref_id"]], w, h)[0], sr["u"], sr["v"], sr["idepth"], sr["hdiF"], exposure=sr["exposure"], aff=sr["aff"]); cur = sr
        T.set_new(O.make_images(imgs[ti["frame_id"]], w, h)[0])
        tries = IDENT if (ti["n_history"] == 2 or not ti["poses_valid"]) else O.make_track_hypotheses(ti["slast_c2w"], ti["sprelast_c2w"], ti["lastF_c2w"])
        r = T.track_new_coarse(tries, aff_last=ti["aff_last"], lastCoarseRMSE=ti["lastCoarseRMSE"], reTrackThreshold=ti["reTrackThreshold"])
        out.append((np.abs(r["pose7"] - to["refToNew"]).max(), np.abs(r["aff"] - to["aff"]).max(), np.nanmax(np.abs(r["achievedRes"] - to["lastCoarseRMSE"])),
                    r["good"] == bool(to["good"])))
    return out


def replay_windows(make_window, events, imgs, w, h):
    _, opts = replay.pair_events(events)
    out = []
    for a, b in opts:
        assert np.array_equal(a["idepth"], a["idepth_zero"]) and not a["res_linearized"].any()   # what the ABI assumes about a window at optimize time
        W = make_window(replay.window_case(a, imgs, w, h))
        replay.apply_window_state(W, a)
        r = W.optimize(6)
        dpose = max(np.abs(W.frame_pose(k)[0] - b["frames"][k]["w2c"]).max() for k in range(a["F"]))
        daff = max(np.abs(W.frame_pose(k)[1] - np.array([b["frames"][k]["state"][6] * 10.0, b["frames"][k]["state"][7] * 1000.0])).max() for k in range(a["F"]))
        out.append(dict(F=a["F"], N=a["N"], R=a["R"], rmse=r["rmse"], rmse_ref=b["rmse"], dpose=dpose, daff=daff, prior=float(np.abs(a["HM"]).max()),
                        poses=np.array([W.frame_pose(k)[0] for k in range(a["F"])])))
    return out


def test_oracle_replays_the_committed_reference_run(oracle, synth):
    g = replay.load_golden(GOLDEN)
    w, h = g["w"], g["h"]
    K4, imgs, _ = replay.make_sequence(synth, w, h, g["n_frames"], g["step"])
    tr = replay_tracks_oracle(oracle, g["events"], imgs, w, h)
    assert len(tr) >= 50
    for dp, da, dr, ok in tr:
        assert ok and dp < 1e-15 and da == 0.0 and dr == 0.0      # the recorded pose went through one SE3 inversion: last-bit differences only
    ws = replay_windows(oracle.BAWindow, g["events"], imgs, w, h)
    fs = [x["F"] for x in ws]      # the live run is multi-threaded: a regenerated recording holds 7 or 8 optimisations (one more full window)
    assert fs[:7] == [2, 3, 4, 5, 6, 7, 8] and all(f == 8 for f in fs[7:]) and ws[-1]["prior"] > 1e8
    for x in ws:
        # the windows are re-created from recorded state; an ulp somewhere in that state is amplified by the 68x68 solve (condition ~1e10)
        assert abs(x["rmse"] - x["rmse_ref"]) < 2e-5 * x["rmse_ref"] and x["dpose"] < 2e-6 and x["daff"] < 1e-4, x


def test_oracle_replays_a_fresh_reference_run(oracle, synth):
    import ref_py as R
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not built and /root/reference absent")
    w, h = 320, 240
    run = replay.run_reference(R, synth, w, h, 66, step=1.6, point_density=600)
    st = run["status"][-1]
    assert st["initialized"] and not st["isLost"] and st["window"] >= 7
    del run["system"]
    tr = replay_tracks_oracle(oracle, run["events"], run["imgs"], w, h)
    assert len(tr) >= 40 and all(ok and dp < 1e-15 and da == 0.0 and dr == 0.0 for dp, da, dr, ok in tr)
    ws = replay_windows(oracle.BAWindow, run["events"], run["imgs"], w, h)
    assert max(x["F"] for x in ws) >= 7
    # the same windows through the reference's own optimize, re-created from the recorded state the same way
    wr = replay_windows(R.BAWindow, run["events"], run["imgs"], w, h)
    # (1) oracle == the reference's own code on the SAME re-created window (no live state involved)
    for xo, xr in zip(ws, wr):
        assert abs(xo["rmse"] - xr["rmse"]) <= 1e-6 * xr["rmse"] and np.abs(xo["poses"] - xr["poses"]).max() < 1e-9, (xo, xr)
    # (2) both against what the LIVE system produced.  The live run is multi-threaded (its sums change in the last bits from run to run) and the re-created window starts from
    # recorded state; the 68x68 solve (condition ~1e10) amplifies an ulp to the sixth digit, and once in a few dozen runs an accept / reject decision of one window flips
    # (seen: rmse apart by 4e-5 relative, affine by 0.015) — for the reference's own re-creation exactly as for the oracle.  So: every window close, all but at most one tight.
    def tight(x):
        return abs(x["rmse"] - x["rmse_ref"]) < 2e-5 * x["rmse_ref"] and x["dpose"] < 2e-6 and x["daff"] < 1e-4
    for x in ws + wr:
        assert abs(x["rmse"] - x["rmse_ref"]) < 1e-3 * x["rmse_ref"] and x["dpose"] < 1e-4, x
    assert sum(not tight(x) for x in ws) <= 1 and [tight(x) for x in ws] == [tight(x) for x in wr]


def test_reference_trajectory_follows_the_rendered_motion(synth):
    """Sanity of the recording itself: the reference's monocular trajectory on the synthetic sequence is the rendered camera path up to the
    scale of its initialisation."""
    g = replay.load_golden(GOLDEN)
    K4, imgs, poses = replay.make_sequence(synth, g["w"], g["h"], g["n_frames"], g["step"])
    tr = g["trajectory"]["camToWorld"]
    # rendered poses are worldToCam; compare translation directions of camToWorld relative to frame 0 over the last frame
    def c2w(p):
        Rm, t = synth.pose7_to_Rt(p); return -Rm.T @ t
    true_last = c2w(poses[-1]) - c2w(poses[0])
    est_last = tr[-1][:3] - tr[0][:3]
    cosang = float(true_last @ est_last / (np.linalg.norm(true_last) * np.linalg.norm(est_last)))
    assert cosang > 0.995
