"""Child process of tests/test_leaf_sensitivity_cpu.py: one tracking problem, one bundle adjustment and one short visual-odometry sequence through the CPU oracle —
whichever build DMVIO_ORACLE_VARIANT selects — results into an .npz.  Test infrastructure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402


def main(out):
    graft.load_package()
    import dmvio_amd.synth as synth
    O = graft.load_oracle()
    import vo_harness as vh
    res = {}
    # trackNewestCoarse, 4 levels, 256x256
    w = h = 256
    tc = synth.tracking_case(w, h, n_ref=800, n_frames=2, xi_jitter=0.3)
    T = O.Tracker(w, h); T.make_k(tc["K4"])
    T.set_ref(O.make_images(tc["ref_img"], w, h)[0], tc["u"], tc["v"], tc["idepth"], tc["hdiF"])
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    for k, f in enumerate(tc["frames"]):
        T.set_new(O.make_images(f["img"], w, h)[0])
        r = T.track(ident, [0.0, 0.0])
        res["track%d_pose" % k] = np.asarray(r["pose7"]); res["track%d_res" % k] = np.asarray(r["lastResiduals"]); res["track%d_aff" % k] = np.asarray(r["aff"])
    # the 31 motion hypotheses of trackNewCoarse: products, inverses, exp / log of poses
    rng = np.random.RandomState(5)
    a, b, c = [O.se3_exp(0.1 * rng.standard_normal(6)) for _ in range(3)]
    res["hypotheses"] = np.asarray(O.make_track_hypotheses(O.se3_mul(a, b), a, O.se3_mul(O.se3_mul(a, b), c)))
    # FullSystem::optimize(6) of a 4-keyframe window
    case = synth.ba_case(w=320, h=256, n_frames=4, n_points=240, hosts_share=(100, 80, 60, 0), seed=777)
    W = O.BAWindow(case)
    r = W.optimize(6)
    res["ba_energy"] = np.array([r["finalEnergy"], r["rmse"]]); res["ba_iterations"] = np.array([r["iterations"]])
    res["ba_poses"] = np.array([W.frame_pose(k)[0] for k in range(4)])
    # a 14-frame sequence: tracking + tracing + activation + BA + marginalisation chained
    K4, imgs, id0, c2w_true = vh.make_sequence(synth, 256, 192, 14)
    vo = vh.run(vh.OracleBackend(O, 256, 192, K4), synth, K4, imgs, id0, 256, 192, kf_every=3, max_kf=3, n_new=250)
    res["traj"] = np.array([np.asarray(vo.traj[k]) for k in sorted(vo.traj)])
    # the windows the reference's own FullSystem optimised in the recorded live run (2 ... 8 keyframes, marginalisation priors ~1e8..1e10: the worst-conditioned solves at hand)
    import replay
    import test_ref_replay_cpu as rr
    g = replay.load_golden(rr.GOLDEN)
    _, gi, _ = replay.make_sequence(synth, g["w"], g["h"], g["n_frames"], g["step"])
    ws = rr.replay_windows(O.BAWindow, g["events"], gi, g["w"], g["h"])
    res["recorded_rmse"] = np.array([x["rmse"] for x in ws])
    res["recorded_poses"] = np.concatenate([x["poses"].ravel() for x in ws])
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
