"""End-to-end parity on a synthetic sequence (north_star: trajectory within 1e-3 m of the CPU path): the same keyframe-based
visual-odometry loop (tests/vo_harness.py: trackNewestCoarse, traceNewCoarse, optimizeImmaturePoint, optimize, marginalizePointsF,
marginalizeFrame) driven through the HIP library and through the CPU oracle on identical images."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sequence_trajectory_matches_oracle(pkg, oracle, synth, gpu_required):
    import vo_harness as vh
    w, h, n = 512, 384, 16
    K4, imgs, id0, c2w_true = vh.make_sequence(synth, w, h, n)
    kw = dict(kf_every=3, max_kf=4, n_new=400)
    vo_o = vh.run(vh.OracleBackend(oracle, w, h, K4), synth, K4, imgs, id0, w, h, **kw)
    vo_g = vh.run(vh.HipBackend(pkg, w, h, K4), synth, K4, imgs, id0, w, h, **kw)
    path = np.linalg.norm(c2w_true[-1][:3] - c2w_true[0][:3])
    d = np.array([np.linalg.norm(vo_g.traj[k][:3] - vo_o.traj[k][:3]) for k in range(n)])
    e_g = np.array([np.linalg.norm(vo_g.traj[k][:3] - c2w_true[k][:3]) for k in range(n)])
    e_o = np.array([np.linalg.norm(vo_o.traj[k][:3] - c2w_true[k][:3]) for k in range(n)])
    assert np.sqrt(np.mean(d ** 2)) < 1e-3 and d.max() < 2e-3, d            # trajectory RMSE GPU vs CPU path
    assert e_g.max() < 0.03 * path and e_o.max() < 0.03 * path
    assert len(vo_g.log) == len(vo_o.log)
    for lg, lo in zip(vo_g.log, vo_o.log):
        # the tracked poses agree to ~1e-6 only (fp32 summation order), so a few borderline candidates fall on the other side of a threshold
        assert abs(lg["candidates"] - lo["candidates"]) <= 0.05 * lo["candidates"] + 2
        assert abs(lg["activated"] - lo["activated"]) <= 0.05 * lo["activated"] + 2
        assert abs(lg["energy"] - lo["energy"]) <= 0.15 * lo["energy"]      # different candidate sets -> different residual counts
    assert vo_g.prior is not None and vo_g.prior[0].shape == vo_o.prior[0].shape
    print("trajectory difference GPU vs oracle [m]: rmse %.2e max %.2e; error vs ground truth: gpu %.4f oracle %.4f of a %.3f m path" % (np.sqrt(np.mean(d ** 2)), d.max(), e_g.max(), e_o.max(), path))


def test_long_vga_sequence_matches_oracle(pkg, oracle, synth, gpu_required):
    """120 frames at 640x480 on a closed 3 m path (30 keyframes, sliding window of 6, marginalisation of 24 of them): the HIP path stays on the CPU
    path's trajectory to a millimetre, with the reference's single-threaded summation order (what the oracle replays) and with the library's default
    accumulation order; both stay within 1 % of the path length of the ground truth."""
    import vo_harness as vh
    w, h, n = 640, 480, 120
    K4, imgs, id0, c2w_true = vh.make_sequence(synth, w, h, n, motion="orbit", device="cuda")
    kw = dict(kf_every=4, max_kf=6, n_new=500)
    vo_o = vh.run(vh.OracleBackend(oracle, w, h, K4), synth, K4, imgs, id0, w, h, **kw)
    path = sum(np.linalg.norm(c2w_true[k + 1][:3] - c2w_true[k][:3]) for k in range(n - 1))
    e_o = np.array([np.linalg.norm(vo_o.traj[k][:3] - c2w_true[k][:3]) for k in range(n)])
    assert path > 2.5 and e_o.max() < 0.01 * path
    for accumulators, rmse_tol, max_tol in ((1, 1.5e-3, 3e-3), (4, 2e-3, 4e-3)):
        vo_g = vh.run(vh.HipBackend(pkg, w, h, K4, accumulators=accumulators), synth, K4, imgs, id0, w, h, **kw)
        d = np.array([np.linalg.norm(vo_g.traj[k][:3] - vo_o.traj[k][:3]) for k in range(n)])
        e_g = np.array([np.linalg.norm(vo_g.traj[k][:3] - c2w_true[k][:3]) for k in range(n)])
        assert np.sqrt(np.mean(d ** 2)) < rmse_tol and d.max() < max_tol, (accumulators, np.sqrt(np.mean(d ** 2)), d.max())
        assert e_g.max() < 0.01 * path
