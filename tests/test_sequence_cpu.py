"""End-to-end check of the CPU oracle on a short synthetic sequence: tracking + tracing + activation + sliding-window BA +
marginalisation, chained by tests/vo_harness.py, must follow the rendered ground-truth trajectory."""
import numpy as np


def test_oracle_pipeline_follows_ground_truth(oracle, synth):
    import vo_harness as vh
    w, h, n = 256, 192, 14
    K4, imgs, id0, c2w_true = vh.make_sequence(synth, w, h, n)
    vo = vh.run(vh.OracleBackend(oracle, w, h, K4), synth, K4, imgs, id0, w, h, kf_every=3, max_kf=3, n_new=250)
    err = np.array([np.linalg.norm(vo.traj[k][:3] - c2w_true[k][:3]) for k in range(n)])
    path = np.linalg.norm(c2w_true[-1][:3] - c2w_true[0][:3])
    assert err.max() < 0.03 * path, (err, path)          # first-keyframe depths carry 3 % noise: the scale is only that well known
    assert len(vo.kfs) <= 3 and vo.prior is not None     # marginalisation happened
    assert sum(l.get("activated", 0) for l in vo.log) > 50
