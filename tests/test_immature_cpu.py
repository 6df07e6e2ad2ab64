"""CPU tests of the immature-point oracle (ImmaturePoint constructor + traceOn, oracle/immature_oracle.cpp): on the rendered
plane world the epipolar search must bracket the true inverse depth and shrink the interval frame after frame."""
import numpy as np


def _case(synth, oracle, w=320, h=256, n=400, seed=3):
    world = synth.PlaneWorld(synth.SEED + seed, fmax=22.0)
    K4 = synth.default_intrinsics(w, h)
    rng = np.random.RandomState(seed)
    host_img, host_id = world.render(K4, np.eye(3), np.zeros(3), w, h)
    u, v = synth.select_points(host_img, n, rng, min_grad=8.0)
    u = u.astype(np.int32); v = v.astype(np.int32)
    keep = (u >= 8) & (v >= 8) & (u < w - 8) & (v < h - 8)
    u, v = u[keep], v[keep]
    frames = []
    for k in range(1, 5):
        xi = np.array([0.04 * k, -0.015 * k, 0.01 * k, 0.002 * k, -0.003 * k, 0.001 * k])
        R, t = synth.se3_exp(xi)
        img, _ = world.render(K4, R, t, w, h)
        frames.append(dict(img=img, pose7=synth.pose7(R, t)))
    return dict(w=w, h=h, K4=K4, host_img=host_img, host_id=host_id, u=u, v=v, frames=frames)


def test_constructor_matches_numpy(oracle, synth):
    c = _case(synth, oracle)
    dI = oracle.make_images(c["host_img"], c["w"], c["h"])[0][0]
    P = oracle.ImmaturePoints(dI, c["w"], c["h"], c["u"], c["v"])
    I = c["host_img"].astype(np.float32)
    pat = synth.PATTERN8
    for k in (0, 7, len(c["u"]) - 1):
        for idx in range(8):
            x, y = c["u"][k] + pat[idx][0], c["v"][k] + pat[idx][1]
            assert P.color[k, idx] == I[y, x]
            gx, gy = I[y, x + 1] - I[y, x], I[y + 1, x] - I[y, x]      # forward differences of the bilinear cell at integer positions
            assert np.isclose(P.weights[k, idx], np.sqrt(2500.0 / (2500.0 + gx * gx + gy * gy)), rtol=1e-6)
    assert np.all(P.energyTH == np.float32(8 * 144))
    assert np.all(P.gradH[:, 1] == P.gradH[:, 2]) and np.all(P.gradH[:, 0] >= 0)


def test_trace_brackets_true_idepth_and_shrinks(oracle, synth):
    c = _case(synth, oracle)
    w, h = c["w"], c["h"]
    dI = oracle.make_images(c["host_img"], w, h)[0][0]
    P = oracle.ImmaturePoints(dI, w, h, c["u"], c["v"])
    true_id = c["host_id"][c["v"], c["u"]]
    widths = []
    for f in c["frames"]:
        dIn = oracle.make_images(f["img"], w, h)[0][0]
        KRKi, Kt, aff = oracle.trace_precalc(f["pose7"], np.array([0, 0, 0, 0, 0, 0, 1.0]), c["K4"])
        assert np.allclose(aff, [1, 0])
        st = P.trace_on(dIn, KRKi, Kt, aff).copy()
        good = st == 0
        assert good.sum() > 0.5 * len(st) or len(widths) > 0
        ok = good & np.isfinite(P.idepth_max)
        inside = (P.idepth_min[ok] <= true_id[ok] * 1.02) & (P.idepth_max[ok] >= true_id[ok] * 0.98)
        assert inside.mean() > 0.9, inside.mean()
        widths.append(np.median(P.idepth_max[ok] - P.idepth_min[ok]))
    assert widths[-1] < 0.5 * widths[0]
    # statuses stay within the enum, OOB points are never touched again
    assert set(np.unique(P.lastTraceStatus)) <= {0, 1, 2, 3, 4, 5}


def test_trace_precalc_matches_matrix_algebra(oracle, synth):
    rng = np.random.RandomState(2)
    new_w2c = oracle.se3_exp(0.1 * rng.standard_normal(6)); host_c2w = oracle.se3_exp(0.1 * rng.standard_normal(6))
    K4 = np.array([102.4, 103.0, 254.9, 250.1])
    KRKi, Kt, aff = oracle.trace_precalc(new_w2c, host_c2w, K4, 0.8, 1.2, (0.05, 3.0), (-0.02, 1.0))
    R1, t1 = oracle.se3_matrix(new_w2c); R2, t2 = oracle.se3_matrix(host_c2w)
    R, t = R1 @ R2, R1 @ t2 + t1
    K = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]])
    assert np.allclose(KRKi.reshape(3, 3), K @ R @ np.linalg.inv(K), rtol=1e-5, atol=1e-4)
    assert np.allclose(Kt, K @ t, rtol=1e-5, atol=1e-5)
    a = np.exp(0.05 + 0.02) * 0.8 / 1.2
    assert np.allclose(aff, [a, 3.0 - a * 1.0], rtol=1e-6)


def _window(synth, oracle, w=320, h=256, n=300, seed=6, F=5):
    """Host keyframe 0 + F-1 other keyframes on a smooth trajectory; immature points traced through all of them."""
    world = synth.PlaneWorld(synth.SEED + seed, fmax=22.0)
    K4 = synth.default_intrinsics(w, h)
    rng = np.random.RandomState(seed)
    imgs, w2c = [], []
    for k in range(F):
        xi = np.array([0.05 * k, -0.02 * k, 0.012 * k, 0.003 * k, -0.004 * k, 0.0015 * k])
        R, t = synth.se3_exp(xi)
        img, idm = world.render(K4, R, t, w, h, aff=(0.01 * k, 0.5 * k))
        imgs.append(img); w2c.append(synth.pose7(R, t))
        if k == 0:
            host_id = idm
    u, v = synth.select_points(imgs[0], n, rng, min_grad=8.0)
    u = u.astype(np.int32); v = v.astype(np.int32)
    keep = (u >= 8) & (v >= 8) & (u < w - 8) & (v < h - 8)
    return dict(w=w, h=h, K4=K4, imgs=imgs, w2c=w2c, u=u[keep], v=v[keep], host_id=host_id, F=F,
                aff=np.array([[0.01 * k, 0.5 * k] for k in range(F)]), exposure=np.ones(F, np.float32))


def _oracle_traced(oracle, c):
    w, h = c["w"], c["h"]
    dIs = [oracle.make_images(im, w, h)[0][0] for im in c["imgs"]]
    P = oracle.ImmaturePoints(dIs[0], w, h, c["u"], c["v"])
    c2w0 = oracle.se3_inv(c["w2c"][0])
    for k in range(1, c["F"]):
        KRKi, Kt, aff = oracle.trace_precalc(c["w2c"][k], c2w0, c["K4"], 1.0, 1.0, tuple(c["aff"][k]), tuple(c["aff"][0]))
        P.trace_on(dIs[k], KRKi, Kt, aff)
    return P, dIs, c2w0


def test_optimize_immature_point_recovers_idepth(oracle, synth):
    c = _window(synth, oracle)
    P, dIs, c2w0 = _oracle_traced(oracle, c)
    pre = [oracle.pair_precalc(c["w2c"][k], c2w0, 1.0, 1.0, tuple(c["aff"][0]), tuple(c["aff"][k])) for k in range(1, c["F"])]
    R = np.stack([p[0] for p in pre]); t = np.stack([p[1] for p in pre]); aff = np.stack([p[2] for p in pre])
    usable = np.isfinite(P.idepth_max) & (P.lastTraceStatus != 1)
    P.idepth_max[~usable] = 0.5; P.idepth_min[~usable] = 0.1
    result, idepth, res_state = oracle.immature_optimize(P, c["K4"], dIs[1:], R, t, aff, min_obs=1)
    act = (result == 1) & usable
    assert act.sum() > 0.6 * usable.sum()
    true_id = c["host_id"][c["v"], c["u"]]
    rel = np.abs(idepth[act] - true_id[act]) / true_id[act]
    # the window's baseline is short (<= 3 px of parallax): 0.06 px of photometric precision is ~2 % of inverse depth
    assert np.median(rel) < 0.04 and np.mean(rel < 0.15) > 0.9
    mid = 0.5 * (P.idepth_min + P.idepth_max)
    assert np.median(rel) <= np.median(np.abs(mid[act] - true_id[act]) / true_id[act]) * 1.05
    assert np.all((res_state[act] == 0).sum(axis=1) >= 1)
    assert set(np.unique(result)) <= {-1, 0, 1}
