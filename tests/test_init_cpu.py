"""CPU tests of the initializer oracle (CoarseInitializer::calcResAndGS, oracle/init_oracle.cpp)."""
import numpy as np


def init_case(synth, oracle, w=320, h=256, lvl=1, n=800, seed=12):
    """First frame at identity, new frame slightly moved; initializer-style points (u = x + 0.1, idepth around the true value rescaled
    to mean 1 like the initializer's convention is NOT needed for the algebra test: true inverse depths are used)."""
    world = synth.PlaneWorld(synth.SEED + seed, fmax=14.0)
    K4 = synth.default_intrinsics(w, h)
    rng = np.random.RandomState(seed)
    img0, id0 = world.render(K4, np.eye(3), np.zeros(3), w, h)
    xi = np.array([0.05, -0.02, 0.01, 0.004, -0.006, 0.002])
    R, t = synth.se3_exp(xi)
    img1, _ = world.render(K4, R, t, w, h, aff=(0.02, 1.5))
    wl, hl = w >> lvl, h >> lvl
    s = 2.0 ** lvl
    fx, fy = K4[0] / s, K4[1] / s
    cx, cy = (K4[2] + 0.5) / s - 0.5, (K4[3] + 0.5) / s - 0.5
    x = rng.randint(4, wl - 5, n); y = rng.randint(4, hl - 5, n)
    u = x + 0.1; v = y + 0.1
    true_id = id0[np.minimum((y * s).astype(int), h - 1), np.minimum((x * s).astype(int), w - 1)]
    idepth_new = (true_id * (1 + 0.1 * rng.standard_normal(n))).astype(np.float32)
    good = np.ones(n, np.uint8); good[::13] = 0
    energy = np.stack([rng.uniform(0, 50, n), rng.uniform(0, 1, n)], axis=1).astype(np.float32)
    pts = dict(u=u.astype(np.float32), v=v.astype(np.float32), iR=np.ones(n, np.float32), isGood=good, energy=energy, outlierTH=np.full(n, 8 * 144.0, np.float32))
    Ki = np.linalg.inv(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]))
    return dict(w=w, h=h, lvl=lvl, wl=wl, hl=hl, img0=img0, img1=img1, pts=pts, idepth_new=idepth_new, Ki=Ki, K_lvl=np.array([fx, fy, cx, cy], np.float32),
                pose7=synth.pose7(R, t), aff=(0.02, 1.5), true_id=true_id)


def test_calc_res_and_gs_matches_numpy(oracle, synth):
    c = init_case(synth, oracle)
    dI0 = oracle.make_images(c["img0"], c["w"], c["h"])[0]; dI1 = oracle.make_images(c["img1"], c["w"], c["h"])[0]
    lvl = c["lvl"]
    o = oracle.init_calc_res_and_gs(dI0[lvl], dI1[lvl], c["wl"], c["hl"], c["Ki"], c["K_lvl"], c["pose7"], c["aff"], c["pts"], c["idepth_new"])
    n = len(c["idepth_new"])
    good_in = c["pts"]["isGood"].astype(bool)
    assert not o["isGood_new"][~good_in].any() and o["isGood_new"][good_in].mean() > 0.8
    assert o["res3"][2] == 2 * n
    acc = o["isGood_new"].astype(bool)
    # energy = sum of accepted energies + old energies of the others (fp32 sum vs float64)
    E = o["energy_new"][acc, 0].astype(np.float64).sum() + c["pts"]["energy"][~acc, 0].astype(np.float64).sum()
    assert abs(o["res3"][0] - E) <= 1e-5 * E
    # Schur system from JbBuffer: Hsc = sum_i w_i jb_i jb_i^T with w = JbBuffer[9] (already 1 / (1 + Hdd + alpha / coupling))
    Jb = o["JbBuffer_new"][acc].astype(np.float64)
    Hsc = np.einsum("i,ij,ik->jk", Jb[:, 9], Jb[:, :9], Jb[:, :9])
    assert np.allclose(o["Hsc"], Hsc[:8, :8], rtol=2e-4, atol=1e-3 * np.abs(Hsc).max() * 1e-3)
    assert np.allclose(o["bsc"], Hsc[:8, 8], rtol=2e-4, atol=1e-6 * np.abs(Hsc).max())
    # H symmetric, positive diagonal; lastHessian_new = dd^T dd >= 0; maxstep finite for good points
    assert np.allclose(o["H"], o["H"].T) and np.all(np.diag(o["H"]) > 0)
    assert np.all(o["lastHessian_new"][acc] >= 0) and np.all(np.isfinite(o["maxstep"][acc])) and np.all(o["maxstep"][~good_in] == np.float32(1e10))
    # alpha energy: alphaW * |t|^2 * n capped by alphaK * n
    t = np.asarray(c["pose7"][:3], dtype=np.float64)
    assert np.isclose(o["res3"][1], min(150 * 150 * (t @ t) * n, 2.5 * 2.5 * n), rtol=1e-6)


def test_gauss_newton_direction_reduces_energy(oracle, synth):
    """The 8-dof step of the reduced system (H - Hsc) x = -(b - bsc) must reduce the photometric energy (what trackFrame does with it)."""
    c = init_case(synth, oracle, seed=13)
    dI0 = oracle.make_images(c["img0"], c["w"], c["h"])[0]; dI1 = oracle.make_images(c["img1"], c["w"], c["h"])[0]
    lvl = c["lvl"]
    idn = c["true_id"].astype(np.float32)
    start = oracle.se3_exp(np.array([0.04, -0.015, 0.008, 0.003, -0.005, 0.0015]))
    kw = dict(alphaW=0.0, alphaK=1e9, couplingWeight=0.0)
    o0 = oracle.init_calc_res_and_gs(dI0[lvl], dI1[lvl], c["wl"], c["hl"], c["Ki"], c["K_lvl"], start, c["aff"], c["pts"], idn, **kw)
    H = o0["H"].astype(np.float64); b = o0["b"].astype(np.float64)        # pose-only step with the depths held fixed (alphaOpt = coupling = 0)
    x = -np.linalg.solve(H + 1e-3 * np.diag(np.diag(H)), b)
    new_pose = oracle.se3_mul(oracle.se3_exp(x[:6]), start)
    o1 = oracle.init_calc_res_and_gs(dI0[lvl], dI1[lvl], c["wl"], c["hl"], c["Ki"], c["K_lvl"], new_pose, (c["aff"][0] + x[6], c["aff"][1] + x[7]), c["pts"], idn, **kw)
    assert o1["res3"][0] < o0["res3"][0]
