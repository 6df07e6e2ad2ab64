"""CPU tests (no GPU): validate the ORACLE by construction.

The reference holds no golden vectors / KATs for the photometric-alignment path.  The pin against the reference's own
compiled sources lives in tests/test_ref_pin_cpu.py / test_ref_replay_cpu.py; here the oracle (oracle/*.cpp, an fp32/SSE restatement of the reference) is
checked by construction against
  * an independent float64 NumPy restatement (tests/np_ref.py) and committed fixtures made from it
    (tests/golden/, generator tests/golden/make_golden.py),
  * analytic identities (exp/log round trips, Adj, power-series matrix exponential),
  * finite differences of the residual w.r.t. the left-multiplied se3 increment and the affine parameters,
  * known answers (identical frame -> zero residual; convergence to the rendered ground-truth pose).
"""
import os

import numpy as np
import pytest

import np_ref

IDENT = np.array([0, 0, 0, 0, 0, 0, 1.0])
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ----------------------------------------------------------------------------- Lie group / dense algebra
def test_se3_exp_matches_matrix_power_series(oracle):
    rng = np.random.RandomState(0)
    for scale in (1e-12, 1e-6, 1e-2, 0.5, 2.0):
        for _ in range(5):
            xi = rng.normal(0, scale, 6)
            R, t = oracle.se3_matrix(oracle.se3_exp(xi))
            M = np_ref.se3_exp_matrix(xi)
            assert np.allclose(R, M[:3, :3], atol=1e-12) and np.allclose(t, M[:3, 3], atol=1e-12)


def test_se3_log_exp_roundtrip_and_group_laws(oracle):
    rng = np.random.RandomState(1)
    for _ in range(20):
        xi = rng.normal(0, 0.7, 6)
        T = oracle.se3_exp(xi)
        assert np.allclose(oracle.se3_log(T), xi, atol=1e-10)
        Ti = oracle.se3_inv(T)
        e = oracle.se3_mul(T, Ti)
        assert np.allclose(e[:3], 0, atol=1e-12) and abs(abs(e[6]) - 1) < 1e-12
        # Adj: T exp(x) T^-1 = exp(Adj_T x)
        x = rng.normal(0, 0.1, 6)
        lhs = oracle.se3_mul(oracle.se3_mul(T, oracle.se3_exp(x)), Ti)
        rhs = oracle.se3_exp(oracle.se3_adj(T) @ x)
        Rl, tl = oracle.se3_matrix(lhs); Rr, tr = oracle.se3_matrix(rhs)
        assert np.allclose(Rl, Rr, atol=1e-10) and np.allclose(tl, tr, atol=1e-10)


def test_sophus_test_se3_vectors(oracle):
    """Tangent vectors of the vendored thirdparty/Sophus/sophus/test_se3.cpp (exp/log consistency set)."""
    tangents = [np.array(v, dtype=float) for v in (
        [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0], [-1, 1, 0, 0, 0, 1],
        [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0])]
    for a in tangents:
        T = oracle.se3_exp(a)
        a2 = oracle.se3_log(T)
        R1, t1 = oracle.se3_matrix(T); R2, t2 = oracle.se3_matrix(oracle.se3_exp(a2))
        assert np.allclose(R1, R2, atol=1e-9) and np.allclose(t1, t2, atol=1e-8)
        M = np_ref.se3_exp_matrix(a) if np.linalg.norm(a[3:]) < 3 else None
        if M is not None:
            assert np.allclose(R1, M[:3, :3], atol=1e-9)


def test_ldlt_solve_matches_numpy(oracle):
    rng = np.random.RandomState(2)
    for n in (6, 7, 8, 20, 68):
        A = rng.normal(size=(n, n)); A = A @ A.T + 1e-3 * np.eye(n)
        A *= np.outer(10.0 ** rng.uniform(-3, 3, n), np.ones(n)); A = 0.5 * (A + A.T) + np.diag(np.abs(A).sum(1))
        b = rng.normal(size=n)
        x = oracle.ldlt_solve(A, b)
        assert np.allclose(A @ x, b, rtol=1e-9, atol=1e-9 * np.abs(b).max())
    assert np.all(oracle.ldlt_solve(np.zeros((4, 4)), np.ones(4)) == 0)  # Eigen: zero matrix -> zero solution


# ----------------------------------------------------------------------------- pyramid
def test_make_images_matches_numpy(oracle, synth):
    case = synth.tracking_case(320, 256, n_ref=50)
    assert oracle.pyr_levels(320, 256) == 4 and oracle.pyr_levels(256, 192) == 3 and oracle.pyr_levels(512, 512) == 4 and oracle.pyr_levels(640, 480) == 4
    assert oracle.pyr_levels(800, 400) == 4 and oracle.pyr_levels(1280, 1024) == 6 and oracle.pyr_levels(63, 64) == 1
    dI, ab = oracle.make_images(case["ref_img"], 320, 256)
    ref = np_ref.make_images(case["ref_img"])
    assert len(dI) == len(ref) == 4
    for a, b in zip(dI, ref):
        assert np.array_equal(a, b)
    assert np.allclose(ab[0][1:-1], dI[0][1:-1, :, 1] ** 2 + dI[0][1:-1, :, 2] ** 2)


# ----------------------------------------------------------------------------- tracker template
@pytest.fixture(scope="module")
def small(oracle, synth):
    w, h = 320, 256
    case = synth.tracking_case(w, h, n_ref=600, xi_true=(0.02, -0.012, 0.025, 0.006, -0.009, 0.005))
    T = oracle.Tracker(w, h)
    T.make_k(case["K4"])
    dIr, _ = oracle.make_images(case["ref_img"], w, h)
    dIn, _ = oracle.make_images(case["frames"][0]["img"], w, h)
    T.set_ref(dIr, case["u"], case["v"], case["idepth"], case["hdiF"])
    T.set_new(dIn)
    return dict(case=case, T=T, dIr=dIr, dIn=dIn, w=w, h=h)


def test_make_k(small):
    T, K4 = small["T"], small["case"]["K4"]
    for lvl in range(4):
        k, ki = T.get_k(lvl)
        assert np.allclose(k, np_ref.level_K(K4, lvl), rtol=1e-6)
        K = np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1]], dtype=np.float64)
        assert np.allclose(ki @ K, np.eye(3), atol=1e-5)


def test_coarse_depth_template_properties(small):
    """makeCoarseDepthL0: every selected point lands in the template with its own idepth; dilation adds its
    diagonal neighbours at level 0; template idepths stay inside the range of the inputs; row-major order."""
    T, case = small["T"], small["case"]
    u, v, idp, col = T.get_pc(0)
    n0 = T.pc_n(0)
    assert len(case["u"]) < n0 <= 5 * len(case["u"])
    key = (v * small["w"] + u).astype(int)
    assert np.all(np.diff(key) > 0), "template points must be in row-major order"
    lut = dict(zip(key.tolist(), idp.tolist()))
    hits = 0
    for x, y, d in zip(case["u"].astype(int), case["v"].astype(int), case["idepth"]):
        k = y * small["w"] + x
        if 2 <= x < small["w"] - 2 and 2 <= y < small["h"] - 2:
            assert k in lut
            if abs(lut[k] - d) < 1e-6 * d:
                hits += 1
    assert hits > 0.9 * len(case["u"])  # collisions of two points in one pixel average their idepths
    assert idp.min() >= case["idepth"].min() * (1 - 1e-5) and idp.max() <= case["idepth"].max() * (1 + 1e-5)
    assert np.array_equal(col, small["dIr"][0][v.astype(int), u.astype(int), 0])
    for lvl in range(1, 4):
        ul, vl, il, cl = T.get_pc(lvl)
        assert T.pc_n(lvl) > 0 and il.min() > 0
        assert ul.min() >= 2 and ul.max() < (small["w"] >> lvl) - 2


# ----------------------------------------------------------------------------- residuals / Jacobians
def _np_eval(small, lvl, pose, aff, cutoff=20.0):
    return np_ref.calc_res_gs(small["case"]["K4"], lvl, small["T"].get_pc(lvl), small["dIn"][lvl], pose, aff, cutoff)


@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_calc_res_gs_matches_float64(small, lvl):
    T, case = small["T"], small["case"]
    # not the exact identity: there Ku == x sits exactly on the `Ku > 2` bound and the count depends on fp32 vs fp64 rounding
    near_ident = np.array([1e-3, 2e-3, -1e-3, 5e-4, 2.5e-4, -3.5e-4, 1.0]); near_ident[3:] /= np.linalg.norm(near_ident[3:])
    for pose, aff in ((near_ident, (0.0, 0.0)), (case["frames"][0]["pose7"], (0.0, 0.0)), (case["frames"][0]["pose7"], (0.03, -2.0))):
        rs = T.calc_res(lvl, pose, aff, 20.0)
        H, b = T.calc_gs(lvl, aff)
        ref = _np_eval(small, lvl, pose, aff)
        assert rs[1] == ref["n"] and round(rs[5] * rs[1]) == ref["nsat"]
        assert abs(rs[0] - ref["E"]) < 2e-5 * ref["E"] + 1e-3
        scale = np.sqrt(np.outer(np.diag(ref["H"]), np.diag(ref["H"])))
        assert np.max(np.abs(H - ref["H"]) / scale) < 5e-5
        rmse = np.sqrt(ref["E"] / max(ref["n"], 1)) + 1e-3
        assert np.max(np.abs(b - ref["b"]) / (np.sqrt(np.diag(ref["H"])) * rmse)) < 1e-4  # b is a cancelling sum: scale by |J||r|
        assert np.allclose(H, H.T)
        assert np.all(np.linalg.eigvalsh(H) > -1e-7 * np.abs(H).max())


def test_warped_buffers_match_float64(small):
    """buf_warped_* (the calcRes -> calcGSSSE hand-off of the reference) vs the float64 per-point values."""
    T, case = small["T"], small["case"]
    pose = case["frames"][0]["pose7"]
    T.calc_res(0, pose, (0.0, 0.0), 20.0)
    wb = T.get_warped()  # idepth,u,v,dx,dy,residual,weight,refColor
    ref = _np_eval(small, 0, pose, (0.0, 0.0))
    n = len(ref["r"])
    assert wb.shape[1] == (n + 3) // 4 * 4
    # fp32 projection error (~3e-5 px) times image gradients of up to ~100 grey/px
    assert np.max(np.abs(wb[5, :n] - ref["r"])) < 2e-2 and np.mean(np.abs(wb[5, :n] - ref["r"])) < 5e-4
    assert np.allclose(wb[6, :n], ref["w"], atol=3e-3)
    assert np.all(wb[:, n:] == 0), "padding entries are zero (CoarseTracker.cpp:486-498)"


def test_jacobian_by_finite_differences(oracle, synth):
    """J of calcGSSSE = d r / d(left se3 increment, a, b) up to the reference's scaling: b = J^T W r / n must equal the
    finite-difference gradient of the Huber energy.  Uses a LOW-frequency texture: the reference pairs a bilinear
    intensity interpolant with interpolated central-difference gradients, which are only consistent for smooth images."""
    w, h = 320, 256
    case = synth.tracking_case(w, h, n_ref=500, fmax=2.5, min_grad=1.0, xi_true=(0.02, -0.012, 0.025, 0.006, -0.009, 0.005))
    T = oracle.Tracker(w, h); T.make_k(case["K4"])
    T.set_ref(oracle.make_images(case["ref_img"], w, h)[0], case["u"], case["v"], case["idepth"], case["hdiF"])
    dIn = oracle.make_images(case["frames"][0]["img"], w, h)[0]
    T.set_new(dIn)
    lvl = 0
    pose = oracle.se3_mul(oracle.se3_exp(np.array([0.004, -0.003, 0.002, 0.0015, -0.001, 0.0012])), case["frames"][0]["pose7"])
    aff = (0.01, 0.5)
    pc = T.get_pc(lvl)
    base = np_ref.calc_res_gs(case["K4"], lvl, pc, dIn[lvl], pose, aff, cutoff=1e9)
    n4 = (len(base["r"]) + 3) // 4 * 4
    sc = np.array([1, 1, 1, 1, 1, 1, 10.0, 1000.0])
    g_analytic = 2 * n4 * base["b"] / sc  # gradient of the Huber energy w.r.t. the unscaled increment
    # the oracle's own b agrees with the float64 one
    T.calc_res(lvl, pose, aff, 1e9)
    _, b_o = T.calc_gs(lvl, aff)
    assert np.allclose(2 * n4 * b_o / sc, g_analytic, rtol=2e-3, atol=1e-4 * np.abs(g_analytic).max())

    def energy(dx):
        p = oracle.se3_mul(oracle.se3_exp(dx[:6]), pose)
        r = np_ref.calc_res_gs(case["K4"], lvl, pc, dIn[lvl], p, (aff[0] + dx[6], aff[1] + dx[7]), cutoff=1e9)
        return r["E"], r["n"]

    for k, eps in enumerate([2e-4] * 3 + [1e-4] * 3 + [1e-4, 1e-2]):
        d = np.zeros(8); d[k] = eps
        (ep, n_p), (em, n_m) = energy(d), energy(-d)
        assert n_p == n_m == base["n"]
        g_fd = (ep - em) / (2 * eps)
        assert abs(g_fd - g_analytic[k]) <= 0.05 * abs(g_analytic[k]) + 5e-3 * np.linalg.norm(g_analytic[:6] if k < 6 else g_analytic), (k, g_fd, g_analytic[k])


def test_identity_frame_zero_residual(small):
    T = small["T"]
    T.set_new(small["dIr"])
    rs = T.calc_res(0, IDENT, (0.0, 0.0), 20.0)
    H, b = T.calc_gs(0, (0.0, 0.0))
    assert rs[0] / rs[1] < 1e-5 and rs[5] == 0
    assert np.max(np.abs(b) / np.sqrt(np.diag(H))) < 1e-2
    T.set_new(small["dIn"])


def test_sse_accumulator_vs_float64(small):
    """Accumulator9 (4 SSE lanes + 1k/1M shift-up) against a plain float64 sum of the same fp32 per-point values."""
    T, case = small["T"], small["case"]
    T.calc_res(0, case["frames"][0]["pose7"], (0.0, 0.0), 20.0)
    wb = T.get_warped().astype(np.float64)
    H, b = T.calc_gs(0, (0.0, 0.0))
    k, _ = T.get_k(0)
    idd, u, v, dx, dy, r, w, col = wb
    dx = dx * k[0]; dy = dy * k[1]
    J = np.stack([idd * dx, idd * dy, -idd * (u * dx + v * dy), -(u * v * dx + dy * (1 + v * v)), u * v * dy + dx * (1 + u * u),
                  u * dy - v * dx, 1.0 * (0.0 - col), -np.ones_like(u)], 1)
    sc = np.array([1, 1, 1, 1, 1, 1, 10.0, 1000.0])
    Href = (J * w[:, None]).T @ J / len(r) * sc[:, None] * sc[None, :]
    bref = (J * w[:, None]).T @ r / len(r) * sc
    assert np.max(np.abs(H - Href) / np.sqrt(np.outer(np.diag(Href), np.diag(Href)))) < 2e-6
    assert np.allclose(b, bref, rtol=2e-4, atol=1e-6 * np.abs(bref).max())


# ----------------------------------------------------------------------------- LM loop
def test_track_converges_to_ground_truth(small):
    T, case = small["T"], small["case"]
    r = T.track(IDENT, (0.0, 0.0))
    assert r["good"]
    assert np.linalg.norm(r["pose7"][:3] - case["frames"][0]["pose7"][:3]) < 2e-3
    assert np.linalg.norm(r["pose7"][3:] - case["frames"][0]["pose7"][3:]) < 1e-3
    assert np.all(np.isfinite(r["lastResiduals"][:4])) and np.isnan(r["lastResiduals"][4])
    n_res, n_gs, _ = T.stats()[0], T.stats()[1], T.stats()[2]
    assert n_gs <= n_res


def test_track_failure_semantics(small):
    T = small["T"]
    mr = np.array([0.05, 0.05, 0.05, 0.05, np.nan])
    r = T.track(IDENT, (0.0, 0.0), min_res=mr)
    assert r["good"] is False and np.array_equal(r["pose7"], IDENT)


def test_affine_modes(small):
    """setting_affineOptModeA/B < 0 fixes a and/or b: the returned aff is zeroed (CoarseTracker.cpp:759-760)."""
    T = small["T"]
    r = T.track(IDENT, (0.0, 0.0), modeA=-1.0, modeB=-1.0)
    assert r["good"] and r["aff"][0] == 0 and r["aff"][1] == 0
    r = T.track(IDENT, (0.0, 0.0), modeA=-1.0, modeB=1e8)
    assert r["aff"][0] == 0 and r["aff"][1] != 0
    r = T.track(IDENT, (0.0, 0.0), modeA=1e12, modeB=-1.0)
    assert r["aff"][0] != 0 and r["aff"][1] == 0


# ----------------------------------------------------------------------------- committed fixtures
def test_golden_fixture(oracle):
    """tests/golden/tracker_small.npz: inputs + float64 expectations produced by tests/golden/make_golden.py."""
    g = np.load(os.path.join(GOLD, "tracker_small.npz"))
    w, h = int(g["w"]), int(g["h"])
    T = oracle.Tracker(w, h)
    T.make_k(g["K4"])
    dIr, _ = oracle.make_images(g["ref_img"], w, h)
    dIn, _ = oracle.make_images(g["new_img"], w, h)
    T.set_ref(dIr, g["u"], g["v"], g["idepth"], g["hdiF"])
    T.set_new(dIn)
    assert [T.pc_n(l) for l in range(T.levels)] == g["pc_n"].tolist()
    for lvl in range(T.levels):
        rs = T.calc_res(lvl, g["pose7"], g["aff"], 20.0)
        H, b = T.calc_gs(lvl, g["aff"])
        assert rs[1] == g["n"][lvl]
        assert abs(rs[0] - g["E"][lvl]) < 2e-5 * g["E"][lvl]
        Hg = g["H"][lvl]
        assert np.max(np.abs(H - Hg) / np.sqrt(np.outer(np.diag(Hg), np.diag(Hg)))) < 5e-5
    r = T.track(IDENT, (0.0, 0.0))
    assert np.linalg.norm(r["pose7"][:3] - g["pose7"][:3]) < 2e-3


def _T(p7):
    import numpy as np
    x, y, z, w = p7[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = p7[:3]
    return M


def test_track_hypotheses_match_matrix_algebra(oracle, pkg):
    """lastF_2_fh_tries (FullSystem.cpp:364-402): oracle list vs independent 4x4 matrix algebra; the library's host-side
    generator (no device work) must give the same list."""
    import numpy as np
    from scipy.linalg import expm, logm
    rng = np.random.RandomState(5)
    sprelast = oracle.se3_exp(0.1 * rng.standard_normal(6))
    slast = oracle.se3_mul(sprelast, oracle.se3_exp(0.05 * rng.standard_normal(6)))
    lastF = oracle.se3_exp(0.1 * rng.standard_normal(6))
    tries = oracle.make_track_hypotheses(slast, sprelast, lastF)
    assert tries.shape == (31, 7)
    A = np.linalg.inv(_T(sprelast)) @ _T(slast)            # fh_2_slast
    Bm = np.linalg.inv(_T(slast)) @ _T(lastF)              # lastF_2_slast
    Ai = np.linalg.inv(A)
    expect = [Ai @ Bm, Ai @ Ai @ Bm, np.linalg.inv(expm(0.5 * np.real(logm(A)))) @ Bm, Bm, np.eye(4)]
    for k, E in enumerate(expect):
        assert np.allclose(_T(tries[k]), E, atol=1e-9), k
    # 26 rotation tries: base * small rotation with unit quaternion ~ (1, +-d, +-d, +-d), all distinct, no translation change
    base = Ai @ Bm
    seen = set()
    for k in range(5, 31):
        D = np.linalg.inv(base) @ _T(tries[k])
        assert np.allclose(D[:3, 3], 0, atol=1e-12)
        ang = np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1))
        assert 0.03 < ang < 0.08
        seen.add(tuple(np.round(D[:3, :3].ravel(), 6)))
    assert len(seen) == 26
    g = pkg.make_track_hypotheses(slast, sprelast, lastF)
    assert g.shape == (31, 7)
    assert np.max(np.abs(g - tries)) < 1e-14
