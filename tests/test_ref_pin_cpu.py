"""Pins the oracle (oracle/*.cpp, the dependency-free restatement) to the REFERENCE ITSELF: oracle/_ref/libref.so holds the reference's
own sources — CoarseTracker.cpp, HessianBlocks.cpp, Residuals.cpp, MatrixAccumulators.h, Accumulated{Top,SC}Hessian.cpp,
EnergyFunctional.cpp, FullSystemOptimize.cpp, ... — compiled unmodified from /root/reference by oracle/Makefile.ref against stand-in
headers for Eigen / Sophus / Boost (oracle/ref_shim).  Every comparison here is bitwise unless it says otherwise.

Runs where libref.so exists or can be built (this container); skipped on a machine that has neither."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_py as R  # noqa: E402

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so not built and /root/reference absent")


def bits(a):
    a = np.atleast_1d(np.ascontiguousarray(a))
    return a.view(np.uint8)


def same(a, b):
    a = np.atleast_1d(np.asarray(a)); b = np.atleast_1d(np.asarray(b))
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(bits(a), bits(b))


@pytest.fixture(scope="module")
def O(oracle):
    L = oracle.lib()
    c_f, c_d = oracle.c_f, oracle.c_d
    L.orc_interp33.argtypes = [c_f, C.c_int, C.c_int, c_f, c_f, c_f]
    L.orc_aff_from_to.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double, C.c_double, C.c_double, c_d]
    L.orc_acc9_stream.argtypes = [C.c_int, c_f, c_f, C.c_int, c_f, c_f, C.c_int, c_f, C.POINTER(C.c_long)]
    L.orc_accapprox_stream.argtypes = [C.c_int, c_f, C.c_int, c_f, C.POINTER(C.c_long)]
    L.orc_accxx_stream.argtypes = [C.c_int, c_f, C.c_int, c_f, c_f, c_f, C.POINTER(C.c_long)]
    L.orc_project_point_short.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, c_f, c_f, c_f]
    L.orc_project_point_long.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, c_f, c_f, c_f]
    return oracle


# ---------------------------------------------------------------------------------------------------------------- primitives
def test_interpolation_bitwise(O):
    """getInterpolatedElement33 (globalFuncs.h:103-118)."""
    rng = np.random.default_rng(1)
    h, w = 96, 128
    img = (rng.random((h, w, 3), dtype=np.float32) * 255).astype(np.float32)
    n = 20000
    x = (rng.random(n) * (w - 2)).astype(np.float32); y = (rng.random(n) * (h - 2)).astype(np.float32)
    x[:50] = np.floor(x[:50]); y[25:75] = np.floor(y[25:75])  # exact pixel centres too
    ref = R.interp33(img, x, y)
    out = np.zeros((n, 3), np.float32)
    O.lib().orc_interp33(O._f(img), w, n, O._f(x), O._f(y), O._f(out))
    assert same(ref, out)


def test_affine_transfer_bitwise(O):
    """AffLight::fromToVecExposure (NumType.h:174-186), incl. the zero-exposure rule."""
    rng = np.random.default_rng(2)
    for _ in range(200):
        eF, eT = [float(np.float32(v)) for v in rng.uniform(0.2, 30, 2)]
        if rng.random() < 0.1:
            eF = 0.0
        aF, aT = rng.uniform(-0.5, 0.5, 2); bF, bT = rng.uniform(-20, 20, 2)
        ref = R.aff_from_to(eF, eT, aF, bF, aT, bT)
        out = np.zeros(2); O.lib().orc_aff_from_to(eF, eT, aF, bF, aT, bT, O._d(out))
        assert same(ref, out)


def test_project_point_both_forms_bitwise(O, synth):
    """projectPoint (ResidualProjections.h:47-58 and :62-87): same accept / reject, same bits."""
    rng = np.random.default_rng(3)
    w, h = 320, 240
    K4 = np.array([150.0, 160.0, 158.3, 121.7], np.float32)
    R.pyr_levels(w, h, K4)
    case = dict(K4=K4, w=w, h=h, n_frames=0, u=[], v=[], host=[], color=np.zeros((0, 8)), weights=np.zeros((0, 8)), res_point=[], res_target=[], poses0=[], idepth0=[],
                imgs=[], dI0=[])
    W = O.BAWindow(case)
    n_ok = 0
    for _ in range(3000):
        ang = rng.normal(0, 0.05, 3)
        Rm = (np.eye(3) + np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]])).astype(np.float32)
        t = rng.normal(0, 0.2, 3).astype(np.float32)
        u, v = float(np.float32(rng.uniform(-10, w + 10))), float(np.float32(rng.uniform(-10, h + 10)))
        idepth = float(np.float32(rng.uniform(-0.1, 2.0)))
        ok_r, out_r = R.project_point_long(u, v, idepth, 0, 0, K4, Rm, t)
        out_o = np.zeros(9, np.float32)
        ok_o = O.lib().orc_project_point_long(W.p, u, v, idepth, O._f(Rm.reshape(-1).copy()), O._f(t), O._f(out_o))
        assert ok_r == bool(ok_o)
        # the reference leaves its outputs untouched past the point of rejection; compare what both computed
        if ok_r:
            assert same(out_r, out_o); n_ok += 1
        Kf = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float32)
        KRKi = (Kf @ Rm @ np.linalg.inv(Kf)).astype(np.float32); Kt = (Kf @ t).astype(np.float32)
        ok_r2, o_r2 = R.project_point_short(u, v, idepth, KRKi, Kt)
        o_o2 = np.zeros(2, np.float32)
        ok_o2 = O.lib().orc_project_point_short(W.p, u, v, idepth, O._f(KRKi.reshape(-1).copy()), O._f(Kt), O._f(o_o2))
        assert ok_r2 == bool(ok_o2) and same(o_r2, o_o2)
    assert n_ok > 500


def test_accumulators_across_both_shift_up_levels(O):
    """Accumulator9 / AccumulatorApprox / AccumulatorXX / AccumulatorX (MatrixAccumulators.h:36-237, 595-972, 982-1345) fed with
    streams long enough to cross the 1k shift-up many times and the 1M shift-up once (1001 x 1001 = 1,002,001 updates)."""
    rng = np.random.default_rng(4)
    L = O.lib()
    # Accumulator9: 10,100 groups of 4 points x 100 repetitions = 1,010,000 SSE updates, then 7 single weighted points
    n4 = 10100
    J = rng.normal(0, 3, (4 * n4, 9)).astype(np.float32); w = rng.uniform(0.1, 1, 4 * n4).astype(np.float32)
    Js = rng.normal(0, 3, (7, 9)).astype(np.float32); ws = rng.uniform(0.1, 1, 7).astype(np.float32)
    for reps in (1, 100):
        Hr, nr = R.acc9_stream(J, w, Js, ws, reps=reps)
        Ho, no = R.acc9_stream(J, w, Js, ws, reps=reps, _fn=L.orc_acc9_stream)
        assert nr == no == 4 * n4 * reps + 7
        assert same(Hr, Ho)
    rec = rng.normal(0, 2, (10100, 35)).astype(np.float32)
    for reps in (1, 100):
        Hr, nr = R.accapprox_stream(rec, reps=reps)
        Ho, no = R.accapprox_stream(rec, reps=reps, _fn=L.orc_accapprox_stream)
        assert nr == no and same(Hr, Ho)
    rec = rng.normal(0, 2, (10100, 21)).astype(np.float32)
    for reps in (1, 100):
        r = R.accxx_stream(rec, reps=reps); o = R.accxx_stream(rec, reps=reps, _fn=L.orc_accxx_stream)
        assert r[3] == o[3]
        for a, b in zip(r[:3], o[:3]):
            assert same(a, b)


# ---------------------------------------------------------------------------------------------------------------- images
@pytest.mark.parametrize("wh", [(256, 256), (640, 480), (200, 120)])
def test_make_images_bitwise(O, synth, wh):
    """FrameHessian::makeImages (HessianBlocks.cpp:128-191): dIp of every level, absSquaredGrad on the rows the reference writes
    (it leaves rows 0 and h-1 of the gradient channels and of absSquaredGrad uninitialised)."""
    w, h = wh
    rng = np.random.default_rng(5)
    img = (rng.random((h, w)) * 255).astype(np.float32)
    dIr, abr = R.make_images(img, w, h)
    dIo, abo = O.make_images(img, w, h)
    assert len(dIr) == len(dIo) == O.pyr_levels(w, h)
    for l in range(len(dIr)):
        assert same(dIr[l][:, :, 0], dIo[l][:, :, 0])
        assert same(dIr[l][1:-1], dIo[l][1:-1])
        assert same(abr[l][1:-1], abo[l][1:-1])


def test_abs_squared_grad_with_response_table_bitwise(O, synth):
    """absSquaredGrad weighted by CalibHessian::getBGradOnly^2 (setting_gammaWeightsPixelSelect == 1, HessianBlocks.cpp:184-188) for a non-linear response."""
    w, h = 320, 240
    rng = np.random.default_rng(6)
    img = (rng.random((h, w)) * 255).astype(np.float32)
    B = (255.0 * (np.arange(256) / 255.0) ** 0.7).astype(np.float32)
    _, abr = R.make_images(img, w, h, B=B)
    _, abo = O.make_images(img, w, h, B=B)
    _, plain = O.make_images(img, w, h)
    for l in range(len(abr)):
        assert same(abr[l][1:-1], abo[l][1:-1])
    assert not same(abo[0][1:-1], plain[0][1:-1])


# ---------------------------------------------------------------------------------------------------------------- tracker
def _trackers(O, synth, w, h, n_ref, seed=0, exposure=(1.0, 1.0), aff_ref=(0.0, 0.0)):
    case = synth.tracking_case(w, h, n_ref=n_ref, seed=seed) if seed else synth.tracking_case(w, h, n_ref=n_ref)
    K4 = case["K4"]
    dIr, _ = O.make_images(case["ref_img"], w, h)
    T = O.Tracker(w, h); T.make_k(K4)
    T.set_ref(dIr, case["u"], case["v"], case["idepth"], case["hdiF"], exposure=exposure[0], aff=aff_ref)
    RT = R.Tracker(w, h, K4)
    RT.set_ref(case["ref_img"], case["u"], case["v"], case["idepth"], case["hdiF"], exposure=exposure[0], aff=aff_ref)
    return case, T, RT


def test_tracker_template_bitwise(O, synth):
    """makeK + setCoarseTrackingRef / makeCoarseDepthL0 (CoarseTracker.cpp:105-134, 138-295): intrinsics, idepth / weight maps, pc_* lists."""
    case, T, RT = _trackers(O, synth, 320, 240, 1800)
    for l in range(T.levels):
        ko, kio = T.get_k(l); kr, kir = RT.get_k(l)
        assert same(ko, kr) and same(kio, kir)
        assert T.pc_n(l) == RT.pc_n(l) > 0
        for a, b in zip(T.get_pc(l), RT.get_pc(l)):
            assert same(a, b)
        for a, b in zip(T.get_idepth(l), RT.get_idepth(l)):
            assert same(a, b)


def test_tracker_evaluations_bitwise(O, synth):
    """calcRes + calcGSSSE (CoarseTracker.cpp:361-517, 299-356): result vector, the 8 warped buffers, H and b, at every level for
    identity, the true motion, a large motion (saturated residuals) and non-trivial brightness / exposure."""
    case, T, RT = _trackers(O, synth, 256, 256, 1500, exposure=(1.3, 0.9), aff_ref=(0.02, -3.0))
    fr = case["frames"][0]
    dIn, _ = O.make_images(fr["img"], 256, 256)
    T.set_new(dIn, exposure=0.9); RT.set_new(fr["img"], exposure=0.9)
    rng = np.random.default_rng(6)
    poses = [np.array([0, 0, 0, 0, 0, 0, 1.0]), np.asarray(fr["pose7"], dtype=np.float64)]
    for _ in range(4):
        poses.append(O.se3_exp(rng.normal(0, [0.05, 0.05, 0.05, 0.02, 0.02, 0.02])))
    for pose in poses:
        for aff in ([0.0, 0.0], [0.03, 4.0]):
            for l in range(T.levels - 1, -1, -1):
                for cutoff in (20.0, 40.0):
                    ro = T.calc_res(l, pose, aff, cutoff); rr = RT.calc_res(l, pose, aff, cutoff)
                    assert same(ro, rr), (l, ro, rr)
                    assert same(T.get_warped(), RT.get_warped())
                Ho, bo = T.calc_gs(l, aff); Hr, br = RT.calc_gs(l, aff)
                assert same(Ho, Hr) and same(bo, br)


@pytest.mark.parametrize("frame", [0, 1, 2])
def test_track_newest_coarse_bitwise(O, synth, frame):
    """trackNewestCoarse (CoarseTracker.cpp:539-770, useimu=0): pose, affine parameters, residuals, flow indicators, return value."""
    case, T, RT = _trackers(O, synth, 256, 256, 1500)
    fr = case["frames"][frame % len(case["frames"])]
    dIn, _ = O.make_images(fr["img"], 256, 256)
    T.set_new(dIn); RT.set_new(fr["img"])
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    for modeA, modeB in ((1e12, 1e8), (-1.0, -1.0), (1e12, -1.0)):
        ro = T.track(ident, [0.0, 0.0], modeA=modeA, modeB=modeB); rr = RT.track(ident, [0.0, 0.0], modeA=modeA, modeB=modeB)
        assert ro["good"] == rr["good"]
        assert same(ro["pose7"], rr["pose7"]) and same(ro["aff"], rr["aff"])
        assert same(ro["lastResiduals"], rr["lastResiduals"]) and same(ro["flow"], rr["flow"])
    # abort rule: thresholds just under the residual of the coarsest level make both give up at the same place
    base = T.track(ident, [0.0, 0.0])
    mr = np.array(base["lastResiduals"]) * 0.5
    mr[np.isnan(mr)] = 100.0
    ro = T.track(ident, [0.0, 0.0], min_res=mr); rr = RT.track(ident, [0.0, 0.0], min_res=mr)
    assert ro["good"] == rr["good"] and same(ro["lastResiduals"], rr["lastResiduals"]) and same(ro["pose7"], rr["pose7"])


def test_vio_branch_hands_over_the_same_systems(O, synth):
    """The reference's default branch (setting_useIMU, CoarseTracker.cpp:612-637): with a computeCoarseUpdate that evaluates the
    visual-only step, the track ends exactly where the useimu=0 branch ends, and every (H, b) it handed over is calcGSSSE's output."""
    case, T, RT = _trackers(O, synth, 256, 256, 1500)
    fr = case["frames"][0]
    dIn, _ = O.make_images(fr["img"], 256, 256)
    T.set_new(dIn); RT.set_new(fr["img"])
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    plain = RT.track(ident, [0.0, 0.0])
    vio = RT.track(ident, [0.0, 0.0], vio=True)
    assert vio["vio_calls"] > 5 and vio["vio_accepts"] > 3 and vio["vio_visual"] == 1
    assert same(plain["pose7"], vio["pose7"]) and same(plain["aff"], vio["aff"]) and same(plain["lastResiduals"], vio["lastResiduals"])
    oracle_run = T.track(ident, [0.0, 0.0])
    assert same(oracle_run["pose7"], vio["pose7"])
    # the system given to addVisualToCoarseGraph at the end (CoarseTracker.cpp:763-767) is the last accepted level-0 system
    assert vio["vio_visual_Hb"][72] == 1.0


# ---------------------------------------------------------------------------------------------------------------- bundle adjustment
def _windows(O, synth, **kw):
    case = synth.ba_case(**kw)
    return case, O.BAWindow(case), R.BAWindow(case)


@pytest.mark.parametrize("kw", [dict(w=256, h=192, n_frames=4, n_points=150, hosts_share=(60, 50, 40, 0), seed=7),
                                dict(w=320, h=240, n_frames=6, n_points=400, seed=11)])
def test_ba_linearize_accumulate_solve_bitwise(O, synth, kw):
    """FrameFramePrecalc::set, PointFrameResidual::linearize, setNewFrameEnergyTH, applyRes / takeDataF, accumulateAF/LF/SCF_MT with
    stitching, solveSystemF incl. resubstitution (single-threaded order)."""
    case, WO, WR = _windows(O, synth, **kw)
    assert same(WR.sampled_color, np.asarray(case["color"], np.float32))  # ImmaturePoint's constructor samples the same colours
    F = case["n_frames"]
    for hh in range(F):
        for tt in range(F):
            if hh != tt:
                assert same(WO.precalc(hh, tt)["KRKi"], WR.precalc(hh, tt)["KRKi"])
                for k in ("Kt", "R0", "t0", "aff", "b0", "R"):
                    assert same(np.float32(WO.precalc(hh, tt)[k]), np.float32(WR.precalc(hh, tt)[k])), k
    ao, at, ad = WO.adjoints(); bo, bt, bd = WR.adjoints()
    assert same(ao, bo) and same(at, bt) and same(ad, bd)
    WO.activate_all(); WR.activate_all()
    eo = WO.linearize_all(False); er = WR.linearize_all(False)
    assert eo == er
    so, sr = WO.res_state(), WR.res_state()
    for k in so:
        assert same(so[k], sr[k]), k
    for i in range(WO.R):
        jo, jr = WO.get_J(i), WR.get_J(i)
        for k in ("resF", "Jpdxi", "Jpdc", "Jpdd", "JIdx", "JabF", "JIdx2", "JabJIdx", "Jab2"):
            assert same(jo[k], jr[k]), (i, k)
    assert same(WO.frame_energy_th(), WR.frame_energy_th())
    WO.apply_res(); WR.apply_res()
    for i in range(0, WO.R, 7):
        assert same(WO.get_J(i, 1)["JpJdF"], WR.get_J(i, 1)["JpJdF"])
    ao, ar = WO.accumulate(), WR.accumulate()
    for k in ao:
        assert same(np.asarray(ao[k]), np.asarray(ar[k])), k
    po, pr = WO.point_acc(), WR.point_acc()
    for k in po:
        assert same(po[k], pr[k]), k
    assert WO.lenergy() == WR.lenergy() and WO.menergy() == WR.menergy()
    for it, lam in ((0, 1e-5), (2, 1e-3)):   # iteration >= 2 orthogonalises against the gauge nullspaces (SVD: tolerance, not bits)
        xo = WO.solve(it, lam); xr = WR.solve(it, lam)
        if it == 0:
            assert same(xo, xr)
        else:
            assert np.abs(xo - xr).max() < 1e-9 * max(1.0, np.abs(xo).max())
        Ho, b0 = WO.last_system(); Hr, b1 = WR.last_system()
        assert same(Ho, Hr) and same(b0, b1)
        io, so_ = WO.point_state(); ir, sr_ = WR.point_state()
        if it == 0:
            assert same(so_, sr_)
    no, nr = WO.nullspaces(), WR.nullspaces()
    assert np.abs(no - nr).max() < 1e-12


@pytest.mark.parametrize("kw", [dict(w=256, h=192, n_frames=4, n_points=150, hosts_share=(60, 50, 40, 0), seed=7),
                                dict(w=320, h=240, n_frames=7, n_points=500, seed=3)])
def test_full_system_optimize_bitwise(O, synth, kw):
    """FullSystem::optimize (FullSystemOptimize.cpp:417-647), the reference's own function from first linearisation to the final
    fix-linearisation: same accept / reject sequence, same energies, and the states it leaves behind bit for bit."""
    case, WO, WR = _windows(O, synth, **kw)
    ro = WO.optimize(6); rr = WR.optimize(6)
    tr_o = ro["trace"]; tr_r = rr["trace"]
    assert ro["iterations"] == rr["iterations"] == len(tr_r) - 1
    assert list(tr_o[1:, 3]) == list(tr_r[1:, 1])                 # accept / reject
    acc = tr_r[:, 1] != 0
    # the reference prints the energy of the attempted step with 6 decimals; on accepted steps that is the oracle's lastEnergy
    assert np.allclose(tr_o[acc, 0], tr_r[acc, 0], rtol=0, atol=1e-6)
    assert ro["rmse"] == rr["rmse"]
    for k in range(case["n_frames"]):
        po, pr = WO.frame_pose(k), WR.frame_pose(k)
        # from iteration 2 on the step is orthogonalised against the gauge nullspaces through an SVD pseudo-inverse (Eigen's JacobiSVD in
        # the reference; two different textbook Jacobi SVDs here and in the oracle): last-bit differences of the states, nothing more
        assert np.abs(po[0] - pr[0]).max() < 1e-13 and np.abs(po[1] - pr[1]).max() < 1e-13 and np.abs(po[2] - pr[2]).max() < 1e-13
    io, so_ = WO.point_state(); ir, sr_ = WR.point_state()
    assert np.abs(io - ir).max() < 1e-6 * np.abs(io).max() and np.mean(io == ir) > 0.98


def test_linearised_residuals_bitwise(O, synth):
    """EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:85-113), AccumulatedTopHessianSSE::addPoint<1> through accumulateLF_MT
    (AccumulatedTopHessian.cpp:36-160, EnergyFunctional.cpp:268-299), calcLEnergyPt (EnergyFunctional.cpp:349-409: an Accumulator11 per run
    of 50 points) and FullSystem::optimize with residuals outside activeResiduals (FullSystemOptimize.cpp:436-446).  The windows are brought
    to the same bits first (one-iteration optimisations: no SVD orthogonalisation, and residuals dropped on the way change the order the
    per-point sums run in, EnergyFunctional::dropResidual), then every third residual is linearised on both sides."""
    case, WO, WR = _windows(O, synth, w=256, h=192, n_frames=5, n_points=300, hosts_share=(90, 80, 70, 60, 0), seed=5)
    for _ in range(3):
        ro = WO.optimize(1); rr = WR.optimize(1)
        assert ro["rmse"] == rr["rmse"]
    mask = (np.arange(WO.R) % 3 == 0).astype(np.uint8)
    n_lin = WO.fix_linearization(mask)
    assert n_lin == WR.fix_linearization(mask) and 250 < n_lin < WO.R // 3 + 1
    ao, ar = WO.accumulate(), WR.accumulate()
    for k in ao:
        assert same(np.asarray(ao[k]), np.asarray(ar[k])), k
    F = case["n_frames"]
    off = ar["HL"] - np.diag(np.diag(ar["HL"]))
    assert np.abs(off).max() > 1e3 and np.abs(ar["bL"][4 + 8:]).max() > 1e2      # the L system is populated, not just the priors
    po, pr = WO.point_acc(), WR.point_acc()
    for k in po:
        assert same(po[k], pr[k]), k
    # the fp32 totals of the 50-point runs are added in double in the order the reference's workers finish
    assert abs(WO.lenergy() - WR.lenergy()) <= 1e-12 * abs(WR.lenergy()) and abs(WR.lenergy()) > 1e3
    ro = WO.optimize(2); rr = WR.optimize(2)                                        # iterations 0 and 1: no orthogonalisation
    assert list(ro["trace"][1:, 3]) == list(rr["trace"][1:, 1]) and ro["rmse"] == rr["rmse"]
    for k in range(F):
        po_, pr_ = WO.frame_pose(k), WR.frame_pose(k)
        assert same(po_[0], pr_[0]) and same(po_[1], pr_[1]) and same(po_[2], pr_[2])
    io, so_ = WO.point_state(); ir, sr_ = WR.point_state()
    assert same(io, ir)
    assert abs(WO.lenergy() - WR.lenergy()) <= 1e-12 * abs(WR.lenergy())
    assert WO.fix_linearization(np.zeros(WO.R, np.uint8)) == WR.fix_linearization(np.zeros(WO.R, np.uint8))   # linearised residuals left after the drops


def _marg_pair(O, synth, case, perturb):
    Wr = R.BAWindow(case); Wo = O.BAWindow(case)
    rng = np.random.RandomState(3)
    if perturb:
        for k in range(1, case["n_frames"]):
            st = np.zeros(10); st[:3] = 2e-3 * rng.standard_normal(3); st[3:6] = 1e-3 * rng.standard_normal(3); st[6] = 1e-3 * rng.standard_normal(); st[7] = 1e-4 * rng.standard_normal()
            Wr.set_frame_state(k, st); Wo.set_frame_state(k, st)
    for X in (Wr, Wo):
        X.activate_all(); X.linearize_all(False); X.apply_res(); X.accumulate()
    Wr.set_num_good_residuals(100)      # every point has been an inlier for long: isInlierNew then only asks for >= setting_minGoodActiveResForMarg (3) residuals
    dr, Hr, br, nr = Wr.marginalize_points(np.array([0, 1, 0, 0, 0], np.uint8))
    # the caller's candidate list (FullSystem.cpp:785-827): the points of the flagged keyframe plus the few the reference's own PointHessian::isOOB test selects (host code
    # in an integration, HessianBlocks.h:196-215) — taken from the reference's run; candidates that fail isInlierNew (host bookkeeping) are dropped without device work
    cand = (np.asarray(case["host"]) == 1).astype(np.uint8)
    cand[(dr != 0) & (cand == 0)] = 1
    n_res = np.bincount(case["res_point"], minlength=len(cand))
    few = (cand == 1) & (n_res < 3)
    assert np.all(dr[few] == 2)
    cand[few] = 0
    do, Ho, bo, no = Wo.marginalize_points(cand)
    do[few] = 2
    assert np.array_equal(dr, do) and nr == no
    return Wr, Wo, dr, (Hr, br), (Ho, bo)


def test_marginalisation_against_the_reference(O, synth):
    """FullSystem::flagPointsForRemoval (relinearisation of the points of a keyframe flagged for marginalisation, inlier test), EnergyFunctional::marginalizePointsF
    (addPoint<2> + Schur complement into HM / bM) and EnergyFunctional::marginalizeFrame — the reference's own members vs the oracle.  Decisions and residual counts are
    identical.  The prior's increment agrees to double rounding when the order of the fp32 sums is forced (a handful of points: every (host, target) accumulator sees one
    or two); with 145 candidates the reference adds them in the order its dropPointsF left them in (removal swaps the last point into the hole, EnergyFunctional.cpp:644-676),
    the oracle in index order: same values to fp32 summation rounding."""
    small = synth.ba_case(320, 256, n_frames=5, n_points=400, hosts_share=(200, 1, 110, 89, 0), seed=35)
    Wr, Wo, dr, (Hr, br), (Ho, bo) = _marg_pair(O, synth, small, perturb=True)
    assert (dr == 1).sum() >= 2
    assert np.linalg.norm(Hr - Ho) <= 1e-13 * np.linalg.norm(Hr) and np.linalg.norm(br - bo) <= 1e-12 * np.linalg.norm(br)
    Wo.set_marg_prior(Ho, bo)
    Hn_r, bn_r = Wr.marginalize_frame(1); Hn_o, bn_o = Wo.marginalize_frame(1)
    # marginalizeFrame inverts the frame's 8x8 block (EnergyFunctional.cpp:619): the stand-in header's inverse and the oracle's are different elimination orders
    assert np.linalg.norm(Hn_r - Hn_o) <= 1e-7 * np.linalg.norm(Hn_r) and np.linalg.norm(bn_r - bn_o) <= 1e-7 * np.linalg.norm(bn_r)
    big = synth.ba_case(320, 256, n_frames=5, n_points=500, hosts_share=(160, 140, 110, 90, 0), seed=33)
    Wr, Wo, dr, (Hr, br), (Ho, bo) = _marg_pair(O, synth, big, perturb=True)
    assert (dr == 1).sum() > 50 and (dr == 2).sum() > 20
    assert np.linalg.norm(Hr - Ho) <= 2e-6 * np.linalg.norm(Hr) and np.linalg.norm(br - bo) <= 2e-6 * np.linalg.norm(br)
    Wo.set_marg_prior(Hr, br)          # the same prior on both sides: marginalizeFrame itself
    Hn_r, bn_r = Wr.marginalize_frame(1); Hn_o, bn_o = Wo.marginalize_frame(1)
    assert np.linalg.norm(Hn_r - Hn_o) <= 1e-7 * np.linalg.norm(Hn_r) and np.linalg.norm(bn_r - bn_o) <= 1e-7 * np.linalg.norm(bn_r)


# ---------------------------------------------------------------------------------------------------------------- immature points
def test_immature_points_bitwise(O, synth):
    """ImmaturePoint::ImmaturePoint, FullSystem::traceNewCoarse (per-host KRKi / Kt / affine + ImmaturePoint::traceOn) through four keyframes with exposure / affine
    changes, then FullSystem::optimizeImmaturePoint against the window — the reference's own members on its own FullSystem vs the oracle: every output bit for bit."""
    w, h, F = 320, 256, 5
    world = synth.PlaneWorld(synth.SEED + 6, fmax=22.0)
    K4 = synth.default_intrinsics(w, h)
    rng = np.random.RandomState(6)
    imgs, w2c = [], []
    affs = np.array([[0.01 * k, 0.5 * k] for k in range(F)]); expo = np.array([1.0, 0.9, 1.1, 1.0, 1.2], np.float32)
    for k in range(F):
        R_, t_ = synth.se3_exp(np.array([0.05 * k, -0.02 * k, 0.012 * k, 0.003 * k, -0.004 * k, 0.0015 * k]))
        imgs.append(world.render(K4, R_, t_, w, h, aff=tuple(affs[k]))[0]); w2c.append(synth.pose7(R_, t_))
    u, v = synth.select_points(imgs[0], 400, rng, min_grad=8.0)
    u = u.astype(np.int32); v = v.astype(np.int32)
    keep = (u >= 8) & (v >= 8) & (u < w - 8) & (v < h - 8)
    u, v = u[keep], v[keep]
    case = dict(w=w, h=h, K4=K4, n_frames=F, imgs=imgs, poses0=np.array(w2c), idepth0=np.zeros(0, np.float32), aff=affs, exposure=expo,
                u=np.zeros(0, np.float32), v=np.zeros(0, np.float32), host=np.zeros(0, np.int32), color=np.zeros((0, 8), np.float32), weights=np.zeros((0, 8), np.float32),
                res_point=np.zeros(0, np.int32), res_target=np.zeros(0, np.int32))
    Wr = R.BAWindow(case)
    Wr.immature_add(0, u, v)
    dIs = [O.make_images(im, w, h)[0][0] for im in imgs]
    P = O.ImmaturePoints(dIs[0], w, h, u, v)
    g = Wr.immature_get(0)
    for k in ("color", "weights", "gradH", "energyTH"):
        assert same(g[k], getattr(P, k)), k
    c2w0 = O.se3_inv(w2c[0])
    for k in range(1, F):
        Wr.trace_new_coarse(k)
        KRKi, Kt, aff = O.trace_precalc(w2c[k], c2w0, K4, float(expo[k]), float(expo[0]), tuple(affs[k]), tuple(affs[0]))
        P.trace_on(dIs[k], KRKi, Kt, aff)
        g = Wr.immature_get(0)
        assert np.array_equal(g["lastTraceStatus"], P.lastTraceStatus), k
        for name in ("idepth_min", "idepth_max", "quality", "lastTraceUV", "lastTracePixelInterval"):
            assert same(g[name], getattr(P, name)), (k, name)
    assert (P.lastTraceStatus == 0).sum() > 100
    usable = np.isfinite(P.idepth_max) & (P.lastTraceStatus != 1)
    P.idepth_max[~usable] = 0.5; P.idepth_min[~usable] = 0.1
    Wr.immature_set_interval(0, P.idepth_min, P.idepth_max)
    res_r, id_r, st_r = Wr.optimize_immature(0, min_obs=1)
    pre = [O.pair_precalc(w2c[k], c2w0, float(expo[0]), float(expo[k]), tuple(affs[0]), tuple(affs[k])) for k in range(1, F)]
    Rm = np.stack([q[0] for q in pre]); tm = np.stack([q[1] for q in pre]); am = np.stack([q[2] for q in pre])
    res_o, id_o, st_o = O.immature_optimize(P, K4, dIs[1:], Rm, tm, am, min_obs=1)
    assert np.array_equal(res_r, res_o) and (res_o == 1).sum() > 100
    act = res_o == 1
    assert same(id_r[act], id_o[act])
    assert np.array_equal(st_r[act], st_o[act])


# ---------------------------------------------------------------------------------------------------------------- initializer
@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_initializer_calc_res_and_gs_bitwise(O, synth, lvl):
    """CoarseInitializer::calcResAndGS (CoarseInitializer.cpp:331-624): every per-point output, JbBuffer_new and the Schur system bit for bit; H, b and the energy — which the
    reference itself sums in a run-dependent order — to rounding."""
    from test_init_cpu import init_case
    for kw in (dict(), dict(alphaW=0.0, alphaK=1e9, couplingWeight=0.0), dict(priorY=3.0, priorX=0.5)):
        c = init_case(synth, O, lvl=lvl, n=700, seed=12 + lvl)
        K4 = synth.default_intrinsics(c["w"], c["h"])
        K_lvl, Ki = R.init_make_k(c["w"], c["h"], K4, lvl)
        r = R.init_calc_res_and_gs(c["img0"], c["img1"], c["w"], c["h"], K4, lvl, c["pose7"], c["aff"], c["pts"], c["idepth_new"], **kw)
        dI0 = O.make_images(c["img0"], c["w"], c["h"])[0]; dI1 = O.make_images(c["img1"], c["w"], c["h"])[0]
        o = O.init_calc_res_and_gs(dI0[lvl], dI1[lvl], c["wl"], c["hl"], Ki, K_lvl, c["pose7"], c["aff"], c["pts"], c["idepth_new"], **kw)
        good = o["isGood_new"].astype(bool)
        assert np.array_equal(r["isGood_new"], o["isGood_new"]) and good.sum() > 400
        # deterministic in the reference: everything per point and the Schur system (one sequential loop over the points, CoarseInitializer.cpp:557-583)
        for k in ("Hsc", "bsc", "energy_new", "maxstep"):
            assert same(r[k], o[k]), (k, kw, float(np.abs(r[k] - o[k]).max()))
        assert same(r["res3"][1:], o["res3"][1:])
        # H, b and the energy are summed by the reference's IndexThreadReduce, whose six workers take the 50-point chunks as they come (the single-threaded shortcut is
        # commented out, util/IndexThreadReduce.h:83-87): its own sums vary in the last bits from run to run.  The oracle is the one-worker order.
        sc = np.sqrt(np.outer(np.abs(np.diag(o["H"])), np.abs(np.diag(o["H"])))) + 1e-30
        assert np.max(np.abs(r["H"] - o["H"]) / sc) < 2e-5 and np.allclose(r["b"], o["b"], rtol=1e-4, atol=1e-5 * np.abs(o["b"]).max())
        assert abs(r["res3"][0] - o["res3"][0]) <= 1e-5 * o["res3"][0]
        assert same(r["lastHessian_new"][good], o["lastHessian_new"][good]) and same(r["JbBuffer_new"][good], o["JbBuffer_new"][good])


# ---------------------------------------------------------------------------------------------------------------- result.txt
def test_print_result_bytes(O, synth, tmp_path):
    """FullSystem::printResult (FullSystem.cpp:256-298) of a live run of the reference (keyframes with their optimised camToWorld, the other frames re-based on their
    tracking reference, invalid poses skipped) vs the oracle's writer fed with the shells' data: the files are identical byte for byte."""
    import replay
    run = replay.run_reference(R, synth, 256, 192, 40, step=1.0, point_density=600)
    S = run["system"]
    tr = S.trajectory(); sh = S.shells()
    assert tr["valid"].sum() > 20 and (tr["keyframeId"] >= 0).sum() >= 3
    ref_idx = np.where(tr["keyframeId"] == -1, tr["trackingRef"], -1).astype(np.int32)
    a, b = tmp_path / "ref.txt", tmp_path / "orc.txt"
    S.print_result(a, only_kf=False, use_cam_to_tracking_ref=True)
    O.write_result_txt(b, sh["timestamp"], tr["camToWorld"], pose_valid=tr["valid"].astype(np.uint8), tracking_ref=ref_idx, camToTrackingRef7=sh["camToTrackingRef"],
                       firstPose7=sh["firstPose"])
    assert a.read_bytes() == b.read_bytes() and len(a.read_bytes().splitlines()) == int(tr["valid"].sum())
    S.print_result(a, only_kf=False, use_cam_to_tracking_ref=False)
    O.write_result_txt(b, sh["timestamp"], tr["camToWorld"], pose_valid=tr["valid"].astype(np.uint8), firstPose7=sh["firstPose"])
    assert a.read_bytes() == b.read_bytes()


# ---------------------------------------------------------------------------------------------------------------- input edge
@pytest.mark.parametrize("bits,model", [(8, "RadTan"), (16, "RadTan"), (8, "FOV")])
def test_undistort_bitwise(O, tmp_path, bits, model):
    """PhotometricUndistorter::processFrame (Undistort.cpp:214-250) + Undistort::undistort (:386-481) on the reference's own tables — response table normalised by its
    constructor, inverse vignette from a 16-bit image, remap of a radial-tangential / FOV camera cropped to 320x240: the oracle fed with those tables returns the same
    image bit for bit, with the photometric calibration and with the exposure-less fall-back (factor * raw)."""
    wOrg, hOrg, w, h = 376, 288, 320, 240
    rng = np.random.default_rng(3 + bits)
    cam = tmp_path / "camera.txt"
    if model == "RadTan":
        cam.write_text("RadTan 0.535 0.669 0.493 0.505 -0.28 0.07 0.0002 -0.0003\n%d %d\ncrop\n%d %d\n" % (wOrg, hOrg, w, h))
    else:
        cam.write_text("0.535 0.669 0.493 0.505 0.897\n%d %d\ncrop\n%d %d\n" % (wOrg, hOrg, w, h))
    depth = 256 if bits == 8 else 65536
    g = np.cumsum(rng.uniform(0.5, 1.5, depth)) ** 1.1
    gam = tmp_path / "pcalib.txt"
    gam.write_text(" ".join("%.9g" % x for x in g) + "\n")
    yy, xx = np.mgrid[0:hOrg, 0:wOrg]
    vig = (65535 * (1 - 0.4 * (((xx - wOrg / 2) / wOrg) ** 2 + ((yy - hOrg / 2) / hOrg) ** 2))).astype(np.uint16)
    U = R.Undistorter(cam, gam, vig)
    assert (U.w, U.h, U.wOrg, U.hOrg, U.GDepth, U.valid, U.passthrough) == (w, h, wOrg, hOrg, depth, 1, 0)
    assert (U.remapX >= 0).mean() > 0.99
    raw = rng.integers(0, depth, (hOrg, wOrg)).astype(np.uint8 if bits == 8 else np.uint16)
    ref_img, _ = U.undistort(raw, exposure=0.02)
    orc_img = O.undistort(raw, U.G, U.vignetteMapInv, U.remapX, U.remapY, w, h)
    assert same(ref_img, orc_img) and ref_img.std() > 10
    ref_img, _ = U.undistort(raw, exposure=0.0, factor=0.25)          # exposure <= 0: no photometric calibration, data = factor * raw (:222-229)
    assert same(ref_img, O.undistort(raw, None, None, U.remapX, U.remapY, w, h, factor=0.25))


def test_lie_algebra_against_the_vendored_sophus(oracle):
    """oracle/lie.h — the SE3 / SO3 arithmetic behind every oracle AND behind the stand-in sophus/se3.hpp the reference's sources are compiled against — equals the reference's
    vendored Sophus (thirdparty/Sophus/sophus/so3.hpp + se3.hpp, compiled unmodified into oracle/_ref/libsophus_pin.so) bit for bit: exp incl. the small-angle series branch,
    log incl. its small-angle branch, products (normalisation after every product), inverse, Adj, the 3x4 matrix, point transforms, the normalising constructor."""
    S = R.sophus_pin()
    rng = np.random.RandomState(11)
    tangents = [np.zeros(6)]
    for scale in (1.0, 0.1, 1e-3, 1e-6, 3e-9, 1e-11, 1e-13, 0.0):       # 1e-10 = SophusConstants<double>::epsilon(): both branches of exp / log / V
        for _ in range(40):
            a = rng.standard_normal(6); a[3:] *= scale
            tangents.append(a)
    for k in range(40):                                                      # large rotations up to just below pi (log's atan branch near w = 0)
        w = rng.standard_normal(3); w *= (np.pi * (1 - 10.0 ** -(k % 8 + 1))) / np.linalg.norm(w)
        tangents.append(np.r_[rng.standard_normal(3), w])
    # the tangent set of the reference's sophus/test_se3.cpp
    for a in ([0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0], [-1, 1, 0, 0, 0, 1], [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0]):
        tangents.append(np.array(a, dtype=np.float64))
    poses = []
    for a in tangents:
        p_o, p_s = oracle.se3_exp(a), S.exp(a)
        assert same(p_o, p_s), a
        poses.append(p_s)
    for p in poses:
        assert same(oracle.se3_log(p), S.log(p))
        assert same(oracle.se3_inv(p), S.inverse(p))
        assert same(oracle.se3_adj(p), S.adj(p))
        Rm, t = oracle.se3_matrix(p)
        assert same(np.c_[Rm, t], S.matrix3x4(p))
    # transformation of points: Sophus computes unit_quaternion()._transformVector(p) + translation, the oracle's trackers use the rotation matrix (as the
    # reference's CoarseTracker does: rotationMatrix().cast<float>()); the pin here is of the product, whose translation is exactly such a transform
    idx = rng.randint(0, len(poses), (400, 2))
    for i, j in idx:
        ab_o, ab_s = oracle.se3_mul(poses[i], poses[j]), S.mul(poses[i], poses[j])
        assert same(ab_o, ab_s)
        assert same(oracle.se3_log(ab_o), S.log(ab_s))
    # chains (a trajectory: every product renormalises) stay identical
    T_o, T_s = poses[1].copy(), poses[1].copy()
    for i in rng.randint(0, len(poses), 300):
        T_o, T_s = oracle.se3_mul(T_o, poses[i]), S.mul(T_s, poses[i])
        T_o, T_s = oracle.se3_mul(oracle.se3_inv(poses[(i * 7) % len(poses)]), T_o), S.mul(S.inverse(poses[(i * 7) % len(poses)]), T_s)
    assert same(T_o, T_s)
    # a pose built from stored numbers that are not unit to rounding is normalised by the constructor, one that is unit keeps its bits through the oracle's import
    q = np.r_[rng.standard_normal(3), rng.standard_normal(4)]
    norm_s = S.from_quaternion(q)
    assert same(oracle.se3_mul(q, np.r_[0, 0, 0, 0, 0, 0, 1.0])[:3], norm_s[:3])
    assert np.allclose(oracle.se3_inv(oracle.se3_inv(q)), norm_s, rtol=0, atol=1e-15)


def test_track_new_coarse_hypothesis_list_bitwise(oracle, synth, pkg):
    """The motion hypotheses of FullSystem::trackNewCoarse (lastF_2_fh_tries, FullSystem.cpp:364-402) as the reference's own loop hands them to trackNewestCoarse, one try
    after the other: FullSystem.cpp is compiled so that its calls go through oracle/ref_trackhook.cpp, which in ENUMERATE mode records each initial guess and reports failure,
    so the unmodified loop walks its whole list.  The oracle's make_track_hypotheses (and, in tests/test_tracker_gpu.py, the HIP library's) must give the same 31 poses in
    the same order, bit for bit, from the same three camera poses."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import replay
    w, h = 256, 192
    K4, imgs, _ = replay.make_sequence(synth, w, h, 40, 1.6)
    S = R.System(w, h, K4, point_density=600, max_frames=7)
    checked = 0
    for k, img in enumerate(imgs):
        probe = k == 25                           # well after initialisation: three frames in the history, a reference keyframe set
        if probe:
            R.enumerate_tries(True)
        S.add_frame(img)
        if not probe:
            continue
        tries = R.enumerated_tries()
        R.enumerate_tries(False)
        ti = [e for e in S.events() if e["kind"] == "track_in"][-1]
        assert ti["frame_id"] == k and ti["n_history"] > 2 and ti["poses_valid"]
        mine = np.asarray(oracle.make_track_hypotheses(ti["slast_c2w"], ti["sprelast_c2w"], ti["lastF_c2w"]))
        assert len(tries) == 31 and mine.shape == (31, 7)
        assert same(mine, tries), np.abs(mine - tries).max()
        # the product's own list (dmvio_hip_make_track_hypotheses: host pose algebra of csrc/lie_dev.h, no device involved) against the reference directly
        hip = np.asarray(pkg.make_track_hypotheses(ti["slast_c2w"], ti["sprelast_c2w"], ti["lastF_c2w"]))
        assert same(hip, tries), [(i, np.abs(hip[i] - tries[i]).tolist()) for i in range(31) if not same(hip[i], tries[i])]
        checked += 1
        break                                      # the enumerated frame was "lost" on purpose: the run ends here
    assert checked == 1
