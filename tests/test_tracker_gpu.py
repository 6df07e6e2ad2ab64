"""GPU parity tests of the coarse tracker: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Tolerances (north_star): final photometric energy within 1e-4 relative, translation within 1e-3 m.
Single evaluations are compared much tighter: per-point arithmetic is bit-identical to the oracle
(integer counts must match EXACTLY); only the fp32 summation order differs (tree vs the reference's
sequential 4-lane SSE order), which bounds sums at ~1e-6 relative.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IDENT = np.array([0, 0, 0, 0, 0, 0, 1.0])


@pytest.fixture(scope="module")
def setup(pkg, oracle, synth, gpu_required):
    w = h = 512
    case = synth.tracking_case(w, h, n_ref=2000, n_frames=3, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=8)
    trk = pkg.CoarseTrackerHip(ctx)
    trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"])
    for k, f in enumerate(case["frames"]):
        ctx.frame_upload(1 + k, f["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    dIr, _ = oracle.make_images(case["ref_img"], w, h)
    dIn = [oracle.make_images(f["img"], w, h)[0] for f in case["frames"]]
    T = oracle.Tracker(w, h)
    T.make_k(case["K4"])
    T.set_ref(dIr, case["u"], case["v"], case["idepth"], case["hdiF"])
    return dict(case=case, ctx=ctx, trk=trk, T=T, dIr=dIr, dIn=dIn, w=w, h=h)


def test_pyramid_bit_exact(setup):
    """FrameHessian::makeImages: every level of (I,dx,dy) identical to the oracle, bit for bit."""
    ctx = setup["ctx"]
    assert ctx.levels == 4
    for lvl in range(ctx.levels):
        g = ctx.frame_download(0, lvl)
        o = setup["dIr"][lvl]
        assert g.shape == o.shape
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), "level %d differs" % lvl


@pytest.mark.parametrize("wh", [(512, 512), (640, 480), (200, 120)])
def test_abs_squared_grad_bit_exact(pkg, oracle, gpu_required, wh):
    """FrameHessian::absSquaredGrad of levels 0..2 (the pixel selector's input), with and without the response-table weights: identical to the CPU path,
    rows 0 and h-1 zero."""
    w, h = wh
    rng = np.random.default_rng(8)
    img = (rng.random((h, w)) * 255).astype(np.float32)
    img[5, 7] = 0.2; img[9, 3] = 254.9          # clamp ends of getBGradOnly
    ctx = pkg.Context(w, h, n_slots=1)
    ctx.frame_upload(0, img)
    B = (255.0 * (np.arange(256) / 255.0) ** 0.7).astype(np.float32)
    for lut in (None, B):
        got = ctx.abs_squared_grad(0, 3, B=lut)
        _, want = oracle.make_images(img, w, h, B=lut)
        for l in range(3):
            assert got[l].tobytes() == want[l].tobytes(), (l, lut is not None)
            assert not got[l][0].any() and not got[l][-1].any()
    ctx.close()


def test_pyramid_batch_from_device(setup, pkg):
    """Batched makeImages from device-resident raw images == per-frame upload path, bit for bit."""
    import torch
    ctx, case = setup["ctx"], setup["case"]
    imgs = np.stack([f["img"] for f in case["frames"]])
    raw = torch.from_numpy(imgs).cuda()
    torch.cuda.synchronize()
    ctx.frames_from_device_batch([5, 6, 7], raw.data_ptr(), imgs.shape[1] * imgs.shape[2] * 4)
    ctx.synchronize()
    for k in range(3):
        for lvl in range(ctx.levels):
            a = ctx.frame_download(5 + k, lvl); b = ctx.frame_download(1 + k, lvl)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_frames_attached_in_place(setup, pkg):
    """Zero-copy attach: level 0 of the slot is the caller's resident image, the coarser levels are built — same pyramids, same tracking
    results as the copying path, bit for bit; a later upload into the slot returns it to its own storage."""
    import torch
    ctx, trk, case = setup["ctx"], setup["trk"], setup["case"]
    imgs = np.stack([f["img"] for f in case["frames"]])
    raw = torch.from_numpy(imgs).cuda()
    torch.cuda.synchronize()
    ref = trk.track_batch([1, 2, 3], [IDENT] * 3, [(0.0, 0.0)] * 3)
    ctx.frames_attach_device_batch([5, 6, 7], raw.data_ptr(), imgs.shape[1] * imgs.shape[2] * 4)
    ctx.synchronize()
    for k in range(3):
        for lvl in range(ctx.levels):
            a = ctx.frame_download(5 + k, lvl); b = ctx.frame_download(1 + k, lvl)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    got = trk.track_batch([5, 6, 7], [IDENT] * 3, [(0.0, 0.0)] * 3)
    for key in ("pose7", "aff", "lastResiduals", "flow", "H", "b", "good", "iterations"):
        assert np.array_equal(got[key], ref[key], equal_nan=True), key
    # the slot really references the caller's memory ...
    raw[0].mul_(0.5); torch.cuda.synchronize()
    assert np.array_equal(ctx.frame_download(5, 0)[:, :, 0], imgs[0] * np.float32(0.5))
    # ... until it is rebuilt
    ctx.frame_upload(5, imgs[1])
    assert np.array_equal(ctx.frame_download(5, 0).view(np.uint32), ctx.frame_download(2, 0).view(np.uint32))


def test_set_ref_bit_exact(setup):
    """makeCoarseDepthL0: same number of template points per level, same order, same bits."""
    trk, T = setup["trk"], setup["T"]
    for lvl in range(setup["ctx"].levels):
        assert trk.pc_n(lvl) == T.pc_n(lvl)
        g = trk.get_pc(lvl)
        o = T.get_pc(lvl)
        for a, b, name in zip(g, o, "u v idepth color".split()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "pc_%s level %d" % (name, lvl)
        # the dense maps next to the template (CoarseTracker::idepth / weightSums, what debugPlotIDepthMap reads): every pixel, borders and invalid ones included
        for a, b, name in zip(trk.get_idepth_map(lvl), T.get_idepth(lvl), ("idepth", "weightSums")):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "dense %s level %d" % (name, lvl)


def test_set_ref_many_points_on_one_pixel_follow_the_reference_order(pkg, oracle, synth, gpu_required):
    """Three and more points that round to the same pixel: makeCoarseDepthL0 adds them in index order (CoarseTracker.cpp:151-165); float sums of three
    terms do not commute, so the template must come out identical to the sequential CPU loop run after run."""
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=600, seed=77, n_frames=1)
    rng = np.random.RandomState(4)
    u, v, idp, hd = [np.array(case[k], dtype=np.float32) for k in ("u", "v", "idepth", "hdiF")]
    # pile groups of 3..7 points onto single pixels, with idepths / weights spread over several orders of magnitude
    pos = 0
    for g in range(40):
        m = 3 + g % 5
        sel = np.arange(pos, pos + m); pos += m
        u[sel] = u[sel[0]] + rng.uniform(-0.4, 0.4, m).astype(np.float32); v[sel] = v[sel[0]] + rng.uniform(-0.4, 0.4, m).astype(np.float32)
        u[sel] = np.round(u[sel[0]]) + rng.uniform(-0.45, 0.45, m).astype(np.float32); v[sel] = np.round(v[sel[0]]) + rng.uniform(-0.45, 0.45, m).astype(np.float32)
        idp[sel] = (10.0 ** rng.uniform(-2, 0.5, m)).astype(np.float32); hd[sel] = (10.0 ** rng.uniform(-6, -1, m)).astype(np.float32)
    perm = rng.permutation(len(u))          # the groups' members far apart in index
    u, v, idp, hd = u[perm], v[perm], idp[perm], hd[perm]
    ctx = pkg.Context(w, h, n_slots=1)
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"])
    T = oracle.Tracker(w, h); T.make_k(case["K4"])
    T.set_ref(oracle.make_images(case["ref_img"], w, h)[0], u, v, idp, hd)
    for rep in range(3):
        trk.setCoarseTrackingRef(0, u, v, idp, hd)
        for lvl in range(ctx.levels):
            assert trk.pc_n(lvl) == T.pc_n(lvl)
            for a, b, name in zip(trk.get_pc(lvl), T.get_pc(lvl), "u v idepth color".split()):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "pc_%s level %d, repetition %d" % (name, lvl, rep)


def _cmp_eval(rs_g, H_g, b_g, rs_o, H_o, b_o, tol=2e-5):
    assert rs_g[1] == rs_o[1], "numTermsInE must match exactly"
    assert abs(rs_g[5] - rs_o[5]) < 1e-7, "saturated ratio"
    assert abs(rs_g[0] - rs_o[0]) <= tol * abs(rs_o[0]) + 1e-6
    for k in (2, 4):
        assert abs(rs_g[k] - rs_o[k]) <= 1e-4 * abs(rs_o[k]) + 1e-9
    scale = np.sqrt(np.outer(np.diag(H_o), np.diag(H_o))) + 1e-30
    assert np.max(np.abs(H_g - H_o) / scale) < tol
    assert np.max(np.abs(b_g - b_o) / (np.sqrt(np.diag(H_o)) * np.sqrt(rs_o[0] / max(rs_o[1], 1)) + 1e-30)) < 10 * tol


@pytest.mark.parametrize("lvl", [3, 2, 1, 0])
def test_eval_parity(setup, lvl):
    """One calcRes+calcGSSSE evaluation at identity, at the true pose and at a perturbed pose / affine."""
    trk, T, case = setup["trk"], setup["T"], setup["case"]
    T.set_new(setup["dIn"][0])
    poses = [IDENT, case["frames"][0]["pose7"], case["frames"][1]["pose7"]]
    affs = [(0.0, 0.0), (0.0, 0.0), (0.02, -3.0)]
    for pose, aff in zip(poses, affs):
        for cutoff in (20.0, 40.0):
            rs_o = T.calc_res(lvl, pose, aff, cutoff)
            H_o, b_o = T.calc_gs(lvl, aff)
            rs_g, H_g, b_g = trk.eval(lvl, 1, pose, aff, cutoff)
            _cmp_eval(rs_g, H_g, b_g, rs_o, H_o, b_o)


def test_eval_deterministic(setup):
    """Fixed summation order: two evaluations give the same bits."""
    trk, case = setup["trk"], setup["case"]
    a = trk.eval(0, 1, case["frames"][0]["pose7"], (0.0, 0.0))
    b = trk.eval(0, 1, case["frames"][0]["pose7"], (0.0, 0.0))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_eval_identity_frame_zero_residual(setup):
    """Identical frame, identity pose: r ~ 0, b ~ 0, E ~ 0 (known answer up to the fp32 rounding of K*K^-1*x)."""
    trk, T = setup["trk"], setup["T"]
    rs, H, b = trk.eval(0, 0, IDENT, (0.0, 0.0))
    assert rs[0] / rs[1] < 1e-5 and rs[5] == 0.0 and rs[1] > 9000
    assert np.max(np.abs(b) / np.sqrt(np.diag(H))) < 1e-2
    assert np.all(np.linalg.eigvalsh(H) > -1e-9 * np.max(np.abs(H)))
    T.set_new(setup["dIr"])
    rs_o = T.calc_res(0, IDENT, (0.0, 0.0), 20.0)
    assert rs_o[1] == rs[1] and abs(rs_o[0] - rs[0]) < 1e-4


def _cmp_track(g, o):
    assert g["good"] == o["good"]
    if not o["good"] and not np.all(np.isfinite(o["lastResiduals"][:1])):
        return
    dt = np.linalg.norm(g["pose7"][:3] - o["pose7"][:3])
    dq = min(np.linalg.norm(g["pose7"][3:] - o["pose7"][3:]), np.linalg.norm(g["pose7"][3:] + o["pose7"][3:]))
    assert dt < 1e-3, "translation differs by %g m" % dt
    assert dq < 1e-3
    for lvl in range(4):
        eo, eg = o["lastResiduals"][lvl] ** 2, g["lastResiduals"][lvl] ** 2
        if np.isfinite(eo):
            assert abs(eg - eo) <= 1e-4 * eo, "level %d energy rel diff %g" % (lvl, abs(eg - eo) / eo)
    assert np.allclose(g["aff"], o["aff"], rtol=1e-3, atol=1e-3)


def test_track_parity(setup):
    """Full 4-level trackNewestCoarse (device-resident LM) vs the oracle, from the identity guess."""
    trk, T, case = setup["trk"], setup["T"], setup["case"]
    for k in range(len(case["frames"])):
        T.set_new(setup["dIn"][k])
        o = T.track(IDENT, (0.0, 0.0))
        g = trk.trackNewestCoarse(1 + k, IDENT, (0.0, 0.0))
        _cmp_track(g, o)
        # and it actually converged to the ground truth
        assert np.linalg.norm(g["pose7"][:3] - case["frames"][k]["pose7"][:3]) < 2e-3
        assert g["iterations"] == o["iterations"]


def test_track_batch_hypotheses(setup, synth):
    """B pose hypotheses in one launch (FullSystem::trackNewCoarse's try list) == B sequential oracle runs."""
    trk, T, case = setup["trk"], setup["T"], setup["case"]
    rng = np.random.RandomState(7)
    B = 12
    poses = []
    for i in range(B):
        xi = case["frames"][0]["xi"] * rng.uniform(0.0, 1.6) + rng.normal(0, 0.004, 6)
        R, t = synth.se3_exp(xi)
        poses.append(synth.pose7(R, t))
    slots = [1 + (i % 3) for i in range(B)]
    affs = [(0.0, 0.0)] * B
    g = trk.track_batch(slots, poses, affs)
    for i in range(B):
        T.set_new(setup["dIn"][slots[i] - 1])
        o = T.track(poses[i], affs[i])
        gi = dict(good=bool(g["good"][i]), pose7=g["pose7"][i], aff=g["aff"][i], lastResiduals=g["lastResiduals"][i])
        _cmp_track(gi, o)


def test_track_abort_and_failure_semantics(setup):
    """minResForAbort: a level RMSE above 1.5x the threshold aborts, returns false and leaves pose/aff untouched."""
    trk, T = setup["trk"], setup["T"]
    T.set_new(setup["dIn"][0])
    mr = np.array([0.1, 0.1, 0.1, 0.1, np.nan])
    o = T.track(IDENT, (0.0, 0.0), min_res=mr)
    g = trk.trackNewestCoarse(1, IDENT, (0.0, 0.0), minResForAbort=mr)
    assert o["good"] is False and g["good"] is False
    assert np.array_equal(g["pose7"], IDENT)
    assert np.isnan(g["lastResiduals"][0]) and np.isfinite(g["lastResiduals"][3])
    assert abs(g["lastResiduals"][3] - o["lastResiduals"][3]) < 1e-4 * o["lastResiduals"][3]


def test_affine_brightness_recovered(pkg, oracle, synth, gpu_required):
    """New frame rendered with an affine brightness change: a,b are estimated like the oracle does."""
    w = h = 512
    case = synth.tracking_case(w, h, n_ref=1500, aff_new=(0.04, 4.0))
    ctx = pkg.Context(w, h, n_slots=2)
    trk = pkg.CoarseTrackerHip(ctx)
    trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, case["frames"][0]["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    T = oracle.Tracker(w, h); T.make_k(case["K4"])
    T.set_ref(oracle.make_images(case["ref_img"], w, h)[0], case["u"], case["v"], case["idepth"], case["hdiF"])
    T.set_new(oracle.make_images(case["frames"][0]["img"], w, h)[0])
    o = T.track(IDENT, (0.0, 0.0)); g = trk.trackNewestCoarse(1, IDENT, (0.0, 0.0))
    _cmp_track(g, o)
    assert abs(g["aff"][0] - 0.04) < 0.02 and abs(g["aff"][1] - 4.0) < 3.0


def test_track_new_coarse_try_loop(setup, oracle, pkg):
    """FullSystem::trackNewCoarse (FullSystem.cpp:419-489): hypothesis list + sequential try loop with achievedRes thresholds.
    The library runs try 0 alone and the rest as one batch, replaying the abort rule; winner, number of tries, achieved
    residuals and the winning pose must match the sequential oracle."""
    case, trk, T = setup["case"], setup["trk"], setup["T"]
    f = case["frames"][0]
    T.set_new(setup["dIn"][0])
    true = f["pose7"]
    # (1) good constant-motion prediction: first try wins and ends the loop
    lastF = IDENT.copy()
    slast = oracle.se3_exp(-0.5 * f["xi"])      # camToWorld of a frame half-way: try 0 = exp(xi/2)^2
    sprelast = IDENT.copy()
    tries_o = oracle.make_track_hypotheses(slast, sprelast, lastF)
    tries_g = pkg.make_track_hypotheses(slast, sprelast, lastF)
    assert np.max(np.abs(tries_o - tries_g)) < 1e-14
    o = T.track_new_coarse(tries_o, lastCoarseRMSE=np.full(5, 100.0))
    g = trk.trackNewCoarse(1, tries_g, lastCoarseRMSE=np.full(5, 100.0))
    assert o["winner"] == g["winner"] == 0 and o["tries_used"] == g["tries_used"] == 1 and g["good"] and o["good"]
    assert np.max(np.abs(g["pose7"] - o["pose7"])) < 1e-5
    assert np.allclose(g["achievedRes"], o["achievedRes"], rtol=1e-4, equal_nan=True)
    assert np.max(np.abs(g["pose7"][:3] - true[:3])) < 1e-3
    # (2) tight re-track threshold forces the loop through every hypothesis: winner selection + threshold replay
    rm = np.full(5, 1e-3)
    o = T.track_new_coarse(tries_o, lastCoarseRMSE=rm.copy())
    g = trk.trackNewCoarse(1, tries_g, lastCoarseRMSE=rm.copy())
    assert o["tries_used"] == g["tries_used"] == 31
    # many hypotheses converge to the same minimum and differ in the last bits of lastResiduals[0] (fp32 summation order), so the
    # winner INDEX is not a stable quantity here — the winning pose and the achieved residuals are
    assert g["winner"] >= 0 and o["winner"] >= 0
    assert np.max(np.abs(g["pose7"] - o["pose7"])) < 1e-5
    assert np.allclose(g["achievedRes"], o["achievedRes"], rtol=1e-4, equal_nan=True)
    assert np.allclose(g["flow"], o["flow"], rtol=1e-4, atol=1e-5)
    # (3) bad motion model (large wrong rotation as "constant motion"): first tries fail or are worse, a later one wins
    bad = oracle.se3_exp(np.array([0.0, 0, 0, 0.0, 0.35, 0.0]))
    tries_bad = np.concatenate([bad[None], oracle.se3_mul(bad, bad)[None], tries_o])
    o = T.track_new_coarse(tries_bad, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    g = trk.trackNewCoarse(1, tries_bad, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    assert o["winner"] == g["winner"] and o["tries_used"] == g["tries_used"] and o["good"] == g["good"]
    assert np.max(np.abs(g["pose7"] - o["pose7"])) < 1e-5
    assert np.allclose(g["achievedRes"], o["achievedRes"], rtol=1e-4, equal_nan=True)


def test_shared_reciprocal_division_is_ieee_division(pkg, gpu_required):
    """The evaluation loop divides by the projected depth through one refined reciprocal (LLVM's fdiv expansion without range scaling and
    fix-up): in the operand range of the path the quotient has the bits of the IEEE division — of the device's own and of the host's."""
    rng = np.random.RandomState(12)
    n = 1 << 20
    ctx = pkg.Context(64, 64, n_slots=1)
    cases = [
        (rng.uniform(-4.0, 4.0, n), rng.uniform(0.05, 8.0, n)),                     # pt0 / pt2, pt1 / pt2
        (rng.uniform(1e-3, 10.0, n), rng.uniform(0.05, 8.0, n)),                    # id / pt2
        (np.full(n, 9.0), rng.uniform(9.0, 255.0, n)),                              # huberTH / |residual|
        (rng.normal(size=n) * 10.0 ** rng.uniform(-12, 12, n), rng.normal(size=n) * 10.0 ** rng.uniform(-12, 12, n)),   # wide, still unscaled
    ]
    for a, b in cases:
        a = a.astype(np.float32); b = b.astype(np.float32)
        qs, qi = ctx.selftest_divide(a, b)
        host = a / b
        assert np.array_equal(qi.view(np.uint32), host.view(np.uint32))             # the device's IEEE division == the host's
        assert np.array_equal(qs.view(np.uint32), qi.view(np.uint32))               # the shared-reciprocal form == IEEE division


def test_eval_parity_random_poses_and_brightness(setup, oracle):
    """Sixty random evaluations: poses up to a few centimetres / degrees away (many points leave the image), affine brightness, exposure ratios,
    cut-off thresholds, every level and frame — the same term counts as the oracle and the same sums to the tolerance of test_eval_parity."""
    trk, T, case = setup["trk"], setup["T"], setup["case"]
    rng = np.random.RandomState(77)
    for trial in range(60):
        k = int(rng.randint(0, 3)); lvl = int(rng.randint(0, 4))
        xi = rng.normal(0, 1, 6) * np.array([0.05, 0.05, 0.05, 0.03, 0.03, 0.03]) * rng.choice([0.2, 1.0, 3.0])
        pose = oracle.se3_exp(xi)
        aff = (float(rng.normal(0, 0.05)), float(rng.normal(0, 5.0)))
        cutoff = float(rng.choice([5.0, 20.0, 80.0]))
        T.set_new(setup["dIn"][k])
        rs_o = T.calc_res(lvl, pose, aff, cutoff)
        H_o, b_o = T.calc_gs(lvl, aff)
        rs_g, H_g, b_g = trk.eval(lvl, 1 + k, pose, aff, cutoff)
        assert rs_g[1] == rs_o[1], (trial, "numTermsInE")
        if rs_o[1] < 8:
            continue                                   # next to nothing left in the image: counts compared, sums too small to scale
        _cmp_eval(rs_g, H_g, b_g, rs_o, H_o, b_o, tol=5e-5)


def test_track_new_coarse_hypotheses_split_over_ranks(pkg, synth, oracle, gpu_required):
    """SURVEY 8e, the one split coarse tracking has: the tries of FullSystem::trackNewCoarse over ranks (dmvio_hip_tracker_set_comm*).  Two ranks emulated by two threads of
    this process, each with a context and tracker of its own on the device, the all-reduce of the per-try records done between the threads: both ranks return the same bits,
    and the unsplit call's answer within the rounding of a different cluster size (batches of 15 instead of 30 problems group their partial sums differently)."""
    import threading
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=600, n_frames=1)
    f = case["frames"][0]
    slast = oracle.se3_exp(-0.5 * f["xi"]); ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    bad = oracle.se3_exp(np.array([0.0, 0, 0, 0.0, 0.35, 0.0]))
    tries = np.concatenate([bad[None], oracle.se3_mul(bad, bad)[None], pkg.make_track_hypotheses(slast, ident, ident)])     # a wrong motion model first: later tries decide

    def make():
        ctx = pkg.Context(w, h, n_slots=2)
        trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
        ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, f["img"])
        trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
        return ctx, trk
    ctx0, trk0 = make()
    single = [trk0.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2), trk0.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 1e-3))]
    world = 2
    bar = threading.Barrier(world); bufs = [None] * world; calls = [0] * world

    def allreduce_for(rank):
        def allreduce(a):
            calls[rank] += 1
            bufs[rank] = a.copy(); bar.wait()
            tot = bufs[0].copy()
            for r in range(1, world):
                tot = tot + bufs[r]
            bar.wait()
            a[:] = tot
        return allreduce
    out = [None] * world; err = []

    def rank_main(rank):
        try:
            ctx, trk = make()
            trk.set_comm_allreduce(allreduce_for(rank), rank, world)
            out[rank] = [trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2), trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 1e-3))]
        except Exception as e:      # a dead rank would leave the other at the barrier
            err.append(e); bar.abort()
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(300) for t in th]
    assert not err, err
    assert calls == [2, 2]                                            # one exchange per call (the loop went past try 0 both times)
    for k in range(2):
        a, b = out[0][k], out[1][k]
        for key in ("pose7", "aff", "achievedRes", "flow"):
            assert np.array_equal(np.asarray(a[key]).view(np.uint64), np.asarray(b[key]).view(np.uint64)), key
        assert a["winner"] == b["winner"] and a["tries_used"] == b["tries_used"] and a["good"] == b["good"]
        s = single[k]
        assert a["tries_used"] == s["tries_used"] and a["good"] == s["good"]
        assert np.max(np.abs(a["pose7"] - s["pose7"])) < 1e-5 and np.allclose(a["achievedRes"], s["achievedRes"], rtol=1e-4, equal_nan=True)
    assert out[0][0]["winner"] == single[0]["winner"] >= 2             # well-conditioned: a later try wins


def test_track_new_coarse_exchange_over_rccl_on_one_device(pkg, synth, oracle, gpu_required):
    """The RCCL transport of the hypothesis split (dmvio_hip_tracker_set_comm) on a one-device box: a communicator of ONE rank, the split path forced by the library's test hook
    (every try is this rank's, the all-reduce over one rank is the identity): the records go through the pinned staging area, the device buffer and ncclAllReduce on the
    context's stream and come back — the answer must be the unsplit call's, within the rounding of a different cluster size (one batch of 32 instead of 1 + 32)."""
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=600, n_frames=1)
    f = case["frames"][0]
    slast = oracle.se3_exp(-0.5 * f["xi"]); ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    bad = oracle.se3_exp(np.array([0.0, 0, 0, 0.0, 0.35, 0.0]))
    tries = np.concatenate([bad[None], oracle.se3_mul(bad, bad)[None], pkg.make_track_hypotheses(slast, ident, ident)])
    ctx = pkg.Context(w, h, n_slots=2)
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, f["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    single = trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    comm = pkg.RcclCommunicator(ctx, pkg.RcclCommunicator.unique_id(ctx.L), 0, 1)
    assert comm.info() == (1, 0)
    trk.set_comm(comm, 0, 1)                         # world 1 without the hook: no exchange is installed, the call is the unsplit one bit for bit
    plain = trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    assert all(np.array_equal(np.asarray(plain[k]), np.asarray(single[k]), equal_nan=True) for k in ("pose7", "aff", "achievedRes", "flow"))
    trk.debug_split_single_rank(True)                # the library's explicit test hook (an environment variable cannot switch it on)
    trk.set_comm(comm, 0, 1)
    a = trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    b = trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    trk.set_comm(None, 0, 0)
    for key in ("pose7", "aff", "achievedRes", "flow"):
        assert np.array_equal(np.asarray(a[key]).view(np.uint64), np.asarray(b[key]).view(np.uint64)), key          # run to run: the same bits
    assert a["winner"] == single["winner"] >= 2 and a["tries_used"] == single["tries_used"] and a["good"] == single["good"]
    assert np.max(np.abs(a["pose7"] - single["pose7"])) < 1e-5 and np.allclose(a["achievedRes"], single["achievedRes"], rtol=1e-4, equal_nan=True)
