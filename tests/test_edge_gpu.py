"""Edge cases through the C ABI: empty inputs, degenerate windows, out-of-range arguments — the library must answer with the
reference's conventions (tracking failed / nothing to do) or a clean error, never crash."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IDENT = np.array([0, 0, 0, 0, 0, 0, 1.0])


def test_tracker_empty_reference_and_bad_slots(pkg, oracle, synth, gpu_required):
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=300)
    ctx = pkg.Context(w, h, n_slots=3)
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, case["frames"][0]["img"])
    # no reference points at all: every level is empty -> NaN residuals, tracking reported as failed, outputs finite or untouched
    e = np.zeros(0, np.float32)
    trk.setCoarseTrackingRef(0, e, e, e, e)
    assert all(trk.pc_n(l) == 0 for l in range(ctx.levels))
    r = trk.trackNewestCoarse(1, IDENT, [0.0, 0.0])
    T = oracle.Tracker(w, h); T.make_k(case["K4"])
    dIr, _ = oracle.make_images(case["ref_img"], w, h); dIn, _ = oracle.make_images(case["frames"][0]["img"], w, h)
    T.set_ref(dIr, e, e, e, e); T.set_new(dIn)
    o = T.track(IDENT, [0.0, 0.0])
    assert bool(r["good"]) == bool(o["good"]) == False
    # a single reference point, far outside after the warp
    one = lambda x: np.array([x], np.float32)
    trk.setCoarseTrackingRef(0, one(5), one(5), one(0.5), one(1e-3))
    far = oracle.se3_exp(np.array([5.0, 0, 0, 0, 0, 0]))
    r = trk.trackNewestCoarse(1, far, [0.0, 0.0])
    assert not r["good"]
    with pytest.raises(pkg.HipLibraryError):
        ctx.frame_upload(7, case["ref_img"])
    with pytest.raises(pkg.HipLibraryError):
        trk.trackNewestCoarse(9, IDENT, [0.0, 0.0])


def test_ba_degenerate_windows(pkg, oracle, synth, gpu_required):
    # two keyframes (the reference forces 20 iterations), a host without points, a point with a single residual
    case = synth.ba_case(256, 192, n_frames=2, n_points=60, hosts_share=(60, 0), seed=41)
    ctx = pkg.Context(256, 192, n_slots=2)
    for k in range(2):
        ctx.frame_upload(k, case["imgs"][k])
    ba = pkg.BundleAdjusterHip(ctx); ba.set_case(case, [0, 1])
    W = oracle.BAWindow(case)
    rg = ba.optimize(6); ro = W.optimize(6)
    assert rg["iterations"] == ro["iterations"]
    assert abs(rg["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * max(ro["finalEnergy"], 1e-9)
    # marginalising with no candidates is a no-op
    dec, H, b, n = ba.marginalize_points(np.zeros(ba.N, np.uint8))
    assert n == 0 and not dec.any() and not H.any() and not b.any()
    with pytest.raises(pkg.HipLibraryError):
        ba.set_frame_state(5, np.zeros(10))


def test_immature_empty_and_initializer_empty(pkg, oracle, synth, gpu_required):
    w = h = 128
    ctx = pkg.Context(w, h, n_slots=2)
    img = synth.PlaneWorld(3).render(synth.default_intrinsics(w, h), np.eye(3), np.zeros(3), w, h)[0]
    ctx.frame_upload(0, img); ctx.frame_upload(1, img)
    imm = pkg.ImmaturePointsHip(ctx, capacity=16)
    assert imm.n == 0
    counts = imm.traceNewCoarse(1, IDENT, IDENT[None], synth.default_intrinsics(w, h))      # nothing to trace
    assert sum(counts.values()) == 0
    res, idp, rs = imm.optimize([0, 1], np.stack([IDENT, IDENT]), synth.default_intrinsics(w, h))
    assert len(res) == 0
    with pytest.raises(pkg.HipLibraryError):
        imm.add_points(0, 0, np.arange(40) % 100 + 10, np.arange(40) % 100 + 10)                # capacity exceeded
    imm.add_points(3, 0, np.array([20, 30]), np.array([20, 30]))
    with pytest.raises(pkg.HipLibraryError):
        imm.traceNewCoarse(1, IDENT, IDENT[None], synth.default_intrinsics(w, h))               # host_tag 3 without a table row
    with pytest.raises(pkg.HipLibraryError):
        imm.optimize([0, 1], np.stack([IDENT, IDENT]), synth.default_intrinsics(w, h))           # host_tag 3 >= F
    ini = pkg.CoarseInitializerHip(ctx, capacity=8)
    ini.set_points(dict(u=np.zeros(0), v=np.zeros(0), iR=np.zeros(0), isGood=np.zeros(0, np.uint8), energy=np.zeros((0, 2)), outlierTH=np.zeros(0)))
    K4 = synth.default_intrinsics(w, h)
    Ki = np.linalg.inv(np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1.0]]))
    g = ini.calcResAndGS(0, 0, 1, Ki, K4, IDENT, (0.0, 0.0), np.zeros(0, np.float32))
    assert not g["H"][3:, 3:].any() and g["res3"][0] == 0 and g["res3"][2] == 0


def test_non_finite_pixels_propagate_like_the_reference(pkg, oracle, synth, gpu_required):
    """NaN / Inf irradiance in the new frame (SURVEY §5 failure semantics): the pyramid carries them to the coarser levels exactly as
    makeImages does, points whose interpolated colour is not finite are skipped (CoarseTracker.cpp:455), gradients that are not finite
    count as zero (HessianBlocks.cpp:172-181) — same term counts, same sums, same alignment as the oracle."""
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=600, n_frames=1)
    img = case["frames"][0]["img"].copy()
    img[100:140, 60:110] = np.nan
    img[30, 200] = np.inf; img[31, 201] = -np.inf; img[200:203, 17] = np.nan
    ctx = pkg.Context(w, h, n_slots=2)
    ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, img)
    dIr, _ = oracle.make_images(case["ref_img"], w, h); dIn, _ = oracle.make_images(img, w, h)
    for lvl in range(ctx.levels):
        assert np.array_equal(ctx.frame_download(1, lvl), dIn[lvl], equal_nan=True)
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    T = oracle.Tracker(w, h); T.make_k(case["K4"]); T.set_ref(dIr, case["u"], case["v"], case["idepth"], case["hdiF"]); T.set_new(dIn)
    for lvl in range(ctx.levels):
        rs_o = T.calc_res(lvl, case["frames"][0]["pose7"], (0.0, 0.0), 20.0)
        H_o, b_o = T.calc_gs(lvl, (0.0, 0.0))
        rs_g, H_g, b_g = trk.eval(lvl, 1, case["frames"][0]["pose7"], (0.0, 0.0), 20.0)
        assert rs_g[1] == rs_o[1] and rs_o[1] < trk.pc_n(lvl)                      # the same points dropped
        assert np.isfinite(rs_g[0]) and abs(rs_g[0] - rs_o[0]) <= 2e-5 * abs(rs_o[0]) + 1e-6
        assert np.all(np.isfinite(H_g)) and np.allclose(H_g, H_o, rtol=1e-4, atol=1e-6 * np.abs(H_o).max())
    g = trk.trackNewestCoarse(1, IDENT, [0.0, 0.0]); o = T.track(IDENT, [0.0, 0.0])
    assert bool(g["good"]) == bool(o["good"])
    assert np.linalg.norm(np.asarray(g["pose7"])[:3] - np.asarray(o["pose7"])[:3]) < 1e-3


def test_fetch_begin_keeps_consecutive_batches_apart(pkg, oracle, synth, gpu_required):
    """The two alternating halves of problem / result memory: batch B is staged and launched before the results of batch A are fetched."""
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=500, n_frames=6, xi_jitter=0.4)
    ctx = pkg.Context(w, h, n_slots=7)
    ctx.frame_upload(0, case["ref_img"])
    for k, f in enumerate(case["frames"]):
        ctx.frame_upload(1 + k, f["img"])
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    A, Bb = [1, 2, 3], [4, 5, 6]
    aff = [(0.0, 0.0)] * 3
    trk.stage(A, [IDENT] * 3, aff); trk.launch(); refA = trk.fetch()
    trk.stage(Bb, [IDENT] * 3, aff); trk.launch(); refB = trk.fetch()
    assert not np.allclose(refA["pose7"], refB["pose7"])
    for _ in range(3):   # several rounds: the halves keep alternating
        trk.stage(A, [IDENT] * 3, aff); trk.launch(); trk.fetch_begin()
        trk.stage(Bb, [IDENT] * 3, aff); trk.launch()
        rA = trk.fetch(); rB = trk.fetch()
        for k in ("pose7", "aff", "lastResiduals", "flow", "H", "b", "good", "iterations"):
            assert np.array_equal(rA[k], refA[k], equal_nan=True) and np.array_equal(rB[k], refB[k], equal_nan=True)
    with pytest.raises(pkg.HipLibraryError):   # a larger batch cannot be staged while results are pending
        trk.stage(A, [IDENT] * 3, aff); trk.launch(); trk.fetch_begin()
        trk.stage(list(range(1, 7)) * 20, [IDENT] * 120, [(0.0, 0.0)] * 120)


def test_second_batch_behind_an_unfetched_one_is_refused(pkg, synth, gpu_required):
    """fetch_begin keeps the results of launch k in one pinned half while launch k+1 runs into the other; staging launch k+2 before fetching k would
    overwrite them: the library refuses instead."""
    case = synth.tracking_case(256, 256, n_ref=400, seed=3, n_frames=2)
    ctx = pkg.Context(256, 256, n_slots=3)
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, case["frames"][0]["img"]); ctx.frame_upload(2, case["frames"][1]["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    ident = np.array([[0, 0, 0, 0, 0, 0, 1.0]]); aff = np.zeros((1, 2))
    trk.stage([1], ident, aff); trk.launch()
    trk.fetch_begin()
    trk.stage([2], ident, aff); trk.launch()
    with pytest.raises(pkg.HipLibraryError):
        trk.stage([1], ident, aff)
    a = trk.fetch()                      # results of the first launch, intact
    b = trk.fetch()                      # then those of the second
    ra = trk.track_batch([1], ident, aff); rb = trk.track_batch([2], ident, aff)
    # (a, b: the device-resident LM through stage / launch; ra, rb: single problems run the host LM against the evaluation server — same evaluation sums, the pose
    # agrees to the last bit or two of its fp64 components)
    assert np.abs(a["pose7"] - ra["pose7"]).max() < 1e-14 and np.abs(b["pose7"] - rb["pose7"]).max() < 1e-14
    assert np.array_equal(a["lastResiduals"], ra["lastResiduals"], equal_nan=True) and np.array_equal(b["H"], rb["H"])


def test_guarded_and_guard_free_paths_agree_bit_for_bit(pkg, synth, gpu_required):
    """A frame whose pixels are all finite is stamped clean by its pyramid build and tracked without the isfinite guards; withdrawing the
    stamp selects the guarded loop — same bits in every output."""
    w = h = 256
    case = synth.tracking_case(w, h, n_ref=800, n_frames=3, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=4)
    ctx.frame_upload(0, case["ref_img"])
    for k, f in enumerate(case["frames"]):
        ctx.frame_upload(1 + k, f["img"])
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    slots = [1, 2, 3]
    a = trk.track_batch(slots, [IDENT] * 3, [(0.0, 0.0)] * 3)
    for s in slots:
        ctx.frame_mark_unclean(s)
    b = trk.track_batch(slots, [IDENT] * 3, [(0.0, 0.0)] * 3)
    assert a["good"].all()
    for k in ("pose7", "aff", "lastResiduals", "flow", "H", "b", "good", "iterations"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    ctx.frame_upload(1, case["frames"][0]["img"])          # a rebuild stamps the slot clean again
    c = trk.track_batch([1], [IDENT], [(0.0, 0.0)])          # a single problem: host LM + evaluation server, same sums, pose to the last bit or two
    assert np.abs(c["pose7"][0] - a["pose7"][0]).max() < 1e-14 and np.array_equal(c["H"][0], a["H"][0]) and c["iterations"][0] == a["iterations"][0]


def test_plain_c_demo_recovers_the_known_pose(pkg, gpu_required, tmp_path):
    """examples/c_abi_demo.c run on the device: plain C, no Python in the loop, a camera translation in front of a textured wall recovered
    through dmvio_hip_frame_upload / tracker_make_k / tracker_set_ref / tracker_track."""
    import subprocess
    from test_capi_cpu import _build_c_demo
    exe = _build_c_demo(pkg, tmp_path / "c_abi_demo")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: translation error" in r.stdout


def test_cpp_adapter_demo_tracks_like_the_reference_surface(pkg, gpu_required, tmp_path):
    """examples/cpp_adapter_demo.cpp: the tracking-thread calls through the C++ mirror (makeImages, makeK, setCoarseTrackingRef,
    trackNewestCoarse returning trackingIsGood, lastResiduals / lastFlowIndicators members); a bad slot reads as tracking failed; then the
    mapping side: a three-keyframe window through WindowOptimizer::optimize (FullSystem::optimize) brings the energy down."""
    import subprocess
    from test_capi_cpu import _build_cpp_demo
    exe = _build_cpp_demo(pkg, tmp_path / "cpp_adapter_demo")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: translation error" in r.stdout and "bad slot -> false" in r.stdout and "ok: energy reduced" in r.stdout


def test_host_arrays_larger_than_the_pinned_staging_area(pkg, oracle, gpu_required):
    """Caller-owned arrays cross PCIe through pinned memory the library owns (csrc/internal.h: DmvBounce, 4 MB to begin with).  A 1280x1024 frame is 5 MB up and 15.7 MB down
    (level 0 as Vec3f): the staging area has to drain and grow in the middle of a call, several downloads of one call have to land in their own arrays, and the same handle
    has to keep working afterwards.  Checked bit for bit against the oracle's makeImages."""
    w, h = 1280, 1024
    rng = np.random.RandomState(4)
    img = (rng.rand(h, w) * 255).astype(np.float32)
    ctx = pkg.Context(w, h, n_slots=2)
    for rep in range(2):
        src = img.copy()
        ctx.frame_upload(rep, src)
        src[:] = -1.0                                           # the call has returned: the caller's array is its own again
    ref, ref_abs = oracle.make_images(img, w, h)
    for lvl in (0, 2):
        for slot in (0, 1):
            got = ctx.frame_download(slot, lvl)
            assert np.array_equal(got.view(np.uint32), ref[lvl].view(np.uint32)), (slot, lvl)
    g = ctx.abs_squared_grad(1, n_levels=3)                     # three arrays out of one call
    for lvl in range(3):
        inner = (slice(1, -1), slice(1, -1))                    # makeImages writes rows 1 .. h-2 (HessianBlocks.cpp:169-189)
        assert np.array_equal(g[lvl][inner].view(np.uint32), ref_abs[lvl][inner].view(np.uint32)), lvl


def test_graft_entry_smoke(gpu_required):
    """__graft_entry__.smoke(): what the driver runs on the GPU box before the bench — one small tracking problem and one small window against the oracle."""
    import __graft_entry__ as graft
    graft.smoke()
