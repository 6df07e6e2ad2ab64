"""The device-resident Gauss-Newton loop of the sliding-window BA and its batched form (round 5: csrc/ba_batch_kernels.hpp, dmvio_hip_ba_optimize_batch): FullSystem::optimize
(FullSystemOptimize.cpp:417-647) with the 68x68 solve (EnergyFunctional.cpp:841-996), doStepFromBackup, FrameFramePrecalc, E_L / E_M and the accept test ON THE DEVICE, W windows
per launch sequence.  Checked against (i) the oracle (= the reference's arithmetic, same bars as test_ba_gpu.py::test_optimize_parity), (ii) the library's host-driven loop
(the same window, same kernels for the photometric part: the two loops differ in the elementary functions of the frame step and, by default, in the association of the back
substitution), (iii) itself: a window's result must not depend on what else is in the batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx_with(pkg, case, n_slots=None):
    F = case["n_frames"]
    ctx = pkg.Context(case["w"], case["h"], n_slots=n_slots or F)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])
    return ctx


def _poses(ba, F):
    return np.stack([np.concatenate(ba.frame_pose(k)[:2]) for k in range(F)])


@pytest.mark.parametrize("accumulators", [1, None])
def test_device_loop_against_oracle_and_host_loop(pkg, oracle, synth, gpu_required, accumulators):
    """One window (BASELINE config 3's shape: 8 keyframes, 2000 points, ~12.8k residuals) through dmvio_hip_ba_optimize with the device-resident loop: the oracle's accept
    sequence, energies within 1e-4, poses within 1e-3 m (north_star); against the host-driven loop of the same library the energy trace within 1e-9 relative and the solve's x of
    the last iteration within 1e-9 of its norm (exact back substitution: within 1e-11)."""
    case = synth.ba_case(512, 512, n_frames=8, n_points=2000, seed=4321)
    ctx = _ctx_with(pkg, case)
    F = 8
    W = oracle.BAWindow(case)
    ro = W.optimize(6)
    host = pkg.BundleAdjusterHip(ctx, accumulators=accumulators); host.set_case(case, list(range(F)))
    rh = host.optimize(6)
    dev = pkg.BundleAdjusterHip(ctx, accumulators=accumulators); dev.set_case(case, list(range(F))); dev.set_device_loop(True)
    rd = dev.optimize(6)
    assert rd["iterations"] == ro["iterations"] == 6
    assert np.array_equal(rd["trace"][:, 3], ro["trace"][:, 3]), (rd["trace"], ro["trace"])
    assert np.allclose(rd["trace"][:, 0], ro["trace"][:, 0], rtol=1e-4)
    assert abs(rd["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"] and abs(rd["rmse"] - ro["rmse"]) <= 1e-4 * ro["rmse"]
    for k in range(F):
        pg, ag, _ = dev.frame_pose(k); po, ao, _ = W.frame_pose(k)
        assert np.linalg.norm(pg[:3] - po[:3]) < 1e-3 and np.allclose(ag, ao, atol=1e-3)
    # device loop vs host loop
    assert np.array_equal(rd["trace"][:, 3], rh["trace"][:, 3])
    dE = np.abs(rd["trace"][:, :3] - rh["trace"][:, :3]) / np.maximum(np.abs(rh["trace"][:, :3]), 1e-30)
    xd, xh = dev.last_x(), host.last_x()
    dx = np.abs(xd - xh).max() / np.abs(xh).max()
    dp = np.abs(_poses(dev, F) - _poses(host, F)).max()
    ig, _ = dev.point_state(); ih, _ = host.point_state()
    print("device loop vs host loop (accumulators %s): trace within %.2e relative, last x within %.2e of its largest entry, poses / affine within %.2e, idepth within %.2e; "
          "final energy %.9g vs %.9g (oracle %.9g)" % (accumulators, dE.max(), dx, dp, np.abs(ig - ih).max(), rd["finalEnergy"], rh["finalEnergy"], ro["finalEnergy"]))
    assert dE.max() < 1e-7 and dx < 1e-6 and dp < 1e-8
    assert abs(rd["finalEnergy"] - rh["finalEnergy"]) <= 1e-7 * rh["finalEnergy"]
    # the exact back substitution (the host's order)
    ex = pkg.BundleAdjusterHip(ctx, accumulators=accumulators); ex.set_case(case, list(range(F)))
    B = pkg.BundleAdjusterBatch(ctx, 1); B.set_exact_backsub(True)
    re_ = B.optimize([ex], 6)[0]
    dxe = np.abs(ex.last_x() - xh).max() / np.abs(xh).max()
    print("exact back substitution: last x within %.2e; default association within %.2e" % (dxe, dx))
    assert np.array_equal(re_["trace"][:, 3], rh["trace"][:, 3]) and dxe < 1e-7
    for o in (host, dev, ex, B):
        o.close()
    ctx.close()


def test_first_solve_of_the_device_loop_equals_the_host_solve_bitwise(pkg, synth, gpu_required):
    """One iteration from the same state: the device's x (k_ba_solve: assembly, Jacobi scaling, pivot order, LDL^T, forward substitution, exact back substitution) against the
    host's (BAHost::solveSystem + ldltSolveTransposed) — the same system, no frame step in between, so every operation is specified: bit for bit.  The default back
    substitution (column-oriented) is bounded against it."""
    case = synth.ba_case(320, 256, n_frames=6, n_points=500, hosts_share=(120, 110, 100, 90, 80, 0), seed=11)
    ctx = _ctx_with(pkg, case)
    F = 6
    host = pkg.BundleAdjusterHip(ctx, accumulators=1); host.set_case(case, list(range(F)))
    host.optimize(1); xh = host.last_x()
    ex = pkg.BundleAdjusterHip(ctx, accumulators=1); ex.set_case(case, list(range(F)))
    B = pkg.BundleAdjusterBatch(ctx, 1); B.set_exact_backsub(True)
    B.optimize([ex], 1); xe = ex.last_x()
    fa = pkg.BundleAdjusterHip(ctx, accumulators=1); fa.set_case(case, list(range(F)))
    B.set_exact_backsub(False)
    B.optimize([fa], 1); xf = fa.last_x()
    print("first solve: exact back substitution differs from the host's x in %d of %d entries (max %.2e); column-oriented: max %.2e relative to the largest entry"
          % (int((xe != xh).sum()), len(xh), np.abs(xe - xh).max(), np.abs(xf - xh).max() / np.abs(xh).max()))
    assert np.array_equal(xe, xh)
    assert np.abs(xf - xh).max() <= 1e-12 * np.abs(xh).max()
    for o in (host, ex, fa, B):
        o.close()
    ctx.close()


def test_batched_windows_equal_single_window_calls_bitwise(pkg, synth, gpu_required):
    """W = 5 windows in one dmvio_hip_ba_optimize_batch call — different sizes, one of them converged (its steps get rejected: the gated restore path), two keyframe counts
    (grouped by the library) — against the same windows optimised one at a time: traces, poses, affine parameters, inverse depths bit for bit."""
    cases = [synth.ba_case(320, 256, n_frames=6, n_points=500, hosts_share=(120, 110, 100, 90, 80, 0), seed=11),
             synth.ba_case(320, 256, n_frames=6, n_points=300, hosts_share=(80, 70, 60, 50, 40, 0), seed=12),
             synth.ba_case(320, 256, n_frames=4, n_points=150, hosts_share=(60, 50, 40, 0), seed=7),
             synth.ba_case(320, 256, n_frames=6, n_points=400, hosts_share=(100, 90, 80, 70, 60, 0), seed=13),
             synth.ba_case(320, 256, n_frames=4, n_points=200, hosts_share=(80, 70, 50, 0), seed=8)]
    ctx = pkg.Context(320, 256, n_slots=32)
    slots = []
    nxt = 0
    for cs in cases:
        sl = list(range(nxt, nxt + cs["n_frames"])); nxt += cs["n_frames"]
        for k, s in enumerate(sl):
            ctx.frame_upload(s, cs["imgs"][k])
        slots.append(sl)

    def fresh(i, pre=False):
        ba = pkg.BundleAdjusterHip(ctx); ba.set_case(cases[i], slots[i])
        if pre:
            ba.set_device_loop(True); ba.optimize(12)      # window 3 enters the comparison converged: most of its steps are rejected
        return ba
    B1 = pkg.BundleAdjusterBatch(ctx, 1); B5 = pkg.BundleAdjusterBatch(ctx, 8)
    single = [fresh(i, pre=(i == 3)) for i in range(5)]
    rs = [B1.optimize([b], 6)[0] for b in single]
    batch = [fresh(i, pre=(i == 3)) for i in range(5)]
    rb = B5.optimize(batch, 6)
    n_rej = 0
    for i in range(5):
        F = cases[i]["n_frames"]
        assert rs[i]["iterations"] == rb[i]["iterations"]
        assert np.array_equal(rs[i]["trace"], rb[i]["trace"]), i
        assert rs[i]["finalEnergy"] == rb[i]["finalEnergy"] and rs[i]["rmse"] == rb[i]["rmse"], i
        assert np.array_equal(_poses(single[i], F), _poses(batch[i], F)), i
        assert np.array_equal(single[i].point_state()[0], batch[i].point_state()[0]), i
        n_rej += int((rb[i]["trace"][1:, 3] == 0).sum())
    assert n_rej >= 1, "no rejected step in the batch: the gated restore path was not exercised"
    # diagnostics of the call: how every window's last solve found Eigen's pivot order (0 = ranks of a distinct diagonal, 1 = the parallel tie replay — the usual case in a real
    # window: the prior-dominated diagonal entries tie — never 2, the NaN rule; the three branches themselves are pinned by tests/test_ba_solve_gpu.py), the in-kernel stamps of
    # a solve and the host clock at the call's phase boundaries
    branches = [B5.last_pivot_branch(i) for i in range(3)]           # (the last group of the call: the three windows of six keyframes)
    assert all(b in (0, 1) for b in branches), branches
    ticks = B5.last_solve_ticks()
    assert len(ticks) >= 12 and ticks[0] > 0 and 0 < ticks[5] <= ticks[10] < 1000, ticks       # microseconds since the kernel began: a solve takes < 1 ms
    hu = B5.last_host_us()
    assert len(hu) == 8 and all(hu[k] <= hu[k + 1] for k in range(7)) and 0 < hu[7] < 1e6, hu
    print("batch of 5 windows (F = 6, 6, 4, 6, 4) == 5 single calls bit for bit; %d rejected steps among them; loop %.3f ms, final linearisation %.3f ms" % ((n_rej,) + B5.last_ms()[:2]))
    for o in single + batch + [B1, B5]:
        o.close()
    ctx.close()


def test_groups_streams_and_the_one_lane_linearisation_change_no_bit(pkg, synth, gpu_required):
    """Eight windows of one keyframe count (so that they form ONE group of the call: from 4 windows on the library cuts it into up to three stream groups with interleaved
    launches and linearises with k_ba_linearize_b1, one lane per residual) — three different graphs, one window entering converged (rejected steps: the gated restore
    path of the one-lane kernel): every configuration of streams x linearisation kernel gives each window the bits of its single-window call (eight-lane kernel, one stream)."""
    cases = [synth.ba_case(320, 256, n_frames=6, n_points=500, hosts_share=(120, 110, 100, 90, 80, 0), seed=11),
             synth.ba_case(320, 256, n_frames=6, n_points=300, hosts_share=(80, 70, 60, 50, 40, 0), seed=12),
             synth.ba_case(320, 256, n_frames=6, n_points=400, hosts_share=(100, 90, 80, 70, 60, 0), seed=13)]
    which = [0, 1, 2, 0, 1, 2, 2, 1]
    converged = 3
    ctx = pkg.Context(320, 256, n_slots=18)
    slots = []
    for c, cs in enumerate(cases):
        sl = list(range(6 * c, 6 * c + 6))
        for k, s in enumerate(sl):
            ctx.frame_upload(s, cs["imgs"][k])
        slots.append(sl)

    def fresh(w):
        ba = pkg.BundleAdjusterHip(ctx); ba.set_case(cases[which[w]], slots[which[w]])
        if w == converged:
            ba.set_device_loop(True); ba.optimize(12)
        return ba

    def state(ba):
        return _poses(ba, 6), ba.point_state()[0]
    B1 = pkg.BundleAdjusterBatch(ctx, 1); B8 = pkg.BundleAdjusterBatch(ctx, 8)
    ref = []
    for w in range(8):
        b = fresh(w); r = B1.optimize([b], 6)[0]; ref.append((r, state(b))); b.close()
    assert sum(int((r["trace"][1:, 3] == 0).sum()) for r, _ in ref) >= 1, "no rejected step: the gated restore path was not exercised"
    for lanes, streams in ((8, 1), (1, 1), (1, 0), (1, 2), (8, 3), (1, 4)):
        B8.set_linearize_lanes(lanes); B8.set_streams(streams)
        hs = [fresh(w) for w in range(8)]
        rb = B8.optimize(hs, 6)
        for w in range(8):
            r0, (p0, d0) = ref[w]
            assert np.array_equal(r0["trace"], rb[w]["trace"]), (lanes, streams, w)
            assert r0["finalEnergy"] == rb[w]["finalEnergy"] and r0["rmse"] == rb[w]["rmse"], (lanes, streams, w)
            p1, d1 = state(hs[w])
            assert np.array_equal(p0, p1) and np.array_equal(d0, d1), (lanes, streams, w)
        for h in hs:
            h.close()
    B1.close(); B8.close(); ctx.close()


def test_device_loop_with_a_marginalisation_prior_and_fej_deltas(pkg, synth, gpu_required):
    """The terms a fresh window leaves at zero: a marginalisation prior (H_M / b_M: bM_top = b_M + H_M delta in every solve, E_M = delta . (2 b_M + H_M delta) in every accept
    test, EnergyFunctional.cpp:322-345, 864-907) with keyframe states off their linearisation points (state != state_zero: non-zero delta from the first iteration on).  The
    device-resident loop (k_ba_solve) against the host-driven loop of the same library — same kernels for the photometric part —: same accept sequence, E_M trace within 1e-9
    relative, states within 1e-8; a batch of four such windows equals its single-window calls bit for bit."""
    case = synth.ba_case(320, 256, n_frames=6, n_points=500, hosts_share=(120, 110, 100, 90, 80, 0), seed=11)
    F = 6
    n = 4 + 8 * F
    ctx = _ctx_with(pkg, case)
    rng = np.random.RandomState(7)
    A = rng.standard_normal((n, n)) * 30.0
    HM = A @ A.T; bM = rng.standard_normal(n) * 10.0
    deltas = []
    for k in range(F):
        st = np.zeros(10)
        if k > 0:
            st[:3] = 1e-3 * rng.standard_normal(3); st[3:6] = 5e-4 * rng.standard_normal(3); st[6] = 1e-3 * rng.standard_normal(); st[7] = 1e-4 * rng.standard_normal()
        deltas.append(st)

    def window():
        ba = pkg.BundleAdjusterHip(ctx, accumulators=1); ba.set_case(case, list(range(F)))
        ba.set_marg_prior(HM, bM)
        for k in range(F):
            ba.set_frame_state(k, deltas[k])
        return ba
    host = window(); rh = host.optimize(6)
    dev = window(); dev.set_device_loop(True); rd = dev.optimize(6)
    assert rd["iterations"] == rh["iterations"] and np.array_equal(rd["trace"][:, 3], rh["trace"][:, 3])
    assert np.abs(rh["trace"][:, 2]).min() > 0.1 and np.abs(rh["trace"][:, 2]).max() > 10.0, "E_M is not exercised"
    dE = np.abs(rd["trace"][:, :3] - rh["trace"][:, :3]) / np.maximum(np.abs(rh["trace"][:, :3]), 1e-30)
    dp = np.abs(_poses(dev, F) - _poses(host, F)).max()
    dx = np.abs(dev.last_x() - host.last_x()).max() / np.abs(host.last_x()).max()
    print("prior + FEJ deltas, device vs host loop: E_A / E_L / E_M within %.2e / %.2e / %.2e relative, last x within %.2e, states within %.2e; %d of %d steps accepted"
          % (dE[:, 0].max(), dE[:, 1].max(), dE[:, 2].max(), dx, dp, int(rh["trace"][1:, 3].sum()), rh["iterations"]))
    assert dE.max() < 1e-7 and dx < 1e-6 and dp < 1e-8
    assert abs(rd["finalEnergy"] - rh["finalEnergy"]) <= 1e-7 * rh["finalEnergy"]
    # four such windows (different priors) in one call == four single calls, bit for bit
    B1 = pkg.BundleAdjusterBatch(ctx, 1); B4 = pkg.BundleAdjusterBatch(ctx, 4)

    def scaled(f):
        ba = window(); ba.set_marg_prior(f * HM, f * bM); return ba
    fs = (1.0, 0.5, 2.0, 0.1)
    single = [scaled(f) for f in fs]; rs = [B1.optimize([b], 6)[0] for b in single]
    batch = [scaled(f) for f in fs]; rb = B4.optimize(batch, 6)
    for i in range(4):
        assert np.array_equal(rs[i]["trace"], rb[i]["trace"]) and rs[i]["finalEnergy"] == rb[i]["finalEnergy"], i
        assert np.array_equal(_poses(single[i], F), _poses(batch[i], F)) and np.array_equal(single[i].point_state()[0], batch[i].point_state()[0]), i
    for o in [host, dev, B1, B4] + single + batch:
        o.close()
    ctx.close()


def test_jacobians_of_an_earlier_host_loop_are_not_taken_for_the_batched_loops(pkg, synth, gpu_required):
    """The batched / device-resident loop relinearises and applies every residual WITHOUT writing the 74-float Jacobians (dmvio_hip_ba_keep_jacobians): after it the buffer still
    holds what an earlier host-driven optimize left there, for another state.  dmvio_hip_ba_fix_linearization (EFResidual::fixLinearizationF needs the Jacobian of the APPLIED
    linearisation) must refuse then instead of freezing stale rows — and serve again once a host-driven linearisation + apply has refreshed them."""
    case = synth.ba_case(256, 192, n_frames=5, n_points=300, hosts_share=(90, 80, 70, 60, 0), seed=5)
    F = case["n_frames"]; R = len(case["res_point"])
    ctx = _ctx_with(pkg, case)
    mask = (np.arange(R) % 3 == 0).astype(np.uint8)
    ba = pkg.BundleAdjusterHip(ctx, accumulators=1, keep_jacobians=True)
    ba.set_case(case, list(range(F)))
    ba.optimize(2)                                            # host-driven: the Jacobians of its last applied linearisation are resident
    B = pkg.BundleAdjusterBatch(ctx, 1)
    B.optimize([ba], 2)                                       # the batched loop moves the window on; the buffer is not rewritten
    with pytest.raises(pkg.HipLibraryError, match="not resident"):
        ba.fix_linearization(mask)
    ba.set_device_loop(True)
    ba.optimize(1)                                            # the same loop behind dmvio_hip_ba_optimize
    with pytest.raises(pkg.HipLibraryError, match="not resident"):
        ba.fix_linearization(mask)
    ba.set_device_loop(False)
    ba.activate_all(); ba.linearize_all(False); ba.apply_res()   # host-driven linearisation + applyRes: resident again
    assert ba.fix_linearization(mask) > 0
    ba.close(); B.close()
