"""GPU parity tests of the sliding-window bundle adjustment: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Per-residual linearisation is compared bit for bit (states, energies, the full RawResidualJacobian); the block accumulators
replay the reference's sequential fp32 order, so the stitched systems agree to double rounding; the full optimize()
must meet the north-star tolerances: final energy within 1e-4 relative, poses within 1e-3 m.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _window(pkg, oracle, case, n_extra_slots=0, accumulators=1):
    """accumulators=1: the reference's single-threaded accumulation order, which these tests compare bit for bit / to double rounding;
    the library's default (4 partial accumulators per bucket) is covered by test_default_accumulation_order_*"""
    F = case["n_frames"]
    ctx = pkg.Context(case["w"], case["h"], n_slots=F + n_extra_slots)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])
    ba = pkg.BundleAdjusterHip(ctx, accumulators=accumulators, keep_jacobians=True)
    ba.set_case(case, list(range(F)))
    W = oracle.BAWindow(case)
    return ctx, ba, W


@pytest.fixture(scope="module")
def big(pkg, oracle, synth, gpu_required):
    case = synth.ba_case(512, 512, n_frames=8, n_points=2000)
    ctx, ba, W = _window(pkg, oracle, case)
    return dict(case=case, ctx=ctx, ba=ba, W=W)


def test_linearize_bit_exact(big):
    """PointFrameResidual::linearize: same state, same energies, same Jacobians for every residual."""
    ba, W = big["ba"], big["W"]
    ba.activate_all(); W.activate_all()
    e_g = ba.linearize_all(False); e_o = W.linearize_all(False)
    sg, so = ba.res_state(), W.res_state()
    assert np.array_equal(sg["newState"], so["newState"].astype(np.uint8))
    assert (so["newState"] == 0).sum() > 0.25 * W.R
    assert np.array_equal(sg["newEnergy"], so["newEnergy"].astype(np.float32))
    assert np.array_equal(sg["newEnergyWO"], so["newEnergyWO"].astype(np.float32))
    ok = so["newState"] != 1
    assert np.array_equal(sg["center"][ok], so["center"][ok])
    assert abs(e_g - e_o) <= 1e-9 * e_o
    assert np.array_equal(ba.frame_energy_th(), W.frame_energy_th())
    J = ba.jacobians()
    rng = np.random.RandomState(0)
    for ri in rng.choice(np.nonzero(so["newState"] == 0)[0], 300, replace=False):
        Jo = W.get_J(int(ri), 0)
        flat = np.concatenate([Jo[k].reshape(-1) for k in ("resF", "Jpdxi", "Jpdc", "Jpdd", "JIdx", "JabF", "JIdx2", "JabJIdx", "Jab2")])
        assert np.array_equal(J[ri].view(np.uint32), flat.view(np.uint32)), "residual %d" % ri


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_linearize_and_system_random_windows(pkg, oracle, synth, gpu_required, seed):
    """Random windows: 3..8 keyframes, affine brightness and exposure times that differ per frame, 640x480 / 256x256 images — per-residual
    linearisation bit-identical, the stitched system and one optimisation step like the oracle's."""
    rng = np.random.RandomState(seed)
    F = int(rng.randint(3, 9))
    w, h = (640, 480) if seed % 2 else (256, 256)
    case = synth.ba_case(w, h, n_frames=F, n_points=int(rng.randint(200, 500)), seed=seed)
    case["aff"] = np.column_stack([rng.normal(0, 0.02, F), rng.normal(0, 2.0, F)])
    case["exposure"] = rng.uniform(0.7, 1.4, F).astype(np.float32)
    ctx, ba, W = _window(pkg, oracle, case)
    ba.activate_all(); W.activate_all()
    e_g = ba.linearize_all(False); e_o = W.linearize_all(False)
    sg, so = ba.res_state(), W.res_state()
    assert np.array_equal(sg["newState"], so["newState"].astype(np.uint8))
    assert np.array_equal(sg["newEnergy"], so["newEnergy"].astype(np.float32))
    assert np.array_equal(sg["newEnergyWO"], so["newEnergyWO"].astype(np.float32))
    assert abs(e_g - e_o) <= 1e-9 * abs(e_o)
    ba.apply_res(); W.apply_res()
    ag, ao = ba.accumulate(), W.accumulate()
    assert ag["resInA"] == ao["resInA"]
    for k in ("HA", "bA", "Hsc", "bsc"):
        scale = np.abs(ao[k]).max() + 1e-30
        assert np.max(np.abs(ag[k] - ao[k])) <= 1e-9 * scale, k
    EL, EM = ba.energy_terms()            # the affine priors pull a, b towards zero: with non-zero brightness parameters E_L starts above zero
    assert abs(EL - W.lenergy()) <= 1e-9 * abs(W.lenergy()) + 1e-12 and EL > 0
    lam, lE = 1e-5, [e_g, EL, EM]
    lamo, lEo = 1e-5, [e_o, W.lenergy(), W.menergy()]
    for it in range(3):
        acc, lam, lE = ba.gn_iteration(it, lam, lE)
        acco, lamo, lEo = W.gn_iteration(it, lamo, lEo)
        assert bool(acc) == bool(acco) and abs(lE[0] - lEo[0]) <= 1e-4 * abs(lEo[0])


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_accumulate_and_solve_parity(big, pkg, oracle, mode, monkeypatch):
    """accumulateAF / accumulateSCF + adjoint stitching, solveSystemF, resubstitute.
    exact (dmvio_hip_ba_set_accumulators(1)): one accumulator per bucket replays the single-threaded reference order (incl. 1k/1M shift-up)
    -> systems agree to double rounding.  fast (the library's default, 4 partial accumulators per bucket, summed in double like the
    reference's six per-worker accumulators) -> agreement at fp32 summation level."""
    case = big["case"]
    ba = pkg.BundleAdjusterHip(big["ctx"], accumulators=1 if mode == "exact" else None)
    ba.set_case(case, list(range(case["n_frames"])))
    W = oracle.BAWindow(case)
    tolH, tolx = (1e-11, 1e-6) if mode == "exact" else (2e-6, 1e-2)
    ba.activate_all(); W.activate_all()
    ba.linearize_all(False); W.linearize_all(False)
    ba.apply_res(); W.apply_res()
    ag, ao = ba.accumulate(), W.accumulate()
    assert ag["resInA"] == ao["resInA"] and ag["resInA"] > 3000
    pg, po = ba.point_acc(), W.point_acc()
    for k in ("Hdd", "bd", "Hcd", "HdiF", "bdSumF"):
        assert np.array_equal(pg[k], po[k]), k
    sc = np.sqrt(np.outer(np.diag(ao["HA"]) + 1e-12, np.diag(ao["HA"]) + 1e-12))
    assert np.max(np.abs(ag["HA"] - ao["HA"]) / sc) < tolH
    assert np.max(np.abs(ag["Hsc"] - ao["Hsc"]) / sc) < tolH
    bs = np.sqrt(np.diag(ao["HA"]) + 1e-12) * np.sqrt(ao["HA"].shape[0])
    assert np.max(np.abs(ag["bA"] - ao["bA"]) / (np.abs(ao["bA"]) + bs)) < max(tolH, 1e-10) * 1e3
    assert np.max(np.abs(ag["bsc"] - ao["bsc"]) / (np.abs(ao["bsc"]) + bs)) < max(tolH, 1e-10) * 1e3
    for it, lam in ((0, 1e-5), (2, 1e-3)):
        xg = ba.solve(it, lam); xo = W.solve(it, lam)
        assert np.linalg.norm(xg - xo) <= tolx * np.linalg.norm(xo)
        assert np.allclose(xg, xo, rtol=1e3 * tolx, atol=tolx * np.abs(xo).max())
        _, sg = ba.point_state(); _, so = W.point_state()
        assert np.linalg.norm(sg - so) <= 100 * tolx * np.linalg.norm(so)
    ba.close()


def test_optimize_parity(pkg, oracle, synth, gpu_required):
    """FullSystem::optimize, 8-keyframe window, 2000 points: energy trace, final energy (1e-4 rel) and poses (1e-3 m) vs the oracle."""
    case = synth.ba_case(512, 512, n_frames=8, n_points=2000, seed=4321)
    ctx, ba, W = _window(pkg, oracle, case)
    rg = ba.optimize(6); ro = W.optimize(6)
    assert rg["iterations"] == ro["iterations"] == 6
    assert np.array_equal(rg["trace"][:, 3], ro["trace"][:, 3]), "same accept / reject decisions"
    assert np.allclose(rg["trace"][:, 0], ro["trace"][:, 0], rtol=1e-4)
    assert abs(rg["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"]
    assert abs(rg["rmse"] - ro["rmse"]) <= 1e-4 * ro["rmse"]
    for k in range(8):
        pg, ag, _ = ba.frame_pose(k); po, ao, _ = W.frame_pose(k)
        assert np.linalg.norm(pg[:3] - po[:3]) < 1e-3
        assert min(np.linalg.norm(pg[3:] - po[3:]), np.linalg.norm(pg[3:] + po[3:])) < 1e-4
        assert np.allclose(ag, ao, atol=1e-3)
    ig, _ = ba.point_state(); io, _ = W.point_state()
    seen = io != 0   # the plane world leaves a few rays without a depth: those points have no residual and keep idepth 0 on both sides
    assert np.array_equal(ig[~seen], io[~seen]) and np.median(np.abs(ig[seen] - io[seen]) / io[seen]) < 1e-4
    # and the optimisation did its job: poses closer to the ground truth than the initial guess
    e0 = np.mean([np.linalg.norm(np.asarray(case["poses0"][k][:3]) - case["poses_true"][k][:3]) for k in range(1, 8)])
    e1 = np.mean([np.linalg.norm(ba.frame_pose(k)[0][:3] - case["poses_true"][k][:3]) for k in range(1, 8)])
    assert e1 < e0


def test_resident_graph_through_a_keyframe_cycle(pkg, synth, gpu_required):
    """dmvio_hip_ba_set_graph_from: a window built from the resident graph (dmvio_hip_graph_*: the mutators of EnergyFunctional.cpp:435-518, 641-646, 766-782) optimises to
    the same bits as the same window handed over as flat arrays — at first, and again after the graph went through what a keyframe does to it: residuals dropped, points
    removed (their host's last point takes their index), the oldest keyframe marginalised (indices move down, its observations dropped), a new keyframe with residuals from
    the old points, new points with new residuals; the inverse depths of the optimisation carried over with dmvio_hip_graph_set_idepths."""
    F = 6
    case = synth.ba_case(320, 256, n_frames=F, n_points=600, hosts_share=(150, 130, 120, 110, 90, 0), seed=31)
    ctx = pkg.Context(320, 256, n_slots=F)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])
    g = pkg.WindowGraph.from_case(case)

    def both(cs, slots, frames):
        """optimize(4) of window `cs` through flat arrays and through the graph: traces, poses and points must be identical bit for bit"""
        out = []
        for how in ("flat", "graph"):
            ba = pkg.BundleAdjusterHip(ctx, accumulators=1)
            if how == "flat":
                ba.set_case(cs, slots)
            else:
                ba.set_window(slots, cs["poses0"], np.zeros((frames, 2)), np.ones(frames, np.float32), np.arange(frames, dtype=np.int32), cs["K4"])
                ba.set_graph_from(g)
            r = ba.optimize(4)
            out.append((r["trace"].copy(), np.stack([ba.frame_pose(k)[0] for k in range(frames)]), ba.point_state()[0].copy()))
            ba.close()
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
        return out[1]

    _, _, idepth = both(case, list(range(F)), F)
    g.set_idepths(idepth)
    # ---- a keyframe: the mirror mutated call by call, a Python list model of the same statements beside it
    rng = np.random.RandomState(5)
    o = g.export()
    pts = [[dict(u=o["u"][p], v=o["v"][p], idepth=o["idepth"][p], color=o["color"][p], weights=o["weights"][p], prior=bool(o["hasDepthPrior"][p]),
                 res=list(o["res_target"][o["res_point"] == p])) for p in range(len(o["u"])) if o["host"][p] == f] for f in range(F)]
    for f in range(F):                                          # FullSystem::optimize's tail: inactive residuals dropped
        for i, p in enumerate(pts[f]):
            if p["res"] and rng.rand() < 0.1:
                k = rng.randint(len(p["res"])); g.drop_residual(f, i, k); p["res"][k] = p["res"][-1]; p["res"].pop()
    for f in range(F):                                          # removeOutliers / flagPointsForRemoval: points without residuals, all points of the frame to be marginalised
        i = 0
        while i < len(pts[f]):
            if f == 0 or not pts[f][i]["res"] or rng.rand() < 0.05:
                g.remove_point(f, i); pts[f][i] = pts[f][-1]; pts[f].pop()
            else:
                i += 1
    g.remove_frame(0); del pts[0]                               # ef->marginalizeFrame, then FullSystem::marginalizeFrame drops what still targets the frame
    for f in range(F - 1):
        for i, p in enumerate(pts[f]):
            p["res"] = [t - 1 for t in p["res"]]               # target 0 -> -1 (dangling), the others move down
            while -1 in p["res"]:
                k = p["res"].index(-1); g.drop_residual(f, i, k); p["res"][k] = p["res"][-1]; p["res"].pop()
    new = g.insert_frame(); pts.append([]); assert new == F - 1  # the new keyframe: a residual from every old point, then freshly activated points
    for f in range(F - 1):
        for i, p in enumerate(pts[f]):
            assert g.insert_residual(f, i, new) == len(p["res"]); p["res"].append(new)
    for _ in range(80):
        h = rng.randint(0, F - 1)
        rec = dict(u=np.float32(rng.uniform(20, 300)), v=np.float32(rng.uniform(20, 236)), idepth=np.float32(rng.uniform(0.15, 0.5)), color=(rng.rand(8) * 200 + 20).astype(np.float32),
                   weights=(rng.rand(8) * 0.5 + 0.5).astype(np.float32), prior=False, res=[])
        i = g.insert_point(h, rec["u"], rec["v"], rec["idepth"], rec["color"], rec["weights"]); pts[h].append(rec); assert i == len(pts[h]) - 1
        for t in range(F):
            if t != h and rng.rand() < 0.7:
                g.insert_residual(h, i, t); rec["res"].append(t)
    # the same window as flat arrays, from the list model
    host, rp, rt = [], [], []
    flat = dict(u=[], v=[], idepth=[], color=[], weights=[], prior=[])
    for f in range(F):
        for p in pts[f]:
            for t in p["res"]:
                rp.append(len(host)); rt.append(t)
            host.append(f)
            for key in flat:
                flat[key].append(p[key])
    cs = dict(case)
    cs.update(host=np.array(host, np.int32), u=np.array(flat["u"], np.float32), v=np.array(flat["v"], np.float32), idepth0=np.array(flat["idepth"], np.float32),
              color=np.stack(flat["color"]), weights=np.stack(flat["weights"]), hasDepthPrior=np.array(flat["prior"], np.uint8), res_point=np.array(rp, np.int32),
              res_target=np.array(rt, np.int32), poses0=np.asarray(case["poses0"])[[1, 2, 3, 4, 5, 0]], aff=None, exposure=None, frameIDs=None)
    assert g.counts() == (F, len(host), len(rp))
    both(cs, [1, 2, 3, 4, 5, 0], F)                             # keyframe 0's slot serves as the "new" image: any image will do for bit equality of the two paths


def test_small_window_and_ragged_graph(pkg, oracle, synth, gpu_required):
    """3-keyframe window (forces 15 iterations), hosts without points, points with a single residual."""
    case = synth.ba_case(320, 256, n_frames=3, n_points=150, hosts_share=(90, 60, 0), seed=99)
    ctx, ba, W = _window(pkg, oracle, case)
    rg = ba.optimize(6); ro = W.optimize(6)
    assert rg["iterations"] == ro["iterations"] == 15
    assert abs(rg["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"]
    for k in range(3):
        assert np.linalg.norm(ba.frame_pose(k)[0][:3] - W.frame_pose(k)[0][:3]) < 1e-3


def test_shift_up_boundaries_inside_buckets(pkg, oracle, synth, gpu_required):
    """> 1000 members per (host,target) and per (host,t1,t2) bucket: the accumulators' 1000-entry shiftUp (MatrixAccumulators.h
    numIn1 / Data1k / Data1m) happens INSIDE buckets, twice; exact mode must still reproduce the oracle's sums."""
    case = synth.ba_case(512, 512, n_frames=3, n_points=4200, hosts_share=(4000, 200, 0), seed=5)
    ctx, ba, W = _window(pkg, oracle, case)
    ba.activate_all(); eg = ba.linearize_all(False); ba.apply_res()
    W.activate_all(); eo = W.linearize_all(False); W.apply_res()
    assert abs(eg - eo) <= 1e-6 * eo
    a = ba.accumulate(); o = W.accumulate()
    assert a["resInA"] == o["resInA"] and a["resInA"] > 2200   # > 1000 active members in both buckets of host 0
    for k in ("HA", "bA", "Hsc", "bsc"):
        assert np.linalg.norm(a[k] - o[k]) <= 1e-11 * np.linalg.norm(o[k]), k


def test_sharded_driver_world1_equals_gn_iteration(pkg, oracle, synth, gpu_required):
    """dm-vio_amd/sharding.ShardedBA with a single rank == the library's own GN iteration (same accept sequence, same energies)."""
    import dmvio_amd.sharding as sh
    case = synth.ba_case(512, 512, n_frames=8, n_points=1000, seed=11)
    ctx, ba, W = _window(pkg, oracle, case)
    ba2 = pkg.BundleAdjusterHip(ctx, accumulators=1); ba2.set_case(case, list(range(8)))
    s = sh.ShardedBA(ba, sh.Collective(None, None))
    lastE = list(s.begin())
    ba2.activate_all(); e = ba2.linearize_all(False); ba2.apply_res()
    assert e == lastE[0]
    lam, lE = 1e-5, [e, 0.0, 0.0]
    for it in range(6):
        acc_s = s.iteration(it)
        acc_l, lam, lE = ba2.gn_iteration(it, lam, lE)
        assert acc_s == acc_l and abs(s.lastE[0] - lE[0]) <= 1e-9 * abs(lE[0]) and abs(s.lam - lam) < 1e-15
    for k in range(8):
        assert np.allclose(ba.frame_pose(k)[0], ba2.frame_pose(k)[0], atol=1e-12)


def test_gn_iterations_through_rejected_steps_match_oracle(pkg, oracle, synth, gpu_required):
    """Thirty GN iterations on one window: once converged, steps are rejected (restore + re-linearise; the library then skips the per-point
    sums of the next iteration because nothing they depend on has changed).  Accept sequence, damping and energies follow the oracle throughout,
    and calls in between (a query, a re-linearisation) do not disturb the sequence."""
    case = synth.ba_case(512, 512, n_frames=8, n_points=1000, seed=13)
    ctx, ba, W = _window(pkg, oracle, case)
    ba.activate_all(); e = ba.linearize_all(False); ba.apply_res()
    W.activate_all(); eo = W.linearize_all(False); W.apply_res()
    assert abs(e - eo) <= 1e-6 * abs(eo)
    lam, lE = 1e-5, [e, 0.0, 0.0]
    lamo, lEo = 1e-5, [eo, W.lenergy(), W.menergy()]
    rejected = 0
    for it in range(30):
        acc, lam, lE = ba.gn_iteration(it % 6, lam, lE)
        acco, lamo, lEo = W.gn_iteration(it % 6, lamo, lEo)
        assert bool(acc) == bool(acco), it
        assert lam == lamo and abs(lE[0] - lEo[0]) <= 1e-4 * abs(lEo[0]), it
        rejected += 0 if acc else 1
        if it % 7 == 3:
            ba.res_state()              # any other entry point in between: the skip must not survive it
    assert rejected >= 10
    for k in range(8):
        assert np.linalg.norm(ba.frame_pose(k)[0][:3] - W.frame_pose(k)[0][:3]) < 1e-3


@pytest.mark.parametrize("F", [5, 8, 10, 12])
def test_window_size_is_a_runtime_setting(pkg, oracle, synth, gpu_required, F):
    """setting_maxFrames is a run-time setting of the reference (util/settings.cpp:100, `maxFrames=` on the command line, util/MainSettings.cpp:223,246): windows of 5, 8, 10
    and 12 keyframes (n = 44 ... 100) on the same handle type — linearisation bit for bit, accumulated + stitched system to double rounding (single-threaded order), the whole
    optimize(6) like the oracle; F = dmvio_hip_ba_max_frames() + 1 is refused with an error."""
    share = tuple([400, 350, 300, 300, 250, 250, 200, 200, 150, 150, 100, 0][:F - 1] + [0]) if F > 2 else (1, 0)
    case = synth.ba_case(384, 320, n_frames=F, n_points=900, seed=60 + F, hosts_share=share, step_t=0.06, step_r=np.deg2rad(1.2))
    assert case["n_frames"] == F and len(set(case["host"].tolist())) == F - 1
    ctx, ba, W = _window(pkg, oracle, case)
    ba.activate_all(); W.activate_all()
    e_g = ba.linearize_all(False); e_o = W.linearize_all(False)
    sg, so = ba.res_state(), W.res_state()
    assert np.array_equal(sg["newState"], so["newState"].astype(np.uint8)) and np.array_equal(sg["newEnergy"], so["newEnergy"].astype(np.float32))
    assert np.array_equal(sg["newEnergyWO"], so["newEnergyWO"].astype(np.float32)) and np.array_equal(ba.frame_energy_th(), W.frame_energy_th())
    J = ba.jacobians()                                                                    # the 74-float RawResidualJacobian, bit for bit
    for ri in np.random.RandomState(F).choice(np.nonzero(so["newState"] == 0)[0], 100, replace=False):
        Jo = W.get_J(int(ri), 0)
        flat = np.concatenate([Jo[k].reshape(-1) for k in ("resF", "Jpdxi", "Jpdc", "Jpdd", "JIdx", "JabF", "JIdx2", "JabJIdx", "Jab2")])
        assert np.array_equal(J[ri].view(np.uint32), flat.view(np.uint32)), "residual %d" % ri
    assert abs(e_g - e_o) <= 1e-9 * abs(e_o)
    ba.apply_res(); W.apply_res()
    ag, ao = ba.accumulate(), W.accumulate()
    n = 4 + 8 * F
    assert ag["HA"].shape == (n, n) and ag["resInA"] == ao["resInA"]
    for k in ("HA", "bA", "Hsc", "bsc"):
        assert np.max(np.abs(ag[k] - ao[k])) <= 1e-9 * (np.abs(ao[k]).max() + 1e-30), k
    ba.set_case(case, list(range(F))); W = oracle.BAWindow(case)
    rg = ba.optimize(6); ro = W.optimize(6)
    assert rg["iterations"] == ro["iterations"] and np.array_equal(rg["trace"][:, 3], ro["trace"][:, 3])
    assert abs(rg["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"]
    for k in range(F):
        assert np.linalg.norm(ba.frame_pose(k)[0][:3] - W.frame_pose(k)[0][:3]) < 1e-3
    # the library's default accumulation order on the same window: same decisions, energy within the bar
    ba4 = pkg.BundleAdjusterHip(ctx); ba4.set_case(case, list(range(F)))
    r4 = ba4.optimize(6)
    assert np.array_equal(r4["trace"][:, 3], ro["trace"][:, 3]) and abs(r4["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"]
    # and the device-resident loop (k_ba_solve<8> up to 8 keyframes, <12> beyond: a 44 ... 100-dimensional solve on one workgroup) against the host-driven one
    bad = pkg.BundleAdjusterHip(ctx, accumulators=1); bad.set_case(case, list(range(F))); bad.set_device_loop(True)
    rd = bad.optimize(6)
    assert np.array_equal(rd["trace"][:, 3], rg["trace"][:, 3]) and np.allclose(rd["trace"][:, :3], rg["trace"][:, :3], rtol=1e-7, atol=1e-9)
    assert abs(rd["finalEnergy"] - rg["finalEnergy"]) <= 1e-7 * rg["finalEnergy"]
    for k in range(F):
        assert np.abs(np.concatenate(bad.frame_pose(k)[:2]) - np.concatenate(ba.frame_pose(k)[:2])).max() < 1e-8
    bad.close()
    if F == 12:
        L = pkg.load_library()
        assert L.dmvio_hip_ba_max_frames() == 12
        big = synth.ba_case(128, 96, n_frames=13, n_points=60, seed=3, hosts_share=tuple([5] * 12 + [0]), step_t=0.02, step_r=np.deg2rad(0.4))
        with pytest.raises(pkg.HipLibraryError, match="dmvio_hip_ba_max_frames"):
            ba4.set_case(big, [0] * 13)


def test_graph_arrives_with_linearised_residuals(pkg, synth, gpu_required):
    """A graph whose residuals ARRIVE linearised (EFResidual::isLinearized with EFResidual::J and ::res_toZeroF, EnergyFunctionalStructs.h:63-87) is served like one whose
    residuals were linearised on the resident graph (next test; that path is the one pinned to the oracle and libref.so):
      - flat hand-over (dmvio_hip_ba_set_linearized_residuals) onto a window standing in the same state: every accumulated system, the per-point sums and E_L bit for bit
        — the record formed from J is the record the linearisation wrote;
      - through the resident graph (dmvio_hip_graph_set_residual_linearized + dmvio_hip_ba_set_graph_from) onto a FRESH window: the optimisation that follows takes the
        same decisions and ends in the same state, bit for bit, on the host loop and on the device loop;
      - the records follow dropResidual / removePoint in the mirror;
      - a bare flag without the Jacobians is still refused (never optimised without the term)."""
    case = synth.ba_case(256, 192, n_frames=5, n_points=300, hosts_share=(90, 80, 70, 60, 0), seed=5)
    R = len(case["res_point"]); F = case["n_frames"]
    mask = (np.arange(R) % 3 == 0).astype(np.uint8)
    ctx = pkg.Context(case["w"], case["h"], n_slots=F)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])

    def perturb(ba, rng, scale):
        for k in range(1, F):
            st = np.zeros(10); st[:3] = 2e-3 * scale * rng.standard_normal(3); st[3:6] = 1e-3 * scale * rng.standard_normal(3)
            st[6] = 1e-3 * scale * rng.standard_normal(); st[7] = 1e-4 * scale * rng.standard_normal()
            ba.set_frame_state(k, st)

    def standing(keep):
        ba = pkg.BundleAdjusterHip(ctx, accumulators=1, keep_jacobians=keep); ba.set_case(case, list(range(F)))
        rng = np.random.RandomState(11)
        perturb(ba, rng, 1.0)
        ba.activate_all(); ba.linearize_all(False); ba.apply_res()
        return ba, rng

    # A: linearised on the resident graph
    A, rngA = standing(True)
    n_lin = A.fix_linearization(mask)
    assert 100 < n_lin <= (R + 2) // 3
    fl, J, rtz = A.linearized_residuals()
    assert int(fl.sum()) == n_lin and np.abs(J[fl == 1]).max() > 0 and not J[fl == 0].any() and not rtz[fl == 0].any()
    perturb(A, rngA, 0.5)
    # B: the same window, the residuals handed over
    B, rngB = standing(False)
    assert B.set_linearized_residuals(fl, J, rtz) == n_lin
    fl2, J2, rtz2 = B.linearized_residuals()
    assert np.array_equal(fl, fl2) and np.array_equal(J, J2) and np.array_equal(rtz, rtz2)
    perturb(B, rngB, 0.5)
    aA, aB = A.accumulate(), B.accumulate()
    assert aA["resInA"] == aB["resInA"]
    for k in ("HA", "bA", "Hsc", "bsc"):
        assert np.array_equal(aA[k], aB[k]), k
    for x, y in zip(A.lf_system(), B.lf_system()):
        assert np.array_equal(x, y) and np.abs(x).max() > 1e2
    pA, pB = A.point_acc(), B.point_acc()
    for k in ("Hdd", "bd", "Hcd", "HdiF", "bdSumF"):
        assert np.array_equal(pA[k], pB[k]), k
    assert A.energy_terms()[0] == B.energy_terms()[0] and abs(A.energy_terms()[0]) > 1.0
    th = A.frame_energy_th()          # FrameHessian::frameEnergyTH as the earlier linearisation left it: frame state, handed over like the poses
    rA = A.optimize(4)
    rB = B.optimize(4)
    assert np.array_equal(rA["trace"], rB["trace"]) and rA["finalEnergy"] == rB["finalEnergy"]
    def frames_of(ba):
        return np.concatenate([np.concatenate([np.ravel(x) for x in ba.frame_pose(k)]) for k in range(F)])
    states = [frames_of(A)]
    assert np.array_equal(states[0], frames_of(B))
    # C: a FRESH window built from the mirror that carries them — nothing linearised or applied before the hand-over
    g = pkg.WindowGraph.from_case(case, ctx.L)
    flat = g.export()
    assert np.array_equal(flat["res_point"], case["res_point"]) and np.array_equal(flat["res_target"], case["res_target"])
    first = np.concatenate([[0], np.cumsum(np.bincount(case["res_point"], minlength=len(case["u"])))])
    idx_in_host = np.zeros(len(case["u"]), dtype=np.int64)
    for h in range(F):
        m = np.where(case["host"] == h)[0]; idx_in_host[m] = np.arange(len(m))
    for ri in np.where(fl == 1)[0]:
        p = case["res_point"][ri]
        g.set_residual_linearized(case["host"][p], idx_in_host[p], ri - first[p], J[ri], rtz[ri])
    assert g.linearized_count() == n_lin
    gf, gJ, gr = g.export_linearized()
    assert np.array_equal(gf, fl) and np.array_equal(gJ, J) and np.array_equal(gr, rtz)
    for device_loop in (False, True):
        Cw = pkg.BundleAdjusterHip(ctx, accumulators=1)
        aff = np.zeros((F, 2)) if case.get("aff") is None else case["aff"]
        Cw.set_window(list(range(F)), np.asarray(case["poses0"], dtype=np.float64).reshape(F, 7), aff, np.ones(F, dtype=np.float32), np.arange(F, dtype=np.int32), case["K4"])
        Cw.set_graph_from(g)
        rng = np.random.RandomState(11)
        perturb(Cw, rng, 1.0); perturb(Cw, rng, 0.5)
        Cw.set_frame_energy_th(th)
        assert Cw.linearized_residuals()[0].sum() == n_lin
        Cw.set_device_loop(device_loop)
        rC = Cw.optimize(4)
        ref, r_ref = A, rA
        if device_loop:      # the resident-graph counterpart on the same loop
            ref, rng2 = standing(True)
            assert ref.fix_linearization(mask) == n_lin
            perturb(ref, rng2, 0.5)
            ref.set_device_loop(True)
            r_ref = ref.optimize(4)
            assert np.array_equal(r_ref["trace"][:, 3], rA["trace"][:, 3]) and np.allclose(r_ref["trace"][:, :2], rA["trace"][:, :2], rtol=1e-6)
        assert np.array_equal(r_ref["trace"], rC["trace"]), (device_loop, r_ref["trace"] - rC["trace"])
        assert np.array_equal(frames_of(ref), frames_of(Cw))
        assert np.array_equal(ref.point_state()[0], Cw.point_state()[0])
        Cw.close()
        if device_loop:
            ref.close()
    # the mirror keeps a record with its residual: dropResidual moves the point's last residual (and its record) into the hole, removePoint gives the records back
    p = int(case["res_point"][np.where(fl == 1)[0][0]])
    nres = g.point_residuals(case["host"][p], idx_in_host[p])
    lin_of_p = [int(fl[first[p] + k]) for k in range(nres)]
    g.drop_residual(case["host"][p], idx_in_host[p], 0)
    assert g.linearized_count() == n_lin - lin_of_p[0]
    gf2 = g.export_linearized()[0]
    moved = lin_of_p[1:]
    if len(moved):
        moved = [moved[-1]] + moved[:-1]
    assert list(gf2[first[p]:first[p] + nres - 1]) == moved
    g.remove_point(case["host"][p], idx_in_host[p])
    assert g.linearized_count() == n_lin - sum(lin_of_p)
    # a bare flag is refused, and the graph with it
    D = pkg.BundleAdjusterHip(ctx); D.set_case(case, list(range(F)))
    D.set_residual_flags(np.zeros(R, dtype=np.uint8))
    with pytest.raises(pkg.HipLibraryError, match="accumulateLF_MT"):
        D.set_residual_flags(fl)
    with pytest.raises(pkg.HipLibraryError, match="set_graph first"):
        D.optimize(2)
    for w in (A, B, D):
        w.close()


@pytest.mark.parametrize("loop", ["host", "device", "batch4"])
def test_residuals_kept_linearised_across_optimize_calls(pkg, oracle, synth, gpu_required, loop):
    """EFResidual::fixLinearizationF outside a marginalisation, accumulateLF_MT / addPoint<1> and calcLEnergyPt (EnergyFunctionalStructs.cpp:85-113,
    AccumulatedTopHessian.cpp:52-58,84-98, EnergyFunctional.cpp:223-233,349-409); the oracle's branch is pinned bit for bit to libref.so (tests/test_ref_pin_cpu.py).
    First with identical bits on both sides — non-zero frame deltas at the fix and other ones at the accumulation, set through the API: H_L / b_L, the A and Schur systems
    as tight as the accumulation itself, the per-point sums bit for bit, E_L to double rounding.  Then through optimisations over the graph that carries them — on the
    host-driven loop, on the device-resident loop (dmvio_hip_ba_set_device_loop: k_ba_solve adds H_L / b_L, k_ba_lin_energy_b the linearised energy, three accumulation
    passes per system on the device) and as one of four windows of a dmvio_hip_ba_optimize_batch call (two more windows carrying linearised residuals, one without)."""
    case = synth.ba_case(256, 192, n_frames=5, n_points=300, hosts_share=(90, 80, 70, 60, 0), seed=5)
    R = len(case["res_point"]); F = case["n_frames"]
    mask = (np.arange(R) % 3 == 0).astype(np.uint8)

    def lin_window(ctx=None, with_oracle=True, check=False):
        """a window with a third of its residuals kept linearised at one set of frame deltas, standing at another one"""
        if ctx is None:
            ctx, ba, W = _window(pkg, oracle, case)
        else:
            ba = pkg.BundleAdjusterHip(ctx, accumulators=1, keep_jacobians=True); ba.set_case(case, list(range(F)))
            W = oracle.BAWindow(case) if with_oracle else None
        rng = np.random.RandomState(11)

        def perturb(scale):
            for k in range(1, F):
                st = np.zeros(10); st[:3] = 2e-3 * scale * rng.standard_normal(3); st[3:6] = 1e-3 * scale * rng.standard_normal(3)
                st[6] = 1e-3 * scale * rng.standard_normal(); st[7] = 1e-4 * scale * rng.standard_normal()
                ba.set_frame_state(k, st)
                if W is not None:
                    W.set_frame_state(k, st)
        perturb(1.0)
        ba.activate_all(); ba.linearize_all(False); ba.apply_res()
        if W is not None:
            W.activate_all(); W.linearize_all(False); W.apply_res()
        n_lin = ba.fix_linearization(mask)                                              # res_toZeroF = resF - J delta at the first deltas
        if W is not None:
            assert n_lin == W.fix_linearization(mask)
        assert 100 < n_lin <= (R + 2) // 3
        perturb(0.5)                                                                    # resApprox = res_toZeroF + J delta at the second ones
        return ctx, ba, W, n_lin

    ctx, ba, W, n_lin = lin_window()
    ag, ao = ba.accumulate(), W.accumulate()
    HLg, bLg = ba.lf_system()
    assert ag["resInA"] == ao["resInA"] and 0 < ag["resInA"] < R - n_lin + 1          # addPoint<0> left the linearised ones out
    off = ao["HL"] - np.diag(np.diag(ao["HL"]))
    assert np.abs(off).max() > 1e3 and np.abs(ao["bL"][12:]).max() > 1e2               # the L system is populated, not just the priors
    for Hg, Ho in ((HLg, ao["HL"]), (ag["HA"], ao["HA"]), (ag["Hsc"], ao["Hsc"])):
        assert np.linalg.norm(Hg - Ho) <= 1e-11 * np.linalg.norm(Ho)
    for bg, bo in ((bLg, ao["bL"]), (ag["bA"], ao["bA"]), (ag["bsc"], ao["bsc"])):
        assert np.linalg.norm(bg - bo) <= 1e-9 * np.linalg.norm(bo)
    pg, po = ba.point_acc(), W.point_acc()
    for k in ("Hdd", "bd", "Hcd", "HdiF", "bdSumF"):                                   # Hdd_accAF .. of addPoint<0>; HdiF / bdSumF carry the LF sums
        assert np.array_equal(pg[k], po[k]), k
    ELg, ELo = ba.energy_terms()[0], W.lenergy()
    assert abs(ELo) > 1.0 and abs(ELg - ELo) <= 1e-12 * abs(ELo)
    x_g = ba.solve(0, 1e-5); x_o = W.solve(0, 1e-5)
    assert np.linalg.norm(x_g - x_o) <= 1e-6 * np.linalg.norm(x_o)
    # a second optimize over the graph that carries them: same decisions, energies and states
    others = []
    if loop == "host":
        rg = ba.optimize(4)
    elif loop == "device":
        ba.set_device_loop(True)
        rg = ba.optimize(4)
    else:
        # the same window twice more (their solve / accumulate calls above replayed, so that they stand where `ba` stands) and a window without linearised residuals
        for _ in range(2):
            _, b2, _, n2 = lin_window(ctx, with_oracle=False)
            assert n2 == n_lin
            b2.accumulate(); b2.solve(0, 1e-5)
            others.append(b2)
        plain = pkg.BundleAdjusterHip(ctx, accumulators=1); plain.set_case(case, list(range(F)))
        single = pkg.BundleAdjusterHip(ctx, accumulators=1); single.set_case(case, list(range(F)))
        B1 = pkg.BundleAdjusterBatch(ctx, 1); r_single = B1.optimize([single], 4)[0]
        B4 = pkg.BundleAdjusterBatch(ctx, 4)
        rs = B4.optimize([ba, others[0], plain, others[1]], 4)
        rg = rs[0]
        for r2 in (rs[1], rs[3]):                                                      # the three windows that carry the same linearised residuals: the same bits
            assert np.array_equal(r2["trace"], rg["trace"]) and r2["finalEnergy"] == rg["finalEnergy"]
        assert np.array_equal(rs[2]["trace"], r_single["trace"]) and rs[2]["finalEnergy"] == r_single["finalEnergy"]   # ... and their neighbour is not touched by them
        assert np.abs(r_single["trace"][:, 1]).max() < np.abs(rg["trace"][:, 1]).min()    # (its E_L has no linearised term)
        others += [plain, single, B1, B4]
    ro = W.optimize(4)
    assert np.array_equal(rg["trace"][:, 3], ro["trace"][:, 3])
    assert np.allclose(rg["trace"][:, 0], ro["trace"][:, 0], rtol=1e-4) and np.allclose(rg["trace"][:, 1], ro["trace"][:, 1], rtol=1e-4)
    assert np.abs(ro["trace"][:, 1]).min() > 1.0                                      # E_L carries the linearised term in every row
    assert abs(rg["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"] and abs(rg["rmse"] - ro["rmse"]) <= 1e-4 * ro["rmse"]
    for k in range(5):
        pg_, ag_, _ = ba.frame_pose(k); po_, ao_, _ = W.frame_pose(k)
        assert np.linalg.norm(pg_[:3] - po_[:3]) < 1e-3 and np.allclose(ag_, ao_, atol=1e-3)
    HL2, bL2 = ba.lf_system()                                                          # accumulateLF_MT's system of the loop's last accumulation came back with the states
    assert np.abs(HL2 - np.diag(np.diag(HL2))).max() > 1e3
    zero = np.zeros(R, np.uint8)
    if loop == "host":
        assert ba.fix_linearization(zero) == W.fix_linearization(zero) == n_lin        # the final linearizeAll(true) removes none of them
        # marginalising the points relinearises their residuals (FullSystem.cpp:840-843): the linearised flags of those points go
        cand = np.zeros(ba.N, np.uint8); cand[: ba.N // 2] = 1
        ba.marginalize_points(cand)
        left = ba.fix_linearization(zero)
        rp = np.asarray(case["res_point"])
        assert left < n_lin and left <= int(np.sum(mask.astype(bool) & (rp >= ba.N // 2)))
    # (a window sharded over ranks: tests/test_sharded_ba_gpu.py)
    for o in others:
        o.close()
    ba.close()


def test_new_graph_right_after_an_accepted_iteration_call(pkg, oracle, synth, gpu_required):
    """An accepted dmvio_hip_ba_gn_iteration leaves the newest keyframe's threshold "on its way" (published a few microseconds behind the decision).  A NEW window set on
    the same handle right afterwards must neither wait for that threshold (the host-coherent record is cleared, its ticket restarts) nor inherit it: the new window
    linearises like a fresh handle does, twenty times in a row."""
    a = synth.ba_case(256, 256, n_frames=5, n_points=300, seed=31)
    b = synth.ba_case(256, 256, n_frames=5, n_points=320, seed=32)
    ctx = pkg.Context(256, 256, n_slots=10)
    for k in range(5):
        ctx.frame_upload(k, a["imgs"][k]); ctx.frame_upload(5 + k, b["imgs"][k])
    fresh = pkg.BundleAdjusterHip(ctx, accumulators=1); fresh.set_case(b, list(range(5, 10)))
    fresh.activate_all(); e_fresh = fresh.linearize_all(False); th_fresh = fresh.frame_energy_th()
    ba = pkg.BundleAdjusterHip(ctx, accumulators=1)
    for rep in range(20):
        ba.set_case(a, list(range(5)))
        ba.activate_all(); e = ba.linearize_all(False); ba.apply_res()
        acc, lam, lE = ba.gn_iteration(0, 1e-5, [e, 0.0, 0.0])
        assert acc                                         # an accepted step: its threshold is still pending when the next graph arrives
        ba.set_case(b, list(range(5, 10)))
        ba.activate_all()
        assert ba.linearize_all(False) == e_fresh and np.array_equal(ba.frame_energy_th(), th_fresh), rep


def test_optimize_loop_equals_iteration_calls_through_rejected_steps(pkg, oracle, synth, gpu_required):
    """dmvio_hip_ba_optimize does not wait for the relinearisation that follows a rejected step (the restored state's energy stays on the device for the next
    accept test); the per-iteration entry point does.  Thirty iterations, most of them rejected: same accept sequence, same energies, bit for bit."""
    case = synth.ba_case(512, 512, n_frames=8, n_points=1000, seed=13)
    ctx, ba, W = _window(pkg, oracle, case)
    ba2 = pkg.BundleAdjusterHip(ctx, accumulators=1); ba2.set_case(case, list(range(8)))
    out = ba.optimize(30)
    ba2.activate_all(); e = ba2.linearize_all(False); ba2.apply_res()
    lam, lE = 1e-5, [e, 0.0, 0.0]
    tr = [[e, 0.0, 0.0, 1.0]]
    for it in range(30):
        acc, lam, lE = ba2.gn_iteration(it, lam, lE)
        tr.append([lE[0], lE[1], lE[2], 1.0 if acc else 0.0])
    tr = np.array(tr)
    assert out["iterations"] == 30 and (tr[:, 3] == 0).sum() >= 10
    assert np.array_equal(out["trace"], tr), np.abs(out["trace"] - tr).max()
    for k in range(8):
        assert np.array_equal(ba.frame_pose(k)[0][:3], ba2.frame_pose(k)[0][:3])


def test_shard_systems_sum_to_full_system(pkg, oracle, synth, gpu_required):
    """Rank emulation on one GPU: the packed systems of two keyframe shards add up to the full window's system."""
    import dmvio_amd.sharding as sh
    case = synth.ba_case(512, 512, n_frames=8, n_points=1000, seed=12)
    ctx, ba, W = _window(pkg, oracle, case)
    ba.activate_all(); e_full = ba.linearize_all(False); ba.apply_res(); full = ba.accumulate()
    parts = sh.partition_points_by_host(case["host"], 2)
    tot, e_sum = None, 0.0
    for idx in parts:
        sub = sh.shard_case(case, idx)
        b = pkg.BundleAdjusterHip(ctx, accumulators=1); b.set_case(sub, list(range(8)))
        b.activate_all(); e_loc, _ = b.linearize_local(False); b.apply_res(); a = b.accumulate()
        buf = sh.pack_system(a["HA"], a["bA"], a["Hsc"], a["bsc"], e_loc, a["resInA"])
        tot = buf if tot is None else tot + buf
        b.close()
    HA, bA, Hsc, bsc, e_sum, res = sh.unpack_system(tot, ba.n)
    sc = np.sqrt(np.outer(np.diag(full["HA"]) + 1e-9, np.diag(full["HA"]) + 1e-9))
    assert np.max(np.abs(HA - full["HA"]) / sc) < 1e-11 and np.max(np.abs(Hsc - full["Hsc"])[4:, 4:] / sc[4:, 4:]) < 1e-11
    assert np.max(np.abs(Hsc - full["Hsc"]) / sc) < 2e-6
    assert res == full["resInA"] and abs(e_sum - e_full) <= 1e-9 * e_full


def test_marginalize_points_parity(pkg, oracle, synth, gpu_required):
    """Relinearisation + fixLinearizationF + addPoint<2> / SC accumulation of the points hosted in keyframe 0, with non-zero frame
    deltas (state != state_zero) set identically on both sides: decisions and residual counts identical, HM / bM increments as tight
    as the BA accumulation itself."""
    case = synth.ba_case(512, 512, n_frames=6, n_points=1200, hosts_share=(350, 300, 250, 200, 100, 0), seed=31)
    ctx, ba, W = _window(pkg, oracle, case)
    rng = np.random.RandomState(3)
    for k in range(1, case["n_frames"]):
        st = np.zeros(10); st[:3] = 2e-3 * rng.standard_normal(3); st[3:6] = 1e-3 * rng.standard_normal(3); st[6] = 1e-3 * rng.standard_normal(); st[7] = 1e-4 * rng.standard_normal()
        ba.set_frame_state(k, st); W.set_frame_state(k, st)
    ba.activate_all(); W.activate_all()
    eg = ba.linearize_all(False); eo = W.linearize_all(False)
    assert abs(eg - eo) <= 1e-6 * eo
    ba.apply_res(); W.apply_res()
    ba.accumulate(); W.accumulate()                 # idepth_hessian of every point (AccumulatedSCHessian.cpp:50)
    cand = (np.asarray(case["host"]) == 0).astype(np.uint8)
    cand[1::9] = 1                                   # a few points of other hosts too (isOOB candidates)
    dg, Hg, bg, ng = ba.marginalize_points(cand)
    do, Ho, bo, no = W.marginalize_points(cand)
    assert np.array_equal(dg, do) and ng == no and (do == 1).sum() > 100
    assert np.linalg.norm(Hg - Ho) <= 1e-9 * np.linalg.norm(Ho)
    assert np.linalg.norm(bg - bo) <= 1e-8 * np.linalg.norm(bo) + 1e-12
    # with update_prior the handle's HM / bM take the increment and the next solve uses it
    x0 = ba.solve(0, 1e-5)
    ba.marginalize_points(cand, update_prior=True)
    ba.accumulate()
    x1 = ba.solve(0, 1e-5)
    assert np.all(np.isfinite(x1)) and np.linalg.norm(x1 - x0) > 0


def test_marginalize_frame_parity(pkg, oracle, synth, gpu_required):
    """marginalizePointsF into the prior, then marginalizeFrame of a middle and of the last keyframe: reduced prior vs the oracle."""
    case = synth.ba_case(512, 512, n_frames=5, n_points=600, hosts_share=(200, 180, 120, 100, 0), seed=33)
    ctx, ba, W = _window(pkg, oracle, case)
    for X in (ba, W):
        X.activate_all(); X.linearize_all(False); X.apply_res(); X.accumulate()
    cand = (np.asarray(case["host"]) == 1).astype(np.uint8)
    dg, Hg, bg, _ = ba.marginalize_points(cand, update_prior=True)
    do, Ho, bo, _ = W.marginalize_points(cand)
    W.set_marg_prior(Ho, bo)
    Hp, bp = ba.get_marg_prior()
    assert np.linalg.norm(Hp - Ho) <= 1e-9 * np.linalg.norm(Ho)
    for k in (1, 4, 0):
        Hn_g, bn_g = ba.marginalize_frame(k)
        Hn_o, bn_o = W.marginalize_frame(k)
        assert np.linalg.norm(Hn_g - Hn_o) <= 1e-7 * np.linalg.norm(Hn_o) + 1e-9
        assert np.linalg.norm(bn_g - bn_o) <= 1e-7 * np.linalg.norm(bn_o) + 1e-9


def test_tracking_and_mapping_overlap_on_two_threads(pkg, oracle, synth, gpu_required):
    """The mapping side (BA handle: own HIP stream, own lock) and the tracking side (context stream) run concurrently from two host
    threads — the reference's tracking / mapping thread structure; results are identical to the sequential runs."""
    import threading
    w = h = 512
    bcase = synth.ba_case(w, h, n_frames=8, n_points=2000, seed=17)
    tcase = synth.tracking_case(w, h, n_ref=2000, n_frames=4, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=8 + 1 + 64)
    for k in range(8):
        ctx.frame_upload(k, bcase["imgs"][k])
    ctx.frame_upload(8, tcase["ref_img"])
    B = 64
    for i in range(B):
        ctx.frame_upload(9 + i, tcase["frames"][i % 4]["img"])
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(tcase["K4"])
    trk.setCoarseTrackingRef(8, tcase["u"], tcase["v"], tcase["idepth"], tcase["hdiF"])
    ba = pkg.BundleAdjusterHip(ctx)
    ident = np.tile(np.array([0, 0, 0, 0, 0, 0, 1.0]), (B, 1)); aff0 = np.zeros((B, 2))

    def do_track(out):
        for _ in range(6):
            out["r"] = trk.track_batch(list(range(9, 9 + B)), ident.copy(), aff0.copy())

    def do_ba(out):
        ba.set_case(bcase, list(range(8)))
        out["r"] = ba.optimize(6)
        out["poses"] = [ba.frame_pose(k)[0] for k in range(8)]

    seq_t, seq_b = {}, {}
    do_track(seq_t); do_ba(seq_b)
    par_t, par_b = {}, {}
    th = [threading.Thread(target=do_track, args=(par_t,)), threading.Thread(target=do_ba, args=(par_b,))]
    for t in th: t.start()
    for t in th: t.join()
    assert np.array_equal(par_t["r"]["pose7"], seq_t["r"]["pose7"]) and np.array_equal(par_t["r"]["good"], seq_t["r"]["good"])
    assert par_b["r"]["iterations"] == seq_b["r"]["iterations"] and par_b["r"]["finalEnergy"] == seq_b["r"]["finalEnergy"]
    for a, b in zip(par_b["poses"], seq_b["poses"]):
        assert np.array_equal(a, b)


def test_live_tracking_server_and_mapping_overlap(pkg, synth, gpu_required):
    """One frame at a time on the tracking thread — the host LM against the resident evaluation kernel, which keeps eight workgroups on the device for the length of a
    frame — while the mapping thread optimises a window on its own stream: neither waits for the other to finish, results identical to the sequential runs."""
    import threading
    w = h = 512
    bcase = synth.ba_case(w, h, n_frames=8, n_points=2000, seed=19)
    tcase = synth.tracking_case(w, h, n_ref=2000, n_frames=4, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=8 + 1 + 4)
    for k in range(8):
        ctx.frame_upload(k, bcase["imgs"][k])
    ctx.frame_upload(8, tcase["ref_img"])
    for i in range(4):
        ctx.frame_upload(9 + i, tcase["frames"][i]["img"])
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(tcase["K4"])
    trk.setCoarseTrackingRef(8, tcase["u"], tcase["v"], tcase["idepth"], tcase["hdiF"])
    ba = pkg.BundleAdjusterHip(ctx)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])

    def do_track(out):
        res = []
        for k in range(200):
            res.append(trk.trackNewestCoarse(9 + k % 4, ident, (0.0, 0.0))["pose7"])
        out["r"] = np.array(res)

    def do_ba(out):
        for _ in range(8):
            ba.set_case(bcase, list(range(8)))
            out["r"] = ba.optimize(6)
        out["poses"] = [ba.frame_pose(k)[0] for k in range(8)]

    seq_t, seq_b = {}, {}
    do_track(seq_t); do_ba(seq_b)
    par_t, par_b = {}, {}
    th = [threading.Thread(target=do_track, args=(par_t,)), threading.Thread(target=do_ba, args=(par_b,))]
    for t in th: t.start()
    for t in th: t.join()
    assert np.array_equal(par_t["r"], seq_t["r"])
    assert par_b["r"]["finalEnergy"] == seq_b["r"]["finalEnergy"] and all(np.array_equal(a, b) for a, b in zip(par_b["poses"], seq_b["poses"]))
    for k in range(4):
        assert np.linalg.norm(seq_t["r"][k][:3] - tcase["frames"][k]["pose7"][:3]) < 5e-3


def test_default_accumulation_order_optimize_parity(pkg, oracle, synth, gpu_required):
    """The library's default accumulation (4 partial accumulators per bucket — the structure of the reference's multi-threaded mode with a
    fixed assignment) through the whole FullSystem::optimize: same accept / reject sequence, final energy within 1e-4, poses within 1e-3 m of
    the single-threaded oracle; and reproducible run to run."""
    case = synth.ba_case(512, 512, n_frames=8, n_points=2000, seed=4321)
    ctx, ba, W = _window(pkg, oracle, case, accumulators=None)
    rg = ba.optimize(6); ro = W.optimize(6)
    assert rg["iterations"] == ro["iterations"] == 6
    assert np.array_equal(rg["trace"][:, 3], ro["trace"][:, 3])
    assert abs(rg["finalEnergy"] - ro["finalEnergy"]) <= 1e-4 * ro["finalEnergy"] and abs(rg["rmse"] - ro["rmse"]) <= 1e-4 * ro["rmse"]
    for k in range(8):
        assert np.linalg.norm(ba.frame_pose(k)[0][:3] - W.frame_pose(k)[0][:3]) < 1e-3
    ba2 = pkg.BundleAdjusterHip(ctx); ba2.set_case(case, list(range(8)))
    r2 = ba2.optimize(6)
    assert np.array_equal(r2["trace"], rg["trace"]) and all(np.array_equal(ba2.frame_pose(k)[0], ba.frame_pose(k)[0]) for k in range(8))
    ba.close(); ba2.close(); ctx.close()


def test_host_solver_equals_the_oracle_bitwise(pkg, oracle, synth, gpu_required):
    """EnergyFunctional::solveSystemF on the host side of the ABI (csrc/ba_host.hpp: priors, Jacobi-scaled 68x68 LDL^T, gauge nullspaces from SE3 exp / log differences, the
    orthogonalisation from iteration 2 on) fed with the ORACLE's accumulated system: the step x must be the oracle's — which is pinned to the reference's — bit for bit up to the
    orthogonalisation, and to 1e-14 behind it."""
    case = synth.ba_case(320, 256, n_frames=5, n_points=400, hosts_share=(130, 110, 90, 70, 0), seed=31)
    ctx, ba, W = _window(pkg, oracle, case)
    ba.activate_all(); W.activate_all()
    ba.linearize_all(False); W.linearize_all(False)
    ba.apply_res(); W.apply_res()
    ba.accumulate(); ao = W.accumulate()
    worst = 0.0
    for it, lam in ((0, 1e-5), (1, 1e-4), (2, 1e-3), (3, 1e-5), (5, 1e-1)):
        xo = W.solve(it, lam)
        xg = ba.solve_system(it, lam, ao["HA"], ao["bA"], ao["Hsc"], ao["bsc"])
        worst = max(worst, float(np.abs(xg - xo).max() / np.abs(xo).max()))
        if it < 2:
            assert np.array_equal(xg.view(np.uint64), xo.view(np.uint64)), (it, lam, worst)
        else:
            # from iteration 2 on x is projected off the gauge nullspaces through an SVD — Eigen's JacobiSVD in the reference, unpinned third-party arithmetic:
            # the library keeps unit singular vectors, the oracle scales by the singular values; same projector, rounding apart
            assert np.abs(xg - xo).max() <= 1e-14 * np.abs(xo).max(), (it, lam, worst)
    ba.close()


def test_linearised_residuals_in_a_window_of_ten_keyframes_on_every_loop(pkg, synth, gpu_required):
    """Residuals kept linearised in a window of MORE than eight keyframes (the wide instantiations: k_ba_solve<12>, the 8-member Schur tiles of the stitch, 100 ticket slots):
    the device-resident loop and a batch of two such windows against the host-driven loop on the same graph — the accept sequence, E_A / E_L / E_M of every iteration and the
    final states (the loops differ in the elementary functions of the frame step: rounding level)."""
    F = 10
    share = tuple([120, 110, 100, 90, 80, 70, 60, 50, 40, 0])
    case = synth.ba_case(320, 256, n_frames=F, n_points=720, seed=77, hosts_share=share, step_t=0.05, step_r=np.deg2rad(1.0))
    R = len(case["res_point"])
    mask = (np.arange(R) % 4 == 1).astype(np.uint8)
    ctx = pkg.Context(case["w"], case["h"], n_slots=F)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])

    def window():
        ba = pkg.BundleAdjusterHip(ctx, accumulators=1, keep_jacobians=True); ba.set_case(case, list(range(F)))
        ba.optimize(2)
        n = ba.fix_linearization(mask)
        assert 0.15 * R < n <= (R + 3) // 4
        return ba, n
    host, n_lin = window()
    r_h = host.optimize(4)
    assert np.abs(r_h["trace"][:5, 1]).min() > 0            # E_L carries the linearised term
    dev, _ = window(); dev.set_device_loop(True)
    r_d = dev.optimize(4)
    b0, _ = window(); b1, _ = window()
    B = pkg.BundleAdjusterBatch(ctx, 2)
    r_b = B.optimize([b0, b1], 4)
    assert np.array_equal(r_b[0]["trace"], r_d["trace"]) and np.array_equal(r_b[1]["trace"], r_d["trace"])      # the batch is the device loop, bit for bit
    for r in (r_d,):
        assert r["iterations"] == r_h["iterations"] and np.array_equal(r["trace"][:, 3], r_h["trace"][:, 3])
        assert np.allclose(r["trace"][:5, :3], r_h["trace"][:5, :3], rtol=1e-6, atol=1e-9)
    for k in range(F):
        assert np.linalg.norm(dev.frame_pose(k)[0][:3] - host.frame_pose(k)[0][:3]) < 1e-6
        assert np.array_equal(b0.frame_pose(k)[0], dev.frame_pose(k)[0])
    assert np.abs(dev.point_state()[0] - host.point_state()[0]).max() < 1e-5
    for o in (host, dev, b0, b1, B):
        o.close()
    ctx.close()


def test_marginalising_points_whose_residuals_arrived_linearised(pkg, synth, gpu_required):
    """FullSystem::flagPointsForRemoval relinearises the residuals of the points it marginalises with isLinearized = false (FullSystem.cpp:840-843) — also when they were kept
    linearised before.  A window that RECEIVED its linearised residuals (dmvio_hip_ba_set_linearized_residuals) must go through dmvio_hip_ba_marginalize_points exactly like the
    window they were linearised on: the decisions, the prior increment and the flags that are left, bit for bit."""
    case = synth.ba_case(256, 192, n_frames=5, n_points=300, hosts_share=(90, 80, 70, 60, 0), seed=5)
    R = len(case["res_point"]); F = case["n_frames"]; N = len(case["u"])
    mask = (np.arange(R) % 3 == 0).astype(np.uint8)
    ctx = pkg.Context(case["w"], case["h"], n_slots=F)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])

    def standing(keep):
        ba = pkg.BundleAdjusterHip(ctx, accumulators=1, keep_jacobians=keep); ba.set_case(case, list(range(F)))
        ba.optimize(3)
        return ba
    A = standing(True)
    n_lin = A.fix_linearization(mask)
    fl, J, rtz = A.linearized_residuals()
    B = standing(True)
    assert B.set_linearized_residuals(fl, J, rtz) == n_lin
    cand = np.zeros(N, np.uint8); cand[::3] = 1
    dA, HA, bA, rA = A.marginalize_points(cand)
    dB, HB, bB, rB = B.marginalize_points(cand)
    assert np.array_equal(dA, dB) and rA == rB and rA > 0
    assert np.array_equal(HA, HB) and np.array_equal(bA, bB) and np.abs(HA).max() > 0
    zero = np.zeros(R, np.uint8)
    left = A.fix_linearization(zero)
    assert left == B.fix_linearization(zero) and 0 < left < n_lin          # the marginalised points' residuals lost the flag, the others kept it
    fa, fb = A.linearized_residuals()[0], B.linearized_residuals()[0]
    assert np.array_equal(fa, fb) and not fa[cand[case["res_point"]] == 1].any()
    for o in (A, B):
        o.close()
    ctx.close()
