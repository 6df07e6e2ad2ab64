"""The host side of the C ABI does its pose algebra (SE3 exp / log / products / inverses: motion hypotheses, frame states, LM steps) with the C library's sincos / exp like the
reference's Sophus compiled by g++ — so it equals the oracle, which is pinned to the reference (tests/test_ref_pin_cpu.py), bit for bit.  No device involved."""
import numpy as np


def test_motion_hypotheses_equal_the_oracle_bitwise(pkg, oracle):
    rng = np.random.RandomState(3)
    for it in range(4000):
        sc = 10.0 ** rng.uniform(-4, 0.3)          # from sub-millimetre inter-frame motion to more than a radian
        a = oracle.se3_exp(0.3 * rng.standard_normal(6)); b = oracle.se3_exp(sc * rng.standard_normal(6)); c = oracle.se3_exp(sc * rng.standard_normal(6))
        slast = oracle.se3_mul(a, b); lastF = oracle.se3_mul(slast, c)
        m = np.asarray(oracle.make_track_hypotheses(slast, a, lastF)); h = np.asarray(pkg.make_track_hypotheses(slast, a, lastF))
        assert m.shape == (31, 7) and np.array_equal(m.view(np.uint64), h.view(np.uint64)), (it, sc, np.abs(m - h).max())


def test_visual_lm_step_equals_the_oracle_bitwise(pkg, oracle):
    """dmvio_hip_coarse_update_visual (what a computeCoarseUpdate callback falls back to, and what the library's own host LM runs) against the step inside the oracle's
    trackNewestCoarse — which is pinned to the reference's (test_track_newest_coarse_bitwise): damped 8x8 system, the 8 / 7 / 6-dof LDL^T of the four affine modes,
    extrapolation, scaling, SE3::exp * current — pose, affine increments and norm, bit for bit."""
    rng = np.random.RandomState(8)
    modes = [(1e12, 1e8), (-1.0, -1.0), (1e12, -1.0), (-1.0, 1e8)]
    for it in range(2000):
        A = rng.standard_normal((8, 12)) * (10.0 ** rng.uniform(-1, 3, (8, 1)))
        H = A @ A.T
        H = 0.5 * (H + H.T)
        b = rng.standard_normal(8) * 10.0 ** rng.uniform(-2, 3)
        lam = np.float32(0.01 * 4.0 ** rng.randint(-6, 6))
        extrap = np.float32(1.0) if lam >= 1e-3 else np.float32(np.sqrt(np.sqrt(np.float32(1e-3) / lam)))
        cur = oracle.se3_exp(0.2 * rng.standard_normal(6))
        mA, mB = modes[it % 4]
        po, ao, bo, no = oracle.coarse_update_visual(H, b, float(extrap), float(lam), cur, mA, mB)
        pg, ag, bg, ng = pkg.coarse_update_visual(H, b, float(extrap), float(lam), cur, settings=(9.0, 20.0, mA, mB))
        assert np.array_equal(po.view(np.uint64), pg.view(np.uint64)), (it, np.abs(po - pg).max())
        assert (ao, bo, no) == (ag, bg, ng), (it, ao - ag, bo - bg, no - ng)
    # a non-finite system: the increment is dropped (pose unchanged), like the reference's isfinite guard
    H = np.eye(8); b = np.full(8, np.nan)
    cur = oracle.se3_exp(np.array([0.1, 0.2, 0.3, 0.01, 0.02, 0.03]))
    pg, ag, bg, ng = pkg.coarse_update_visual(H, b, 1.0, 0.01, cur)
    po, ao, bo, no = oracle.coarse_update_visual(H, b, 1.0, 0.01, cur)
    assert np.array_equal(po.view(np.uint64), pg.view(np.uint64)) and ag == 0.0 and bg == 0.0 and ao == 0.0


def test_lie_dev_host_functions_equal_lie_h_bitwise(tmp_path, oracle):
    """csrc/lie_dev.h compiled for the host (hipcc, the flags of csrc/Makefile) against oracle/lie.h as g++ compiled it (liboracle.so's orc_se3_*), function by function: exp,
    log, product, inverse, Adj over 200000 random tangents — the pose algebra every host path of the library uses is the oracle's, hence the vendored Sophus', bit for bit."""
    import os
    import shutil
    import subprocess
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "lie_compare")
    libdir = os.path.dirname(oracle.build())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "dm-vio_amd", "csrc"),
                           os.path.join(root, "tests", "lie_compare.hip"), "-L" + libdir, "-loracle", "-Wl,-rpath," + libdir, "-o", exe])
    p = subprocess.run([exe], stdout=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stdout.decode()
    assert b"exp 0 log 0 mul 0 inv 0 adj 0" in p.stdout


def test_ba_host_algebra_equals_the_oracle_bitwise(tmp_path, oracle, synth):
    """csrc/ba_host.hpp's BAHost — everything the library computes on the host around the BA kernels — compiled host-only (tests/ba_host_harness.hip) and set up like
    dmvio_hip_ba_set_window does: FrameFramePrecalc tables, adjoints, gauge nullspaces, frame poses after state changes, prior / marginalisation energies and the damped
    Jacobi-scaled solve of solveSystemF (with and without a marginalisation prior) equal the oracle's — which is pinned to the reference's — bit for bit; behind the SVD-based
    orthogonalisation (iteration >= 2, unpinned Eigen arithmetic) the step agrees to 1e-14."""
    import ctypes as C
    import os
    import shutil
    import subprocess
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not available")
    so = str(tmp_path / "libbah.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(root, "dm-vio_amd", "csrc"),
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "ba_host_harness.hip"), "-o", so])
    L = C.CDLL(so)
    cd, cf, ci = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.bah_create.restype = C.c_void_p; L.bah_create.argtypes = [C.c_int, cd, cd, cf, ci, cd]
    L.bah_destroy.argtypes = [C.c_void_p]
    L.bah_get_precalc.argtypes = [C.c_void_p, C.c_int, C.c_int, cf]; L.bah_get_adjoints.argtypes = [C.c_void_p, cd, cd]; L.bah_get_nullspaces.argtypes = [C.c_void_p, cd]
    L.bah_set_frame_state.argtypes = [C.c_void_p, C.c_int, cd]; L.bah_get_frame.argtypes = [C.c_void_p, C.c_int, cd, cd]; L.bah_energies.argtypes = [C.c_void_p, cd, cd]
    L.bah_set_marg_prior.argtypes = [C.c_void_p, cd, cd]; L.bah_solve_system.argtypes = [C.c_void_p, C.c_int, C.c_double, cd, cd, cd, cd, cd]

    def d(a):
        return a.ctypes.data_as(cd)

    def same(a, b):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    case = synth.ba_case(320, 256, n_frames=5, n_points=400, hosts_share=(130, 110, 90, 70, 0), seed=31)
    F = case["n_frames"]; n = 4 + 8 * F
    W = oracle.BAWindow(case)
    poses = np.ascontiguousarray(np.array(case["poses0"], dtype=np.float64).reshape(F, 7))
    aff = np.zeros((F, 2)); ex = np.ones(F, np.float32); ids = np.arange(F, dtype=np.int32)
    K = np.ascontiguousarray(np.array(case["K4"], dtype=np.float64))
    H = L.bah_create(F, d(poses), d(aff), ex.ctypes.data_as(cf), ids.ctypes.data_as(ci), d(K))

    def tables(tag):
        for h in range(F):
            for t in range(F):
                if h == t:
                    continue
                o = np.zeros(28, np.float32); L.bah_get_precalc(H, h, t, o.ctypes.data_as(cf)); p = W.precalc(h, t)
                ref = np.r_[p["KRKi"].ravel(), p["Kt"], p["R0"].ravel(), p["t0"], p["aff"], p["b0"]].astype(np.float32)
                assert same(o[:27], ref), (tag, "precalc", h, t)
        ah = np.zeros((F * F, 8, 8)); at = np.zeros((F * F, 8, 8)); L.bah_get_adjoints(H, d(ah), d(at)); oh, ot, _ = W.adjoints()
        assert same(ah, oh) and same(at, ot), (tag, "adjoints")
        ns = np.zeros((7, n)); L.bah_get_nullspaces(H, d(ns))
        assert same(ns, W.nullspaces()), (tag, "nullspaces")
    tables("initial")
    rng = np.random.RandomState(2)
    for k in range(1, F):                                   # FrameHessian::setState with a moved pose and brightness
        _, _, s = W.frame_pose(k)
        st = s.copy(); st[:6] += 1e-3 * rng.standard_normal(6); st[6] += 1e-3 * rng.standard_normal(); st[7] += 1e-4 * rng.standard_normal()
        W.set_frame_state(k, st); L.bah_set_frame_state(H, k, d(np.ascontiguousarray(st)))
        pp = np.zeros(7); ss = np.zeros(10); L.bah_get_frame(H, k, d(pp), d(ss))
        assert same(pp, W.frame_pose(k)[0]), ("w2c", k)
    tables("stepped")
    W.activate_all(); W.linearize_all(False); W.apply_res()
    for with_prior in (False, True):
        if with_prior:                                      # a marginalisation prior: HM enters HFinal, bM + HM * delta enters bFinal, and E_M is no longer zero
            A = rng.standard_normal((n, n + 8)) * 30.0; HM = A @ A.T; bM = 50.0 * rng.standard_normal(n)
            W.set_marg_prior(HM, bM); L.bah_set_marg_prior(H, d(np.ascontiguousarray(HM)), d(np.ascontiguousarray(bM)))
        EL = np.zeros(1); EM = np.zeros(1); L.bah_energies(H, d(EL), d(EM))
        assert EL[0] == W.lenergy() and EM[0] == W.menergy() and (EM[0] != 0.0) == with_prior
        ao = W.accumulate()
        sysm = [np.ascontiguousarray(ao[k]) for k in ("HA", "bA", "Hsc", "bsc")]
        for it, lam in ((0, 1e-5), (1, 1e-4), (2, 1e-3), (4, 1e-1)):
            xo = W.solve(it, lam); x = np.zeros(n)
            L.bah_solve_system(H, it, lam, d(sysm[0]), d(sysm[1]), d(sysm[2]), d(sysm[3]), d(x))
            if it < 2:
                assert same(x, xo), (with_prior, it, np.abs(x - xo).max())
            else:
                assert np.abs(x - xo).max() <= 1e-14 * np.abs(xo).max(), (with_prior, it)
    # residuals kept linearised (EFResidual::fixLinearizationF outside a marginalisation): H_L / b_L of accumulateLF_MT enter HFinal_top = (H_L + priors) + H_M + H_A and
    # bFinal_top likewise (EnergyFunctional.cpp:906-907) — the host side adds the stitched L system it is handed in that order
    L.bah_set_lf_raw.argtypes = [C.c_void_p, cd, cd]
    mask = (np.arange(W.R) % 3 == 0).astype(np.uint8)
    assert W.fix_linearization(mask) > 100
    ao = W.accumulate()
    HLr, bLr = W.accumulate_lf_raw()
    assert np.abs(HLr).max() > 1e3 and not same(ao["HL"], HLr)
    L.bah_set_lf_raw(H, d(np.ascontiguousarray(HLr)), d(np.ascontiguousarray(bLr)))
    sysm = [np.ascontiguousarray(ao[k]) for k in ("HA", "bA", "Hsc", "bsc")]
    for it, lam in ((0, 1e-5), (1, 1e-4)):
        xo = W.solve(it, lam); x = np.zeros(n)
        L.bah_solve_system(H, it, lam, d(sysm[0]), d(sysm[1]), d(sysm[2]), d(sysm[3]), d(x))
        assert same(x, xo), ("linearised", it, np.abs(x - xo).max())
    L.bah_destroy(H)


def test_ba_solve_ldlt_is_the_references_non_gtsam_solve(pkg, oracle):
    """dmvio_hip_ba_solve_ldlt / dmvio_hip_ba_hook_ldlt (host-only: what a computeBAUpdate hook falls back to): EnergyFunctional.cpp:971-973 — scale by (H_ii + 10)^-1/2 from both
    sides, pivoted LDL^T, scale back.  Against the oracle's LDL^T (oracle/dense.h) on the scaled system bit for bit, and against a float64 dense solve to rounding."""
    import ctypes as C
    L = pkg.load_library()
    c_d = C.POINTER(C.c_double)
    L.dmvio_hip_ba_solve_ldlt.argtypes = [C.c_int, c_d, c_d, c_d]; L.dmvio_hip_ba_solve_ldlt.restype = C.c_int
    rng = np.random.RandomState(12)
    for n in (12, 36, 68):
        A = rng.standard_normal((n, 3 * n)); H = A @ A.T + np.diag(rng.uniform(0.0, 50.0, n)); H = 0.5 * (H + H.T)
        b = rng.standard_normal(n) * 10.0
        x = np.zeros(n)
        Hc = np.ascontiguousarray(H); bc = np.ascontiguousarray(b)
        assert L.dmvio_hip_ba_solve_ldlt(n, Hc.ctypes.data_as(c_d), bc.ctypes.data_as(c_d), x.ctypes.data_as(c_d)) == 0
        sv = 1.0 / np.sqrt(np.diag(H) + 10.0)
        Hs = sv[:, None] * H * sv[None, :]                       # SVecI.asDiagonal() * HFinal_top * SVecI.asDiagonal()
        # the library forms sv_i * H_ij * sv_j left to right over the lower triangle and mirrors it
        Hs_lib = np.zeros_like(H)
        for i in range(n):
            for j in range(i + 1):
                Hs_lib[i, j] = Hs_lib[j, i] = (sv[i] * H[i, j]) * sv[j]
        want = sv * oracle.ldlt_solve(Hs_lib, sv * b)
        assert np.array_equal(x.view(np.uint64), want.view(np.uint64)), n
        assert np.allclose(x, np.linalg.solve(H, b), rtol=1e-8, atol=1e-10)
    assert L.dmvio_hip_ba_solve_ldlt(0, Hc.ctypes.data_as(c_d), bc.ctypes.data_as(c_d), x.ctypes.data_as(c_d)) < 0
