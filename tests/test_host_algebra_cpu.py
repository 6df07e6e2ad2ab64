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
