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
