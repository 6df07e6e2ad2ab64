"""k_ba_solve's pivoted LDL^T (csrc/ba_batch_kernels.hpp: baLdltSolveCore) on crafted systems, through dmvio_hip_ba_debug_solve, against the host's
BAHost::ldltSolveTransposed (dmvio_hip_ba_solve_ldlt) — which restates Eigen's ldlt().solve() as the reference calls it (EnergyFunctional.cpp:971-973; Eigen's unblocked
LDLT picks the largest |diagonal| of the not yet updated trailing diagonal, first one on ties, and swaps it to the front).  Every branch of the device's pivot-order search is
forced and ASSERTED to have run: ranks (all |diagonal| distinct), ties displaced by swaps, NaN on the diagonal, a zero matrix; n = 36 / 68 / 100 (4 / 8 / 12 keyframes: one
and two row groups, both kernel instantiations).  x with the exact back substitution: bit for bit; the permutation: equal to a literal replay of the selection loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RANKS, TIES, NAN = 0, 1, 2


def literal_pivot_order(H):
    """the selection-with-swaps of Eigen's unblocked LDLT on the SCALED diagonal, as BAHost::ldltSolveTransposed runs it (strict >: the first maximum wins)"""
    n = len(H)
    sv = 1.0 / np.sqrt(np.diag(H) + 10)
    d = [float(abs((sv[i] * H[i, i]) * sv[i])) for i in range(n)]
    idx = list(range(n))
    for k in range(n):
        big = k; bigv = d[k]
        for i in range(k + 1, n):
            if d[i] > bigv:
                bigv = d[i]; big = i
        d[k], d[big] = d[big], d[k]; idx[k], idx[big] = idx[big], idx[k]
    return np.array(idx)


def spd(rng, n, scale=1.0):
    A = rng.standard_normal((n, n + 8))
    return scale * (A @ A.T)


@pytest.fixture(scope="module")
def ctx(pkg, gpu_required):
    c = pkg.Context(64, 64, n_slots=1)
    yield c
    c.close()


@pytest.mark.parametrize("n", [36, 68, 100])
def test_distinct_diagonal_takes_the_rank_shortcut(pkg, ctx, n):
    rng = np.random.RandomState(100 + n)
    for trial in range(4):
        H = spd(rng, n, 10.0 ** (trial - 1)); b = rng.standard_normal(n) * 5
        xh = pkg.host_solve_ldlt(ctx.L, H, b)
        x, perm, branch, zero = pkg.debug_solve(ctx, H, b, exact_backsub=True)
        assert branch == RANKS and zero == 0
        assert np.array_equal(perm, literal_pivot_order(H))
        assert np.array_equal(x, xh), (n, trial, np.abs(x - xh).max())
        xf, _, _, _ = pkg.debug_solve(ctx, H, b, exact_backsub=False)
        assert np.abs(xf - xh).max() <= 1e-11 * np.abs(xh).max()
        assert np.abs(H @ x - b).max() <= 1e-7 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("n", [36, 68, 100])
def test_ties_displaced_by_swaps_replay_the_selection(pkg, ctx, n):
    """tie groups at several positions of the sequence, among them the [5, 5, 9] pattern (selection with swaps takes 9, then the SECOND 5: the swap brought it to the
    front), groups that a swap splits, a group spanning both lane registers (positions below and above 64) and huge diagonals that all scale to the same value"""
    rng = np.random.RandomState(200 + n)
    seen_ties = 0
    for trial in range(8):
        H = spd(rng, n, 0.05)
        dg = np.diag(H).copy()
        if trial == 0:
            dg[:3] = [5.0, 5.0, 9.0]                                   # the verdict's pattern at the front
        elif trial == 1:
            dg[n - 3:] = [5.0, 5.0, 9.0]                               # ... at the end
            dg[n // 2] = 9.0                                           # and a second 9 in the middle
        elif trial == 2:
            dg[::3] = 7.0                                              # one large group over the whole range (both registers when n > 64)
        elif trial == 3:
            dg[:] = np.repeat(rng.permutation(n // 4 + 1) + 1.0, 4)[:n]   # many groups of four, in scrambled order of size
        elif trial == 4:
            dg[1::2] = 1e13; dg[0] = 1e13                              # priors of fixed parameters: v / (v + 10) rounds to the same double
        elif trial == 5:
            dg[:] = 3.0                                                # everything tied: the order is the identity
        elif trial == 6:
            dg[:] = 3.0; dg[n - 1] = 4.0; dg[n // 2] = 4.0             # two leaders behind a tied field: both swaps displace members of the big group
        else:
            k = rng.permutation(n)[: n // 2]; dg[k] = np.round(dg[k])  # random small integer collisions
        H[np.arange(n), np.arange(n)] = dg + n                         # keep it comfortably definite
        b = rng.standard_normal(n)
        xh = pkg.host_solve_ldlt(ctx.L, H, b)
        x, perm, branch, zero = pkg.debug_solve(ctx, H, b, exact_backsub=True)
        lit = literal_pivot_order(H)
        sv = 1.0 / np.sqrt(np.diag(H) + 10); ds = np.abs((sv * np.diag(H)) * sv)
        has_ties = len(np.unique(ds)) < n
        assert branch == (TIES if has_ties else RANKS), (n, trial, branch)
        seen_ties += int(has_ties)
        assert np.array_equal(perm, lit), (n, trial, perm, lit)
        assert np.array_equal(x, xh), (n, trial, np.abs(x - xh).max())
    assert seen_ties >= 6


def test_the_5_5_9_pattern_is_eigens_not_the_stable_order(pkg, ctx):
    """diag [5a, 5b, 9, ...smaller]: Eigen's swaps give 9, 5b, 5a (the swap of step 0 moved 5a behind 5b); a stable sort would give 9, 5a, 5b"""
    n = 12
    H = np.eye(n) * 1.0 + 0.01 * np.ones((n, n))
    H[0, 0] = 5.0; H[1, 1] = 5.0; H[2, 2] = 9.0
    b = np.arange(1.0, n + 1)
    x, perm, branch, _ = pkg.debug_solve(ctx, H, b)
    assert branch == TIES
    assert list(perm[:3]) == [2, 1, 0], perm
    assert np.array_equal(x, pkg.host_solve_ldlt(ctx.L, H, b))


@pytest.mark.parametrize("n", [36, 68, 100])
def test_nan_on_the_diagonal_takes_the_literal_loop(pkg, ctx, n):
    rng = np.random.RandomState(300 + n)
    for pos in (0, n // 2, n - 1):
        H = spd(rng, n); b = rng.standard_normal(n)
        H[pos, pos] = np.nan
        x, perm, branch, zero = pkg.debug_solve(ctx, H, b)
        xh = pkg.host_solve_ldlt(ctx.L, H, b)
        assert branch == NAN
        # the host's loop: a NaN is never selected over a number, and nothing is selected over a NaN standing at position k
        sv = 1.0 / np.sqrt(np.diag(H) + 10); d = list(np.abs((sv * np.diag(H)) * sv)); idx = list(range(n))
        for k in range(n):
            big = k; bigv = d[k]
            for i in range(k + 1, n):
                if d[i] > bigv:
                    bigv = d[i]; big = i
            d[k], d[big] = d[big], d[k]; idx[k], idx[big] = idx[big], idx[k]
        assert np.array_equal(perm, np.array(idx))
        assert np.array_equal(np.isnan(x), np.isnan(xh))
        assert np.array_equal(x[~np.isnan(x)], xh[~np.isnan(xh)])


@pytest.mark.parametrize("n", [36, 68, 100])
def test_zero_matrix_gives_zero(pkg, ctx, n):
    """H = 0: the scaled matrix is zero (all diagonal entries tie), the first pivot is zero, ldltSolveTransposed returns d = 0"""
    H = np.zeros((n, n)); b = np.arange(1.0, n + 1)
    x, perm, branch, zero = pkg.debug_solve(ctx, H, b)
    assert zero == 1 and branch == TIES
    assert np.array_equal(x, np.zeros(n)) and np.array_equal(x, pkg.host_solve_ldlt(ctx.L, H, b))


@pytest.mark.parametrize("n", [12, 20, 28, 44, 52, 60, 76, 84, 92])
def test_every_window_size(pkg, ctx, n):
    """n = 4 + 8 F for F = 1 .. 11 (the sizes the two tests above leave out), with a semidefinite direction (a zero pivot past the first: the column is not divided)"""
    rng = np.random.RandomState(400 + n)
    H = spd(rng, n); b = rng.standard_normal(n)
    x, perm, branch, _ = pkg.debug_solve(ctx, H, b)
    assert np.array_equal(x, pkg.host_solve_ldlt(ctx.L, H, b)) and np.array_equal(perm, literal_pivot_order(H))
    # rank-deficient: two identical rows / columns -> an exactly zero pivot in the trailing block
    H2 = H.copy(); H2[:, 1] = H2[:, 0]; H2[1, :] = H2[0, :]; H2[1, 1] = H2[0, 0]
    x2, perm2, _, _ = pkg.debug_solve(ctx, H2, b)
    xh2 = pkg.host_solve_ldlt(ctx.L, H2, b)
    assert np.array_equal(np.isfinite(x2), np.isfinite(xh2)) and np.array_equal(x2[np.isfinite(x2)], xh2[np.isfinite(xh2)])
