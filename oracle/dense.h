// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.h header).  UNPINNED third-party arithmetic: the compiled reference (oracle/_ref) uses the same restatement through oracle/ref_shim, so upstream Eigen's last bits are not checked.
//
// Small dense double-precision helpers restating the Eigen 3 routines the reference calls at the
// edge of the hot path.  Eigen is a system dependency of the reference (version unpinned by
// CMakeLists.txt:16, absent from /root/reference); the algorithm restated here is Eigen's published
// LDLT<_, Lower>: in-place robust Cholesky with symmetric diagonal pivoting
// (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked and LDLT::_solve_impl).
// Call sites in the reference: CoarseTracker.cpp:639-658 (8x8/7x7/6x6), EnergyFunctional.cpp:971-973.
#pragma once
#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>

namespace orc {

// Solve A x = rhs with Eigen-style pivoted LDLT.  A is n x n row-major symmetric (only lower read).
#ifdef ORC_ALT_EIGEN_LEAF
// oracle/_build/liboracle_altleaf.so only (tests/test_leaf_sensitivity_cpu.py): a DIFFERENT factorisation of the same systems — LDL^T without pivoting, the inner products
// accumulated from the far end — to measure how far the solver's internal rounding can move tracking / BA results.  The systems are symmetric positive definite (damped).
inline void ldltSolve(const double* Ain, const double* rhs, double* x, int n) {
  std::vector<double> L(Ain, Ain + n * n), d(n), y(rhs, rhs + n);
  auto M = [&](int r, int c) -> double& { return L[r * n + c]; };
  for (int j = 0; j < n; j++) {
    double s = 0;
    for (int k = j - 1; k >= 0; k--) s += M(j, k) * M(j, k) * d[k];
    d[j] = M(j, j) - s;
    for (int i = j + 1; i < n; i++) {
      double t = 0;
      for (int k = j - 1; k >= 0; k--) t += M(i, k) * M(j, k) * d[k];
      M(i, j) = d[j] != 0 ? (M(i, j) - t) / d[j] : 0.0;
    }
  }
  for (int i = 0; i < n; i++) { double s = y[i]; for (int k = i - 1; k >= 0; k--) s -= M(i, k) * y[k]; y[i] = s; }
  for (int i = 0; i < n; i++) y[i] = d[i] != 0 ? y[i] / d[i] : 0.0;
  for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = n - 1; k > i; k--) s -= M(k, i) * y[k]; y[i] = s; }
  for (int i = 0; i < n; i++) x[i] = y[i];
}
#else
inline void ldltSolve(const double* Ain, const double* rhs, double* x, int n) {
  std::vector<double> m(Ain, Ain + n * n);
  std::vector<int> tr(n);
  std::vector<double> temp(n);
  auto M = [&](int r, int c) -> double& { return m[r * n + c]; };
  bool zeroMatrix = false;
  for (int k = 0; k < n; k++) {
    // largest |diag| in the trailing corner
    int big = k;
    double bigv = std::fabs(M(k, k));
    for (int i = k + 1; i < n; i++) {
      double v = std::fabs(M(i, i));
      if (v > bigv) { bigv = v; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      // symmetric swap of rows/cols k and big in lower-triangular storage
      int s = n - big - 1;
      for (int c = 0; c < k; c++) std::swap(M(k, c), M(big, c));
      for (int r = 0; r < s; r++) std::swap(M(big + 1 + r, k), M(big + 1 + r, big));
      std::swap(M(k, k), M(big, big));
      for (int i = k + 1; i < big; i++) std::swap(M(i, k), M(big, i));
    }
    int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += M(k, j) * temp[j];
      M(k, k) -= s;
      for (int r = 0; r < rs; r++) {
        double s2 = 0;
        for (int j = 0; j < k; j++) s2 += M(k + 1 + r, j) * temp[j];
        M(k + 1 + r, k) -= s2;
      }
    }
    double akk = M(k, k);
    bool pivot_ok = std::fabs(akk) > 0;
    if (k == 0 && !pivot_ok) { zeroMatrix = true; for (int j = 0; j < n; j++) tr[j] = j; break; }
    if (rs > 0 && pivot_ok)
      for (int r = 0; r < rs; r++) M(k + 1 + r, k) /= akk;
  }
  // solve: x = P^T L^-T D^-1 L^-1 P rhs
  std::vector<double> d(rhs, rhs + n);
  if (zeroMatrix) { for (int i = 0; i < n; i++) x[i] = 0; return; }
  for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  for (int i = 0; i < n; i++) {
    double s = d[i];
    for (int j = 0; j < i; j++) s -= M(i, j) * d[j];
    d[i] = s;
  }
  const double tol = DBL_MIN;  // (std::numeric_limits<double>::min)()
  for (int i = 0; i < n; i++) {
    if (std::fabs(M(i, i)) > tol) d[i] /= M(i, i); else d[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = d[i];
    for (int j = i + 1; j < n; j++) s -= M(j, i) * d[j];
    d[i] = s;
  }
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  for (int i = 0; i < n; i++) x[i] = d[i];
}
#endif

}  // namespace orc
