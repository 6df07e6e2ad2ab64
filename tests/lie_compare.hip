// Test infrastructure (tests/test_host_algebra_cpu.py builds and runs it, host code only): the HOST side of csrc/lie_dev.h — SE3 exp, log, product, inverse, adjoint as the
// C ABI's host code uses them, compiled by hipcc like the library — against the oracle's lie.h as g++ compiled it into liboracle.so (orc_se3_*), which is pinned to the
// reference's vendored Sophus; 200000 random tangents from 1e-4 to ~3 rad, compared bit for bit.
#include "lie_dev.h"
#include <cstdio>
#include <cstring>
#include <random>
using namespace dmv;
extern "C" {
void orc_se3_exp(const double a[6], double pose7[7]);
void orc_se3_log(const double pose7[7], double a[6]);
void orc_se3_mul(const double a7[7], const double b7[7], double out7[7]);
void orc_se3_inv(const double a7[7], double out7[7]);
void orc_se3_adj(const double a7[7], double A[36]);
}
static bool same(const double* a, const double* b, int n) { return memcmp(a, b, 8 * n) == 0; }
int main() {
  std::mt19937_64 g(5); std::normal_distribution<double> N(0, 1); std::uniform_real_distribution<double> U(-4, 0.5);
  long bad_exp = 0, bad_log = 0, bad_mul = 0, bad_inv = 0, bad_adj = 0, n = 200000;
  for (long it = 0; it < n; it++) {
    const double sc = pow(10.0, U(g));
    double a[6], b[6]; for (int i = 0; i < 6; i++) { a[i] = sc * N(g); b[i] = 0.5 * N(g); }
    const Pose P = poseExp(a), Q = poseExp(b);
    double p7[7], q7[7], o7[7], r7[7];
    poseTo7(P, p7); poseTo7(Q, q7);
    orc_se3_exp(a, o7); if (!same(p7, o7, 7)) { if (bad_exp < 3) printf("exp differs at scale %.3g\n", sc); bad_exp++; continue; }
    orc_se3_exp(b, o7); if (!same(q7, o7, 7)) { bad_exp++; continue; }
    double l1[6], l2[6]; poseLogHost(P, l1); orc_se3_log(p7, l2); if (!same(l1, l2, 6)) bad_log++;
    poseTo7(poseMul(P, Q), r7); orc_se3_mul(p7, q7, o7); if (!same(r7, o7, 7)) bad_mul++;
    poseTo7(poseInv(P), r7); orc_se3_inv(p7, o7); if (!same(r7, o7, 7)) bad_inv++;
    double A1[36], A2[36]; poseAdj(P, A1); orc_se3_adj(p7, A2); if (!same(A1, A2, 36)) bad_adj++;
  }
  printf("of %ld: exp %ld log %ld mul %ld inv %ld adj %ld\n", n, bad_exp, bad_log, bad_mul, bad_inv, bad_adj);
  return (bad_exp || bad_log || bad_mul || bad_inv || bad_adj) ? 1 : 0;
}
