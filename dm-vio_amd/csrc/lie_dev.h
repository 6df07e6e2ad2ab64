// SE3 / small dense algebra used by the device-resident Levenberg–Marquardt loop and by the host
// side of the C ABI.  double precision, __host__ __device__.
//
// Follows the conventions of the reference's Sophus v0.9a (thirdparty/Sophus/sophus/se3.hpp:407-428
// exp with [translation; rotation] tangent order, so3.hpp:343-369 quaternion exp, group product =
// quaternion product + renormalisation so3.hpp:266-269) and of Eigen's pivoted LDLT that
// CoarseTracker.cpp:639-658 calls, so results track the reference CPU path to rounding.
#pragma once
#include <hip/hip_runtime.h>

#define DMV_HD __host__ __device__ __forceinline__

namespace dmv {

// ---- small-footprint fp64 elementary functions -------------------------------------------------------------
// The device libm's double sin/cos/exp carry a Payne-Hanek / double-double path that costs >150 VGPRs; since the LM
// control step shares its kernel with the evaluation loop, that footprint would halve the occupancy of every wave.
// These are the classic fdlibm kernels (Sun Microsystems' freely distributable libm: k_sin.c, k_cos.c, e_exp.c) with a
// Cody-Waite reduction — ~1 ulp for the |x| < 1e5 arguments that occur here (rotation increments, affine exponents).
// DEVICE code only.  The host side of the C ABI (motion hypotheses, the BA's frame states and precalc tables, the host-driven LM of single frames and of the VIO hand-off)
// calls the C library's sin / cos / exp like the reference's Sophus and AffLight do: its pose algebra equals the reference's bit for bit
// (tests/test_ref_pin_cpu.py::test_track_new_coarse_hypothesis_list_bitwise); the device-resident LM can differ from it by an ulp of a pose coefficient per exp.
DMV_HD void dsincos(const double x, double* sn, double* cs) {
#if !defined(__HIP_DEVICE_COMPILE__)
  ::sincos(x, sn, cs);   // one call, as g++ compiles Sophus' sin(h) / cos(h) pairs: glibc's sincos and its separate sin / cos differ in the last bit for ~0.1 % of arguments near 1 rad
  return;
#else
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double n = rint(x * invpio2);
  double r = x - n * pio2_1;
  r = r - n * pio2_1t;
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ks = r + (r * z) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
  const double kc = 1.0 - 0.5 * z + (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const int q = ((int)(long long)n) & 3;
  const double s0 = (q & 1) ? kc : ks, c0 = (q & 1) ? ks : kc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
#endif
}
DMV_HD double dexp(const double x) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return exp(x);
#else
  if (!(x < 709.0)) return x != x ? x : __builtin_huge_val();
  if (x < -745.0) return 0.0;
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const double k = rint(invln2 * x);
  const double hi = x - k * ln2HI, lo = k * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  return ldexp(y, (int)k);
#endif
}

struct Quatd { double w, x, y, z; };
struct Pose { Quatd q; double t[3]; };

DMV_HD Quatd qmul(const Quatd& a, const Quatd& b) {
  Quatd r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
DMV_HD Quatd qnormalize(const Quatd& a) {
  const double n = sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
#if !defined(__HIP_DEVICE_COMPILE__)
  Quatd r = {a.w / n, a.x / n, a.y / n, a.z / n};   // Eigen's normalize(): one division per coefficient — the host side equals the reference bit for bit
#else
  // device (the serial tail of the LM control step, k_ba_solve's frame step): one reciprocal, four products — three IEEE fp64 divisions (~30 dependent instructions each)
  // less per normalisation; differs from the host's quotient by at most an ulp of a coefficient, like the device's sin / cos / exp already do
  const double inv = 1.0 / n;
  Quatd r = {a.w * inv, a.x * inv, a.y * inv, a.z * inv};
#endif
  return r;
}
DMV_HD void quatToR(const Quatd& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
DMV_HD void quatRotate(const Quatd& q, const double v[3], double out[3]) {
  double ux = q.y * v[2] - q.z * v[1];
  double uy = q.z * v[0] - q.x * v[2];
  double uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  out[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  out[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}

// exp of a = [upsilon; omega]
DMV_HD Pose poseExp(const double a[6]) {
  Pose r;
  const double ox = a[3], oy = a[4], oz = a[5];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  const double theta = sqrt(theta_sq);
  double imag, real;
  double sh = 0.0, ch = 1.0;   // sin / cos of theta / 2 (theta >= 1e-10)
  if (theta < 1e-10) {
    const double p4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * p4;
    real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * p4;
  } else {
    dsincos(0.5 * theta, &sh, &ch);
#if !defined(__HIP_DEVICE_COMPILE__)
    imag = sh / theta;
#else
    imag = sh * (1.0 / theta);   // device: the reciprocal is shared with c1 / c2 below (the compiler keeps one)
#endif
    real = ch;
  }
  Quatd q = {real, imag * ox, imag * oy, imag * oz};
  r.q = qnormalize(q);
  // V = I + c1*Om + c2*Om^2   (Om = hat(omega));  theta ~ 0: V = R
  double V[9];
  if (theta < 1e-10) {
    quatToR(r.q, V);
  } else {
#if !defined(__HIP_DEVICE_COMPILE__)
    double sth, cth;
    dsincos(theta, &sth, &cth);
    const double tsq = theta * theta;   // se3.hpp:417 squares the theta that so3's expAndTheta returned (sqrt of the sum of squares): not the sum itself in the last bit
    const double c1 = (1.0 - cth) / tsq;
    const double c2 = (theta - sth) / (tsq * theta);
#else
    // device (the serial tail of the LM control step): sin / cos of theta from the half-angle pair already at hand (sin = 2 s c, 1 - cos = 2 s^2) and ONE reciprocal for the
    // three quotients — a second polynomial evaluation and two ~30-instruction IEEE divisions less on a single lane's dependent chain; within an ulp or two of the host's values,
    // like the device's sin / cos / exp themselves
    const double sth = 2.0 * sh * ch;
    const double inv = 1.0 / theta, inv2 = inv * inv;
    const double c1 = (2.0 * sh * sh) * inv2;
    const double c2 = (theta - sth) * (inv2 * inv);
#endif
    const double O[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    double O2[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3 + 0] * O[0 * 3 + j] + O[i * 3 + 1] * O[1 * 3 + j] + O[i * 3 + 2] * O[2 * 3 + j];
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int i = 0; i < 3; i++) r.t[i] = V[i * 3 + 0] * a[0] + V[i * 3 + 1] * a[1] + V[i * 3 + 2] * a[2];
  return r;
}

DMV_HD Pose poseMul(const Pose& a, const Pose& b) {
  Pose r;
  double rt[3];
  quatRotate(a.q, b.t, rt);
  r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
  r.q = qnormalize(qmul(a.q, b.q));
  return r;
}

DMV_HD Pose poseInv(const Pose& a) {
  Pose r;
  // SO3Group::inverse() builds SO3Group(conjugate), whose constructor normalises (thirdparty/Sophus/sophus/so3.hpp:173-175, 631-633)
  const Quatd c = {a.q.w, -a.q.x, -a.q.y, -a.q.z};
  r.q = qnormalize(c);
  const double nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
  quatRotate(r.q, nt, r.t);
  return r;
}

// 6x6 adjoint [R, hat(t) R; 0, R], row-major (se3.hpp:131-140)
DMV_HD void poseAdj(const Pose& T, double A[36]) {
  double R[9];
  quatToR(T.q, R);
  const double H[9] = {0, -T.t[2], T.t[1], T.t[2], 0, -T.t[0], -T.t[1], T.t[0], 0};
  for (int i = 0; i < 36; i++) A[i] = 0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      A[r * 6 + c] = R[r * 3 + c];
      A[(r + 3) * 6 + c + 3] = R[r * 3 + c];
      A[r * 6 + c + 3] = H[r * 3 + 0] * R[0 * 3 + c] + H[r * 3 + 1] * R[1 * 3 + c] + H[r * 3 + 2] * R[2 * 3 + c];
    }
}

DMV_HD Pose poseFrom7(const double p[7]) {
  Pose T;
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
  Quatd q = {p[6], p[3], p[4], p[5]};
  // A quaternion that is already unit to rounding (it came out of the reference's SE3, which normalises after every product) is taken
  // over bit for bit: normalising it a second time would move its last bits, and the sliding-window solve amplifies such a change
  // (condition ~1e10) into 1e-7 of the poses.  Anything else is normalised like Sophus' constructor does (so3.hpp setQuaternion).
  const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  T.q = (fabs(n2 - 1.0) <= 1e-14) ? q : qnormalize(q);
  return T;
}
DMV_HD void poseTo7(const Pose& T, double p[7]) {
  p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2];
  p[3] = T.q.x; p[4] = T.q.y; p[5] = T.q.z; p[6] = T.q.w;
}

// log of an SE3 element (se3.hpp:560-586, so3.hpp:497-540) — host only (nullspace construction)
inline void poseLogHost(const Pose& T, double out[6]) {
  const double n2 = T.q.x * T.q.x + T.q.y * T.q.y + T.q.z * T.q.z, n = sqrt(n2), w = T.q.w;
  double f;
  if (n < 1e-10) f = 2.0 / w - 2.0 * n2 / (w * w * w);
  else if (fabs(w) < 1e-10) f = (w > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
  else f = 2.0 * atan(n / w) / n;
  const double theta = f * n;
  out[3] = f * T.q.x; out[4] = f * T.q.y; out[5] = f * T.q.z;
  const double O[9] = {0, -out[5], out[4], out[5], 0, -out[3], -out[4], out[3], 0};
  double O2[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
  const double c = fabs(theta) < 1e-10 ? 1.0 / 12.0 : (1.0 - theta / (2.0 * tan(theta / 2.0))) / (theta * theta);
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int j = 0; j < 3; j++) s += (((i == j) ? 1.0 : 0.0) - 0.5 * O[i * 3 + j] + c * O2[i * 3 + j]) * T.t[j];
    out[i] = s;
  }
}

// AffLight::fromToVecExposure (src/dso/util/NumType.h:174-186)
DMV_HD void affFromTo(float exposureF, float exposureT, double aF, double bF, double aT, double bT, double out[2]) {
  if (exposureF == 0 || exposureT == 0) { exposureT = exposureF = 1; }
  const double a = dexp(aT - aF) * exposureT / exposureF;
  out[0] = a;
  out[1] = bT - a * bF;
}

// Symmetric n x n (n <= NMAX) solve by LDL^T with diagonal pivoting (largest |d_kk| first), the
// decomposition CoarseTracker.cpp:639 uses through Eigen.  m: row-major, leading dimension ld, destroyed.
template <int NMAX>
DMV_HD void ldltSolveInPlace(double* m, int ld, double* d /*rhs in, x out*/, int n) {
  int tr[NMAX];
  double temp[NMAX];
  bool zero = false;
  for (int k = 0; k < n; k++) {
    int big = k;
    double bigv = fabs(m[k * ld + k]);
    for (int i = k + 1; i < n; i++) {
      const double v = fabs(m[i * ld + i]);
      if (v > bigv) { bigv = v; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      for (int c = 0; c < k; c++) { double s = m[k * ld + c]; m[k * ld + c] = m[big * ld + c]; m[big * ld + c] = s; }
      for (int r = big + 1; r < n; r++) { double s = m[r * ld + k]; m[r * ld + k] = m[r * ld + big]; m[r * ld + big] = s; }
      { double s = m[k * ld + k]; m[k * ld + k] = m[big * ld + big]; m[big * ld + big] = s; }
      for (int i = k + 1; i < big; i++) { double s = m[i * ld + k]; m[i * ld + k] = m[big * ld + i]; m[big * ld + i] = s; }
    }
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = m[j * ld + j] * m[k * ld + j];
      double s = 0;
      for (int j = 0; j < k; j++) s += m[k * ld + j] * temp[j];
      m[k * ld + k] -= s;
      for (int r = k + 1; r < n; r++) {
        double s2 = 0;
        for (int j = 0; j < k; j++) s2 += m[r * ld + j] * temp[j];
        m[r * ld + k] -= s2;
      }
    }
    const double akk = m[k * ld + k];
    const bool ok = fabs(akk) > 0;
    if (k == 0 && !ok) { zero = true; break; }
    if (ok) for (int r = k + 1; r < n; r++) m[r * ld + k] /= akk;
  }
  if (zero) { for (int i = 0; i < n; i++) d[i] = 0; return; }
  for (int k = 0; k < n; k++) if (tr[k] != k) { double s = d[k]; d[k] = d[tr[k]]; d[tr[k]] = s; }
  for (int i = 0; i < n; i++) {
    double s = d[i];
    for (int j = 0; j < i; j++) s -= m[i * ld + j] * d[j];
    d[i] = s;
  }
  for (int i = 0; i < n; i++) {
    if (fabs(m[i * ld + i]) > 2.2250738585072014e-308) d[i] /= m[i * ld + i]; else d[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = d[i];
    for (int j = i + 1; j < n; j++) s -= m[j * ld + i] * d[j];
    d[i] = s;
  }
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) { double s = d[k]; d[k] = d[tr[k]]; d[tr[k]] = s; }
}

}  // namespace dmv
