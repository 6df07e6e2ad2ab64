// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
// PINNED to the reference's vendored Sophus: thirdparty/Sophus/sophus/so3.hpp + se3.hpp compile unmodified (oracle/Makefile.ref -> oracle/_ref/libsophus_pin.so, and
// as part of oracle/_ref/libref.so) and every function below equals them bit for bit (tests/test_ref_pin_cpu.py::test_lie_algebra_against_the_vendored_sophus).
// Unpinned underneath: Eigen's quaternion leaf arithmetic (product, toRotationMatrix, _transformVector, the order in which norm() adds the four squares), which the
// stand-in Eigen/src/QuaternionStandin.h and this file restate alike — DESIGN.md §2.
//
// Minimal double-precision SO3/SE3 restating the vendored Sophus v0.9a used by the
// reference (thirdparty/Sophus/sophus/so3.hpp, se3.hpp).  Quaternion-backed like Sophus:
//   SO3::expAndTheta   so3.hpp:343-369        SO3::logAndTheta  so3.hpp:497-540
//   SO3 * SO3          so3.hpp:165-167,266-269 (quaternion product + normalize)
//   SE3::exp           se3.hpp:407-428        SE3::log          se3.hpp:560-586
//   SE3 * SE3          se3.hpp:160-163,268-271  SE3::inverse    se3.hpp:169-173
//   SE3::Adj           se3.hpp:131-140
// Quaternion->matrix and quaternion*vector follow Eigen's published Quaternion
// formulas (Eigen is a third-party dependency absent from /root/reference).
#pragma once
#include <cmath>

namespace orc {

static const double kSophusEps = 1e-10;  // SophusConstants<double>::epsilon(), sophus.hpp

struct Quat { double w, x, y, z; };

struct SE3 {
  Quat q{1, 0, 0, 0};
  double t[3]{0, 0, 0};
};

// ORC_ALT_EIGEN_LEAF (oracle/_build/liboracle_altleaf.so only; tests/test_leaf_sensitivity_cpu.py): the two quaternion leaf operations with the sums associated the
// way a packet (SSE2) implementation pairs them — products (ww*x + yy*z) -/+ (zz*y - xx*w) per lane pair, the norm as (x2 + z2) + (y2 + w2).  Eigen's own order is
// unpinned (no Eigen on this image); the variant exists to MEASURE how far a different leaf rounding can move tracking / BA results.
inline Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
#ifdef ORC_ALT_EIGEN_LEAF
  r.x = (a.w * b.x + a.y * b.z) - (a.z * b.y - a.x * b.w);
  r.y = (a.w * b.y + a.y * b.w) + (a.z * b.x - a.x * b.z);
  r.z = (a.w * b.z - a.y * b.x) + (a.z * b.w + a.x * b.y);
  r.w = (a.w * b.w - a.y * b.y) - (a.z * b.z + a.x * b.x);
#else
  // Eigen quat_product (generic path): a*b
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
#endif
  return r;
}
inline Quat qnormalize(const Quat& a) {
#ifdef ORC_ALT_EIGEN_LEAF
  double n = std::sqrt((a.x * a.x + a.z * a.z) + (a.y * a.y + a.w * a.w));
#else
  double n = std::sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
#endif
  return Quat{a.w / n, a.x / n, a.y / n, a.z / n};
}
inline Quat qconj(const Quat& a) { return Quat{a.w, -a.x, -a.y, -a.z}; }
// Import of a quaternion across the test / C boundary: one that is unit already to rounding (a pose exported by a running system, whose
// SE3 normalises after every product) keeps its bits; anything else is normalised like Sophus' constructors do.  Normalising twice moves
// last bits, which the sliding-window solve amplifies (condition ~1e10) — see poseFrom7 in dm-vio_amd/csrc/lie_dev.h.
inline Quat qimport(const Quat& q) {
  const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return (std::fabs(n2 - 1.0) <= 1e-14) ? q : qnormalize(q);
}

// Eigen QuaternionBase::toRotationMatrix
inline void qToR(const Quat& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// Eigen QuaternionBase::_transformVector: v + 2w(q x v) + 2 q x (q x v)
inline void qRot(const Quat& q, const double v[3], double out[3]) {
  double uvx = q.y * v[2] - q.z * v[1];
  double uvy = q.z * v[0] - q.x * v[2];
  double uvz = q.x * v[1] - q.y * v[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  out[0] = v[0] + q.w * uvx + (q.y * uvz - q.z * uvy);
  out[1] = v[1] + q.w * uvy + (q.z * uvx - q.x * uvz);
  out[2] = v[2] + q.w * uvz + (q.x * uvy - q.y * uvx);
}

inline Quat so3ExpAndTheta(const double om[3], double* theta) {
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  *theta = std::sqrt(theta_sq);
  const double half_theta = 0.5 * (*theta);
  double imag_factor, real_factor;
  if ((*theta) < kSophusEps) {
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    const double sin_half_theta = std::sin(half_theta);
    imag_factor = sin_half_theta / (*theta);
    real_factor = std::cos(half_theta);
  }
  // SO3Group(Quaternion) ctor normalizes (so3.hpp setQuaternion/normalize)
  return qnormalize(Quat{real_factor, imag_factor * om[0], imag_factor * om[1], imag_factor * om[2]});
}

inline void so3LogAndTheta(const Quat& q, double out[3], double* theta) {
  const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
  const double n = std::sqrt(squared_n);
  const double w = q.w;
  double two_atan_nbyw_by_n;
  if (n < kSophusEps) {
    const double squared_w = w * w;
    two_atan_nbyw_by_n = 2.0 / w - 2.0 * (squared_n) / (w * squared_w);
  } else {
    if (std::fabs(w) < kSophusEps) {
      two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
    } else {
      two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
    }
  }
  *theta = two_atan_nbyw_by_n * n;
  out[0] = two_atan_nbyw_by_n * q.x;
  out[1] = two_atan_nbyw_by_n * q.y;
  out[2] = two_atan_nbyw_by_n * q.z;
}

inline void hat(const double o[3], double O[9]) {
  O[0] = 0;     O[1] = -o[2]; O[2] = o[1];
  O[3] = o[2];  O[4] = 0;     O[5] = -o[0];
  O[6] = -o[1]; O[7] = o[0];  O[8] = 0;
}
inline void mat3mul(const double A[9], const double B[9], double C[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}

// a = [upsilon(3) ; omega(3)]  (translation first, rotation last — se3.hpp:407)
inline SE3 se3Exp(const double a[6]) {
  SE3 r;
  double theta;
  const double* omega = a + 3;
  r.q = so3ExpAndTheta(omega, &theta);
  double Omega[9], Omega_sq[9], V[9];
  hat(omega, Omega);
  mat3mul(Omega, Omega, Omega_sq);
  if (theta < kSophusEps) {
    qToR(r.q, V);
  } else {
    const double theta_sq = theta * theta;
    const double c1 = (1.0 - std::cos(theta)) / theta_sq;
    const double c2 = (theta - std::sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Omega[i] + c2 * Omega_sq[i];
  }
  for (int i = 0; i < 3; i++) r.t[i] = V[i * 3 + 0] * a[0] + V[i * 3 + 1] * a[1] + V[i * 3 + 2] * a[2];
  return r;
}

inline void se3Log(const SE3& T, double out[6]) {
  double theta;
  so3LogAndTheta(T.q, out + 3, &theta);
  double Omega[9], O2[9], Vinv[9];
  hat(out + 3, Omega);
  mat3mul(Omega, Omega, O2);
  if (std::fabs(theta) < kSophusEps) {
    for (int i = 0; i < 9; i++) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Omega[i] + (1. / 12.) * O2[i];
  } else {
    const double c = (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vinv[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Omega[i] + c * O2[i];
  }
  for (int i = 0; i < 3; i++) out[i] = Vinv[i * 3 + 0] * T.t[0] + Vinv[i * 3 + 1] * T.t[1] + Vinv[i * 3 + 2] * T.t[2];
}

inline SE3 se3Mul(const SE3& a, const SE3& b) {
  SE3 r;
  double rt[3];
  qRot(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.q = qnormalize(qmul(a.q, b.q));
  return r;
}
inline SE3 se3Inv(const SE3& a) {
  SE3 r;
  // so3.hpp:173-175: SO3Group(unit_quaternion().conjugate()) — the constructor from a quaternion normalises (so3.hpp:631-633), so the inverse of a pose
  // whose quaternion is unit only to rounding moves in its last bits (found by the pin against the vendored Sophus, tests/test_ref_pin_cpu.py)
  r.q = qnormalize(qconj(a.q));
  double nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
  qRot(r.q, nt, r.t);
  return r;
}
// 6x6 adjoint, row-major: [R, hat(t) R; 0, R]
inline void se3Adj(const SE3& T, double A[36]) {
  double R[9], H[9], HR[9];
  qToR(T.q, R);
  hat(T.t, H);
  mat3mul(H, R, HR);
  for (int i = 0; i < 36; i++) A[i] = 0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      A[r * 6 + c] = R[r * 3 + c];
      A[(r + 3) * 6 + (c + 3)] = R[r * 3 + c];
      A[r * 6 + (c + 3)] = HR[r * 3 + c];
    }
}

}  // namespace orc
