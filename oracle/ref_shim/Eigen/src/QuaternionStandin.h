// TEST INFRASTRUCTURE (oracle/).  Stand-in for the part of Eigen's Geometry module that the reference's VENDORED Sophus (thirdparty/Sophus/sophus/so3.hpp,
// se3.hpp) is written against, so that those two headers compile UNMODIFIED, from where they lie, into oracle/_ref/libref.so and oracle/_ref/libsophus_pin.so
// (oracle/Makefile.ref, ref_shim/sophus/se3.hpp, oracle/sophus_pin.cpp).  Dense matrices come from the stand-in Core; this header adds Eigen::internal::traits and
// Eigen::Quaternion with the scalar formulas of Eigen 3's Quaternion.h (quat_product, toRotationMatrix, _transformVector, the rotation-matrix constructor)
// written in Eigen's operand order.  What the pin proves: oracle/lie.h == Sophus' own exp / log / product / inverse / Adj code, bit for bit.  What it does
// not prove: that Eigen's quaternion leaf arithmetic (where an SSE build may add in another order) has these bits — DESIGN.md §2 "unpinned for".
#pragma once
#include "../Core"
#include <cmath>

#ifndef EIGEN_DEPRECATED
#define EIGEN_DEPRECATED
#endif
#ifndef EIGEN_INHERIT_ASSIGNMENT_EQUAL_OPERATOR
#define EIGEN_INHERIT_ASSIGNMENT_EQUAL_OPERATOR(X)
#endif

namespace Eigen {
namespace internal {
template <typename T> struct traits;
}

template <typename Scalar_, int Options_ = 0>
class Quaternion {
 public:
  typedef Scalar_ Scalar;
  typedef Matrix<Scalar, 4, 1> Coefficients;   // x, y, z, w (Eigen's storage order)
  typedef Matrix<Scalar, 3, 1> Vector3;
  typedef Matrix<Scalar, 3, 3> Matrix3;

  Quaternion() {}
  Quaternion(const Scalar& w, const Scalar& x, const Scalar& y, const Scalar& z) { m_.data()[0] = x; m_.data()[1] = y; m_.data()[2] = z; m_.data()[3] = w; }
  explicit Quaternion(const Scalar* data) { for (int i = 0; i < 4; i++) m_.data()[i] = data[i]; }
  // QuaternionBase::operator=(MatrixBase) for a 3x3 rotation matrix: quaternionbase_assign_impl<Other,3,3>::run
  Quaternion(const Matrix3& mat) {
    Scalar t = mat(0, 0) + mat(1, 1) + mat(2, 2);
    if (t > Scalar(0)) {
      t = std::sqrt(t + Scalar(1.0));
      w() = Scalar(0.5) * t;
      t = Scalar(0.5) / t;
      x() = (mat(2, 1) - mat(1, 2)) * t;
      y() = (mat(0, 2) - mat(2, 0)) * t;
      z() = (mat(1, 0) - mat(0, 1)) * t;
    } else {
      int i = 0;
      if (mat(1, 1) > mat(0, 0)) i = 1;
      if (mat(2, 2) > mat(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + Scalar(1.0));
      m_.data()[i] = Scalar(0.5) * t;
      t = Scalar(0.5) / t;
      w() = (mat(k, j) - mat(j, k)) * t;
      m_.data()[j] = (mat(j, i) + mat(i, j)) * t;
      m_.data()[k] = (mat(k, i) + mat(i, k)) * t;
    }
  }
  template <typename O> explicit Quaternion(const Quaternion<O, Options_>& o) { for (int i = 0; i < 4; i++) m_.data()[i] = Scalar(o.coeffs().data()[i]); }

  Scalar& x() { return m_.data()[0]; } Scalar& y() { return m_.data()[1]; } Scalar& z() { return m_.data()[2]; } Scalar& w() { return m_.data()[3]; }
  const Scalar& x() const { return m_.data()[0]; } const Scalar& y() const { return m_.data()[1]; } const Scalar& z() const { return m_.data()[2]; } const Scalar& w() const { return m_.data()[3]; }
  Coefficients& coeffs() { return m_; }
  const Coefficients& coeffs() const { return m_; }
  Vector3 vec() const { return Vector3(x(), y(), z()); }

  static Quaternion Identity() { return Quaternion(Scalar(1), Scalar(0), Scalar(0), Scalar(0)); }
  Quaternion& setIdentity() { *this = Identity(); return *this; }
  // Sum order of the four squares: Eigen's depends on its version and instruction set (3.2 unrolls in halves, (x2+y2)+(z2+w2); 3.3+ with SSE2 adds two
  // packets, (x2+z2)+(y2+w2)); this stand-in adds w2, x2, y2, z2 in sequence like oracle/lie.h so that the comparison isolates Sophus' code.
  Scalar squaredNorm() const { return w() * w() + x() * x() + y() * y() + z() * z(); }
  Scalar norm() const { return std::sqrt(squaredNorm()); }
  void normalize() { m_ /= norm(); }
  Quaternion normalized() const { Quaternion r(*this); r.normalize(); return r; }
  Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
  Quaternion inverse() const {
    const Scalar n2 = squaredNorm();
    if (n2 > Scalar(0)) { Quaternion c = conjugate(); c.m_ /= n2; return c; }
    Quaternion r; r.m_.setZero(); return r;
  }
  template <typename NewScalar> Quaternion<NewScalar, Options_> cast() const { return Quaternion<NewScalar, Options_>(*this); }

  // internal::quat_product<Architecture::Generic, ...>::run
  Quaternion operator*(const Quaternion& b) const {
    const Quaternion& a = *this;
    return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                      a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                      a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                      a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  Quaternion& operator*=(const Quaternion& o) { *this = *this * o; return *this; }

  // QuaternionBase::toRotationMatrix
  Matrix3 toRotationMatrix() const {
    Matrix3 res;
    const Scalar tx = Scalar(2) * x(), ty = Scalar(2) * y(), tz = Scalar(2) * z();
    const Scalar twx = tx * w(), twy = ty * w(), twz = tz * w();
    const Scalar txx = tx * x(), txy = ty * x(), txz = tz * x();
    const Scalar tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    res(0, 0) = Scalar(1) - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
    res(1, 0) = txy + twz; res(1, 1) = Scalar(1) - (txx + tzz); res(1, 2) = tyz - twx;
    res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = Scalar(1) - (txx + tyy);
    return res;
  }
  // QuaternionBase::_transformVector:  uv = vec x v;  uv += uv;  v + w * uv + vec x uv
  Vector3 _transformVector(const Vector3& v) const {
    Vector3 uv = vec().cross(v);
    uv += uv;
    return v + this->w() * uv + vec().cross(uv);
  }
  Vector3 operator*(const Vector3& v) const { return _transformVector(v); }

 private:
  Coefficients m_;
};
typedef Quaternion<float> Quaternionf;
typedef Quaternion<double> Quaterniond;

}  // namespace Eigen
