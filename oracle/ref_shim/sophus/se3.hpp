// ORACLE-SIDE TEST INFRASTRUCTURE — stand-in for the vendored Sophus v0.9a (thirdparty/Sophus needs the real Eigen: Quaternion,
// Map specialisations, internal traits).  Same quaternion-backed algorithms, restated in ../../lie.h with the Sophus lines
// they follow (so3.hpp:343-369,497-540, se3.hpp:131-140,169-173,407-428,560-586); here only wrapped in the member surface
// the reference's sources call.  SE3 exp/log are therefore "unpinned" (DESIGN.md §2).
#pragma once
#include "Eigen/Core"
#include "../../lie.h"

namespace Sophus
{
typedef Eigen::Matrix<double, 3, 1> Vector3d_;
typedef Eigen::Matrix<double, 3, 3> Matrix3d_;
typedef Eigen::Matrix<double, 6, 1> Vector6d_;
typedef Eigen::Matrix<double, 6, 6> Matrix6d_;
typedef Eigen::Matrix<double, 4, 4> Matrix4d_;
typedef Eigen::Matrix<double, 3, 4> Matrix34d_;

// quaternion with Eigen's accessor names
struct Quaterniond_
{
	orc::Quat q;
	Quaterniond_() : q{1, 0, 0, 0} {}
	Quaterniond_(double w, double x, double y, double z) : q{w, x, y, z} {}
	explicit Quaterniond_(const orc::Quat& o) : q(o) {}
	explicit Quaterniond_(const Matrix3d_& R)
	{
		// Shepperd's method (the rotation -> quaternion conversion Eigen documents)
		double t = R(0, 0) + R(1, 1) + R(2, 2);
		if (t > 0)
		{
			t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
			q.x = (R(2, 1) - R(1, 2)) * t; q.y = (R(0, 2) - R(2, 0)) * t; q.z = (R(1, 0) - R(0, 1)) * t;
		}
		else
		{
			int i = 0;
			if (R(1, 1) > R(0, 0)) i = 1;
			if (R(2, 2) > R(i, i)) i = 2;
			int j = (i + 1) % 3, k = (j + 1) % 3;
			t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
			double v[3];
			v[i] = 0.5 * t; t = 0.5 / t;
			q.w = (R(k, j) - R(j, k)) * t; v[j] = (R(j, i) + R(i, j)) * t; v[k] = (R(k, i) + R(i, k)) * t;
			q.x = v[0]; q.y = v[1]; q.z = v[2];
		}
	}
	double w() const { return q.w; }
	double x() const { return q.x; }
	double y() const { return q.y; }
	double z() const { return q.z; }
	double& w() { return q.w; }
	double& x() { return q.x; }
	double& y() { return q.y; }
	double& z() { return q.z; }
	Eigen::Matrix<double, 4, 1> coeffs() const { return Eigen::Matrix<double, 4, 1>(q.x, q.y, q.z, q.w); }
	Matrix3d_ toRotationMatrix() const
	{
		double R[9]; orc::qToR(q, R);
		Matrix3d_ m;
		for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = R[r * 3 + c];
		return m;
	}
	Quaterniond_ operator*(const Quaterniond_& o) const { return Quaterniond_(orc::qmul(q, o.q)); }
	Quaterniond_ conjugate() const { return Quaterniond_(orc::qconj(q)); }
	Quaterniond_ inverse() const { return conjugate(); }
	void normalize() { q = orc::qnormalize(q); }
	Quaterniond_ normalized() const { return Quaterniond_(orc::qnormalize(q)); }
	double norm() const { return std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z); }
	double squaredNorm() const { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
	Vector3d_ operator*(const Vector3d_& v) const
	{
		double in[3] = {v(0), v(1), v(2)}, out[3];
		orc::qRot(q, in, out);
		return Vector3d_(out[0], out[1], out[2]);
	}
};
typedef Quaterniond_ Quaterniond;

class SO3d
{
public:
	EIGEN_MAKE_ALIGNED_OPERATOR_NEW;
	typedef Vector3d_ Tangent;
	typedef Vector3d_ Point;
	typedef Matrix3d_ Transformation;
	SO3d() {}
	explicit SO3d(const Quaterniond_& q) : q_(q.normalized()) {}
	explicit SO3d(const Matrix3d_& R) : q_(Quaterniond_(R)) {}
	static SO3d exp(const Vector3d_& om)
	{
		double o[3] = {om(0), om(1), om(2)}, theta;
		SO3d r; r.q_ = Quaterniond_(orc::so3ExpAndTheta(o, &theta));
		return r;
	}
	Vector3d_ log() const { double o[3], theta; orc::so3LogAndTheta(q_.q, o, &theta); return Vector3d_(o[0], o[1], o[2]); }
	static Vector3d_ log(const SO3d& o) { return o.log(); }
	Matrix3d_ matrix() const { return q_.toRotationMatrix(); }
	Matrix3d_ Adj() const { return matrix(); }
	SO3d inverse() const { SO3d r; r.q_ = q_.conjugate(); return r; }
	SO3d operator*(const SO3d& o) const { SO3d r; r.q_ = Quaterniond_(orc::qnormalize(orc::qmul(q_.q, o.q_.q))); return r; }
	SO3d& operator*=(const SO3d& o) { *this = *this * o; return *this; }
	Vector3d_ operator*(const Vector3d_& p) const { return q_ * p; }
	const Quaterniond_& unit_quaternion() const { return q_; }
	void setQuaternion(const Quaterniond_& q) { q_ = q.normalized(); }
	void setQuaternionRaw(const Quaterniond_& q) { q_ = q; }   // stand-in only: takes a unit quaternion over bit for bit (test glue)
	static Matrix3d_ hat(const Vector3d_& o)
	{
		Matrix3d_ O;
		O << 0, -o(2), o(1), o(2), 0, -o(0), -o(1), o(0), 0;
		return O;
	}
private:
	Quaterniond_ q_;
	friend class SE3d;
};

class SE3d
{
public:
	EIGEN_MAKE_ALIGNED_OPERATOR_NEW;
	typedef Vector6d_ Tangent;
	typedef Vector3d_ Point;
	typedef Matrix4d_ Transformation;
	typedef Matrix6d_ Adjoint;
	static const int DoF = 6;
	SE3d() : t_(Vector3d_::Zero()) {}
	SE3d(const SO3d& so3, const Vector3d_& t) : so3_(so3), t_(t) {}
	SE3d(const Matrix3d_& R, const Vector3d_& t) : so3_(R), t_(t) {}
	SE3d(const Quaterniond_& q, const Vector3d_& t) : so3_(q), t_(t) {}
	explicit SE3d(const Matrix4d_& T) : so3_(Matrix3d_(T.topLeftCorner<3, 3>())), t_(T.topRightCorner<3, 1>()) {}

	static SE3d exp(const Vector6d_& a)
	{
		double v[6]; for (int i = 0; i < 6; i++) v[i] = a(i);
		return fromOrc(orc::se3Exp(v));
	}
	Vector6d_ log() const
	{
		double o[6]; orc::se3Log(toOrc(), o);
		Vector6d_ r; for (int i = 0; i < 6; i++) r(i) = o[i];
		return r;
	}
	static Vector6d_ log(const SE3d& T) { return T.log(); }
	SE3d inverse() const { return fromOrc(orc::se3Inv(toOrc())); }
	SE3d operator*(const SE3d& o) const { return fromOrc(orc::se3Mul(toOrc(), o.toOrc())); }
	SE3d& operator*=(const SE3d& o) { *this = *this * o; return *this; }
	Vector3d_ operator*(const Vector3d_& p) const { return so3_ * p + t_; }
	Matrix6d_ Adj() const
	{
		double A[36]; orc::se3Adj(toOrc(), A);
		Matrix6d_ m;
		for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) m(r, c) = A[r * 6 + c];
		return m;
	}
	Matrix3d_ rotationMatrix() const { return so3_.matrix(); }
	Matrix4d_ matrix() const
	{
		Matrix4d_ m = Matrix4d_::Identity();
		m.topLeftCorner<3, 3>() = rotationMatrix();
		m.topRightCorner<3, 1>() = t_;
		return m;
	}
	Matrix34d_ matrix3x4() const
	{
		Matrix34d_ m;
		m.topLeftCorner<3, 3>() = rotationMatrix();
		m.topRightCorner<3, 1>() = t_;
		return m;
	}
	Vector3d_& translation() { return t_; }
	const Vector3d_& translation() const { return t_; }
	SO3d& so3() { return so3_; }
	const SO3d& so3() const { return so3_; }
	const Quaterniond_& unit_quaternion() const { return so3_.unit_quaternion(); }
	void setQuaternion(const Quaterniond_& q) { so3_.setQuaternion(q); }
	void setRotationMatrix(const Matrix3d_& R) { so3_ = SO3d(R); }
	template<typename T> SE3d cast() const { return *this; }
private:
	orc::SE3 toOrc() const
	{
		orc::SE3 r; r.q = so3_.q_.q;
		for (int i = 0; i < 3; i++) r.t[i] = t_(i);
		return r;
	}
	static SE3d fromOrc(const orc::SE3& o)
	{
		SE3d r; r.so3_.q_ = Quaterniond_(o.q);
		r.t_ = Vector3d_(o.t[0], o.t[1], o.t[2]);
		return r;
	}
	SO3d so3_;
	Vector3d_ t_;
};

typedef SE3d SE3;
typedef SO3d SO3;
}  // namespace Sophus

namespace Eigen
{
typedef Sophus::Quaterniond_ Quaterniond;
}
