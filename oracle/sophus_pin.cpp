// TEST INFRASTRUCTURE (oracle/).  The reference's VENDORED Sophus (thirdparty/Sophus/sophus/so3.hpp + se3.hpp, v0.9a), compiled unmodified from where it lies
// under /root/reference (through oracle/ref_shim/sophus/se3.hpp) against the stand-in Eigen of oracle/ref_shim (Core + src/QuaternionStandin.h), exported as plain C so that
// tests/test_ref_pin_cpu.py can hold oracle/lie.h (the restatement every oracle uses) against Sophus' own code.  Built by oracle/Makefile.ref into
// oracle/_ref/libsophus_pin.so (git-ignored, travels with gpurun).  Poses cross the boundary as 7 doubles: translation(3), quaternion x, y, z, w.
#include <sophus/se3.hpp>
#include <cstring>

typedef Sophus::SE3Group<double> SE3d_;
typedef Sophus::SO3Group<double> SO3d_;
typedef Eigen::Matrix<double, 6, 1> Vec6_;
typedef Eigen::Matrix<double, 3, 1> Vec3_;

static void put(const SE3d_& T, double* pose7) {
  for (int i = 0; i < 3; i++) pose7[i] = T.translation()(i);
  const Eigen::Quaternion<double>& q = T.unit_quaternion();
  pose7[3] = q.x(); pose7[4] = q.y(); pose7[5] = q.z(); pose7[6] = q.w();
}
// raw member-wise import: no constructor normalisation (SE3Group(Quaternion, Point) would normalise); the bits of a pose produced by `put` come back unchanged
static SE3d_ get(const double* pose7) {
  SE3d_ T;
  std::memcpy(T.data(), pose7 + 3, 4 * sizeof(double));   // so3_.data() = quaternion coefficients x, y, z, w
  for (int i = 0; i < 3; i++) T.translation()(i) = pose7[i];
  return T;
}

extern "C" {
int sophus_pin_abi() { return 1; }
void sophus_se3_exp(const double* a6, double* pose7) {
  Vec6_ a; for (int i = 0; i < 6; i++) a(i) = a6[i];
  put(SE3d_::exp(a), pose7);
}
int sophus_se3_log(const double* pose7, double* a6) {
  try { Vec6_ a = get(pose7).log(); for (int i = 0; i < 6; i++) a6[i] = a(i); return 0; } catch (const Sophus::SophusException&) { return -1; }
}
void sophus_se3_mul(const double* A7, const double* B7, double* out7) { put(get(A7) * get(B7), out7); }
void sophus_se3_inverse(const double* A7, double* out7) { put(get(A7).inverse(), out7); }
void sophus_se3_adj(const double* A7, double* adj36_rowmajor) {
  const Eigen::Matrix<double, 6, 6> A = get(A7).Adj();
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) adj36_rowmajor[6 * r + c] = A(r, c);
}
void sophus_se3_matrix3x4(const double* A7, double* m12_rowmajor) {
  const Eigen::Matrix<double, 3, 4> M = get(A7).matrix3x4();
  for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) m12_rowmajor[4 * r + c] = M(r, c);
}
void sophus_se3_transform(const double* A7, const double* p3, double* out3) {
  Vec3_ p; for (int i = 0; i < 3; i++) p(i) = p3[i];
  const Vec3_ o = get(A7) * p;
  for (int i = 0; i < 3; i++) out3[i] = o(i);
}
// SE3Group(Quaternion, Point): the normalising constructor the reference uses when it builds a pose from stored numbers
void sophus_se3_from_quaternion(const double* pose7_in, double* pose7_out) {
  Vec3_ t; for (int i = 0; i < 3; i++) t(i) = pose7_in[i];
  put(SE3d_(Eigen::Quaternion<double>(pose7_in[6], pose7_in[3], pose7_in[4], pose7_in[5]), t), pose7_out);
}
// SE3Group(Matrix3, Point): rotation-matrix constructor (Eigen's matrix -> quaternion conversion, stand-in arithmetic)
void sophus_se3_from_matrix(const double* R9_rowmajor, const double* t3, double* pose7_out) {
  Eigen::Matrix<double, 3, 3> R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = R9_rowmajor[3 * r + c];
  Vec3_ t; for (int i = 0; i < 3; i++) t(i) = t3[i];
  put(SE3d_(R, t), pose7_out);
}
void sophus_so3_exp(const double* w3, double* quat_xyzw, double* theta) {
  Vec3_ w; for (int i = 0; i < 3; i++) w(i) = w3[i];
  const SO3d_ R = SO3d_::expAndTheta(w, theta);
  const Eigen::Quaternion<double>& q = R.unit_quaternion();
  quat_xyzw[0] = q.x(); quat_xyzw[1] = q.y(); quat_xyzw[2] = q.z(); quat_xyzw[3] = q.w();
}
}
