// Test infrastructure (tests/test_lie_device_gpu.py builds and runs it on the GPU box): the DEVICE instantiation of csrc/lie_dev.h — SE3 exp, product, inverse as the device-resident
// LM of the tracker (k_track_lm) and k_ba_solve's frame step evaluate them: fdlibm-kernel sin / cos / exp polynomials, sin / cos of theta from the half-angle pair, one
// reciprocal for a quaternion's normalisation — against the HOST instantiation of the same header, which equals the reference's Sophus bit for bit
// (tests/lie_compare.hip).  Prints the largest deviation in units in the last place of the larger coefficient magnitude of each result.
#include <hip/hip_runtime.h>
#include "lie_dev.h"
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
using namespace dmv;
__global__ void k_lie(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x[6], y[6];
  for (int k = 0; k < 6; k++) { x[k] = a[6 * i + k]; y[k] = b[6 * i + k]; }
  const Pose P = poseExp(x), Q = poseExp(y);
  const Pose M = poseMul(P, Q), I = poseInv(P);
  double* o = out + 21 * (size_t)i;
  poseTo7(P, o); poseTo7(M, o + 7); poseTo7(I, o + 14);
}
static double ulps(const double* dev, const double* host, int n) {
  double scale = 0, d = 0;
  for (int k = 0; k < n; k++) { scale = fmax(scale, fabs(host[k])); d = fmax(d, fabs(dev[k] - host[k])); }
  return scale > 0 ? d / (scale * 2.220446049250313e-16) : d;
}
int main() {
  const int n = 1 << 18;
  std::mt19937_64 g(17); std::normal_distribution<double> N(0, 1); std::uniform_real_distribution<double> U(-6, 0.5);
  std::vector<double> a(6 * n), b(6 * n), out(21 * (size_t)n);
  for (int i = 0; i < n; i++) { const double sc = pow(10.0, U(g)); for (int k = 0; k < 6; k++) { a[6 * i + k] = sc * N(g); b[6 * i + k] = 0.5 * N(g); } }
  double *da, *db, *dout;
  if (hipMalloc((void**)&da, 48 * n) != hipSuccess || hipMalloc((void**)&db, 48 * n) != hipSuccess || hipMalloc((void**)&dout, 168 * (size_t)n) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
  hipMemcpy(da, a.data(), 48 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 48 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_lie, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n);
  if (hipMemcpy(out.data(), dout, 168 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  double mq = 0, mt = 0, mmq = 0, mmt = 0, miq = 0, mit = 0, mt_big = 0, mit_big = 0;
  for (int i = 0; i < n; i++) {
    const Pose P = poseExp(&a[6 * i]), Q = poseExp(&b[6 * i]);
    double p7[7], m7[7], i7[7];
    poseTo7(P, p7); poseTo7(poseMul(P, Q), m7); poseTo7(poseInv(P), i7);
    const double* o = &out[21 * (size_t)i];
    mt = fmax(mt, ulps(o, p7, 3)); mq = fmax(mq, ulps(o + 3, p7 + 3, 4));
    const double th = sqrt(a[6 * i + 3] * a[6 * i + 3] + a[6 * i + 4] * a[6 * i + 4] + a[6 * i + 5] * a[6 * i + 5]);
    if (th >= 0.1) { mt_big = fmax(mt_big, ulps(o, p7, 3)); mit_big = fmax(mit_big, ulps(o + 14, i7, 3)); }
    mmt = fmax(mmt, ulps(o + 7, m7, 3)); mmq = fmax(mmq, ulps(o + 10, m7 + 3, 4));
    mit = fmax(mit, ulps(o + 14, i7, 3)); miq = fmax(miq, ulps(o + 17, i7 + 3, 4));
  }
  printf("max ulp (of the result's largest coefficient) over %d tangents 1e-6 .. 3 rad: exp q %.2f t %.2f (t for theta >= 0.1: %.2f) | mul q %.2f t %.2f | inv q %.2f t %.2f (theta >= 0.1: %.2f)\n",
         n, mq, mt, mt_big, mmq, mmt, miq, mit, mit_big);
  // Quaternions: a few ulp everywhere.  Translations: V = I + c1 Om + c2 Om^2 with c1 = (1 - cos theta) / theta^2 — the host (= the reference's Sophus, se3.hpp:417-421) forms
  // 1 - cos(theta), which cancels below theta ~ 1e-3 (relative error 1e-16 / theta^2 on c1, i.e. ~1e-16 / theta on t); the device forms 2 sin^2(theta / 2) / theta^2, which
  // does not.  The two therefore differ by up to ~4e-10 of |t| at theta = 1e-6 (|t| itself ~1e-6 there: 4e-16 absolute); for theta >= 0.1 they agree to ~100 ulp (measured; both sides' theta - sin(theta) also cancels there).
  const bool ok = mq <= 8.0 && mmq <= 8.0 && miq <= 8.0 && mt_big <= 512.0 && mit_big <= 512.0 && mmt <= 512.0 && mt <= 1e-9 / 2.220446049250313e-16 && mit <= 1e-9 / 2.220446049250313e-16;
  return ok ? 0 : 1;
}
