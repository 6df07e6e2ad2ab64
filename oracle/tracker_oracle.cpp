// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// PARITY PINNED: interpolation, projection, accumulators, makeImages, makeCoarseDepthL0, calcRes, calcGSSSE and trackNewestCoarse agree bit for bit with the
// reference's own sources compiled by oracle/Makefile.ref (oracle/_ref/libref.so; tests/test_ref_pin_cpu.py), all 53 trackNewCoarse calls of a recorded live run of
// the reference are reproduced to the last bit (tests/test_ref_replay_cpu.py); unpinned only for Eigen's ldlt and Sophus' exp / log.
//
// Restates the TRACKING half of the hot path of lukasvst/dm-vio, function by function:
//   orc_make_images        <- FrameHessian::makeImages           src/dso/FullSystem/HessianBlocks.cpp:128-191
//   OrcTracker::makeK      <- CoarseTracker::makeK               src/dso/FullSystem/CoarseTracker.cpp:105-134
//   OrcTracker::setRef     <- setCoarseTrackingRef/makeCoarseDepthL0  CoarseTracker.cpp:524-538 / 138-295
//   OrcTracker::calcRes    <- CoarseTracker::calcRes             CoarseTracker.cpp:361-517
//   OrcTracker::calcGS     <- CoarseTracker::calcGSSSE           CoarseTracker.cpp:299-356
//   Acc9                   <- Accumulator9 (SSE, 1k/1M shiftUp)  src/dso/OptimizationBackend/MatrixAccumulators.h:975-1345
//   OrcTracker::track      <- CoarseTracker::trackNewestCoarse   CoarseTracker.cpp:539-770 (useimu=0 branch)
//   interp33               <- getInterpolatedElement33           src/dso/util/globalFuncs.h:103-118
//   affFromTo              <- AffLight::fromToVecExposure        src/dso/util/NumType.h:174-186
// Build: see oracle/Makefile (-O3 -msse2 -ffp-contract=off, matching CMakeLists.txt:45-55: SSE only).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <vector>
#include <emmintrin.h>
#include "lie.h"
#include "acc9.h"
#include "dense.h"

namespace orc {

// ---- constants (src/dso/util/settings.cpp, src/dso/FullSystem/HessianBlocks.h:60-68) ----
static const float SCALE_XI_ROT = 1.0f, SCALE_XI_TRANS = 1.0f, SCALE_A = 10.0f, SCALE_B = 1000.0f;
static const float setting_huberTH = 9;          // settings.cpp:148
static const float setting_coarseCutoffTH = 20;  // settings.cpp:160
static const int PYR_LEVELS = 6;                 // settings.h:52

struct V3f { float v[3]; };

static inline void affFromTo(float exposureF, float exposureT, double aF, double bF, double aT, double bT, double out[2]) {
  if (exposureF == 0 || exposureT == 0) { exposureT = exposureF = 1; }
  double a = std::exp(aT - aF) * exposureT / exposureF;
  double b = bT - a * bF;
  out[0] = a; out[1] = b;
}

static inline V3f interp33(const V3f* mat, float x, float y, int width) {
  int ix = (int)x, iy = (int)y;
  float dx = x - ix, dy = y - iy;
  float dxdy = dx * dy;
  const V3f* bp = mat + ix + iy * width;
  V3f r;
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++)
    r.v[c] = w11 * bp[1 + width].v[c] + w01 * bp[width].v[c] + w10 * bp[1].v[c] + w00 * bp[0].v[c];
  return r;
}

// Accumulator9: 45 upper-triangular sums x 4 SSE lanes, hierarchical 1k/1M shift-up.
// The visual-only LM step of trackNewestCoarse (CoarseTracker.cpp:639-682): damped system, 8 / 7 / 6-dof LDL^T by affineOptModeA / B, extrapolation, scaling, SE3::exp * current.
// inc = the extrapolated increment before the SCALE_* factors, incScaled = after them (zeroed when not finite).  OrcTracker::track calls it every iteration.
static bool coarseUpdateVisual(float affineOptModeA, float affineOptModeB, const double H[64], const double b[8], float extrapFac, float lambda, const SE3& refToNew_current,
                               SE3& refToNew_new, double inc[8], double incScaled[8], double& incNorm) {
  double Hl[64]; memcpy(Hl, H, sizeof(Hl));
  for (int i = 0; i < 8; i++) Hl[i * 8 + i] *= (1 + lambda);
  double nb[8]; for (int i = 0; i < 8; i++) nb[i] = -b[i];
  ldltSolve(Hl, nb, inc, 8);
  if (affineOptModeA < 0 && affineOptModeB < 0) {  // fix a, b
    double H6[36], x6[6];
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) H6[r * 6 + c] = Hl[r * 8 + c];
    ldltSolve(H6, nb, x6, 6);
    for (int i = 0; i < 6; i++) inc[i] = x6[i];
    inc[6] = inc[7] = 0;
  }
  if (!(affineOptModeA < 0) && affineOptModeB < 0) {  // fix b
    double H7[49], x7[7];
    for (int r = 0; r < 7; r++) for (int c = 0; c < 7; c++) H7[r * 7 + c] = Hl[r * 8 + c];
    ldltSolve(H7, nb, x7, 7);
    for (int i = 0; i < 7; i++) inc[i] = x7[i];
    inc[7] = 0;
  }
  if (affineOptModeA < 0 && !(affineOptModeB < 0)) {  // fix a
    double Hs[64], bs[8]; memcpy(Hs, Hl, sizeof(Hs)); memcpy(bs, b, sizeof(bs));
    for (int r = 0; r < 8; r++) Hs[r * 8 + 6] = Hs[r * 8 + 7];
    for (int c = 0; c < 8; c++) Hs[6 * 8 + c] = Hs[7 * 8 + c];
    bs[6] = bs[7];
    double H7[49], nb7[7], x7[7];
    for (int r = 0; r < 7; r++) { for (int c = 0; c < 7; c++) H7[r * 7 + c] = Hs[r * 8 + c]; nb7[r] = -bs[r]; }
    ldltSolve(H7, nb7, x7, 7);
    for (int i = 0; i < 8; i++) inc[i] = 0;
    for (int i = 0; i < 6; i++) inc[i] = x7[i];
    inc[6] = 0; inc[7] = x7[6];
  }
  for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
  for (int i = 0; i < 3; i++) incScaled[i] = inc[i] * SCALE_XI_ROT;
  for (int i = 3; i < 6; i++) incScaled[i] = inc[i] * SCALE_XI_TRANS;
  incScaled[6] = inc[6] * SCALE_A; incScaled[7] = inc[7] * SCALE_B;
  double ssum = 0; for (int i = 0; i < 8; i++) ssum += incScaled[i];
  if (!std::isfinite(ssum)) for (int i = 0; i < 8; i++) incScaled[i] = 0;
  refToNew_new = se3Mul(se3Exp(incScaled), refToNew_current);
  incNorm = 0; for (int i = 0; i < 8; i++) incNorm += inc[i] * inc[i];
  incNorm = std::sqrt(incNorm);
  return std::isfinite(ssum);
}

struct OrcTracker {
  int w[PYR_LEVELS], h[PYR_LEVELS], levels;
  float fx[PYR_LEVELS], fy[PYR_LEVELS], cx[PYR_LEVELS], cy[PYR_LEVELS];
  float Ki[PYR_LEVELS][9];
  std::vector<float> idepth[PYR_LEVELS], weightSums[PYR_LEVELS], weightSums_bak[PYR_LEVELS];
  std::vector<float> pc_u[PYR_LEVELS], pc_v[PYR_LEVELS], pc_idepth[PYR_LEVELS], pc_color[PYR_LEVELS];
  int pc_n[PYR_LEVELS];
  std::vector<float> bw_idepth, bw_u, bw_v, bw_dx, bw_dy, bw_residual, bw_weight, bw_refColor;
  int buf_warped_n;
  // reference frame / new frame state
  const V3f* refDIp[PYR_LEVELS];
  const V3f* newDIp[PYR_LEVELS];
  float ref_exposure, new_exposure;
  double ref_aff_a, ref_aff_b;  // lastRef_aff_g2l
  double lastResiduals[5];
  double lastFlowIndicators[3];
  Acc9 acc;
  // statistics for the CPU baseline
  long n_calcRes, n_calcGS, n_points_evaluated;

  OrcTracker(int ww, int hh, int lv) {
    levels = lv;
    for (int l = 0; l < levels; l++) {
      w[l] = ww >> l; h[l] = hh >> l;
      size_t n = (size_t)w[l] * h[l];
      idepth[l].assign(n, 0); weightSums[l].assign(n, 0); weightSums_bak[l].assign(n, 0);
      pc_u[l].assign(n, 0); pc_v[l].assign(n, 0); pc_idepth[l].assign(n, 0); pc_color[l].assign(n, 0);
      pc_n[l] = 0; refDIp[l] = newDIp[l] = nullptr;
    }
    size_t n0 = (size_t)ww * hh + 8;
    bw_idepth.assign(n0, 0); bw_u.assign(n0, 0); bw_v.assign(n0, 0); bw_dx.assign(n0, 0); bw_dy.assign(n0, 0);
    bw_residual.assign(n0, 0); bw_weight.assign(n0, 0); bw_refColor.assign(n0, 0);
    buf_warped_n = 0; ref_exposure = new_exposure = 1; ref_aff_a = ref_aff_b = 0;
    n_calcRes = n_calcGS = n_points_evaluated = 0;
  }

  void makeK(float fx0, float fy0, float cx0, float cy0) {
    fx[0] = fx0; fy[0] = fy0; cx[0] = cx0; cy[0] = cy0;
    for (int level = 1; level < levels; ++level) {
      fx[level] = fx[level - 1] * 0.5;
      fy[level] = fy[level - 1] * 0.5;
      cx[level] = (cx[0] + 0.5) / ((int)1 << level) - 0.5;
      cy[level] = (cy[0] + 0.5) / ((int)1 << level) - 0.5;
    }
    for (int level = 0; level < levels; ++level) {
      // Eigen Matrix3f::inverse() (cofactors * 1/det) on K = [fx 0 cx; 0 fy cy; 0 0 1]
      const float a = fx[level], e = fy[level], c = cx[level], f = cy[level];
      float c00 = e * 1.0f - f * 0.0f;           // cofactor(0,0)
      float c10 = -(0.0f * 1.0f - f * 0.0f);     // cofactor feeding det along col 0
      float c20 = 0.0f * 0.0f - e * 0.0f;
      float det = a * c00 + 0.0f * c10 + c * c20;
      float invdet = 1.0f / det;
      float* Kiv = Ki[level];
      Kiv[0] = c00 * invdet;                      // (0,0)
      Kiv[1] = (c * 0.0f - 0.0f * 1.0f) * invdet; // (0,1) = cofactor(1,0)
      Kiv[2] = (0.0f * f - c * e) * invdet;       // (0,2)
      Kiv[3] = (f * 0.0f - 0.0f * 1.0f) * invdet; // (1,0)
      Kiv[4] = (a * 1.0f - c * 0.0f) * invdet;    // (1,1)
      Kiv[5] = (c * 0.0f - a * f) * invdet;       // (1,2)
      Kiv[6] = (0.0f * 0.0f - e * 0.0f) * invdet; // (2,0)
      Kiv[7] = (0.0f * 0.0f - a * 0.0f) * invdet; // (2,1)
      Kiv[8] = (a * e - 0.0f * 0.0f) * invdet;    // (2,2)
    }
  }

  // points: active points of the window whose newest residual (target == lastRef) is IN:
  // (Ku, Kv, new_idepth) = centerProjectedTo, hdiF = efPoint->HdiF   (CoarseTracker.cpp:144-161)
  void setRef(const V3f* const* dIpRef, float exposure, double affA, double affB,
              int n, const float* cpu, const float* cpv, const float* cpid, const float* hdiF) {
    for (int l = 0; l < levels; l++) refDIp[l] = dIpRef[l];
    ref_exposure = exposure; ref_aff_a = affA; ref_aff_b = affB;
    memset(idepth[0].data(), 0, sizeof(float) * w[0] * h[0]);
    memset(weightSums[0].data(), 0, sizeof(float) * w[0] * h[0]);
    for (int i = 0; i < n; i++) {
      int u = cpu[i] + 0.5f;
      int v = cpv[i] + 0.5f;
      float new_idepth = cpid[i];
      float weight = sqrtf(1e-3 / (hdiF[i] + 1e-12));
      idepth[0][u + w[0] * v] += new_idepth * weight;
      weightSums[0][u + w[0] * v] += weight;
    }
    for (int lvl = 1; lvl < levels; lvl++) {
      int lvlm1 = lvl - 1;
      int wl = w[lvl], hl = h[lvl], wlm1 = w[lvlm1];
      float* idepth_l = idepth[lvl].data(); float* weightSums_l = weightSums[lvl].data();
      float* idepth_lm = idepth[lvlm1].data(); float* weightSums_lm = weightSums[lvlm1].data();
      for (int y = 0; y < hl; y++)
        for (int x = 0; x < wl; x++) {
          int bidx = 2 * x + 2 * y * wlm1;
          idepth_l[x + y * wl] = idepth_lm[bidx] + idepth_lm[bidx + 1] + idepth_lm[bidx + wlm1] + idepth_lm[bidx + wlm1 + 1];
          weightSums_l[x + y * wl] = weightSums_lm[bidx] + weightSums_lm[bidx + 1] + weightSums_lm[bidx + wlm1] + weightSums_lm[bidx + wlm1 + 1];
        }
    }
    // dilate idepth by 1 (diagonal neighbours) on levels 0,1
    for (int lvl = 0; lvl < 2 && lvl < levels; lvl++) {
      int wh = w[lvl] * h[lvl] - w[lvl];
      int wl = w[lvl];
      float* weightSumsl = weightSums[lvl].data(); float* weightSumsl_bak = weightSums_bak[lvl].data();
      memcpy(weightSumsl_bak, weightSumsl, w[lvl] * h[lvl] * sizeof(float));
      float* idepthl = idepth[lvl].data();
      for (int i = w[lvl] + 1; i < wh - 1; i++) {
        if (weightSumsl_bak[i] <= 0) {
          float sum = 0, num = 0, numn = 0;
          if (weightSumsl_bak[i + 1 + wl] > 0) { sum += idepthl[i + 1 + wl]; num += weightSumsl_bak[i + 1 + wl]; numn++; }
          if (weightSumsl_bak[i - 1 - wl] > 0) { sum += idepthl[i - 1 - wl]; num += weightSumsl_bak[i - 1 - wl]; numn++; }
          if (weightSumsl_bak[i + wl - 1] > 0) { sum += idepthl[i + wl - 1]; num += weightSumsl_bak[i + wl - 1]; numn++; }
          if (weightSumsl_bak[i - wl + 1] > 0) { sum += idepthl[i - wl + 1]; num += weightSumsl_bak[i - wl + 1]; numn++; }
          if (numn > 0) { idepthl[i] = sum / numn; weightSumsl[i] = num / numn; }
        }
      }
    }
    // 4-neighbourhood on the coarser levels
    for (int lvl = 2; lvl < levels; lvl++) {
      int wh = w[lvl] * h[lvl] - w[lvl];
      int wl = w[lvl];
      float* weightSumsl = weightSums[lvl].data(); float* weightSumsl_bak = weightSums_bak[lvl].data();
      memcpy(weightSumsl_bak, weightSumsl, w[lvl] * h[lvl] * sizeof(float));
      float* idepthl = idepth[lvl].data();
      for (int i = w[lvl] + 1; i < wh - 1; i++) {
        if (weightSumsl_bak[i] <= 0) {
          float sum = 0, num = 0, numn = 0;
          if (weightSumsl_bak[i + 1] > 0) { sum += idepthl[i + 1]; num += weightSumsl_bak[i + 1]; numn++; }
          if (weightSumsl_bak[i - 1] > 0) { sum += idepthl[i - 1]; num += weightSumsl_bak[i - 1]; numn++; }
          if (weightSumsl_bak[i + wl] > 0) { sum += idepthl[i + wl]; num += weightSumsl_bak[i + wl]; numn++; }
          if (weightSumsl_bak[i - wl] > 0) { sum += idepthl[i - wl]; num += weightSumsl_bak[i - wl]; numn++; }
          if (numn > 0) { idepthl[i] = sum / numn; weightSumsl[i] = num / numn; }
        }
      }
    }
    // normalise + compact
    for (int lvl = 0; lvl < levels; lvl++) {
      float* weightSumsl = weightSums[lvl].data(); float* idepthl = idepth[lvl].data();
      const V3f* dIRefl = refDIp[lvl];
      int wl = w[lvl], hl = h[lvl];
      int lpc_n = 0;
      float* lpc_u = pc_u[lvl].data(); float* lpc_v = pc_v[lvl].data();
      float* lpc_idepth = pc_idepth[lvl].data(); float* lpc_color = pc_color[lvl].data();
      for (int y = 2; y < hl - 2; y++)
        for (int x = 2; x < wl - 2; x++) {
          int i = x + y * wl;
          if (weightSumsl[i] > 0) {
            idepthl[i] /= weightSumsl[i];
            lpc_u[lpc_n] = x; lpc_v[lpc_n] = y;
            lpc_idepth[lpc_n] = idepthl[i];
            lpc_color[lpc_n] = dIRefl[i].v[0];
            if (!std::isfinite(lpc_color[lpc_n]) || !(idepthl[i] > 0)) { idepthl[i] = -1; continue; }
            lpc_n++;
          } else
            idepthl[i] = -1;
          weightSumsl[i] = 1;
        }
      pc_n[lvl] = lpc_n;
    }
  }

  void setNewFrame(const V3f* const* dIpNew, float exposure) {
    for (int l = 0; l < levels; l++) newDIp[l] = dIpNew[l];
    new_exposure = exposure;
  }

  // rs[6] = {E, numTermsInE, flowT, 0, flowRT, saturatedRatio}
  void calcRes(int lvl, const SE3& refToNew, double affA, double affB, float cutoffTH, double rs[6]) {
    float E = 0;
    int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
    int wl = w[lvl], hl = h[lvl];
    const V3f* dINewl = newDIp[lvl];
    float fxl = fx[lvl], fyl = fy[lvl], cxl = cx[lvl], cyl = cy[lvl];
    double Rd[9]; qToR(refToNew.q, Rd);
    float Rf[9]; for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
    const float* Kil = Ki[lvl];
    float RKi[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        RKi[r * 3 + c] = Rf[r * 3 + 0] * Kil[0 * 3 + c] + Rf[r * 3 + 1] * Kil[1 * 3 + c] + Rf[r * 3 + 2] * Kil[2 * 3 + c];
    float t[3] = {(float)refToNew.t[0], (float)refToNew.t[1], (float)refToNew.t[2]};
    double affd[2]; affFromTo(ref_exposure, new_exposure, ref_aff_a, ref_aff_b, affA, affB, affd);
    float affLL[2] = {(float)affd[0], (float)affd[1]};

    float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
    float maxEnergy = 2 * setting_huberTH * cutoffTH - setting_huberTH * setting_huberTH;

    int nl = pc_n[lvl];
    const float* lpc_u = pc_u[lvl].data(); const float* lpc_v = pc_v[lvl].data();
    const float* lpc_idepth = pc_idepth[lvl].data(); const float* lpc_color = pc_color[lvl].data();
    n_calcRes++; n_points_evaluated += nl;

    for (int i = 0; i < nl; i++) {
      float id = lpc_idepth[i], x = lpc_u[i], y = lpc_v[i];
      float pt0 = RKi[0] * x + RKi[1] * y + RKi[2] * 1.0f + t[0] * id;
      float pt1 = RKi[3] * x + RKi[4] * y + RKi[5] * 1.0f + t[1] * id;
      float pt2 = RKi[6] * x + RKi[7] * y + RKi[8] * 1.0f + t[2] * id;
      float u = pt0 / pt2, v = pt1 / pt2;
      float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
      float new_idepth = id / pt2;

      if (lvl == 0 && i % 32 == 0) {
        float k0 = Kil[0] * x + Kil[1] * y + Kil[2] * 1.0f;
        float k1 = Kil[3] * x + Kil[4] * y + Kil[5] * 1.0f;
        float k2 = Kil[6] * x + Kil[7] * y + Kil[8] * 1.0f;
        float r0 = RKi[0] * x + RKi[1] * y + RKi[2] * 1.0f;
        float r1 = RKi[3] * x + RKi[4] * y + RKi[5] * 1.0f;
        float r2 = RKi[6] * x + RKi[7] * y + RKi[8] * 1.0f;
        // translation only (positive)
        float ptT0 = k0 + t[0] * id, ptT1 = k1 + t[1] * id, ptT2 = k2 + t[2] * id;
        float KuT = fxl * (ptT0 / ptT2) + cxl, KvT = fyl * (ptT1 / ptT2) + cyl;
        // translation only (negative)
        float pT20 = k0 - t[0] * id, pT21 = k1 - t[1] * id, pT22 = k2 - t[2] * id;
        float KuT2 = fxl * (pT20 / pT22) + cxl, KvT2 = fyl * (pT21 / pT22) + cyl;
        // translation and rotation (negative)
        float p30 = r0 - t[0] * id, p31 = r1 - t[1] * id, p32 = r2 - t[2] * id;
        float Ku3 = fxl * (p30 / p32) + cxl, Kv3 = fyl * (p31 / p32) + cyl;
        sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
        sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
        sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
        sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
        sumSquaredShiftNum += 2;
      }

      if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;

      float refColor = lpc_color[i];
      V3f hitColor = interp33(dINewl, Ku, Kv, wl);
      if (!std::isfinite((float)hitColor.v[0])) continue;
      float residual = hitColor.v[0] - (float)(affLL[0] * refColor + affLL[1]);
      float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);

      if (fabs(residual) > cutoffTH) {
        E += maxEnergy; numTermsInE++; numSaturated++;
      } else {
        E += hw * residual * residual * (2 - hw);
        numTermsInE++;
        bw_idepth[numTermsInWarped] = new_idepth;
        bw_u[numTermsInWarped] = u; bw_v[numTermsInWarped] = v;
        bw_dx[numTermsInWarped] = hitColor.v[1]; bw_dy[numTermsInWarped] = hitColor.v[2];
        bw_residual[numTermsInWarped] = residual; bw_weight[numTermsInWarped] = hw;
        bw_refColor[numTermsInWarped] = lpc_color[i];
        numTermsInWarped++;
      }
    }
    while (numTermsInWarped % 4 != 0) {
      bw_idepth[numTermsInWarped] = 0; bw_u[numTermsInWarped] = 0; bw_v[numTermsInWarped] = 0;
      bw_dx[numTermsInWarped] = 0; bw_dy[numTermsInWarped] = 0; bw_residual[numTermsInWarped] = 0;
      bw_weight[numTermsInWarped] = 0; bw_refColor[numTermsInWarped] = 0;
      numTermsInWarped++;
    }
    buf_warped_n = numTermsInWarped;
    rs[0] = E; rs[1] = numTermsInE;
    rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
    rs[3] = 0;
    rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
    rs[5] = numSaturated / (float)numTermsInE;
  }

  // H_out 8x8 row-major, b_out 8 (double)
  void calcGS(int lvl, double affA, double affB, double H_out[64], double b_out[8]) {
    acc.initialize();
    n_calcGS++;
    __m128 fxl = _mm_set1_ps(fx[lvl]);
    __m128 fyl = _mm_set1_ps(fy[lvl]);
    __m128 b0 = _mm_set1_ps((float)ref_aff_b);
    double affd[2]; affFromTo(ref_exposure, new_exposure, ref_aff_a, ref_aff_b, affA, affB, affd);
    __m128 a = _mm_set1_ps((float)(affd[0]));
    __m128 one = _mm_set1_ps(1), minusOne = _mm_set1_ps(-1), zero = _mm_set1_ps(0);
    int n = buf_warped_n;
    for (int i = 0; i < n; i += 4) {
      __m128 dx = _mm_mul_ps(_mm_loadu_ps(&bw_dx[i]), fxl);
      __m128 dy = _mm_mul_ps(_mm_loadu_ps(&bw_dy[i]), fyl);
      __m128 u = _mm_loadu_ps(&bw_u[i]);
      __m128 v = _mm_loadu_ps(&bw_v[i]);
      __m128 id = _mm_loadu_ps(&bw_idepth[i]);
      __m128 J[9];
      J[0] = _mm_mul_ps(id, dx);
      J[1] = _mm_mul_ps(id, dy);
      J[2] = _mm_sub_ps(zero, _mm_mul_ps(id, _mm_add_ps(_mm_mul_ps(u, dx), _mm_mul_ps(v, dy))));
      J[3] = _mm_sub_ps(zero, _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dx), _mm_mul_ps(dy, _mm_add_ps(one, _mm_mul_ps(v, v)))));
      J[4] = _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dy), _mm_mul_ps(dx, _mm_add_ps(one, _mm_mul_ps(u, u))));
      J[5] = _mm_sub_ps(_mm_mul_ps(u, dy), _mm_mul_ps(v, dx));
      J[6] = _mm_mul_ps(a, _mm_sub_ps(b0, _mm_loadu_ps(&bw_refColor[i])));
      J[7] = minusOne;
      J[8] = _mm_loadu_ps(&bw_residual[i]);
      acc.updateSSE_eighted(J, _mm_loadu_ps(&bw_weight[i]));
    }
    acc.finish();
    const float invn = 1.0f / n;
    for (int r = 0; r < 8; r++) {
      for (int c = 0; c < 8; c++) H_out[r * 8 + c] = (double)acc.H[r][c] * invn;
      b_out[r] = (double)acc.H[r][8] * invn;
    }
    const float sc[8] = {SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_A, SCALE_B};
    for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) H_out[r * 8 + c] *= sc[c];
    for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) H_out[r * 8 + c] *= sc[r];
    for (int r = 0; r < 8; r++) b_out[r] *= sc[r];
  }

  // trackNewestCoarse, useimu=0 branch.  affineOptModeA/B follow settings.cpp:139-140 (1e12 / 1e8 by default).
  bool track(SE3& lastToNew_out, double aff_io[2], int coarsestLvl, const double minResForAbort[5],
             float affineOptModeA, float affineOptModeB, double H[64], double b[8], int* iterationsOut) {
    for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = 1000;
    int maxIterations[] = {10, 20, 50, 50, 50};
    float lambdaExtrapolationLimit = 0.001;
    SE3 refToNew_current = lastToNew_out;
    double affA_cur = aff_io[0], affB_cur = aff_io[1];
    bool haveRepeated = false;
    int totalIts = 0;
    for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
      float levelCutoffRepeat = 1;
      double resOld[6];
      calcRes(lvl, refToNew_current, affA_cur, affB_cur, setting_coarseCutoffTH * levelCutoffRepeat, resOld);
      while (resOld[5] > 0.6 && (levelCutoffRepeat < 50 || resOld[5] > 0.99)) {
        levelCutoffRepeat *= 2;
        calcRes(lvl, refToNew_current, affA_cur, affB_cur, setting_coarseCutoffTH * levelCutoffRepeat, resOld);
      }
      calcGS(lvl, affA_cur, affB_cur, H, b);
      float lambda = 0.01;
      for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
        float extrapFac = 1;
        if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));
        double inc[8], incScaled[8], incNorm;
        SE3 refToNew_new;
        coarseUpdateVisual(affineOptModeA, affineOptModeB, H, b, extrapFac, lambda, refToNew_current, refToNew_new, inc, incScaled, incNorm);
        double affA_new = affA_cur + incScaled[6], affB_new = affB_cur + incScaled[7];

        double resNew[6];
        calcRes(lvl, refToNew_new, affA_new, affB_new, setting_coarseCutoffTH * levelCutoffRepeat, resNew);
        bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
        totalIts++;
        if (accept) {
          calcGS(lvl, affA_new, affB_new, H, b);
          memcpy(resOld, resNew, sizeof(resOld));
          affA_cur = affA_new; affB_cur = affB_new;
          refToNew_current = refToNew_new;
          lambda *= 0.5;
        } else {
          lambda *= 4;
          if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
        }
        if (!(incNorm > 1e-3)) break;
      }
      lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
      lastFlowIndicators[0] = resOld[2]; lastFlowIndicators[1] = resOld[3]; lastFlowIndicators[2] = resOld[4];
      if (std::isnan(lastResiduals[lvl])) { if (iterationsOut) *iterationsOut = totalIts; return false; }
      if (lastResiduals[lvl] > 1.5 * minResForAbort[lvl]) { if (iterationsOut) *iterationsOut = totalIts; return false; }
      if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    }
    lastToNew_out = refToNew_current;
    aff_io[0] = affA_cur; aff_io[1] = affB_cur;
    bool trackingGood = true;
    if ((affineOptModeA != 0 && (fabsf((float)aff_io[0]) > 1.2)) || (affineOptModeB != 0 && (fabsf((float)aff_io[1]) > 200)))
      trackingGood = false;
    double rel[2]; affFromTo(ref_exposure, new_exposure, ref_aff_a, ref_aff_b, aff_io[0], aff_io[1], rel);
    float relAff0 = (float)rel[0], relAff1 = (float)rel[1];
    if ((affineOptModeA == 0 && (fabsf(logf(relAff0)) > 1.5)) || (affineOptModeB == 0 && (fabsf(relAff1) > 200)))
      trackingGood = false;
    if (affineOptModeA < 0) aff_io[0] = 0;
    if (affineOptModeB < 0) aff_io[1] = 0;
    if (iterationsOut) *iterationsOut = totalIts;
    return trackingGood;
  }
};

}  // namespace orc

using namespace orc;

// ------------------------------------------------------------------------------------------------
// flat C interface for ctypes (tests / bench cpu_baseline)
// ------------------------------------------------------------------------------------------------
extern "C" {

// pyramid-level rule of setGlobalCalib (src/dso/util/globalCalib.cpp:47-55)
int orc_pyr_levels(int w, int h) {
  int wlvl = w, hlvl = h, used = 1;
  while (wlvl % 2 == 0 && hlvl % 2 == 0 && wlvl * hlvl > 5000 && used < PYR_LEVELS) { wlvl /= 2; hlvl /= 2; used++; }
  return used;
}

// makeImages: color[w*h] -> per level dIp (float3 AoS), abs (absSquaredGrad, identity gamma).
// dIp_out[l] must hold 3*w_l*h_l floats; abs_out[l] w_l*h_l floats (may be NULL).
// Rows 0 and h-1 of dx/dy (never written by the reference) are zero here.
void orc_make_images_gamma(const float* color, int w0, int h0, int levels, float** dIp_out, float** abs_out, const float* B256);
void orc_make_images(const float* color, int w0, int h0, int levels, float** dIp_out, float** abs_out) { orc_make_images_gamma(color, w0, h0, levels, dIp_out, abs_out, nullptr); }
// B256 = CalibHessian::B: absSquaredGrad is weighted by getBGradOnly(color)^2 (HessianBlocks.cpp:184-188, HessianBlocks.h:394-400); NULL = unweighted
void orc_make_images_gamma(const float* color, int w0, int h0, int levels, float** dIp_out, float** abs_out, const float* B256) {
  for (int lvl = 0; lvl < levels; lvl++) {
    int wl = w0 >> lvl, hl = h0 >> lvl;
    V3f* dI_l = (V3f*)dIp_out[lvl];
    memset(dI_l, 0, sizeof(V3f) * wl * hl);
    float* dabs_l = abs_out ? abs_out[lvl] : nullptr;
    if (dabs_l) memset(dabs_l, 0, sizeof(float) * wl * hl);
    if (lvl == 0) {
      for (int i = 0; i < wl * hl; i++) dI_l[i].v[0] = color[i];
    } else {
      int wlm1 = w0 >> (lvl - 1);
      const V3f* dI_lm = (const V3f*)dIp_out[lvl - 1];
      for (int y = 0; y < hl; y++)
        for (int x = 0; x < wl; x++)
          dI_l[x + y * wl].v[0] = 0.25f * (dI_lm[2 * x + 2 * y * wlm1].v[0] + dI_lm[2 * x + 1 + 2 * y * wlm1].v[0] +
                                           dI_lm[2 * x + 2 * y * wlm1 + wlm1].v[0] + dI_lm[2 * x + 1 + 2 * y * wlm1 + wlm1].v[0]);
    }
    for (int idx = wl; idx < wl * (hl - 1); idx++) {
      float dx = 0.5f * (dI_l[idx + 1].v[0] - dI_l[idx - 1].v[0]);
      float dy = 0.5f * (dI_l[idx + wl].v[0] - dI_l[idx - wl].v[0]);
      if (!std::isfinite(dx)) dx = 0;
      if (!std::isfinite(dy)) dy = 0;
      dI_l[idx].v[1] = dx; dI_l[idx].v[2] = dy;
      if (dabs_l) {
        dabs_l[idx] = dx * dx + dy * dy;
        if (B256) {
          int c = dI_l[idx].v[0] + 0.5f;
          if (c < 5) c = 5;
          if (c > 250) c = 250;
          float gw = B256[c + 1] - B256[c];
          dabs_l[idx] *= gw * gw;
        }
      }
    }
  }
}

void* orc_tracker_create(int w, int h) { return new OrcTracker(w, h, orc_pyr_levels(w, h)); }
void orc_tracker_destroy(void* p) { delete (OrcTracker*)p; }
int orc_tracker_levels(void* p) { return ((OrcTracker*)p)->levels; }
void orc_tracker_make_k(void* p, float fx, float fy, float cx, float cy) { ((OrcTracker*)p)->makeK(fx, fy, cx, cy); }
void orc_tracker_get_k(void* p, int lvl, float out[4], float Ki[9]) {
  OrcTracker* t = (OrcTracker*)p;
  out[0] = t->fx[lvl]; out[1] = t->fy[lvl]; out[2] = t->cx[lvl]; out[3] = t->cy[lvl];
  memcpy(Ki, t->Ki[lvl], sizeof(float) * 9);
}
void orc_tracker_set_ref(void* p, float** dIpRef, float exposure, double affA, double affB,
                         int n, const float* u, const float* v, const float* idepth, const float* hdiF) {
  ((OrcTracker*)p)->setRef((const V3f* const*)dIpRef, exposure, affA, affB, n, u, v, idepth, hdiF);
}
void orc_tracker_set_new(void* p, float** dIpNew, float exposure) { ((OrcTracker*)p)->setNewFrame((const V3f* const*)dIpNew, exposure); }
int orc_tracker_pc_n(void* p, int lvl) { return ((OrcTracker*)p)->pc_n[lvl]; }
void orc_tracker_get_pc(void* p, int lvl, float* u, float* v, float* idepth, float* color) {
  OrcTracker* t = (OrcTracker*)p; int n = t->pc_n[lvl];
  memcpy(u, t->pc_u[lvl].data(), n * 4); memcpy(v, t->pc_v[lvl].data(), n * 4);
  memcpy(idepth, t->pc_idepth[lvl].data(), n * 4); memcpy(color, t->pc_color[lvl].data(), n * 4);
}
void orc_tracker_get_idepth(void* p, int lvl, float* idepth, float* weightSums) {
  OrcTracker* t = (OrcTracker*)p; size_t n = (size_t)t->w[lvl] * t->h[lvl];
  memcpy(idepth, t->idepth[lvl].data(), n * 4); memcpy(weightSums, t->weightSums[lvl].data(), n * 4);
}
static SE3 poseFrom7(const double p7[7]) {  // tx ty tz qx qy qz qw
  SE3 T; T.t[0] = p7[0]; T.t[1] = p7[1]; T.t[2] = p7[2];
  T.q = qimport(Quat{p7[6], p7[3], p7[4], p7[5]});
  return T;
}
static void poseTo7(const SE3& T, double p7[7]) {
  p7[0] = T.t[0]; p7[1] = T.t[1]; p7[2] = T.t[2]; p7[3] = T.q.x; p7[4] = T.q.y; p7[5] = T.q.z; p7[6] = T.q.w;
}
void orc_tracker_calc_res(void* p, int lvl, const double pose7[7], const double aff[2], float cutoffTH, double rs[6]) {
  ((OrcTracker*)p)->calcRes(lvl, poseFrom7(pose7), aff[0], aff[1], cutoffTH, rs);
}
int orc_tracker_warped_n(void* p) { return ((OrcTracker*)p)->buf_warped_n; }
// out: 8 arrays of buf_warped_n floats: idepth,u,v,dx,dy,residual,weight,refColor
void orc_tracker_get_warped(void* p, float* out) {
  OrcTracker* t = (OrcTracker*)p; int n = t->buf_warped_n;
  const std::vector<float>* src[8] = {&t->bw_idepth, &t->bw_u, &t->bw_v, &t->bw_dx, &t->bw_dy, &t->bw_residual, &t->bw_weight, &t->bw_refColor};
  for (int k = 0; k < 8; k++) memcpy(out + (size_t)k * n, src[k]->data(), n * 4);
}
void orc_tracker_calc_gs(void* p, int lvl, const double aff[2], double H[64], double b[8]) {
  ((OrcTracker*)p)->calcGS(lvl, aff[0], aff[1], H, b);
}
// returns trackingGood (1/0).  pose7/aff in-out (only written on success like the reference),
// lastResiduals[5], lastFlow[3], H[64], b[8] = state at exit.
int orc_tracker_track(void* p, double pose7[7], double aff[2], int coarsestLvl, const double minResForAbort[5],
                      float affineOptModeA, float affineOptModeB,
                      double lastResiduals[5], double lastFlow[3], double H[64], double b[8], int* iterations) {
  OrcTracker* t = (OrcTracker*)p;
  SE3 T = poseFrom7(pose7);
  bool good = t->track(T, aff, coarsestLvl, minResForAbort, affineOptModeA, affineOptModeB, H, b, iterations);
  poseTo7(T, pose7);
  memcpy(lastResiduals, t->lastResiduals, sizeof(double) * 5);
  memcpy(lastFlow, t->lastFlowIndicators, sizeof(double) * 3);
  return good ? 1 : 0;
}
void orc_tracker_stats(void* p, long out[3]) {
  OrcTracker* t = (OrcTracker*)p; out[0] = t->n_calcRes; out[1] = t->n_calcGS; out[2] = t->n_points_evaluated;
}

// FullSystem::trackNewCoarse (src/dso/FullSystem/FullSystem.cpp:300-539), visual-only path without IMU hint.
// Hypothesis list (:364-402): constant / double / half / zero motion, zero motion from the keyframe, 26 small rotations
// (note the reference's `for(rotDelta=0.02; rotDelta<0.05; rotDelta++)` runs once).  Inputs are camToWorld poses.
int orc_make_track_hypotheses(const double slast_c2w[7], const double sprelast_c2w[7], const double lastF_c2w[7], double* out /* 31 x 7 */) {
  SE3 slast = poseFrom7(slast_c2w), sprelast = poseFrom7(sprelast_c2w), lastF = poseFrom7(lastF_c2w);
  SE3 slast_2_sprelast = se3Mul(se3Inv(sprelast), slast);
  SE3 lastF_2_slast = se3Mul(se3Inv(slast), lastF);
  SE3 fh_2_slast = slast_2_sprelast;
  std::vector<SE3> tries;
  SE3 fhInv = se3Inv(fh_2_slast);
  tries.push_back(se3Mul(fhInv, lastF_2_slast));
  tries.push_back(se3Mul(se3Mul(fhInv, fhInv), lastF_2_slast));
  { double lg[6]; se3Log(fh_2_slast, lg); for (int i = 0; i < 6; i++) lg[i] *= 0.5; tries.push_back(se3Mul(se3Inv(se3Exp(lg)), lastF_2_slast)); }
  tries.push_back(lastF_2_slast);
  tries.push_back(SE3());
  const double d = (double)0.02f;   // `float rotDelta=0.02` widened by Sophus::Quaterniond (FullSystem.cpp:376)
  const double rot[26][3] = {{d,0,0},{0,d,0},{0,0,d},{-d,0,0},{0,-d,0},{0,0,-d},{d,d,0},{0,d,d},{d,0,d},{-d,d,0},{0,-d,d},{-d,0,d},{d,-d,0},{0,d,-d},{d,0,-d},
                             {-d,-d,0},{0,-d,-d},{-d,0,-d},{-d,-d,-d},{-d,-d,d},{-d,d,-d},{-d,d,d},{d,-d,-d},{d,-d,d},{d,d,-d},{d,d,d}};
  SE3 base = se3Mul(fhInv, lastF_2_slast);
  for (int k = 0; k < 26; k++) {
    SE3 R; R.q = qnormalize(Quat{1, rot[k][0], rot[k][1], rot[k][2]});  // SE3(Quaterniond(1,x,y,z), 0): Sophus normalises the quaternion
    tries.push_back(se3Mul(base, R));
  }
  for (size_t i = 0; i < tries.size(); i++) poseTo7(tries[i], out + 7 * i);
  return (int)tries.size();
}
// the try loop (:419-489): returns the index of the winning hypothesis (or -1), outputs as the reference leaves them
int orc_tracker_track_new_coarse(void* p, int n_tries, const double* tries7, const double aff_last[2], double lastCoarseRMSE_io[5], double reTrackThreshold,
                                 double pose7_out[7], double aff_out[2], double flow_out[3], int* tries_used, int* tracking_good) {
  OrcTracker* t = (OrcTracker*)p;
  double achievedRes[5]; for (int i = 0; i < 5; i++) achievedRes[i] = NAN;
  bool haveOneGood = false, trackingGoodRet = false;
  int winner = -1, used = 0;
  double flow[3] = {100, 100, 100};
  SE3 best; double bestAff[2] = {0, 0};
  for (int i = 0; i < n_tries; i++) {
    double aff[2] = {aff_last[0], aff_last[1]};
    SE3 T = poseFrom7(tries7 + 7 * i);
    double H[64], b[8]; int its;
    bool good = t->track(T, aff, t->levels - 1, achievedRes, 1e12f, 1e8f, H, b, &its);
    used++;
    if (good) trackingGoodRet = true;
    if (good && std::isfinite((float)t->lastResiduals[0]) && !(t->lastResiduals[0] >= achievedRes[0])) {
      for (int k = 0; k < 3; k++) flow[k] = t->lastFlowIndicators[k];
      bestAff[0] = aff[0]; bestAff[1] = aff[1]; best = T; haveOneGood = true; winner = i;
    }
    if (haveOneGood)
      for (int k = 0; k < 5; k++)
        if (!std::isfinite((float)achievedRes[k]) || achievedRes[k] > t->lastResiduals[k]) achievedRes[k] = t->lastResiduals[k];
    if (haveOneGood && achievedRes[0] < lastCoarseRMSE_io[0] * reTrackThreshold) break;
  }
  if (!haveOneGood) { for (int k = 0; k < 3; k++) flow[k] = 0; bestAff[0] = aff_last[0]; bestAff[1] = aff_last[1]; best = poseFrom7(tries7); }
  for (int k = 0; k < 5; k++) lastCoarseRMSE_io[k] = achievedRes[k];
  poseTo7(best, pose7_out); aff_out[0] = bestAff[0]; aff_out[1] = bestAff[1];
  for (int k = 0; k < 3; k++) flow_out[k] = flow[k];
  if (tries_used) *tries_used = used;
  if (tracking_good) *tracking_good = trackingGoodRet ? 1 : 0;
  return winner;
}

// --- Lie helpers exposed for tests ---
void orc_coarse_update_visual(float affineOptModeA, float affineOptModeB, const double H[64], const double b[8], float extrapFac, float lambda, const double pose7_cur[7],
                               double pose7_new[7], double* incA, double* incB, double* incNorm) {
  double inc[8], incScaled[8], nn; SE3 nxt;
  const bool finite = coarseUpdateVisual(affineOptModeA, affineOptModeB, H, b, extrapFac, lambda, poseFrom7(pose7_cur), nxt, inc, incScaled, nn);
  poseTo7(nxt, pose7_new);
  *incA = finite ? inc[6] : 0.0; *incB = finite ? inc[7] : 0.0; *incNorm = nn;
}
void orc_se3_exp(const double a[6], double pose7[7]) { poseTo7(se3Exp(a), pose7); }
void orc_se3_log(const double pose7[7], double a[6]) { se3Log(poseFrom7(pose7), a); }
void orc_se3_mul(const double a7[7], const double b7[7], double out7[7]) { poseTo7(se3Mul(poseFrom7(a7), poseFrom7(b7)), out7); }
void orc_se3_inv(const double a7[7], double out7[7]) { poseTo7(se3Inv(poseFrom7(a7)), out7); }
void orc_se3_matrix(const double a7[7], double R[9], double t[3]) { SE3 T = poseFrom7(a7); qToR(T.q, R); memcpy(t, T.t, 24); }
void orc_se3_adj(const double a7[7], double A[36]) { se3Adj(poseFrom7(a7), A); }
void orc_ldlt_solve(const double* A, const double* rhs, double* x, int n) { ldltSolve(A, rhs, x, n); }

// --- primitives exposed for the pinning tests (tests/test_ref_pin_cpu.py compares them with the reference's own compiled code) ---
void orc_interp33(const float* img3, int width, int n, const float* x, const float* y, float* out3) {
  for (int i = 0; i < n; i++) { V3f r = interp33((const V3f*)img3, x[i], y[i], width); out3[3 * i] = r.v[0]; out3[3 * i + 1] = r.v[1]; out3[3 * i + 2] = r.v[2]; }
}
void orc_aff_from_to(float eF, float eT, double aF, double bF, double aT, double bT, double out2[2]) { affFromTo(eF, eT, aF, bF, aT, bT, out2); }
// Accumulator9 fed like ref_acc9_stream (oracle/ref_glue.cpp): n4 groups of 4 weighted points, then nsingle single weighted points
void orc_acc9_stream(int n4, const float* J, const float* w, int nsingle, const float* Js, const float* ws, int reps, float* H81, long* num) {
  Acc9 acc; acc.initialize();
  for (int rep = 0; rep < reps; rep++)
  for (int g = 0; g < n4; g++) {
    __m128 j[9];
    for (int k = 0; k < 9; k++) j[k] = _mm_setr_ps(J[(4 * g + 0) * 9 + k], J[(4 * g + 1) * 9 + k], J[(4 * g + 2) * 9 + k], J[(4 * g + 3) * 9 + k]);
    acc.updateSSE_eighted(j, _mm_setr_ps(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]));
  }
  for (int i = 0; i < nsingle; i++) { float j[9]; memcpy(j, Js + 9 * i, 36); acc.updateSingleWeighted(j, ws[i]); }
  acc.finish();
  for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) H81[r * 9 + c] = acc.H[r][c];
  *num = (long)acc.num;
}

}  // extern "C"
