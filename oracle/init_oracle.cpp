// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY PINNED against the reference's CoarseInitializer::calcResAndGS compiled from its own source (oracle/_ref, tests/test_ref_pin_cpu.py::test_initializer_calc_res_and_gs_bitwise):
// per-point outputs, JbBuffer_new, Hsc, bsc bit for bit; H, b, energy to rounding (the reference's own six-worker sums are run-dependent).
// CPU restatement of CoarseInitializer::calcResAndGS (src/dso/FullSystem/CoarseInitializer.cpp:331-624), single worker
// (the reference's IndexThreadReduce splits the points in chunks of 50 over its workers, each with its own Accumulator9; run with
// one worker the order below is the reference's).  Accumulators: oracle/acc9.h.  Interpolators: globalFuncs.h:103-118,160-176.
#include <cmath>
#include <cstring>
#include <vector>
#include "lie.h"
#include "acc9.h"

namespace {
const int patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};
const int patternNum = 8;
const float setting_huberTH = 9;

inline float interp31(const float* mat, float x, float y, int width) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float* bp = mat + 3 * (ix + iy * width);
  return dxdy * bp[3 * (1 + width)] + (dy - dxdy) * bp[3 * width] + (dx - dxdy) * bp[3] + (1 - dx - dy + dxdy) * bp[0];
}
inline void interp33(const float* mat, float x, float y, int width, float out[3]) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float* bp = mat + 3 * (ix + iy * width);
  for (int c = 0; c < 3; c++)
    out[c] = dxdy * bp[3 * (1 + width) + c] + (dy - dxdy) * bp[3 * width + c] + (dx - dxdy) * bp[3 + c] + (1 - dx - dy + dxdy) * bp[c];
}
}  // namespace

extern "C" {

// Point arrays mirror struct Pnt (CoarseInitializer.h:44-83).  Outputs: H_out / H_sc 8x8 row-major, b_out / b_sc 8, res3 = (E.A, alphaEnergy, E.num),
// per point: energy_new[2n], isGood_new[n], maxstep[n], lastHessian_new[n] (only written for good points), JbBuffer_new[10n].
void orc_init_calc_res_and_gs(const float* dI_ref, const float* dI_new, int wl, int hl, const double* Ki9, float fxl, float fyl, float cxl, float cyl,
                              const double* refToNew7, double aff_a, double aff_b, int npts, const float* pu, const float* pv, const float* idepth_new,
                              const float* iR, const unsigned char* isGood, const float* energy /*2n*/, const float* outlierTH, float alphaW, float alphaK,
                              float couplingWeight, double priorY, double priorX, float* H_out, float* b_out, float* H_sc, float* b_sc, float* res3,
                              float* energy_new, unsigned char* isGood_new, float* maxstep_out, float* lastHessian_new, float* JbBuffer_new) {
  orc::SE3 T; T.t[0] = refToNew7[0]; T.t[1] = refToNew7[1]; T.t[2] = refToNew7[2];
  T.q = orc::qimport(orc::Quat{refToNew7[6], refToNew7[3], refToNew7[4], refToNew7[5]});
  double Rd[9], RKid[9];
  orc::qToR(T.q, Rd);
  orc::mat3mul(Rd, Ki9, RKid);
  float RKi[9], t[3];
  for (int i = 0; i < 9; i++) RKi[i] = (float)RKid[i];
  for (int i = 0; i < 3; i++) t[i] = (float)T.t[i];
  const float r2new_aff[2] = {(float)exp(aff_a), (float)aff_b};
  Acc9 acc9; acc9.initialize();
  Acc11 E; E.initialize();
  for (int i = 0; i < npts; i++) {
    maxstep_out[i] = 1e10;
    if (!isGood[i]) {
      E.updateSingle((float)(energy[2 * i]));
      energy_new[2 * i] = energy[2 * i]; energy_new[2 * i + 1] = energy[2 * i + 1];
      isGood_new[i] = 0;
      continue;
    }
    alignas(16) float dp[8][8], dd[8], r[8];   // dp[k][idx]
    float* Jb = JbBuffer_new + 10 * i;
    for (int k = 0; k < 10; k++) Jb[k] = 0;
    bool good = true;
    float en = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      const int dx = patternP[idx][0], dy = patternP[idx][1];
      const float X = pu[i] + dx, Y = pv[i] + dy;
      const float pt0 = RKi[0] * X + RKi[1] * Y + RKi[2] * 1.0f + t[0] * idepth_new[i];
      const float pt1 = RKi[3] * X + RKi[4] * Y + RKi[5] * 1.0f + t[1] * idepth_new[i];
      const float pt2 = RKi[6] * X + RKi[7] * Y + RKi[8] * 1.0f + t[2] * idepth_new[i];
      const float u = pt0 / pt2, v = pt1 / pt2;
      const float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
      const float new_idepth = idepth_new[i] / pt2;
      if (!(Ku > 1 && Kv > 1 && Ku < wl - 2 && Kv < hl - 2 && new_idepth > 0)) { good = false; break; }
      float hit[3];
      interp33(dI_new, Ku, Kv, wl, hit);
      const float rlR = interp31(dI_ref, pu[i] + dx, pv[i] + dy, wl);
      if (!std::isfinite(rlR) || !std::isfinite(hit[0])) { good = false; break; }
      const float residual = hit[0] - r2new_aff[0] * rlR - r2new_aff[1];
      float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      en += hw * residual * residual * (2 - hw);
      const float dxdd = (t[0] - t[2] * u) / pt2, dydd = (t[1] - t[2] * v) / pt2;
      if (hw < 1) hw = sqrtf(hw);
      const float dxInterp = hw * hit[1] * fxl, dyInterp = hw * hit[2] * fyl;
      dp[0][idx] = new_idepth * dxInterp;
      dp[1][idx] = new_idepth * dyInterp;
      dp[2][idx] = -new_idepth * (u * dxInterp + v * dyInterp);
      dp[3][idx] = -u * v * dxInterp - (1 + v * v) * dyInterp;
      dp[4][idx] = (1 + u * u) * dxInterp + u * v * dyInterp;
      dp[5][idx] = -v * dxInterp + u * dyInterp;
      dp[6][idx] = -hw * r2new_aff[0] * rlR;
      dp[7][idx] = -hw * 1;
      dd[idx] = dxInterp * dxdd + dyInterp * dydd;
      r[idx] = hw * residual;
      const float a = dxdd * fxl, b = dydd * fyl;
      const float ms = 1.0f / sqrtf(a * a + b * b);
      if (ms < maxstep_out[i]) maxstep_out[i] = ms;
      for (int k = 0; k < 8; k++) Jb[k] += dp[k][idx] * dd[idx];
      Jb[8] += r[idx] * dd[idx];
      Jb[9] += dd[idx] * dd[idx];
    }
    if (!good || en > outlierTH[i] * 20) {
      E.updateSingle((float)(energy[2 * i]));
      isGood_new[i] = 0;
      energy_new[2 * i] = energy[2 * i]; energy_new[2 * i + 1] = energy[2 * i + 1];
      continue;
    }
    E.updateSingle(en);
    isGood_new[i] = 1;
    energy_new[2 * i] = en;
    energy_new[2 * i + 1] = energy[2 * i + 1];   // energy_new[1] is overwritten below for good points; the struct member keeps its old value until then
    for (int k4 = 0; k4 + 3 < patternNum; k4 += 4) {
      __m128 J[9];
      for (int k = 0; k < 8; k++) J[k] = _mm_load_ps(&dp[k][k4]);
      J[8] = _mm_load_ps(&r[k4]);
      acc9.updateSSE(J);
    }
  }
  acc9.finish();
  E.finish();
  // alpha energy (EAlpha is never updated in the reference: its A stays 0, dso issue 52); accE[0].num grows by npts
  for (int i = 0; i < npts; i++)
    if (isGood_new[i]) energy_new[2 * i + 1] = (idepth_new[i] - 1) * (idepth_new[i] - 1);
  const size_t Enum = E.num + (size_t)npts;
  const float EAlphaA = 0;
  const double tsq = T.t[0] * T.t[0] + T.t[1] * T.t[1] + T.t[2] * T.t[2];
  float alphaEnergy = alphaW * (EAlphaA + tsq * npts);
  float alphaOpt;
  if (alphaEnergy > alphaK * npts) { alphaOpt = 0; alphaEnergy = alphaK * npts; }
  else alphaOpt = alphaW;
  Acc9 acc9SC; acc9SC.initialize();
  for (int i = 0; i < npts; i++) {
    if (!isGood_new[i]) continue;
    float* Jb = JbBuffer_new + 10 * i;
    lastHessian_new[i] = Jb[9];
    Jb[8] += alphaOpt * (idepth_new[i] - 1);
    Jb[9] += alphaOpt;
    if (alphaOpt == 0) { Jb[8] += couplingWeight * (idepth_new[i] - iR[i]); Jb[9] += couplingWeight; }
    Jb[9] = 1 / (1 + Jb[9]);
    float J[9];
    for (int k = 0; k < 9; k++) J[k] = Jb[k];
    acc9SC.updateSingleWeighted(J, Jb[9]);
  }
  acc9SC.finish();
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) { H_out[r * 8 + c] = acc9.H[r][c]; H_sc[r * 8 + c] = acc9SC.H[r][c]; }
    b_out[r] = acc9.H[r][8]; b_sc[r] = acc9SC.H[r][8];
  }
  H_out[0] += alphaOpt * npts; H_out[9] += alphaOpt * npts; H_out[18] += alphaOpt * npts;
  double lg[6];
  orc::se3Log(T, lg);
  const float tlog[3] = {(float)lg[0], (float)lg[1], (float)lg[2]};
  b_out[0] += tlog[0] * alphaOpt * npts; b_out[1] += tlog[1] * alphaOpt * npts; b_out[2] += tlog[2] * alphaOpt * npts;
  // zero prior on the translation (DM-VIO: setting_weightZeroPriorDSOInitY / X, settings.cpp:40-41)
  H_out[9] += priorY; b_out[1] += priorY * T.t[1];
  H_out[0] += priorX; b_out[0] += priorX * T.t[0];
  res3[0] = E.A; res3[1] = alphaEnergy; res3[2] = (float)Enum;
}

}  // extern "C"
