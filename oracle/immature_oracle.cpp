// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY PINNED: constructor, traceNewCoarse tables + traceOn through four keyframes and optimizeImmaturePoint agree bit for bit with the reference's own members run on its own FullSystem (oracle/_ref, tests/test_ref_pin_cpu.py::test_immature_points_bitwise).
// CPU restatement of the immature-point path of DM-VIO / DSO:
//   orc_immature_init   <- ImmaturePoint::ImmaturePoint        src/dso/FullSystem/ImmaturePoint.cpp:34-62
//   orc_immature_trace  <- ImmaturePoint::traceOn              src/dso/FullSystem/ImmaturePoint.cpp:76-437
//   orc_trace_precalc   <- FullSystem::traceNewCoarse          src/dso/FullSystem/FullSystem.cpp:541-584 (per-host KRKi, Kt, aff)
//   orc_immature_optimize <- FullSystem::optimizeImmaturePoint src/dso/FullSystem/FullSystemOptPoint.cpp:51-205
//                            ImmaturePoint::linearizeResidual   src/dso/FullSystem/ImmaturePoint.cpp:498-565
//                            projectPoint / derive_idepth       src/dso/FullSystem/ResidualProjections.h:36-87
//   orc_pair_precalc    <- FrameFramePrecalc::set              src/dso/FullSystem/HessianBlocks.cpp:193-223 (PRE_RTll, PRE_tTll, PRE_aff_mode)
// Interpolators: getInterpolatedElement31 / 33 / 33BiLin       src/dso/util/globalFuncs.h:160-176,103-118,203-227
// Settings: src/dso/util/settings.cpp:111-112,159,178-187,296 (pattern 8), src/dso/util/settings.h:227-228.
// Images are the reference's Eigen::Vector3f (I, dx, dy) level-0 arrays as produced by orc_make_images.
#include <cmath>
#include <cstring>
#include <algorithm>
#include "lie.h"

namespace {

const int patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};
const int patternNum = 8;
enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };

const float setting_outlierTH = 12 * 12, setting_outlierTHSumComponent = 50 * 50, setting_overallEnergyTHWeight = 1;
const float setting_maxPixSearch = 0.027f, setting_huberTH = 9;
const int setting_minTraceTestRadius = 2, setting_trace_GNIterations = 3;
const float setting_trace_stepsize = 1.0f, setting_trace_GNThreshold = 0.1f, setting_trace_extraSlackOnTH = 1.2f;
const float setting_trace_slackInterval = 1.5f, setting_trace_minImprovementFactor = 2;

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
inline void interp33BiLin(const float* mat, float x, float y, int width, float out[3]) {
  const int ix = (int)x, iy = (int)y;
  const float* bp = mat + 3 * (ix + iy * width);
  const float tl = bp[0], tr = bp[3], bl = bp[3 * width], br = bp[3 * (width + 1)];
  const float dx = x - ix, dy = y - iy;
  const float topInt = dx * tr + (1 - dx) * tl, botInt = dx * br + (1 - dx) * bl;
  const float leftInt = dy * bl + (1 - dy) * tl, rightInt = dy * br + (1 - dy) * tr;
  out[0] = dx * rightInt + (1 - dx) * leftInt; out[1] = rightInt - leftInt; out[2] = botInt - topInt;
}

struct ImmState {   // the mutable part of ImmaturePoint
  float idepth_min, idepth_max, quality, lastTraceU, lastTraceV, lastTracePixelInterval;
  int lastTraceStatus;
};

int traceOn(const float* dI, int w, int h, float u, float v, const float* color, const float* weights, const float* gradH /*00 01 10 11*/,
            float energyTH, const float* KRKi /*row-major 3x3*/, const float* Kt, const float* aff, ImmState& s) {
  if (s.lastTraceStatus == IPS_OOB) return s.lastTraceStatus;
  const float maxPixSearch = (w + h) * setting_maxPixSearch;
  // 1. both ends of the depth interval must project inside the image
  const float pr[3] = {KRKi[0] * u + KRKi[1] * v + KRKi[2] * 1.0f, KRKi[3] * u + KRKi[4] * v + KRKi[5] * 1.0f, KRKi[6] * u + KRKi[7] * v + KRKi[8] * 1.0f};
  const float ptpMin[3] = {pr[0] + Kt[0] * s.idepth_min, pr[1] + Kt[1] * s.idepth_min, pr[2] + Kt[2] * s.idepth_min};
  const float uMin = ptpMin[0] / ptpMin[2], vMin = ptpMin[1] / ptpMin[2];
  int maxRotPatX = 0, maxRotPatY = 0;
  float rotPat[8][2];
  for (int idx = 0; idx < patternNum; idx++) {
    rotPat[idx][0] = KRKi[0] * (float)patternP[idx][0] + KRKi[1] * (float)patternP[idx][1];
    rotPat[idx][1] = KRKi[3] * (float)patternP[idx][0] + KRKi[4] * (float)patternP[idx][1];
    const int absX = (int)std::abs(rotPat[idx][0]), absY = (int)std::abs(rotPat[idx][1]);
    maxRotPatX = std::max(absX, maxRotPatX); maxRotPatY = std::max(absY, maxRotPatY);
  }
  const int boundU = std::max(4, maxRotPatX + 2), boundV = std::max(4, maxRotPatY + 2);
  auto oob = [&]() { s.lastTraceU = -1; s.lastTraceV = -1; s.lastTracePixelInterval = 0; return s.lastTraceStatus = IPS_OOB; };
  if (!(uMin > boundU && vMin > boundV && uMin < w - boundU - 1 && vMin < h - boundV - 1)) return oob();
  float dist, uMax, vMax;
  if (std::isfinite(s.idepth_max)) {
    const float ptpMax[3] = {pr[0] + Kt[0] * s.idepth_max, pr[1] + Kt[1] * s.idepth_max, pr[2] + Kt[2] * s.idepth_max};
    uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
    if (!(uMax > boundU && vMax > boundV && uMax < w - boundU - 1 && vMax < h - boundV - 1)) return oob();
    // 2. an interval that already spans less than slackInterval pixels is left alone
    dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
    dist = sqrtf(dist);
    if (dist < setting_trace_slackInterval) {
      s.lastTraceU = (uMax + uMin) * 0.5f; s.lastTraceV = (vMax + vMin) * 0.5f;
      s.lastTracePixelInterval = dist;
      return s.lastTraceStatus = IPS_SKIPPED;
    }
  } else {
    dist = maxPixSearch;
    // project to arbitrary depth to get direction.
    const float ptpMax[3] = {pr[0] + Kt[0] * 0.01f, pr[1] + Kt[1] * 0.01f, pr[2] + Kt[2] * 0.01f};
    uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
    const float dx = uMax - uMin, dy = vMax - vMin;
    const float d = 1.0f / sqrtf(dx * dx + dy * dy);
    uMax = uMin + dist * dx * d; vMax = vMin + dist * dy * d;
    if (!(uMax > boundU && vMax > boundV && uMax < w - boundU - 1 && vMax < h - boundV - 1)) return oob();
  }
  // set OOB if scale change too big.
  if (!(s.idepth_min < 0 || (ptpMin[2] > 0.75f && ptpMin[2] < 1.5f))) return oob();
  // 3. localisation error from the gradient structure tensor; no trace when it cannot shrink the interval
  float dx = setting_trace_stepsize * (uMax - uMin), dy = setting_trace_stepsize * (vMax - vMin);
  const float a = (dx * gradH[0] + dy * gradH[2]) * dx + (dx * gradH[1] + dy * gradH[3]) * dy;
  const float b = (dy * gradH[0] + (-dx) * gradH[2]) * dy + (dy * gradH[1] + (-dx) * gradH[3]) * (-dx);
  float errorInPixel = 0.2f + 0.2f * (a + b) / a;
  if (errorInPixel * setting_trace_minImprovementFactor > dist && std::isfinite(s.idepth_max)) {
    s.lastTraceU = (uMax + uMin) * 0.5f; s.lastTraceV = (vMax + vMin) * 0.5f;
    s.lastTracePixelInterval = dist;
    return s.lastTraceStatus = IPS_BADCONDITION;
  }
  if (errorInPixel > 10) errorInPixel = 10;
  // 4. discrete search along the epipolar segment
  dx /= dist; dy /= dist;
  if (dist > maxPixSearch) { uMax = uMin + maxPixSearch * dx; vMax = vMin + maxPixSearch * dy; dist = maxPixSearch; }
  int numSteps = 1.9999f + dist / setting_trace_stepsize;
  const float randShift = uMin * 1000 - floorf(uMin * 1000);
  float ptx = uMin - randShift * dx, pty = vMin - randShift * dy;
  if (!std::isfinite(dx) || !std::isfinite(dy)) { s.lastTracePixelInterval = 0; s.lastTraceU = -1; s.lastTraceV = -1; return s.lastTraceStatus = IPS_OOB; }
  float errors[100];
  float bestU = 0, bestV = 0, bestEnergy = 1e10;
  int bestIdx = -1;
  if (numSteps >= 100) numSteps = 99;
  for (int i = 0; i < numSteps; i++) {
    float energy = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      const float hitColor = interp31(dI, (float)(ptx + rotPat[idx][0]), (float)(pty + rotPat[idx][1]), w);
      if (!std::isfinite(hitColor)) { energy += 1e5; continue; }
      const float residual = hitColor - (float)(aff[0] * color[idx] + aff[1]);
      const float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      energy += hw * residual * residual * (2 - hw);
    }
    errors[i] = energy;
    if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = i; }
    ptx += dx; pty += dy;
  }
  // find best score outside a +-2px radius.
  float secondBest = 1e10;
  for (int i = 0; i < numSteps; i++)
    if ((i < bestIdx - setting_minTraceTestRadius || i > bestIdx + setting_minTraceTestRadius) && errors[i] < secondBest) secondBest = errors[i];
  const float newQuality = secondBest / bestEnergy;
  if (newQuality < s.quality || numSteps > 10) s.quality = newQuality;
  // 5. Gauss-Newton refinement along the line
  float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
  if (setting_trace_GNIterations > 0) bestEnergy = 1e5;
  for (int it = 0; it < setting_trace_GNIterations; it++) {
    float H = 1, bb = 0, energy = 0;
    for (int idx = 0; idx < patternNum; idx++) {
      const float posU = (float)(bestU + rotPat[idx][0]), posV = (float)(bestV + rotPat[idx][1]);
      if (posU < 0 || posV < 0 || posU >= w - 1 || posV >= h - 1) return oob();
      float hit[3];
      interp33(dI, posU, posV, w, hit);
      if (!std::isfinite(hit[0])) { energy += 1e5; continue; }
      const float residual = hit[0] - (aff[0] * color[idx] + aff[1]);
      const float dResdDist = dx * hit[1] + dy * hit[2];
      const float hw = fabs(residual) < setting_huberTH ? 1 : setting_huberTH / fabs(residual);
      H += hw * dResdDist * dResdDist;
      bb += hw * residual * dResdDist;
      energy += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
    }
    if (energy > bestEnergy) {
      stepBack *= 0.5;
      bestU = uBak + stepBack * dx; bestV = vBak + stepBack * dy;
    } else {
      float step = -gnstepsize * bb / H;
      if (step < -0.5) step = -0.5; else if (step > 0.5) step = 0.5;
      if (!std::isfinite(step)) step = 0;
      uBak = bestU; vBak = bestV; stepBack = step;
      bestU += step * dx; bestV += step * dy;
      bestEnergy = energy;
    }
    if (fabsf(stepBack) < setting_trace_GNThreshold) break;
  }
  // 6. energy-based outlier test
  if (!(bestEnergy < energyTH * setting_trace_extraSlackOnTH)) {
    s.lastTracePixelInterval = 0; s.lastTraceU = -1; s.lastTraceV = -1;
    if (s.lastTraceStatus == IPS_OUTLIER) return s.lastTraceStatus = IPS_OOB;
    return s.lastTraceStatus = IPS_OUTLIER;
  }
  // 7. new inverse-depth interval from the refined position +- the localisation error
  if (dx * dx > dy * dy) {
    s.idepth_min = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    s.idepth_max = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    s.idepth_min = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    s.idepth_max = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (s.idepth_min > s.idepth_max) std::swap(s.idepth_min, s.idepth_max);
  if (!std::isfinite(s.idepth_min) || !std::isfinite(s.idepth_max) || (s.idepth_max < 0)) {
    s.lastTracePixelInterval = 0; s.lastTraceU = -1; s.lastTraceV = -1;
    return s.lastTraceStatus = IPS_OUTLIER;
  }
  s.lastTracePixelInterval = 2 * errorInPixel;
  s.lastTraceU = bestU; s.lastTraceV = bestV;
  return s.lastTraceStatus = IPS_GOOD;
}


enum { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };   // ResState (Residuals.h:44)
const float setting_minIdepthH_act = 100;
const int setting_GNItsOnPointActivation = 3;
const float SCALE_IDEPTH = 1.0f;

struct TmpRes { int state_state, state_NewState; double state_energy, state_NewEnergy; };
struct PairPre { const float* R; const float* t; const float* aff; const float* dI; };
struct CalibF { float fxl, fyl, cxl, cyl, fxli, fyli; int w, h; };

double linearizeResidual(const CalibF& C, const PairPre& pre, float pu, float pv, const float* color, const float* weights, float energyTH, float outlierTHSlack,
                         TmpRes& tmp, float& Hdd, float& bd, float idepth) {
  if (tmp.state_state == RS_OOB) { tmp.state_NewState = RS_OOB; return tmp.state_energy; }
  float energyLeft = 0;
  const float wM3G = C.w - 3, hM3G = C.h - 3;
  for (int idx = 0; idx < patternNum; idx++) {
    const int dx = patternP[idx][0], dy = patternP[idx][1];
    // projectPoint (ResidualProjections.h:61-87)
    const float Kl0 = (pu + dx - C.cxl) * C.fxli, Kl1 = (pv + dy - C.cyl) * C.fyli, Kl2 = 1;
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = pre.R[r * 3 + 0] * Kl0 + pre.R[r * 3 + 1] * Kl1 + pre.R[r * 3 + 2] * Kl2 + pre.t[r] * idepth;
    const float drescale = 1.0f / ptp[2];
    bool ok = drescale > 0;
    float u = 0, v = 0, Ku = 0, Kv = 0;
    if (ok) {
      u = ptp[0] * drescale; v = ptp[1] * drescale;
      Ku = u * C.fxl + C.cxl; Kv = v * C.fyl + C.cyl;
      ok = Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
    }
    if (!ok) { tmp.state_NewState = RS_OOB; return tmp.state_energy; }
    float hit[3];
    interp33(pre.dI, Ku, Kv, C.w, hit);
    if (!std::isfinite(hit[0])) { tmp.state_NewState = RS_OOB; return tmp.state_energy; }
    const float residual = hit[0] - (pre.aff[0] * color[idx] + pre.aff[1]);
    float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);
    energyLeft += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
    // depth derivatives.
    const float dxInterp = hit[1] * C.fxl, dyInterp = hit[2] * C.fyl;
    const float d_idepth = (dxInterp * drescale * (pre.t[0] - pre.t[2] * u) + dyInterp * drescale * (pre.t[1] - pre.t[2] * v)) * SCALE_IDEPTH;
    hw *= weights[idx] * weights[idx];
    Hdd += (hw * d_idepth) * d_idepth;
    bd += (hw * residual) * d_idepth;
  }
  if (energyLeft > energyTH * outlierTHSlack) { energyLeft = energyTH * outlierTHSlack; tmp.state_NewState = RS_OUTLIER; }
  else tmp.state_NewState = RS_IN;
  tmp.state_NewEnergy = energyLeft;
  return energyLeft;
}

// returns 1 = activated (idepth_out, res_state valid), 0 = not well-constrained (keep immature), -1 = delete the point
int optimizeImmaturePoint(const CalibF& C, int nres, const PairPre* pre, float pu, float pv, const float* color, const float* weights, float energyTH,
                          float idepth_min, float idepth_max, int minObs, float& idepth_out, int* res_state) {
  TmpRes residuals[16];
  for (int i = 0; i < nres; i++) { residuals[i].state_NewEnergy = residuals[i].state_energy = 0; residuals[i].state_NewState = RS_OUTLIER; residuals[i].state_state = RS_IN; }
  float lastEnergy = 0, lastHdd = 0, lastbd = 0;
  float currentIdepth = (idepth_max + idepth_min) * 0.5f;
  for (int i = 0; i < nres; i++) {
    lastEnergy += linearizeResidual(C, pre[i], pu, pv, color, weights, energyTH, 1000, residuals[i], lastHdd, lastbd, currentIdepth);
    residuals[i].state_state = residuals[i].state_NewState;
    residuals[i].state_energy = residuals[i].state_NewEnergy;
  }
  idepth_out = currentIdepth;
  if (!std::isfinite(lastEnergy) || lastHdd < setting_minIdepthH_act) return 0;
  float lambda = 0.1;
  for (int iteration = 0; iteration < setting_GNItsOnPointActivation; iteration++) {
    float H = lastHdd;
    H *= 1 + lambda;
    const float step = (1.0 / H) * lastbd;
    const float newIdepth = currentIdepth - step;
    float newHdd = 0, newbd = 0, newEnergy = 0;
    for (int i = 0; i < nres; i++) newEnergy += linearizeResidual(C, pre[i], pu, pv, color, weights, energyTH, 1, residuals[i], newHdd, newbd, newIdepth);
    if (!std::isfinite(lastEnergy) || newHdd < setting_minIdepthH_act) return 0;
    if (newEnergy < lastEnergy) {
      currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
      for (int i = 0; i < nres; i++) { residuals[i].state_state = residuals[i].state_NewState; residuals[i].state_energy = residuals[i].state_NewEnergy; }
      lambda *= 0.5;
    } else lambda *= 5;
    if (fabsf(step) < 0.0001 * currentIdepth) break;
  }
  idepth_out = currentIdepth;
  for (int i = 0; i < nres; i++) res_state[i] = residuals[i].state_state;
  if (!std::isfinite(currentIdepth)) return -1;
  int numGoodRes = 0;
  for (int i = 0; i < nres; i++) if (residuals[i].state_state == RS_IN) numGoodRes++;
  if (numGoodRes < minObs) return -1;
  if (!std::isfinite(energyTH)) return -1;   // PointHessian::energyTH copied from the immature point (HessianBlocks.cpp:39-57)
  return 1;
}

}  // namespace

extern "C" {

// ImmaturePoint constructor for n candidate pixels (integer u, v) of one host image: color[8n], weights[8n], gradH[4n], energyTH[n]
void orc_immature_init(const float* dI, int w, int h, int n, const int* u, const int* v, float* color, float* weights, float* gradH, float* energyTH) {
  (void)h;
  for (int i = 0; i < n; i++) {
    float g[4] = {0, 0, 0, 0};
    bool bad = false;
    for (int idx = 0; idx < patternNum; idx++) {
      float ptc[3];
      interp33BiLin(dI, (float)(u[i] + patternP[idx][0]), (float)(v[i] + patternP[idx][1]), w, ptc);
      color[8 * i + idx] = ptc[0];
      if (!std::isfinite(ptc[0])) { energyTH[i] = NAN; bad = true; break; }
      g[0] += ptc[1] * ptc[1]; g[1] += ptc[1] * ptc[2]; g[2] += ptc[2] * ptc[1]; g[3] += ptc[2] * ptc[2];
      weights[8 * i + idx] = sqrtf(setting_outlierTHSumComponent / (setting_outlierTHSumComponent + (ptc[1] * ptc[1] + ptc[2] * ptc[2])));
    }
    for (int k = 0; k < 4; k++) gradH[4 * i + k] = g[k];
    if (bad) continue;
    float eth = patternNum * setting_outlierTH;
    eth *= setting_overallEnergyTHWeight * setting_overallEnergyTHWeight;
    energyTH[i] = eth;
  }
}

// traceOn for n points of ONE host against the new frame; state arrays are in/out (7 values per point, see ImmState)
void orc_immature_trace(const float* dI_new, int w, int h, int n, const float* u, const float* v, const float* color, const float* weights, const float* gradH,
                        const float* energyTH, const float* KRKi9, const float* Kt3, const float* aff2, float* idepth_min, float* idepth_max, float* quality,
                        float* lastTraceUV, float* lastTracePixelInterval, int* lastTraceStatus) {
  for (int i = 0; i < n; i++) {
    ImmState s = {idepth_min[i], idepth_max[i], quality[i], lastTraceUV[2 * i], lastTraceUV[2 * i + 1], lastTracePixelInterval[i], lastTraceStatus[i]};
    traceOn(dI_new, w, h, u[i], v[i], color + 8 * i, weights + 8 * i, gradH + 4 * i, energyTH[i], KRKi9, Kt3, aff2, s);
    idepth_min[i] = s.idepth_min; idepth_max[i] = s.idepth_max; quality[i] = s.quality;
    lastTraceUV[2 * i] = s.lastTraceU; lastTraceUV[2 * i + 1] = s.lastTraceV; lastTracePixelInterval[i] = s.lastTracePixelInterval;
    lastTraceStatus[i] = s.lastTraceStatus;
  }
}

// FullSystem::traceNewCoarse's per-host tables: hostToNew = new_w2c * host_c2w; KRKi = K R K^-1 (float), Kt = K t, aff = fromToVecExposure
void orc_trace_precalc(const double* new_w2c7, const double* host_c2w7, const double* fxfycxcy, float new_exposure, float host_exposure,
                       const double* new_aff, const double* host_aff, float* KRKi9, float* Kt3, float* aff2) {
  auto from7 = [](const double* p) { orc::SE3 T; T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q = orc::qimport(orc::Quat{p[6], p[3], p[4], p[5]}); return T; };   // tx ty tz qx qy qz qw
  const orc::SE3 T = orc::se3Mul(from7(new_w2c7), from7(host_c2w7));
  double Rd[9];
  orc::qToR(T.q, Rd);
  float R[9], t[3];
  for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
  for (int i = 0; i < 3; i++) t[i] = (float)T.t[i];
  const float fx = (float)fxfycxcy[0], fy = (float)fxfycxcy[1], cx = (float)fxfycxcy[2], cy = (float)fxfycxcy[3];
  const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
  // Eigen's 3x3 inverse: cofactors / determinant (Eigen/src/LU/InverseImpl.h compute_inverse_size3_helper)
  const float a = K[0], e = K[4], c = K[2], ff = K[5];
  const float det = a * (e * 1.0f - ff * 0.0f), invdet = 1.0f / det;
  const float Ki[9] = {(e * 1.0f - ff * 0.0f) * invdet, (c * 0.0f - 0.0f * 1.0f) * invdet, (0.0f * ff - c * e) * invdet,
                       (ff * 0.0f - 0.0f * 1.0f) * invdet, (a * 1.0f - c * 0.0f) * invdet, (c * 0.0f - a * ff) * invdet,
                       (0.0f * 0.0f - e * 0.0f) * invdet, (0.0f * 0.0f - a * 0.0f) * invdet, (a * e - 0.0f * 0.0f) * invdet};
  float KR[9];
  for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) KR[r * 3 + cc] = K[r * 3 + 0] * R[cc] + K[r * 3 + 1] * R[3 + cc] + K[r * 3 + 2] * R[6 + cc];
  for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) KRKi9[r * 3 + cc] = KR[r * 3 + 0] * Ki[cc] + KR[r * 3 + 1] * Ki[3 + cc] + KR[r * 3 + 2] * Ki[6 + cc];
  for (int r = 0; r < 3; r++) Kt3[r] = K[r * 3 + 0] * t[0] + K[r * 3 + 1] * t[1] + K[r * 3 + 2] * t[2];
  // AffLight::fromToVecExposure (src/dso/util/NumType.h:174-186)
  const double ea = exp(new_aff[0] - host_aff[0]) * new_exposure / host_exposure;
  const double eb = new_aff[1] - ea * host_aff[1];
  aff2[0] = (float)ea; aff2[1] = (float)eb;
}

// FrameFramePrecalc::set (HessianBlocks.cpp:193-223): PRE_RTll, PRE_tTll (current state), PRE_aff_mode of the pair (host -> target)
void orc_pair_precalc(const double* target_w2c7, const double* host_c2w7, float host_exposure, float target_exposure, const double* host_aff, const double* target_aff,
                      float* R9, float* t3, float* aff2) {
  auto from7 = [](const double* p) { orc::SE3 T; T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q = orc::qimport(orc::Quat{p[6], p[3], p[4], p[5]}); return T; };
  const orc::SE3 T = orc::se3Mul(from7(target_w2c7), from7(host_c2w7));
  double Rd[9];
  orc::qToR(T.q, Rd);
  for (int i = 0; i < 9; i++) R9[i] = (float)Rd[i];
  for (int i = 0; i < 3; i++) t3[i] = (float)T.t[i];
  float eF = host_exposure, eT = target_exposure;
  if (eF == 0 || eT == 0) { eT = eF = 1; }
  const double a = exp(target_aff[0] - host_aff[0]) * eT / eF;
  aff2[0] = (float)a; aff2[1] = (float)(target_aff[1] - a * host_aff[1]);
}

// optimizeImmaturePoint for the n points of ONE host against nres targets (the other keyframes in window order).
// result[n]: 1 activated / 0 skip / -1 delete; idepth[n]; res_state[n x nres] (ResState of the residual to every target, valid when result != 0)
void orc_immature_optimize(int w, int h, const float* fxfycxcy, int nres, const float* const* dI_targets, const float* R9, const float* t3, const float* aff2, int n,
                           const float* u, const float* v, const float* color, const float* weights, const float* energyTH, const float* idepth_min,
                           const float* idepth_max, int minObs, int* result, float* idepth, int* res_state) {
  CalibF C;
  C.fxl = fxfycxcy[0]; C.fyl = fxfycxcy[1]; C.cxl = fxfycxcy[2]; C.cyl = fxfycxcy[3];
  C.fxli = 1.0f / C.fxl; C.fyli = 1.0f / C.fyl;   // CalibHessian::setValueScaled (HessianBlocks.h:374-387)
  C.w = w; C.h = h;
  PairPre pre[16];
  for (int i = 0; i < nres; i++) { pre[i].R = R9 + 9 * i; pre[i].t = t3 + 3 * i; pre[i].aff = aff2 + 2 * i; pre[i].dI = dI_targets[i]; }
  for (int k = 0; k < n; k++) {
    for (int i = 0; i < nres; i++) res_state[(size_t)k * nres + i] = RS_OOB;
    result[k] = optimizeImmaturePoint(C, nres, pre, u[k], v[k], color + 8 * k, weights + 8 * k, energyTH[k], idepth_min[k], idepth_max[k], minObs, idepth[k],
                                      res_state + (size_t)k * nres);
  }
}

}  // extern "C"
