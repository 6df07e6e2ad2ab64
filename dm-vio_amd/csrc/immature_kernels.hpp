// Immature-point path on the device: candidate points of every keyframe are traced along their epipolar line in each new frame.
//   k_immature_init   <- ImmaturePoint::ImmaturePoint   (src/dso/FullSystem/ImmaturePoint.cpp:34-62)
//   k_immature_trace  <- ImmaturePoint::traceOn         (src/dso/FullSystem/ImmaturePoint.cpp:76-437)
// Every point is independent and the reference's arithmetic is sequential per point, so the results are BIT-IDENTICAL to the CPU
// path: one WAVEFRONT traces one point — lane i evaluates step i of the discrete epipolar search (its position is produced by the
// same i sequential fp32 additions the reference performs), the argmin / second-best are wave reductions with the reference's
// first-minimum tie rule, and the three Gauss-Newton refinements evaluate the 8 pattern pixels on 8 lanes and add them in
// pattern order.  Images are the intensity-only planes of common.h FrameStore; gradients = the reference's central differences.
#pragma once
#include "common.h"
#include "interp.hpp"

namespace dmv {

enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };

struct ImmaturePts {
  int n;
  // static part (constructor)
  float *u, *v;              // pixel position in the host (integers stored as float, ImmaturePoint.h:66)
  int* host;                 // index into the per-host tables of a trace call
  float *color, *weights;    // n x 8
  float* gradH;              // n x 4 (00 01 10 11)
  float* energyTH;
  // mutable part
  float *idepth_min, *idepth_max, *quality, *lastTraceUV /* n x 2 */, *lastTracePixelInterval;
  int* lastTraceStatus;
};

struct ImmatureSettings {
  float outlierTH = 12 * 12, outlierTHSumComponent = 50 * 50, overallEnergyTHWeight = 1;
  float maxPixSearch = 0.027f, huberTH = 9;
  int minTraceTestRadius = 2, GNIterations = 3;
  float stepsize = 1.0f, GNThreshold = 0.1f, extraSlackOnTH = 1.2f, slackInterval = 1.5f, minImprovementFactor = 2;
};

// status histogram of FullSystem::traceNewCoarse's printout / bookkeeping (FullSystem.cpp:562-583): one workgroup, counts stored straight into
// pinned host memory (out6)
__global__ void __launch_bounds__(1024) k_status_hist(const int* __restrict__ status, const int n, int* __restrict__ out6) {
  __shared__ int s_c[6];
  if (threadIdx.x < 6) s_c[threadIdx.x] = 0;
  __syncthreads();
  int c[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int v = status[i];
#pragma unroll
    for (int k = 0; k < 6; k++) c[k] += (v == k) ? 1 : 0;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int t = c[k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_c[k], t);
  }
  __syncthreads();
  if (threadIdx.x < 6) out6[threadIdx.x] = s_c[threadIdx.x];
}

struct TraceTables { const float *KRKi /* H x 9 */, *Kt /* H x 3 */, *aff /* H x 2 */; };
// the same tables by value, for windows of up to 16 host keyframes: they travel as kernel arguments (896 B) — no upload, no staging buffer
enum { IMM_ARG_HOSTS = 16 };
struct TraceTablesArg { float KRKi[IMM_ARG_HOSTS * 9], Kt[IMM_ARG_HOSTS * 3], aff[IMM_ARG_HOSTS * 2]; };

__constant__ int c_pattern8[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};   // settings.cpp:296, pattern 8

// getInterpolatedElement31 (globalFuncs.h:160-176) on the intensity plane
__device__ __forceinline__ float interp31(const float* __restrict__ I, const float x, const float y, const int w) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float* bp = I + ix + iy * w;
  return dxdy * bp[1 + w] + (dy - dxdy) * bp[w] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
}
// getInterpolatedElement33 (globalFuncs.h:103-118) anywhere in the image (border rows / columns included): the gradient channels
// of the four taps follow the reference's flat-index rule (gradAt)
__device__ __forceinline__ float3 interp33Any(const float* __restrict__ I, const float x, const float y, const int w, const int h) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  const float* bp = I + ix + iy * w;
  const float2 g00 = gradAt(I, w, h, ix, iy), g10 = gradAt(I, w, h, ix + 1, iy), g01 = gradAt(I, w, h, ix, iy + 1), g11 = gradAt(I, w, h, ix + 1, iy + 1);
  float3 r;
  r.x = w11 * bp[1 + w] + w01 * bp[w] + w10 * bp[1] + w00 * bp[0];
  r.y = w11 * g11.x + w01 * g01.x + w10 * g10.x + w00 * g00.x;
  r.z = w11 * g11.y + w01 * g01.y + w10 * g10.y + w00 * g00.y;
  return r;
}

// ImmaturePoint constructor: thread per point.  getInterpolatedElement33BiLin (globalFuncs.h:203-227) only reads the intensity
// channel (forward differences of the bilinear cell).
__global__ void __launch_bounds__(256) k_immature_init(const float* __restrict__ I, const int w, const int first, const int n, const ImmaturePts P,
                                                        const int host_tag, const ImmatureSettings S) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = first + k;
  const float uf = P.u[i], vf = P.v[i];
  float g00 = 0.f, g01 = 0.f, g10 = 0.f, g11 = 0.f;
  bool bad = false;
  for (int idx = 0; idx < 8 && !bad; idx++) {
    const float x = (float)((int)uf + c_pattern8[idx][0]), y = (float)((int)vf + c_pattern8[idx][1]);
    const int ix = (int)x, iy = (int)y;
    const float* bp = I + ix + iy * w;
    const float tl = bp[0], tr = bp[1], bl = bp[w], br = bp[w + 1];
    const float dx = x - ix, dy = y - iy;
    const float topInt = dx * tr + (1 - dx) * tl, botInt = dx * br + (1 - dx) * bl;
    const float leftInt = dy * bl + (1 - dy) * tl, rightInt = dy * br + (1 - dy) * tr;
    const float c = dx * rightInt + (1 - dx) * leftInt, gx = rightInt - leftInt, gy = botInt - topInt;
    P.color[8 * i + idx] = c;
    if (!isfinite(c)) { P.energyTH[i] = NAN; bad = true; break; }
    g00 += gx * gx; g01 += gx * gy; g10 += gy * gx; g11 += gy * gy;
    P.weights[8 * i + idx] = sqrtf(S.outlierTHSumComponent / (S.outlierTHSumComponent + (gx * gx + gy * gy)));
  }
  P.gradH[4 * i + 0] = g00; P.gradH[4 * i + 1] = g01; P.gradH[4 * i + 2] = g10; P.gradH[4 * i + 3] = g11;
  if (!bad) {
    float eth = 8 * S.outlierTH;
    eth *= S.overallEnergyTHWeight * S.overallEnergyTHWeight;
    P.energyTH[i] = eth;
  }
  P.host[i] = host_tag;
  P.idepth_min[i] = 0.f; P.idepth_max[i] = NAN; P.quality[i] = 10000.f;
  P.lastTraceUV[2 * i] = 0.f; P.lastTraceUV[2 * i + 1] = 0.f; P.lastTracePixelInterval[i] = 0.f;
  P.lastTraceStatus[i] = IPS_UNINITIALIZED;
}

// first-minimum argmin over the wave: (value, index) with the reference's strict '<' scan order (smallest index among equal minima)
__device__ __forceinline__ void waveArgMin(float& v, int& idx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(v, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}
__device__ __forceinline__ float waveMin(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// traceOn: one wavefront per point (4 points per 256-thread workgroup); TT = TraceTables (device memory) or TraceTablesArg (kernel arguments)
template <class TT>
__global__ void __launch_bounds__(256) k_immature_trace(const float* __restrict__ I, const int w, const int h, const ImmaturePts P, const TT T,
                                                         const ImmatureSettings S) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= P.n) return;
  int status = P.lastTraceStatus[i];
  if (status == IPS_OOB) return;
  const int hI = P.host[i];
  const float* __restrict__ KRKi = T.KRKi + 9 * hI;
  const float* __restrict__ Kt = T.Kt + 3 * hI;
  const float aff0 = T.aff[2 * hI], aff1 = T.aff[2 * hI + 1];
  const float u = P.u[i], v = P.v[i];
  float idepth_min = P.idepth_min[i], idepth_max = P.idepth_max[i];
  float outU = -1.f, outV = -1.f, outInterval = 0.f;
  bool writeInterval = false;   // idepth_min / idepth_max updated
  const float maxPixSearch = (w + h) * S.maxPixSearch;
  // 1. both ends of the depth interval must project inside the image
  const float pr0 = KRKi[0] * u + KRKi[1] * v + KRKi[2] * 1.0f, pr1 = KRKi[3] * u + KRKi[4] * v + KRKi[5] * 1.0f, pr2 = KRKi[6] * u + KRKi[7] * v + KRKi[8] * 1.0f;
  const float pmin0 = pr0 + Kt[0] * idepth_min, pmin1 = pr1 + Kt[1] * idepth_min, pmin2 = pr2 + Kt[2] * idepth_min;
  const float uMin = pmin0 / pmin2, vMin = pmin1 / pmin2;
  // rotated pattern of THIS lane's pattern index (lanes >= 8 mirror lane & 7) + the wave-wide maximum extents
  const int pidx = lane & 7;
  const float rpx = KRKi[0] * (float)c_pattern8[pidx][0] + KRKi[1] * (float)c_pattern8[pidx][1];
  const float rpy = KRKi[3] * (float)c_pattern8[pidx][0] + KRKi[4] * (float)c_pattern8[pidx][1];
  int maxRotPatX = (int)fabsf(rpx), maxRotPatY = (int)fabsf(rpy);
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) { maxRotPatX = max(maxRotPatX, __shfl_xor(maxRotPatX, off, 64)); maxRotPatY = max(maxRotPatY, __shfl_xor(maxRotPatY, off, 64)); }
  const int boundU = max(4, maxRotPatX + 2), boundV = max(4, maxRotPatY + 2);
  const float loU = (float)boundU, loV = (float)boundV, hiU = (float)(w - boundU - 1), hiV = (float)(h - boundV - 1);
  float quality = P.quality[i];
  bool done = false;
  float dist = 0.f, uMax = 0.f, vMax = 0.f;
  if (!(uMin > loU && vMin > loV && uMin < hiU && vMin < hiV)) { status = IPS_OOB; done = true; }
  const bool finMax = isfinite(idepth_max);
  if (!done) {
    if (finMax) {
      const float pmax0 = pr0 + Kt[0] * idepth_max, pmax1 = pr1 + Kt[1] * idepth_max, pmax2 = pr2 + Kt[2] * idepth_max;
      uMax = pmax0 / pmax2; vMax = pmax1 / pmax2;
      if (!(uMax > loU && vMax > loV && uMax < hiU && vMax < hiV)) { status = IPS_OOB; done = true; }
      else {
        // 2. an interval that already spans less than slackInterval pixels is left alone
        dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
        dist = sqrtf(dist);
        if (dist < S.slackInterval) { outU = (uMax + uMin) * 0.5f; outV = (vMax + vMin) * 0.5f; outInterval = dist; status = IPS_SKIPPED; done = true; }
      }
    } else {
      dist = maxPixSearch;
      // project to arbitrary depth to get direction.
      const float pmax0 = pr0 + Kt[0] * 0.01f, pmax1 = pr1 + Kt[1] * 0.01f, pmax2 = pr2 + Kt[2] * 0.01f;
      uMax = pmax0 / pmax2; vMax = pmax1 / pmax2;
      const float ddx = uMax - uMin, ddy = vMax - vMin;
      const float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
      uMax = uMin + dist * ddx * d; vMax = vMin + dist * ddy * d;
      if (!(uMax > loU && vMax > loV && uMax < hiU && vMax < hiV)) { status = IPS_OOB; done = true; }
    }
  }
  // set OOB if scale change too big.
  if (!done && !(idepth_min < 0 || (pmin2 > 0.75f && pmin2 < 1.5f))) { status = IPS_OOB; done = true; }
  float dx = 0.f, dy = 0.f, errorInPixel = 0.f;
  if (!done) {
    // 3. localisation error from the gradient structure tensor; no trace when it cannot shrink the interval
    dx = S.stepsize * (uMax - uMin); dy = S.stepsize * (vMax - vMin);
    const float g0 = P.gradH[4 * i], g1 = P.gradH[4 * i + 1], g2 = P.gradH[4 * i + 2], g3 = P.gradH[4 * i + 3];
    const float a = (dx * g0 + dy * g2) * dx + (dx * g1 + dy * g3) * dy;
    const float b = (dy * g0 + (-dx) * g2) * dy + (dy * g1 + (-dx) * g3) * (-dx);
    errorInPixel = 0.2f + 0.2f * (a + b) / a;
    if (errorInPixel * S.minImprovementFactor > dist && finMax) {
      outU = (uMax + uMin) * 0.5f; outV = (vMax + vMin) * 0.5f; outInterval = dist; status = IPS_BADCONDITION; done = true;
    }
  }
  if (!done) {
    if (errorInPixel > 10) errorInPixel = 10;
    // 4. discrete search along the epipolar segment
    dx /= dist; dy /= dist;
    if (dist > maxPixSearch) { uMax = uMin + maxPixSearch * dx; vMax = vMin + maxPixSearch * dy; dist = maxPixSearch; }
    int numSteps = (int)(1.9999f + dist / S.stepsize);
    const float randShift = uMin * 1000 - floorf(uMin * 1000);
    const float ptx0 = uMin - randShift * dx, pty0 = vMin - randShift * dy;
    if (!isfinite(dx) || !isfinite(dy)) { status = IPS_OOB; done = true; }
    if (!done) {
      if (numSteps >= 100) numSteps = 99;
      const float c = P.color[8 * i + pidx];   // lanes use it in the GN phase; the search reads all 8 through shuffles below
      float colors[8];
#pragma unroll
      for (int k = 0; k < 8; k++) colors[k] = __shfl(c, k, 64);
      float rx[8], ry[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { rx[k] = __shfl(rpx, k, 64); ry[k] = __shfl(rpy, k, 64); }
      // lane l evaluates steps l and l + 64; its position is reached by the reference's sequential additions
      float e[2] = {1e30f, 1e30f}, px[2] = {0.f, 0.f}, py[2] = {0.f, 0.f};
      {
        float ptx = ptx0, pty = pty0;
        int stepIdx = 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int target = lane + 64 * r;
          for (; stepIdx < target && stepIdx < numSteps; stepIdx++) { ptx += dx; pty += dy; }
          if (target < numSteps) {
            float energy = 0.f;
#pragma unroll
            for (int idx = 0; idx < 8; idx++) {
              const float hitColor = interp31(I, (float)(ptx + rx[idx]), (float)(pty + ry[idx]), w);
              if (!isfinite(hitColor)) { energy += 1e5f; continue; }
              const float residual = hitColor - (float)(aff0 * colors[idx] + aff1);
              const float hw = fabsf(residual) < S.huberTH ? 1 : S.huberTH / fabsf(residual);
              energy += hw * residual * residual * (2 - hw);
            }
            e[r] = energy < 1e10f ? energy : 1e30f;   // only energies below the initial 1e10 can become best / second best (NaN never does)
            px[r] = ptx; py[r] = pty;
          }
        }
      }
      // best step: first index with the minimal energy among those below 1e10
      float bv = e[0]; int bi = lane;
      if (e[1] < bv) { bv = e[1]; bi = lane + 64; }
      float bestEnergy = bv; int bestIdx = bi;
      waveArgMin(bestEnergy, bestIdx);
      float bestU = 0.f, bestV = 0.f;
      if (!(bestEnergy < 1e10f)) { bestEnergy = 1e10f; bestIdx = -1; }   // no step beat the initial 1e10 (all NaN / huge)
      else {
        const int src = bestIdx & 63, rr = bestIdx >> 6;
        bestU = __shfl(rr ? px[1] : px[0], src, 64); bestV = __shfl(rr ? py[1] : py[0], src, 64);
      }
      // find best score outside a +-2px radius.
      float sb = 1e10f;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int k = lane + 64 * r;
        if (k < numSteps && (k < bestIdx - S.minTraceTestRadius || k > bestIdx + S.minTraceTestRadius) && e[r] < sb) sb = e[r];
      }
      const float secondBest = waveMin(sb);
      const float newQuality = secondBest / bestEnergy;
      if (newQuality < quality || numSteps > 10) quality = newQuality;
      // 5. Gauss-Newton refinement along the line
      float uBak = bestU, vBak = bestV, stepBack = 0.f;
      const float gnstepsize = 1;
      if (S.GNIterations > 0) bestEnergy = 1e5f;
      const float wgt = P.weights[8 * i + pidx];
      for (int it = 0; it < S.GNIterations && !done; it++) {
        // lanes 0..7: pattern pixel idx = lane
        const float posU = (float)(bestU + rpx), posV = (float)(bestV + rpy);
        const bool oobTap = (posU < 0 || posV < 0 || posU >= w - 1 || posV >= h - 1);
        if (__ballot(oobTap && lane < 8) != 0ull) { status = IPS_OOB; done = true; outU = -1.f; outV = -1.f; outInterval = 0.f; break; }
        const float3 hit = interp33Any(I, posU, posV, w, h);
        const float residual = hit.x - (aff0 * c + aff1);
        const float dResdDist = dx * hit.y + dy * hit.z;
        const float hw = fabsf(residual) < S.huberTH ? 1 : S.huberTH / fabsf(residual);
        const float tH = hw * dResdDist * dResdDist, tb = hw * residual * dResdDist, tE = wgt * wgt * hw * residual * residual * (2 - hw);
        const bool fin = isfinite(hit.x);
        float H = 1, bb = 0, energy = 0;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {   // pattern order, every lane redundantly
          const bool f = __shfl((int)fin, idx, 64) != 0;
          const float aH = __shfl(tH, idx, 64), ab = __shfl(tb, idx, 64), aE = __shfl(tE, idx, 64);
          if (!f) { energy += 1e5f; continue; }
          H += aH; bb += ab; energy += aE;
        }
        if (energy > bestEnergy) {
          // do a smaller step from old point.
          stepBack *= 0.5f;
          bestU = uBak + stepBack * dx; bestV = vBak + stepBack * dy;
        } else {
          float step = -gnstepsize * bb / H;
          if (step < -0.5f) step = -0.5f; else if (step > 0.5f) step = 0.5f;
          if (!isfinite(step)) step = 0;
          uBak = bestU; vBak = bestV; stepBack = step;
          bestU += step * dx; bestV += step * dy;
          bestEnergy = energy;
        }
        if (fabsf(stepBack) < S.GNThreshold) break;
      }
      if (!done) {
        // 6. energy-based outlier test
        if (!(bestEnergy < P.energyTH[i] * S.extraSlackOnTH)) {
          status = (status == IPS_OUTLIER) ? IPS_OOB : IPS_OUTLIER;
          done = true;
        }
      }
      if (!done) {
        // 7. new inverse-depth interval from the refined position +- the localisation error
        float nmin, nmax;
        if (dx * dx > dy * dy) {
          nmin = (pr2 * (bestU - errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
          nmax = (pr2 * (bestU + errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
        } else {
          nmin = (pr2 * (bestV - errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
          nmax = (pr2 * (bestV + errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
        }
        if (nmin > nmax) { const float t = nmin; nmin = nmax; nmax = t; }
        idepth_min = nmin; idepth_max = nmax; writeInterval = true;
        if (!isfinite(nmin) || !isfinite(nmax) || (nmax < 0)) { status = IPS_OUTLIER; }
        else { outInterval = 2 * errorInPixel; outU = bestU; outV = bestV; status = IPS_GOOD; }
        done = true;
      }
    }
  }
  if (lane == 0) {
    P.lastTraceStatus[i] = status;
    P.lastTraceUV[2 * i] = outU; P.lastTraceUV[2 * i + 1] = outV;
    P.lastTracePixelInterval[i] = outInterval;
    P.quality[i] = quality;
    if (writeInterval) { P.idepth_min[i] = idepth_min; P.idepth_max[i] = idepth_max; }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:51-205) + ImmaturePoint::linearizeResidual (ImmaturePoint.cpp:498-565):
// Gauss-Newton on the inverse depth of one candidate over its residuals to all other keyframes.  One wavefront per point: lane
// (r, idx) = (lane >> 3, lane & 7) evaluates pattern pixel idx of the residual to the r-th other keyframe; the sums Hdd, bd, the
// residual energies and the accept / reject logic are replayed by every lane in the reference's order (bit-identical results).
struct OptTables {
  int F;
  int slot[8];                 // level-0 image of every keyframe
  const float* R;              // F*F x 9, index host*F + target: PRE_RTll
  const float* t;              // F*F x 3: PRE_tTll
  const float* aff;            // F*F x 2: PRE_aff_mode
  float fxl, fyl, cxl, cyl, fxli, fyli;
};
enum { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };

struct OptPass { float energy, Hdd, bd; };

__global__ void __launch_bounds__(256) k_immature_optimize(const FrameStore fs, const int w, const int h, const ImmaturePts P, const OptTables T,
                                                            const unsigned char* __restrict__ select, const int minObs, const float minIdepthH_act,
                                                            const int GNIts, const float huberTH, int* __restrict__ result, float* __restrict__ idepth_out,
                                                            int* __restrict__ res_state /* n x F, host entry -1 */) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= P.n) return;
  if (select && !select[k]) { if (lane == 0) { result[k] = 0; idepth_out[k] = 0.f; } return; }
  const int F = T.F, hI = P.host[k], nres = F - 1;
  const int r = lane >> 3, idx = lane & 7;
  const bool active = r < nres;
  const int tI = active ? (r < hI ? r : r + 1) : 0;          // r-th keyframe that is not the host
  const float* __restrict__ R = T.R + 9 * (hI * F + tI);
  const float* __restrict__ tt = T.t + 3 * (hI * F + tI);
  const float aff0 = T.aff[2 * (hI * F + tI)], aff1 = T.aff[2 * (hI * F + tI) + 1];
  const float* __restrict__ I = fs.level(T.slot[tI], 0);
  const float pu = P.u[k], pv = P.v[k];
  const float color = P.color[8 * k + idx], wgt = P.weights[8 * k + idx], energyTH = P.energyTH[k];
  const int pdx = c_pattern8[idx][0], pdy = c_pattern8[idx][1];
  const float wM3G = (float)(w - 3), hM3G = (float)(h - 3);
  int state[7], newState[7];
  float energy[7], newEnergy[7];
#pragma unroll
  for (int i = 0; i < 7; i++) { state[i] = RS_IN; newState[i] = RS_OUTLIER; energy[i] = 0.f; newEnergy[i] = 0.f; }

  auto pass = [&](const float idepth, const float slack, float Hdd, float bd) -> OptPass {
    // this lane's pattern pixel (projectPoint, ResidualProjections.h:61-87)
    bool ok = false;
    float eTerm = 0.f, hTerm = 0.f, bTerm = 0.f;
    if (active) {
      const float Kl0 = (pu + pdx - T.cxl) * T.fxli, Kl1 = (pv + pdy - T.cyl) * T.fyli, Kl2 = 1;
      const float p0 = R[0] * Kl0 + R[1] * Kl1 + R[2] * Kl2 + tt[0] * idepth;
      const float p1 = R[3] * Kl0 + R[4] * Kl1 + R[5] * Kl2 + tt[1] * idepth;
      const float p2 = R[6] * Kl0 + R[7] * Kl1 + R[8] * Kl2 + tt[2] * idepth;
      const float drescale = 1.0f / p2;
      if (drescale > 0) {
        const float u = p0 * drescale, v = p1 * drescale;
        const float Ku = u * T.fxl + T.cxl, Kv = v * T.fyl + T.cyl;
        if (Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G) {
          const float3 hit = interp33Any(I, Ku, Kv, w, h);
          if (isfinite(hit.x)) {
            ok = true;
            const float residual = hit.x - (aff0 * color + aff1);
            float hw = fabsf(residual) < huberTH ? 1 : huberTH / fabsf(residual);
            eTerm = wgt * wgt * hw * residual * residual * (2 - hw);
            const float dxInterp = hit.y * T.fxl, dyInterp = hit.z * T.fyl;
            const float d_idepth = (dxInterp * drescale * (tt[0] - tt[2] * u) + dyInterp * drescale * (tt[1] - tt[2] * v)) * 1.0f;   // SCALE_IDEPTH
            hw *= wgt * wgt;
            hTerm = (hw * d_idepth) * d_idepth;
            bTerm = (hw * residual) * d_idepth;
          }
        }
      }
    }
    // sequential replay, identical on every lane
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      if (i < nres) {
        float ret;
        if (state[i] == RS_OOB) { newState[i] = RS_OOB; ret = energy[i]; }
        else {
          float energyLeft = 0.f;
          bool broke = false;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            if (!broke) {
              const int src = i * 8 + q;
              if (!__shfl((int)ok, src, 64)) broke = true;
              else { energyLeft += __shfl(eTerm, src, 64); Hdd += __shfl(hTerm, src, 64); bd += __shfl(bTerm, src, 64); }
            }
          }
          if (broke) { newState[i] = RS_OOB; ret = energy[i]; }
          else {
            if (energyLeft > energyTH * slack) { energyLeft = energyTH * slack; newState[i] = RS_OUTLIER; }
            else newState[i] = RS_IN;
            newEnergy[i] = energyLeft;
            ret = energyLeft;
          }
        }
        total = (float)((double)total + (double)ret);
      }
    }
    OptPass o; o.energy = total; o.Hdd = Hdd; o.bd = bd;
    return o;
  };

  float currentIdepth = (P.idepth_max[k] + P.idepth_min[k]) * 0.5f;
  OptPass last = pass(currentIdepth, 1000.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 7; i++) { state[i] = newState[i]; energy[i] = newEnergy[i]; }
  int res = 1;
  bool finished = false;
  if (!isfinite(last.energy) || last.Hdd < minIdepthH_act) { res = 0; finished = true; }
  float lambda = 0.1f;
  for (int iteration = 0; iteration < GNIts && !finished; iteration++) {
    float H = last.Hdd;
    H *= 1 + lambda;
    const float step = (float)((1.0 / (double)H) * (double)last.bd);
    const float newIdepth = currentIdepth - step;
    const OptPass nw = pass(newIdepth, 1.f, 0.f, 0.f);
    if (!isfinite(last.energy) || nw.Hdd < minIdepthH_act) { res = 0; finished = true; break; }
    if (nw.energy < last.energy) {
      currentIdepth = newIdepth;
      last = nw;
#pragma unroll
      for (int i = 0; i < 7; i++) { state[i] = newState[i]; energy[i] = newEnergy[i]; }
      lambda *= 0.5f;
    } else lambda *= 5.f;
    if ((double)fabsf(step) < 0.0001 * (double)currentIdepth) break;
  }
  if (res == 1) {
    if (!isfinite(currentIdepth)) res = -1;
    else {
      int numGoodRes = 0;
#pragma unroll
      for (int i = 0; i < 7; i++) if (i < nres && state[i] == RS_IN) numGoodRes++;
      if (numGoodRes < minObs || !isfinite(energyTH)) res = -1;
    }
  }
  if (lane == 0) {
    result[k] = res;
    idepth_out[k] = currentIdepth;
    for (int t = 0; t < F; t++) res_state[(size_t)k * F + t] = -1;
#pragma unroll
    for (int i = 0; i < 7; i++) if (i < nres) res_state[(size_t)k * F + (i < hI ? i : i + 1)] = state[i];
  }
}

}  // namespace dmv
