// CDNA4 (gfx950) kernels of the sliding-window photometric bundle adjustment.
//
// Replaces, on the reference side:
//   PointFrameResidual::linearize                src/dso/FullSystem/Residuals.cpp:78-274            -> k_ba_linearize
//   PointFrameResidual::applyRes / takeDataF     Residuals.cpp:306-328, EnergyFunctionalStructs.cpp:39-49 -> k_ba_apply
//   AccumulatedTopHessianSSE::addPoint<0>        OptimizationBackend/AccumulatedTopHessian.cpp:39-159 -> k_ba_point_sums + k_ba_accum_top
//   AccumulatorApprox::update/TopRight/BotRight  OptimizationBackend/MatrixAccumulators.h:754-915
//   AccumulatedSCHessianSSE::addPoint            OptimizationBackend/AccumulatedSCHessian.cpp:34-77   -> k_ba_accum_sc*
//   stitchDoubleInternal (top + SC)              AccumulatedTopHessian.cpp:241-303, AccumulatedSCHessian.cpp:78-157 -> k_ba_stitch_*
//   EnergyFunctional::resubstituteFPt            OptimizationBackend/EnergyFunctional.cpp:295-321     -> k_ba_resubstitute
//   doStepFromBackup / backupState / loadSateBackup (point part)  FullSystemOptimize.cpp:224-388      -> k_ba_point_step
//
// Design (MI355X-first):
//   * the reference walks a pointer graph (frame -> point -> residual -> EFResidual); here points and residuals are flat SoA
//     index arrays built once per window, residuals of a point contiguous, and the per-(host,target) / per-(host,t1,t2)
//     accumulation targets are BUCKETS with precomputed member lists;
//   * every accumulator element is owned by ONE thread that walks its bucket's member list in the reference's traversal
//     order (points in window order, residuals in point order) and replays the reference's fp32 arithmetic including the
//     1k / 1M hierarchical shift-up — no atomics, no reduction tree, results reproduce the single-threaded reference
//     accumulators; member records are staged through LDS in tiles so the 64-128 owners of a bucket share every load;
//   * linearisation keeps only what the accumulation needs: a 52-float compact record per residual (the reference
//     materialises a 74-float RawResidualJacobian and re-derives the rest on every accumulation);
//   * adjoint stitching stays fp64 like the reference (AccumulatedTopHessian.cpp:189-208) — 2.2k tiny 8x8x8 products that
//     cost a CPU ~1 ms per GN iteration and the GPU a few microseconds.
#pragma once
#include "common.h"
#include "interp.hpp"

namespace dmv {

#define BA_MAXF 8
enum { BA_IN = 0, BA_OOB = 1, BA_OUTLIER = 2 };
enum { REC_FLOATS = 52 };
// compact residual record layout (floats)
enum {
  REC_JPDC0 = 0,   // 4
  REC_JPDC1 = 4,   // 4
  REC_JPDXI0 = 8,  // 6
  REC_JPDXI1 = 14, // 6
  REC_JIDX2 = 20,  // 00, 01, 11
  REC_JABJIDX = 23,  // 00, 01, 10, 11
  REC_JAB2 = 27,   // 00, 01, 11
  REC_JI_R = 30,   // 2   (JIdx^T resF)
  REC_JAB_R = 32,  // 2   (JabF^T resF)
  REC_RR = 34,     // resF^T resF
  REC_JPDD = 35,   // 2
  REC_JPJD = 37,   // 8   JpJdF
  REC_HDD = 45, REC_BD = 46, REC_HCD = 47,  // per-residual contributions to the point sums (1, 1, 4)
  REC_PAD = 51
};

struct BAPrecalc {  // FrameFramePrecalc (HessianBlocks.h:80-107), the members linearize reads
  float KRKi[9], Kt[3], R0[9], t0[3], aff0, aff1, b0, pad;
};

struct BAWindow {
  int F, w, h, N, R;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;  // CalibHessian::fxl().. / fxli()..
  float wM3, hM3;
  float huberTH, outlierTHSum, modeA, modeB;
  int slot[BA_MAXF];
  float frameEnergyTH[BA_MAXF];
};

struct BAPoints {   // SoA, N entries
  const int* host;
  const float *u, *v;
  float *idepth, *idepth_zero, *idepth_backup, *step;
  const float* color;    // N x 8
  const float* weights;  // N x 8
  const float* priorF;
  const int* res_begin;  // N+1: residuals of point p are [res_begin[p], res_begin[p+1])
  float *Hdd, *bd, *Hcd, *HdiF, *bdSumF;  // accumulated per point (Hcd: N x 4)
};

struct BARes {      // SoA, R entries
  const int *point, *target;
  unsigned char *state, *newState, *active, *which;  // which: buffer (0/1) holding the APPLIED record
  float *energy, *newEnergy, *newEnergyWO;
  float* center;   // R x 3 centerProjectedTo
  float* rec[2];   // R x REC_FLOATS
};

__constant__ int c_patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};  // settings.cpp:296, pattern 8

// ------------------------------------------------------------------------------------------------ linearize
__global__ void __launch_bounds__(128) k_ba_linearize(const BAWindow W, const BAPoints P, const BARes Rs, const BAPrecalc* __restrict__ pre,
                                                       const FrameStore fs, double* __restrict__ energy_partials, float* __restrict__ fullJ) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  double myE = 0.0;
  if (ri < W.R) {
    float* __restrict__ rec = Rs.rec[Rs.which[ri] ^ 1] + (size_t)ri * REC_FLOATS;  // write the NON-applied buffer
    Rs.newEnergyWO[ri] = -1.0f;
    const int state = Rs.state[ri];
    bool done = false;
    if (state == BA_OOB) { Rs.newState[ri] = BA_OOB; myE = Rs.energy[ri]; done = true; }
    const int pi = Rs.point[ri], ti = Rs.target[ri];
    const int hi = P.host[pi];
    const BAPrecalc& pc = pre[hi + W.F * ti];
    const float pu = P.u[pi], pv = P.v[pi];
    float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x = 0, d_d_y = 0;
    if (!done) {
      // centre pixel at the LINEARISATION point (idepth_zero, evalPT poses)
      const float idz = P.idepth_zero[pi];
      const float Kx = (pu + 0 - W.cx) * W.fxi, Ky = (pv + 0 - W.cy) * W.fyi;
      const float p0 = pc.R0[0] * Kx + pc.R0[1] * Ky + pc.R0[2] * 1.0f + pc.t0[0] * idz;
      const float p1 = pc.R0[3] * Kx + pc.R0[4] * Ky + pc.R0[5] * 1.0f + pc.t0[1] * idz;
      const float p2 = pc.R0[6] * Kx + pc.R0[7] * Ky + pc.R0[8] * 1.0f + pc.t0[2] * idz;
      const float drescale = 1.0f / p2;
      const float new_idepth = idz * drescale;
      bool ok = drescale > 0;
      const float u = p0 * drescale, v = p1 * drescale;
      const float Ku = u * W.fx + W.cx, Kv = v * W.fy + W.cy;
      ok = ok && (Ku > 1.1f && Kv > 1.1f && Ku < W.wM3 && Kv < W.hM3);
      if (!ok) { Rs.newState[ri] = BA_OOB; myE = Rs.energy[ri]; done = true; }
      else {
        Rs.center[3 * ri + 0] = Ku; Rs.center[3 * ri + 1] = Kv; Rs.center[3 * ri + 2] = new_idepth;
        d_d_x = drescale * (pc.t0[0] - pc.t0[2] * u) * 1.0f * W.fx;
        d_d_y = drescale * (pc.t0[1] - pc.t0[2] * v) * 1.0f * W.fy;
        d_C_x[2] = drescale * (pc.R0[6] * u - pc.R0[0]);
        d_C_x[3] = W.fx * drescale * (pc.R0[7] * u - pc.R0[1]) * W.fyi;
        d_C_x[0] = Kx * d_C_x[2];
        d_C_x[1] = Ky * d_C_x[3];
        d_C_y[2] = W.fy * drescale * (pc.R0[6] * v - pc.R0[3]) * W.fxi;
        d_C_y[3] = drescale * (pc.R0[7] * v - pc.R0[4]);
        d_C_y[0] = Kx * d_C_y[2];
        d_C_y[1] = Ky * d_C_y[3];
        d_C_x[0] = (d_C_x[0] + u) * 50.0f; d_C_x[1] *= 50.0f; d_C_x[2] = (d_C_x[2] + 1) * 50.0f; d_C_x[3] *= 50.0f;   // SCALE_F, SCALE_C
        d_C_y[0] *= 50.0f; d_C_y[1] = (d_C_y[1] + v) * 50.0f; d_C_y[2] *= 50.0f; d_C_y[3] = (d_C_y[3] + 1) * 50.0f;
        d_xi_x[0] = new_idepth * W.fx; d_xi_x[1] = 0; d_xi_x[2] = -new_idepth * u * W.fx;
        d_xi_x[3] = -u * v * W.fx; d_xi_x[4] = (1 + u * u) * W.fx; d_xi_x[5] = -v * W.fx;
        d_xi_y[0] = 0; d_xi_y[1] = new_idepth * W.fy; d_xi_y[2] = -new_idepth * v * W.fy;
        d_xi_y[3] = -(1 + v * v) * W.fy; d_xi_y[4] = u * v * W.fy; d_xi_y[5] = u * W.fy;
      }
    }
    if (!done) {
      const float* __restrict__ img = fs.level(W.slot[ti], 0);
      const float ids = P.idepth[pi];
      float JI00 = 0, JI11 = 0, JI10 = 0, Ja00 = 0, Ja01 = 0, Ja10 = 0, Ja11 = 0, Jb00 = 0, Jb01 = 0, Jb11 = 0, wJI2 = 0;
      float JIr0 = 0, JIr1 = 0, Jar0 = 0, Jar1 = 0, rr = 0;
      float energyLeft = 0;
      float* fj = fullJ ? fullJ + (size_t)ri * 74 : nullptr;
#pragma unroll 1
      for (int idx = 0; idx < 8; idx++) {
        const float xu = pu + c_patternP[idx][0], xv = pv + c_patternP[idx][1];
        const float q0 = pc.KRKi[0] * xu + pc.KRKi[1] * xv + pc.KRKi[2] * 1.0f + pc.Kt[0] * ids;
        const float q1 = pc.KRKi[3] * xu + pc.KRKi[4] * xv + pc.KRKi[5] * 1.0f + pc.Kt[1] * ids;
        const float q2 = pc.KRKi[6] * xu + pc.KRKi[7] * xv + pc.KRKi[8] * 1.0f + pc.Kt[2] * ids;
        const float Ku = q0 / q2, Kv = q1 / q2;
        if (!(Ku > 1.1f && Kv > 1.1f && Ku < W.wM3 && Kv < W.hM3)) { done = true; break; }
        float3 hit = interp33(img, Ku, Kv, W.w);
        const float color = P.color[pi * 8 + idx];
        const float residual = hit.x - (pc.aff0 * color + pc.aff1);
        const float drdA = (color - pc.b0);
        if (!isfinite(hit.x)) { done = true; break; }
        float wgt = sqrtf(W.outlierTHSum / (W.outlierTHSum + (hit.y * hit.y + hit.z * hit.z)));
        wgt = 0.5f * (wgt + P.weights[pi * 8 + idx]);
        float hw = fabsf(residual) < W.huberTH ? 1.0f : W.huberTH / fabsf(residual);
        energyLeft += wgt * wgt * hw * residual * residual * (2 - hw);
        if (hw < 1) hw = sqrtf(hw);
        hw = hw * wgt;
        hit.y *= hw; hit.z *= hw;
        const float resF = residual * hw;
        float jab0 = drdA * hw, jab1 = hw;
        JI00 += hit.y * hit.y; JI11 += hit.z * hit.z; JI10 += hit.y * hit.z;
        Ja00 += drdA * hw * hit.y; Ja01 += drdA * hw * hit.z; Ja10 += hw * hit.y; Ja11 += hw * hit.z;
        Jb00 += drdA * drdA * hw * hw; Jb01 += drdA * hw * hw; Jb11 += hw * hw;
        wJI2 += hw * hw * (hit.y * hit.y + hit.z * hit.z);
        if (W.modeA < 0) jab0 = 0;
        if (W.modeB < 0) jab1 = 0;
        // accumulation-side inner products of addPoint<0> (resApprox = resF)  (AccumulatedTopHessian.cpp:103-113)
        JIr0 += resF * hit.y; JIr1 += resF * hit.z; Jar0 += resF * jab0; Jar1 += resF * jab1; rr += resF * resF;
        if (fj) { fj[idx] = resF; fj[30 + idx] = hit.y; fj[38 + idx] = hit.z; fj[46 + idx] = jab0; fj[54 + idx] = jab1; }
      }
      if (done) { Rs.newState[ri] = BA_OOB; myE = Rs.energy[ri]; }
      else {
        Rs.newEnergyWO[ri] = energyLeft;
        const float th = fmaxf(W.frameEnergyTH[hi], W.frameEnergyTH[ti]);
        if (energyLeft > th || wJI2 < 2) { energyLeft = th; Rs.newState[ri] = BA_OUTLIER; }
        else Rs.newState[ri] = BA_IN;
        Rs.newEnergy[ri] = energyLeft;
        myE = energyLeft;
        // compact record
#pragma unroll
        for (int k = 0; k < 4; k++) { rec[REC_JPDC0 + k] = d_C_x[k]; rec[REC_JPDC1 + k] = d_C_y[k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) { rec[REC_JPDXI0 + k] = d_xi_x[k]; rec[REC_JPDXI1 + k] = d_xi_y[k]; }
        rec[REC_JIDX2 + 0] = JI00; rec[REC_JIDX2 + 1] = JI10; rec[REC_JIDX2 + 2] = JI11;
        rec[REC_JABJIDX + 0] = Ja00; rec[REC_JABJIDX + 1] = Ja01; rec[REC_JABJIDX + 2] = Ja10; rec[REC_JABJIDX + 3] = Ja11;
        rec[REC_JAB2 + 0] = Jb00; rec[REC_JAB2 + 1] = Jb01; rec[REC_JAB2 + 2] = Jb11;
        rec[REC_JI_R + 0] = JIr0; rec[REC_JI_R + 1] = JIr1; rec[REC_JAB_R + 0] = Jar0; rec[REC_JAB_R + 1] = Jar1; rec[REC_RR] = rr;
        rec[REC_JPDD + 0] = d_d_x; rec[REC_JPDD + 1] = d_d_y;
        // takeDataF (EnergyFunctionalStructs.cpp:39-49)
        const float v0 = JI00 * d_d_x + JI10 * d_d_y, v1 = JI10 * d_d_x + JI11 * d_d_y;
#pragma unroll
        for (int k = 0; k < 6; k++) rec[REC_JPJD + k] = d_xi_x[k] * v0 + d_xi_y[k] * v1;
        rec[REC_JPJD + 6] = Ja00 * d_d_x + Ja01 * d_d_y;
        rec[REC_JPJD + 7] = Ja10 * d_d_x + Ja11 * d_d_y;
        // per-point contributions (AccumulatedTopHessian.cpp:131-134)
        rec[REC_BD] = JIr0 * d_d_x + JIr1 * d_d_y;
        rec[REC_HDD] = v0 * d_d_x + v1 * d_d_y;
#pragma unroll
        for (int k = 0; k < 4; k++) rec[REC_HCD + k] = d_C_x[k] * v0 + d_C_y[k] * v1;
        if (fj) {
          for (int k = 0; k < 6; k++) { fj[8 + k] = d_xi_x[k]; fj[14 + k] = d_xi_y[k]; }
          for (int k = 0; k < 4; k++) { fj[20 + k] = d_C_x[k]; fj[24 + k] = d_C_y[k]; }
          fj[28] = d_d_x; fj[29] = d_d_y;
          fj[62] = JI00; fj[63] = JI10; fj[64] = JI10; fj[65] = JI11;
          fj[66] = Ja00; fj[67] = Ja01; fj[68] = Ja10; fj[69] = Ja11;
          fj[70] = Jb00; fj[71] = Jb01; fj[72] = Jb01; fj[73] = Jb11;
        }
      }
    }
  }
  // block energy partial, fixed order
  __shared__ double s_e[128];
  s_e[threadIdx.x] = myE;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int k = 0; k < 128; k++) s += s_e[k];
    energy_partials[blockIdx.x] = s;
  }
}

// applyRes(true) for every active residual (Residuals.cpp:306-328): flips the applied-record selector
__global__ void __launch_bounds__(256) k_ba_apply(const int R, const BARes Rs) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= R) return;
  if (Rs.state[ri] == BA_OOB) return;  // can never go back from OOB
  const int ns = Rs.newState[ri];
  if (ns == BA_IN) { Rs.active[ri] = 1; Rs.which[ri] ^= 1; }
  else Rs.active[ri] = 0;
  Rs.state[ri] = (unsigned char)ns;
  Rs.energy[ri] = Rs.newEnergy[ri];
}

// ------------------------------------------------------------------------------------------------ per-point sums
// Hdd_accAF, bd_accAF, Hcd_accAF (sequential over the point's residuals) and the head of AccumulatedSCHessianSSE::addPoint:
// HdiF, bdSumF (AccumulatedSCHessian.cpp:36-54).
__global__ void __launch_bounds__(256) k_ba_point_sums(const BAWindow W, const BAPoints P, const BARes Rs) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= W.N) return;
  float Hdd = 0, bd = 0, Hcd[4] = {0, 0, 0, 0};
  int ngood = 0;
  for (int ri = P.res_begin[pi]; ri < P.res_begin[pi + 1]; ri++) {
    if (!Rs.active[ri]) continue;
    const float* __restrict__ rec = Rs.rec[Rs.which[ri]] + (size_t)ri * REC_FLOATS;
    bd += rec[REC_BD];
    Hdd += rec[REC_HDD];
#pragma unroll
    for (int k = 0; k < 4; k++) Hcd[k] += rec[REC_HCD + k];
    ngood++;
  }
  P.Hdd[pi] = Hdd; P.bd[pi] = bd;
#pragma unroll
  for (int k = 0; k < 4; k++) P.Hcd[4 * pi + k] = Hcd[k];
  if (ngood == 0) { P.HdiF[pi] = 0; P.bdSumF[pi] = 0; return; }
  float H = Hdd + 0.0f + P.priorF[pi];
  if (H < 1e-10) H = 1e-10;
  P.HdiF[pi] = 1.0 / H;
  const float deltaF = P.idepth[pi] - P.idepth_zero[pi];
  P.bdSumF[pi] = (bd + 0.0f) + P.priorF[pi] * deltaF;  // shiftPriorToZero = true
}

// hierarchical fp32 accumulator of the reference (Data / Data1k / Data1m + numIn1 counters), one value per thread
struct Acc3 {
  float d, d1k, d1m;
  int n1, n1k;
  __device__ __forceinline__ void init() { d = d1k = d1m = 0.f; n1 = n1k = 0; }
  __device__ __forceinline__ void add(const float v) { d += v; }
  __device__ __forceinline__ void bump() {  // numIn1++ ; shiftUp(false)
    n1++;
    if (n1 > 1000) { d1k = d + d1k; n1k += n1; n1 = 0; d = 0.f; }
    if (n1k > 1000) { d1m = d1k + d1m; n1k = 0; d1k = 0.f; }
  }
  __device__ __forceinline__ float finish() { d1k = d + d1k; d1m = d1k + d1m; return d1m; }
  // true when nAct more bump()s cannot trigger a shift-up: the tile can then be added without per-member counter checks
  __device__ __forceinline__ bool tileFits(const int nAct) const { return n1 + nAct <= 1000; }
};

// ------------------------------------------------------------------------------------------------ top accumulation
// Two-phase tiles keep the reference's SEQUENTIAL fp32 summation order without serialising the arithmetic:
//   phase A (parallel): the workgroup computes the contribution of every (member, element) pair of a tile into LDS,
//   phase B (sequential, cheap): the thread that owns element e adds the tile's contributions in member order — one LDS read
//   and one dependent add per member — replaying AccumulatorApprox / AccumulatorXX incl. the 1k / 1M shift-up.
// One workgroup per (host,target) bucket; thread e < 91 owns element e of the 13x13 block:
//   e in [0,55): upper triangle of the 10x10 [calib4 | pose6] block, row-major (AccumulatorApprox::update order)
//   e in [55,85): TopRight 10x3, e in [85,91): BotRight (a00,a01,a02,a11,a12,a22).
// gridDim.y = number of PARTIAL accumulators per bucket: partial sp owns a contiguous slice of the member list — the device
// analogue of the reference's per-worker accumulators acc[tid] (AccumulatedTopHessian.h:146), which stitchDoubleInternal
// sums in double (AccumulatedTopHessian.cpp:263-268).  gridDim.y == 1 (default) replays the single-threaded reference bit for bit.
// out: (F*F x nsplit) blocks of 96 floats (91 used) + counts of active members.
struct TopElem { int kind, r, c; };   // decoded once per thread
__device__ __forceinline__ TopElem decodeTop(const int e) {
  TopElem t; t.kind = 3; t.r = 0; t.c = 0;
  if (e < 55) { t.kind = 0; int off = 0, r = 0; while (e >= off + (10 - r)) { off += 10 - r; r++; } t.r = r; t.c = r + (e - off); }
  else if (e < 85) { t.kind = 1; t.r = (e - 55) / 3; t.c = (e - 55) % 3; }
  else if (e < 91) { t.kind = 2; t.r = e - 85; }
  return t;
}
__device__ __forceinline__ float topContribution(const float* q, const TopElem t) {
  // x[i] = (i < 4) ? Jpdc0[i] : Jpdxi0[i-4]  -> q[i < 4 ? i : 4 + i];   y[i] -> q[i < 4 ? 4 + i : 10 + i]
  const int r = t.r, c = t.c;
  if (t.kind == 0) {
    const float xr = q[r < 4 ? r : 4 + r], xc = q[c < 4 ? c : 4 + c];
    const float yr = q[r < 4 ? 4 + r : 10 + r], yc = q[c < 4 ? 4 + c : 10 + c];
    const float a = q[REC_JIDX2], bb = q[REC_JIDX2 + 1], cc = q[REC_JIDX2 + 2];
    return a * xc * xr + cc * yc * yr + bb * (xc * yr + yc * xr);
  } else if (t.kind == 1) {
    const float xr = q[r < 4 ? r : 4 + r], yr = q[r < 4 ? 4 + r : 10 + r];
    // TR col 0: (JabJIdx00, JabJIdx01), col 1: (JabJIdx10, JabJIdx11), col 2: (JI_r0, JI_r1)
    const float t0 = c == 0 ? q[REC_JABJIDX + 0] : (c == 1 ? q[REC_JABJIDX + 2] : q[REC_JI_R + 0]);
    const float t1 = c == 0 ? q[REC_JABJIDX + 1] : (c == 1 ? q[REC_JABJIDX + 3] : q[REC_JI_R + 1]);
    return xr * t0 + yr * t1;
  } else {
    // a00 = Jab2_00, a01 = Jab2_01, a02 = Jab_r0, a11 = Jab2_11, a12 = Jab_r1, a22 = rr
    return r == 0 ? q[REC_JAB2 + 0] : r == 1 ? q[REC_JAB2 + 1] : r == 2 ? q[REC_JAB_R + 0] : r == 3 ? q[REC_JAB2 + 2] : r == 4 ? q[REC_JAB_R + 1] : q[REC_RR];
  }
}

#define TOP_TILE 64
__device__ __forceinline__ void accumTopBlock(const int b, const int sp, const int nsp, const BARes& Rs, const int* __restrict__ bucket_begin,
                                              const int* __restrict__ bucket_members, float* __restrict__ out, int* __restrict__ out_num) {
  __shared__ float s_rec[TOP_TILE][37];
  __shared__ int s_nact;
  const int e = threadIdx.x;
  const int mb = bucket_begin[b], mcnt = bucket_begin[b + 1] - mb;
  const int m0 = mb + (int)(((long long)mcnt * sp) / nsp), m1 = mb + (int)(((long long)mcnt * (sp + 1)) / nsp);
  const TopElem te = decodeTop(e);
  Acc3 acc; acc.init();
  int num = 0;
  for (int base = m0; base < m1; base += TOP_TILE) {
    const int cnt = min(TOP_TILE, m1 - base);
    __syncthreads();
    if (e == 0) s_nact = 0;
    __syncthreads();
    {  // stage up to 64 member records (first 35 floats): 4 threads per record; inactive members become all-zero rows
      const int j = e >> 2, part = e & 3;
      if (j < cnt) {
        const int ri = bucket_members[base + j];
        const bool act = Rs.active[ri] != 0;
        const float* __restrict__ rec = Rs.rec[Rs.which[ri]] + (size_t)ri * REC_FLOATS;
        for (int k = part; k < 35; k += 4) s_rec[j][k] = act ? rec[k] : 0.0f;
        if (part == 0) { s_rec[j][35] = act ? 1.0f : 0.0f; if (act) atomicAdd(&s_nact, 1); }
      }
    }
    __syncthreads();
    const int nact = s_nact;
    if (e < 91) {
      if (acc.tileFits(nact)) {
        // no shift-up can happen inside this tile: contributions of inactive members are exact zeros (x + 0 == x), so the
        // adds run back to back in member order; the shared counters advance by the number of active members
        for (int j0 = 0; j0 < cnt; j0 += 16) {
          float c[16];
#pragma unroll
          for (int j = 0; j < 16; j++) c[j] = (j0 + j < cnt) ? topContribution(s_rec[j0 + j], te) : 0.0f;   // independent: pipelined LDS reads
#pragma unroll
          for (int j = 0; j < 16; j++) acc.d += c[j];                                                       // the sequential part
        }
        acc.n1 += nact;
      } else {
        for (int j = 0; j < cnt; j++) {
          if (s_rec[j][35] == 0.0f) continue;
          const float v = topContribution(s_rec[j], te);
          if (te.kind == 0) { acc.add(v); acc.bump(); }
          else { acc.bump(); acc.add(v); }   // update() advances the shared counters BEFORE updateTopRight/BotRight
        }
      }
    }
    num += nact;
  }
  if (e < 91) out[(b * nsp + sp) * 96 + e] = acc.finish();
  if (e == 91) out_num[b * nsp + sp] = num;
}

// ------------------------------------------------------------------------------------------------ Schur accumulation
// accD[h,t1,t2] (8x8) += (HdiF * JpJd(r1)) JpJd(r2)^T : one workgroup (64 threads) per bucket, member = (r1, r2, point).
__device__ __forceinline__ void accumScDBlock(const int b, const int sp, const int nsp, const BARes& Rs, const BAPoints& P, const int* __restrict__ bucket_begin,
                                              const int* __restrict__ members /* 3 ints each */, float* __restrict__ outD, int* __restrict__ outNum) {
  __shared__ float s_l[64][9], s_r[64][9], s_w[64];
  __shared__ float s_c[64][65];
  const int e = threadIdx.x & 63, i = e >> 3, j = e & 7;
  const bool live = threadIdx.x < 64;   // the block is launched with 256 threads; the first wave does the work
  const int mb = bucket_begin[b], mcnt = bucket_begin[b + 1] - mb;
  const int m0 = mb + (int)(((long long)mcnt * sp) / nsp), m1 = mb + (int)(((long long)mcnt * (sp + 1)) / nsp);
  Acc3 acc; acc.init();
  int num = 0;
  for (int base = m0; base < m1; base += 64) {
    const int cnt = min(64, m1 - base);
    __syncthreads();
    if (live && e < cnt) {
      const int r1 = members[3 * (base + e)], r2 = members[3 * (base + e) + 1], pi = members[3 * (base + e) + 2];
      const bool act = Rs.active[r1] && Rs.active[r2];
      const float* __restrict__ q1 = Rs.rec[Rs.which[r1]] + (size_t)r1 * REC_FLOATS + REC_JPJD;
      const float* __restrict__ q2 = Rs.rec[Rs.which[r2]] + (size_t)r2 * REC_FLOATS + REC_JPJD;
#pragma unroll
      for (int k = 0; k < 8; k++) { s_l[e][k] = q1[k]; s_r[e][k] = q2[k]; }
      s_w[e] = act ? P.HdiF[pi] : -1.0f;
    }
    __syncthreads();
    if (live) for (int m = 0; m < cnt; m++) s_c[m][e] = (s_w[m] * s_l[m][i]) * s_r[m][j];   // A += w*L*R^T (phase A, independent)
    __syncthreads();
    if (live)
      for (int m = 0; m < cnt; m++) {
        if (s_w[m] < 0) continue;
        acc.add(s_c[m][e]);
        acc.bump();
        num++;
      }
  }
  if (live) {
    outD[(b * nsp + sp) * 64 + e] = acc.finish();
    if (e == 0) outNum[b * nsp + sp] = num;
  }
}

// accE[h,t] (8x4) += (HdiF JpJd) Hcd^T ; accEB[h,t] (8) += (HdiF*bdSumF) JpJd : workgroup per (h,t) bucket, 40 owners
__device__ __forceinline__ void accumScEBlock(const int b, const int sp, const int nsp, const BARes& Rs, const BAPoints& P, const int* __restrict__ bucket_begin,
                                              const int* __restrict__ bucket_members, float* __restrict__ outE /* 40 per bucket */) {
  __shared__ float s_l[64][9], s_h[64][5], s_w[64], s_wb[64];
  __shared__ float s_c[64][41];
  const int e = threadIdx.x;
  const int mb = bucket_begin[b], mcnt = bucket_begin[b + 1] - mb;
  const int m0 = mb + (int)(((long long)mcnt * sp) / nsp), m1 = mb + (int)(((long long)mcnt * (sp + 1)) / nsp);
  Acc3 acc; acc.init();
  for (int base = m0; base < m1; base += 64) {
    const int cnt = min(64, m1 - base);
    __syncthreads();
    if (e < cnt) {
      const int ri = bucket_members[base + e];
      const int pi = Rs.point[ri];
      const bool act = Rs.active[ri] != 0;
      const float* __restrict__ q = Rs.rec[Rs.which[ri]] + (size_t)ri * REC_FLOATS + REC_JPJD;
#pragma unroll
      for (int k = 0; k < 8; k++) s_l[e][k] = q[k];
#pragma unroll
      for (int k = 0; k < 4; k++) s_h[e][k] = P.Hcd[4 * pi + k] + 0.0f;
      const float hdi = P.HdiF[pi];
      s_w[e] = act ? hdi : -1.0f;
      s_wb[e] = hdi * P.bdSumF[pi];
    }
    __syncthreads();
    if (e < 40)
      for (int m = 0; m < cnt; m++) s_c[m][e] = (e < 32) ? (s_w[m] * s_l[m][e >> 2]) * s_h[m][e & 3] : s_wb[m] * s_l[m][e - 32];
    __syncthreads();
    if (e < 40)
      for (int m = 0; m < cnt; m++) {
        if (s_w[m] < 0) continue;
        acc.add(s_c[m][e]);
        acc.bump();
      }
  }
  if (e < 40) outE[(b * nsp + sp) * 40 + e] = acc.finish();
}

// accHcc (4x4) += HdiF Hcd Hcd^T ; accbc (4) += (bdSumF*HdiF) Hcd over all points with an active residual: 20 owners per workgroup,
// gridDim.x partial accumulators over contiguous point ranges (1 = the reference's single-threaded order)
__device__ __forceinline__ void accumScCBlock(const int sp, const int nsp, const int N, const BAPoints& P, float* __restrict__ outC /* 20 */) {
  __shared__ float s_c[256][21];
  __shared__ float s_w[256];
  const int e = threadIdx.x;
  const int p0 = (int)(((long long)N * sp) / nsp), p1 = (int)(((long long)N * (sp + 1)) / nsp);
  Acc3 acc; acc.init();
  for (int base = p0; base < p1; base += 256) {
    const int cnt = min(256, p1 - base);
    __syncthreads();
    if (e < cnt) {
      const int pi = base + e;
      const float hdi = P.HdiF[pi];
      float hc[4];
#pragma unroll
      for (int k = 0; k < 4; k++) hc[k] = P.Hcd[4 * pi + k] + 0.0f;
      s_w[e] = hdi > 0 ? hdi : -1.0f;   // HdiF == 0 <=> no active residual (point skipped by addPoint)
      const float wb = P.bdSumF[pi] * hdi;
#pragma unroll
      for (int q = 0; q < 16; q++) s_c[e][q] = (hdi * hc[q >> 2]) * hc[q & 3];
#pragma unroll
      for (int q = 0; q < 4; q++) s_c[e][16 + q] = wb * hc[q];
    }
    const int nact = __syncthreads_count(e < cnt && s_w[e] >= 0);
    if (e < 20) {
      if (acc.tileFits(nact)) {
        for (int m0 = 0; m0 < cnt; m0 += 16) {
          float c[16];
#pragma unroll
          for (int m = 0; m < 16; m++) c[m] = (m0 + m < cnt && s_w[m0 + m] >= 0) ? s_c[m0 + m][e] : 0.0f;
#pragma unroll
          for (int m = 0; m < 16; m++) acc.d += c[m];
        }
        acc.n1 += nact;
      } else {
        for (int m = 0; m < cnt; m++) {
          if (s_w[m] < 0) continue;
          acc.add(s_c[m][e]);
          acc.bump();
        }
      }
    }
  }
  if (e < 20) outC[sp * 20 + e] = acc.finish();
}

// All four accumulations of solveSystemF in ONE launch (they only depend on the applied records and the per-point sums):
// blocks [0, nTop) top buckets, [nTop, nTop+nD) accD buckets, then accE buckets, then the calibration partials.
struct AccumArgs {
  int F, N, nsTop, nsD, nsC;
  const int *top_begin, *top_members, *scd_begin, *scd_members;
  float *accTop, *accD, *accE, *accC;
  int *numTop, *numD;
};
__global__ void __launch_bounds__(256) k_ba_accumulate(const AccumArgs A, const BARes Rs, const BAPoints P) {
  const int F2 = A.F * A.F;
  const int nTop = F2 * A.nsTop, nD = F2 * A.F * A.nsD, nE = F2 * A.nsTop;
  int blk = blockIdx.x;
  if (blk < nTop) { accumTopBlock(blk / A.nsTop, blk % A.nsTop, A.nsTop, Rs, A.top_begin, A.top_members, A.accTop, A.numTop); return; }
  blk -= nTop;
  if (blk < nD) { accumScDBlock(blk / A.nsD, blk % A.nsD, A.nsD, Rs, P, A.scd_begin, A.scd_members, A.accD, A.numD); return; }
  blk -= nD;
  if (blk < nE) { accumScEBlock(blk / A.nsTop, blk % A.nsTop, A.nsTop, Rs, P, A.top_begin, A.top_members, A.accE); return; }
  blk -= nE;
  accumScCBlock(blk, A.nsC, A.N, P, A.accC);
}

// ------------------------------------------------------------------------------------------------ fp64 stitching
// step 1: adjoint sandwiches per bucket (summed over the inner frame index where the destination block is fixed),
// step 2 (k_ba_stitch_gather): every element of H_A, b_A, H_sc, b_sc sums <= 2F+1 slab entries in a fixed order.
struct StitchBufs {
  double* topHH;  // F x 64     sum_t AH B AH^T            -> block (h,h)
  double* topTT;  // F*F x 64   AT B AT^T   per (h,t)      -> block (t,t)
  double* topHT;  // F*F x 64   AH B AT^T   per (h,t)      -> block (h,t)
  double* topHC;  // F x 32     sum_t AH B8C               -> rows of frame h, calib columns
  double* topTC;  // F*F x 32   AT B8C      per (h,t)      -> rows of frame t
  double* topBH;  // F x 8      sum_t AH b8
  double* topBT;  // F*F x 8    AT b8       per (h,t)
  double* topCC;  // F x 20     sum_t [Bcc (16) | bc (4)]
  double* scHH;   // F*F x 64   sum_k AH_ij D AH_ik^T      -> block (i,i)
  double* scTH;   // F*F x 64   sum_k AT_ij D AH_ik^T      -> block (j,i)
  double* scTT;   // F^3 x 64   AT_ij D AT_ik^T            -> block (j,k)
  double* scHT;   // F^3 x 64   AH_ij D AT_ik^T            -> block (i,k)
  double* scHC; double* scTC; double* scBH; double* scBT;  // F*F x 32 / 8
};

__device__ __forceinline__ double rowDotT(const double* T /*8x8: A*B*/, const double* C, int r, int c) {
  double s = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) s += T[r * 8 + q] * C[c * 8 + q];
  return s;
}

// top: one workgroup per host h, loop over targets t; bucket k = h + F*t : B = acc.H (13x13)
__global__ void __launch_bounds__(64) k_ba_stitch_top(const int F, const int nsplit, const float* __restrict__ acc /* F*F x nsplit x 96 */,
                                                       const int* __restrict__ num, const double* __restrict__ adHost, const double* __restrict__ adTarget,
                                                       const StitchBufs S) {
  __shared__ double sB[13][13], sAH[64], sAT[64], sT1[64], sT2[64];
  const int hI = blockIdx.x, e = threadIdx.x, r = e >> 3, c = e & 7;
  double hh = 0, hc = 0, bh = 0, cc = 0;
  for (int t = 0; t < F; t++) {
    const int k = hI + F * t;
    __syncthreads();
    // unpack the 91 sums into the symmetric 13x13 (finish(), MatrixAccumulators.h:621-653)
    for (int q = e; q < 169; q += 64) {
      int i = q / 13, j = q % 13;
      if (i > j) { int tt = i; i = j; j = tt; }
      int slot;
      if (j < 10) slot = i * 10 - (i * (i - 1)) / 2 + (j - i);
      else if (i < 10) slot = 55 + i * 3 + (j - 10);
      else slot = 85 + ((i == 10) ? (j - 10) : (i == 11 ? 3 + (j - 11) : 5));
      double v = 0.0;
      for (int sp = 0; sp < nsplit; sp++) if (num[k * nsplit + sp] > 0) v += (double)acc[(k * nsplit + sp) * 96 + slot];
      sB[q / 13][q % 13] = v;
    }
    sAH[e] = adHost[k * 64 + e]; sAT[e] = adTarget[k * 64 + e];
    __syncthreads();
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { t1 += sAH[r * 8 + q] * sB[4 + q][4 + c]; t2 += sAT[r * 8 + q] * sB[4 + q][4 + c]; }
    sT1[e] = t1; sT2[e] = t2;
    __syncthreads();
    hh += rowDotT(sT1, sAH, r, c);
    S.topTT[k * 64 + e] = rowDotT(sT2, sAT, r, c);
    S.topHT[k * 64 + e] = rowDotT(sT1, sAT, r, c);
    if (c < 4) {
      double h1 = 0, h2 = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) { h1 += sAH[r * 8 + q] * sB[4 + q][c]; h2 += sAT[r * 8 + q] * sB[4 + q][c]; }
      hc += h1; S.topTC[k * 32 + r * 4 + c] = h2;
    }
    if (c == 4) {
      double b1 = 0, b2 = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) { b1 += sAH[r * 8 + q] * sB[4 + q][12]; b2 += sAT[r * 8 + q] * sB[4 + q][12]; }
      bh += b1; S.topBT[k * 8 + r] = b2;
    }
    if (e < 16) cc += sB[e >> 2][e & 3];
    else if (e < 20) cc += sB[e - 16][12];
  }
  S.topHH[hI * 64 + e] = hh;
  if (c < 4) S.topHC[hI * 32 + r * 4 + c] = hc;
  if (c == 4) S.topBH[hI * 8 + r] = bh;
  if (e < 20) S.topCC[hI * 20 + e] = cc;
}

// SC: one workgroup per (i,j), loop over k; accD index = ((i + F*j) + k*F*F)
__global__ void __launch_bounds__(64) k_ba_stitch_sc(const int F, const int nsplit, const int nsplitE, const float* __restrict__ accD, const int* __restrict__ numD,
                                                      const float* __restrict__ accE /* F*F x nsplitE x 40 */,
                                                      const double* __restrict__ adHost, const double* __restrict__ adTarget, const StitchBufs S) {
  __shared__ double sD[64], sAHij[64], sATij[64], sAHik[64], sATik[64], sT1[64], sT2[64], sE[40];
  const int ij = blockIdx.x, e = threadIdx.x, r = e >> 3, c = e & 7;
  const int F2 = F * F, i = ij % F;
  sAHij[e] = adHost[ij * 64 + e]; sATij[e] = adTarget[ij * 64 + e];
  if (e < 40) {
    double v = 0.0;
    for (int sp = 0; sp < nsplitE; sp++) v += (double)accE[(ij * nsplitE + sp) * 40 + e];
    sE[e] = v;
  }
  double hh = 0, th = 0;
  for (int kk = 0; kk < F; kk++) {
    const int ijk = ij + kk * F2, ik = i + F * kk;
    __syncthreads();
    {
      double v = 0.0;
      for (int sp = 0; sp < nsplit; sp++) if (numD[ijk * nsplit + sp] > 0) v += (double)accD[(ijk * nsplit + sp) * 64 + e];
      sD[e] = v;
    }
    sAHik[e] = adHost[ik * 64 + e]; sATik[e] = adTarget[ik * 64 + e];
    __syncthreads();
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { t1 += sAHij[r * 8 + q] * sD[q * 8 + c]; t2 += sATij[r * 8 + q] * sD[q * 8 + c]; }
    sT1[e] = t1; sT2[e] = t2;
    __syncthreads();
    hh += rowDotT(sT1, sAHik, r, c);
    th += rowDotT(sT2, sAHik, r, c);
    S.scTT[ijk * 64 + e] = rowDotT(sT2, sATik, r, c);
    S.scHT[ijk * 64 + e] = rowDotT(sT1, sATik, r, c);
  }
  S.scHH[ij * 64 + e] = hh;
  S.scTH[ij * 64 + e] = th;
  // accE (8x4) and accEB (8)
  if (c < 4) {
    double h1 = 0, h2 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { h1 += sAHij[r * 8 + q] * sE[q * 4 + c]; h2 += sATij[r * 8 + q] * sE[q * 4 + c]; }
    S.scHC[ij * 32 + r * 4 + c] = h1; S.scTC[ij * 32 + r * 4 + c] = h2;
  }
  if (c == 4) {
    double b1 = 0, b2 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { b1 += sAHij[r * 8 + q] * sE[32 + q]; b2 += sATij[r * 8 + q] * sE[32 + q]; }
    S.scBH[ij * 8 + r] = b1; S.scBT[ij * 8 + r] = b2;
  }
}

// step 2: one thread per element of H_A, H_sc (n x n, n = 4+8F) and b_A, b_sc; fixed summation order.
// Output layout: out[0 .. n*n) = H_A, then b_A (n), then H_sc (n*n), then b_sc (n).
__global__ void __launch_bounds__(256) k_ba_stitch_gather(const int F, const int nsC, const float* __restrict__ accC, const StitchBufs S,
                                                           double* __restrict__ out) {
  const int n = 4 + 8 * F, F2 = F * F;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = n * n + n;
  if (tid >= 2 * per) return;
  const bool sc = tid >= per;
  const int t = sc ? tid - per : tid;
  double val = 0;
  if (t < n * n) {
    const int row = t / n, col = t % n;
    if (row < 4 && col < 4) {
      if (!sc) { for (int q = 0; q < F; q++) val += S.topCC[q * 20 + row * 4 + col]; }
      else for (int sp = 0; sp < nsC; sp++) val += (double)accC[sp * 20 + row * 4 + col];
    } else if (row < 4 || col < 4) {
      // calib cross terms: H[fIdx.., 0..4) accumulated, mirrored into H[0..4), fIdx..)
      const int cc = row < 4 ? row : col, fr = (row < 4 ? col : row) - 4;
      const int f = fr >> 3, r = fr & 7;
      if (!sc) { val = S.topHC[f * 32 + r * 4 + cc]; for (int q = 0; q < F; q++) val += S.topTC[(q + F * f) * 32 + r * 4 + cc]; }
      else { for (int q = 0; q < F; q++) val += S.scHC[(f + F * q) * 32 + r * 4 + cc]; for (int q = 0; q < F; q++) val += S.scTC[(q + F * f) * 32 + r * 4 + cc]; }
    } else {
      const int bi = (row - 4) >> 3, bj = (col - 4) >> 3, r = (row - 4) & 7, c = (col - 4) & 7;
      if (!sc) {
        // H[h,h] += HH[h,t], H[t,t] += TT[h,t], H[h,t] += HT[h,t]; then (h<t): H[h,t] += H[t,h]^T, H[t,h] = H[h,t]^T
        if (bi == bj) { val = S.topHH[bi * 64 + r * 8 + c]; for (int q = 0; q < F; q++) val += S.topTT[(q + F * bi) * 64 + r * 8 + c]; val += S.topHT[(bi + F * bi) * 64 + r * 8 + c]; }
        else val = S.topHT[(bi + F * bj) * 64 + r * 8 + c] + S.topHT[(bj + F * bi) * 64 + c * 8 + r];
      } else {
        // H[i,i] += HH[ij] (all j); H[j,k] += TT[ijk] (all i); H[j,i] += TH[ij]; H[i,k] += HT[ijk] (all j)
        if (bi == bj) for (int j = 0; j < F; j++) val += S.scHH[(bi + F * j) * 64 + r * 8 + c];
        for (int i = 0; i < F; i++) val += S.scTT[((i + F * bi) + bj * F2) * 64 + r * 8 + c];
        val += S.scTH[(bj + F * bi) * 64 + r * 8 + c];
        for (int j = 0; j < F; j++) val += S.scHT[((bi + F * j) + bj * F2) * 64 + r * 8 + c];
      }
    }
    out[(sc ? per : 0) + t] = val;
  } else {
    const int row = t - n * n;
    if (row < 4) {
      if (!sc) { for (int q = 0; q < F; q++) val += S.topCC[q * 20 + 16 + row]; }
      else for (int sp = 0; sp < nsC; sp++) val += (double)accC[sp * 20 + 16 + row];
    } else {
      const int f = (row - 4) >> 3, r = (row - 4) & 7;
      if (!sc) { val = S.topBH[f * 8 + r]; for (int q = 0; q < F; q++) val += S.topBT[(q + F * f) * 8 + r]; }
      else { for (int q = 0; q < F; q++) val += S.scBH[(f + F * q) * 8 + r]; for (int q = 0; q < F; q++) val += S.scBT[(q + F * f) * 8 + r]; }
    }
    out[(sc ? per : 0) + n * n + row] = val;
  }
}

// ------------------------------------------------------------------------------------------------ back-substitution / stepping
// xAd: F*F x 8 floats, index h*F + t (EnergyFunctional.cpp:280-282), xc: 4 floats
__global__ void __launch_bounds__(256) k_ba_resubstitute(const BAWindow W, const BAPoints P, const BARes Rs, const float* __restrict__ xc,
                                                          const float* __restrict__ xAd) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= W.N) return;
  int ngood = 0;
  for (int ri = P.res_begin[pi]; ri < P.res_begin[pi + 1]; ri++) if (Rs.active[ri]) ngood++;
  if (ngood == 0) { P.step[pi] = 0; return; }
  float b = P.bdSumF[pi];
  float dotc = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) dotc += xc[k] * (P.Hcd[4 * pi + k] + 0.0f);
  b -= dotc;
  const int hi = P.host[pi];
  for (int ri = P.res_begin[pi]; ri < P.res_begin[pi + 1]; ri++) {
    if (!Rs.active[ri]) continue;
    const float* __restrict__ q = Rs.rec[Rs.which[ri]] + (size_t)ri * REC_FLOATS + REC_JPJD;
    const float* __restrict__ xa = xAd + (size_t)(hi * W.F + Rs.target[ri]) * 8;
    float d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += xa[k] * q[k];
    b -= d;
  }
  P.step[pi] = -b * P.HdiF[pi];
}

// mode 0: backupState (idepth_backup = idepth);  mode 1: doStepFromBackup (idepth = idepth_zero = backup + fac*step),
// mode 2: loadSateBackup (idepth = idepth_zero = backup).  Mode 1 also emits per-block partial sums of step^2, |backup|.
__global__ void __launch_bounds__(256) k_ba_point_step(const int N, const BAPoints P, const int mode, const float stepfacD, float* __restrict__ partials) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  float s2 = 0, sn = 0;
  if (pi < N) {
    if (mode == 0) P.idepth_backup[pi] = P.idepth[pi];
    else if (mode == 1) {
      const float st = P.step[pi], bk = P.idepth_backup[pi];
      const float v = bk + stepfacD * st;
      P.idepth[pi] = v; P.idepth_zero[pi] = v;
      s2 = st * st; sn = fabsf(bk);
    } else { const float bk = P.idepth_backup[pi]; P.idepth[pi] = bk; P.idepth_zero[pi] = bk; }
  }
  if (mode == 1) {
    __shared__ float a[256], b[256];
    a[threadIdx.x] = s2; b[threadIdx.x] = sn;
    __syncthreads();
    if (threadIdx.x == 0) {
      float x = 0, y = 0;
      for (int k = 0; k < 256; k++) { x += a[k]; y += b[k]; }
      partials[2 * blockIdx.x] = x; partials[2 * blockIdx.x + 1] = y;
    }
  }
}

}  // namespace dmv
