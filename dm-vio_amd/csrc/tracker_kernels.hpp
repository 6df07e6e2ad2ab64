// CDNA4 (gfx950) kernels of the coarse tracker: fused calcRes+calcGS evaluation and the
// device-resident Levenberg–Marquardt loop.
//
// Replaces, on the reference side (all under src/dso/FullSystem/):
//   CoarseTracker::calcRes          CoarseTracker.cpp:361-517   (per-point warp, bilinear tap, Huber, energy)
//   CoarseTracker::calcGSSSE        CoarseTracker.cpp:299-356   (9x9 weighted outer-product reduction)
//   Accumulator9::updateSSE_eighted OptimizationBackend/MatrixAccumulators.h:1091-1166
//   CoarseTracker::trackNewestCoarse CoarseTracker.cpp:539-770  (LM loop, useimu=0 branch)
//
// Design (MI355X-first, not a translation):
//   * calcRes and calcGS are ONE pass: the reference writes 8 "warped" SoA buffers and re-reads them
//     in calcGSSSE; here the Jacobian row never leaves registers.  Algorithmic traffic per template
//     point: 16 B point record + 4 taps x 12 B = 64 B.
//   * per-thread register accumulators (52 sums), then a transposing wave64 butterfly (63 cross-lane
//     exchanges for 64 values instead of 64 x 6), one LDS hop across waves — fixed summation order,
//     bitwise reproducible run to run.
//   * the LM loop (8x8 pivoted LDLT in fp64, SE3 exp, accept/reject, lambda schedule, level logic)
//     runs inside the same launch: one workgroup per alignment problem, thread 0 is a small state
//     machine that asks the workgroup for evaluations.  B problems (pose hypotheses / frames) = B
//     workgroups in one launch.
//   * per-point arithmetic is written in the reference's operation order and the library is built
//     with -ffp-contract=off, so residuals / bounds decisions are bit-identical to the CPU path;
//     only the summation order differs.
#pragma once
#include "common.h"
#include "lie_dev.h"

namespace dmv {

__device__ __forceinline__ float3 interp33(const float4* __restrict__ img, float x, float y, int width) {
  // getInterpolatedElement33 (src/dso/util/globalFuncs.h:103-118)
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy;
  const float dxdy = dx * dy;
  const float4* bp = img + ix + iy * width;
  const float4 p00 = bp[0], p10 = bp[1], p01 = bp[width], p11 = bp[1 + width];
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  float3 r;
  r.x = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
  r.y = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
  r.z = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
  return r;
}

// Accumulator9 slot of H(r,c), r <= c: rows of the upper triangle back to back (MatrixAccumulators.h:1091-1166).
__host__ __device__ constexpr int accIdx(int r, int c) { return ACC_H + r * 9 - (r * (r - 1)) / 2 + (c - r); }

// acc[H(R,C..8)] += (J[R]*w) * J[C]; all indices are compile-time constants so the 64 accumulators stay in VGPRs.
template <int R, int C>
__device__ __forceinline__ void accumulateCols(float (&acc)[ACC_PAD], const float (&J)[9], const float Jw) {
  acc[accIdx(R, C)] = __builtin_fmaf(Jw, J[C], acc[accIdx(R, C)]);
  if constexpr (C < 8) accumulateCols<R, C + 1>(acc, J, Jw);
}
template <int R>
__device__ __forceinline__ void accumulateRows(float (&acc)[ACC_PAD], const float (&J)[9], const float w) {
  accumulateCols<R, R>(acc, J, J[R] * w);
  if constexpr (R < 8) accumulateRows<R + 1>(acc, J, w);
}

// One template point: everything calcRes does for it plus its calcGS row, accumulated in registers.
__device__ __forceinline__ void evalPoint(const float4 P, const int i, const EvalP& e, const LevelGeom& g,
                                          const float4* __restrict__ img, const float huberTH, float (&acc)[ACC_PAD]) {
  const float x = P.x, y = P.y, id = P.z, refColor = P.w;
  const float pt0 = e.RKi[0] * x + e.RKi[1] * y + e.RKi[2] * 1.0f + e.t[0] * id;
  const float pt1 = e.RKi[3] * x + e.RKi[4] * y + e.RKi[5] * 1.0f + e.t[1] * id;
  const float pt2 = e.RKi[6] * x + e.RKi[7] * y + e.RKi[8] * 1.0f + e.t[2] * id;
  const float u = pt0 / pt2, v = pt1 / pt2;
  const float Ku = g.fx * u + g.cx, Kv = g.fy * v + g.cy;
  const float new_idepth = id / pt2;

  if (e.lvl == 0 && (i & 31) == 0) {
    // flow indicators (CoarseTracker.cpp:416-447)
    const float k0 = g.Ki[0] * x + g.Ki[1] * y + g.Ki[2] * 1.0f;
    const float k1 = g.Ki[3] * x + g.Ki[4] * y + g.Ki[5] * 1.0f;
    const float k2 = g.Ki[6] * x + g.Ki[7] * y + g.Ki[8] * 1.0f;
    const float r0 = e.RKi[0] * x + e.RKi[1] * y + e.RKi[2] * 1.0f;
    const float r1 = e.RKi[3] * x + e.RKi[4] * y + e.RKi[5] * 1.0f;
    const float r2 = e.RKi[6] * x + e.RKi[7] * y + e.RKi[8] * 1.0f;
    const float a0 = k0 + e.t[0] * id, a1 = k1 + e.t[1] * id, a2 = k2 + e.t[2] * id;
    const float KuT = g.fx * (a0 / a2) + g.cx, KvT = g.fy * (a1 / a2) + g.cy;
    const float b0 = k0 - e.t[0] * id, b1 = k1 - e.t[1] * id, b2 = k2 - e.t[2] * id;
    const float KuT2 = g.fx * (b0 / b2) + g.cx, KvT2 = g.fy * (b1 / b2) + g.cy;
    const float c0 = r0 - e.t[0] * id, c1 = r1 - e.t[1] * id, c2 = r2 - e.t[2] * id;
    const float Ku3 = g.fx * (c0 / c2) + g.cx, Kv3 = g.fy * (c1 / c2) + g.cy;
    float sT = (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
    sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
    float sRT = (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
    sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
    acc[ACC_FT] += sT;
    acc[ACC_FRT] += sRT;
    acc[ACC_FN] += 2.0f;
  }

  if (!(Ku > 2 && Kv > 2 && Ku < g.w - 3 && Kv < g.h - 3 && new_idepth > 0)) return;

  const float3 hit = interp33(img, Ku, Kv, g.w);
  if (!isfinite(hit.x)) return;
  const float residual = hit.x - (e.aff0 * refColor + e.aff1);
  const float ar = fabsf(residual);
  const float hw = ar < huberTH ? 1.0f : huberTH / ar;

  acc[ACC_NE] += 1.0f;
  if (ar > e.cutoff) {
    acc[ACC_E] += e.maxEnergy;
    acc[ACC_NSAT] += 1.0f;
    return;
  }
  acc[ACC_E] += hw * residual * residual * (2 - hw);
  acc[ACC_NW] += 1.0f;

  // calcGSSSE row (CoarseTracker.cpp:314-338)
  const float dx = hit.y * g.fx, dy = hit.z * g.fy;
  float J[9];
  J[0] = new_idepth * dx;
  J[1] = new_idepth * dy;
  J[2] = 0.0f - new_idepth * (u * dx + v * dy);
  J[3] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
  J[4] = (u * v) * dy + dx * (1.0f + u * u);
  J[5] = u * dy - v * dx;
  J[6] = e.aff0 * (e.b0 - refColor);
  J[7] = -1.0f;
  J[8] = residual;
  accumulateRows<0>(acc, J, hw);
}

// Transposing butterfly: on entry every lane holds 64 partial sums v[0..63]; on exit lane L holds the
// wave-wide total of slot L in v[0].  63 cross-lane exchanges.  Template recursion keeps every register
// index a compile-time constant (a runtime-indexed array would be demoted to scratch memory).
template <int HALF, int I>
__device__ __forceinline__ void butterflyStep(float (&v)[ACC_PAD], const bool hi) {
  const float lo_v = v[I], hi_v = v[I + HALF];
  const float keep = hi ? hi_v : lo_v;
  const float send = hi ? lo_v : hi_v;
  v[I] = keep + __shfl_xor(send, HALF, 64);
  if constexpr (I + 1 < HALF) butterflyStep<HALF, I + 1>(v, hi);
}
template <int HALF>
__device__ __forceinline__ void butterflyLevel(float (&v)[ACC_PAD], const int lane) {
  butterflyStep<HALF, 0>(v, (lane & HALF) != 0);
  if constexpr (HALF > 1) butterflyLevel<HALF / 2>(v, lane);
}
__device__ __forceinline__ float waveReduceTranspose(float (&v)[ACC_PAD]) {
  butterflyLevel<32>(v, __lane_id());
  return v[0];
}

// Workgroup-wide evaluation over points [first, n) with the given stride.  Result: s_tot[0..63] (LDS)
// valid for all threads after return.  T = threads per workgroup (multiple of 64).
template <int T>
__device__ __forceinline__ void blockEval(const EvalP& e, const LevelGeom& g, const float4* __restrict__ pc, const int n,
                                          const int first, const int stride, const float4* __restrict__ img,
                                          const float huberTH, float (*s_part)[ACC_PAD], float* s_tot) {
  float acc[ACC_PAD];
#pragma unroll
  for (int k = 0; k < ACC_PAD; k++) acc[k] = 0.0f;
  for (int i = first; i < n; i += stride) {
    const float4 P = pc[i];
    evalPoint(P, i, e, g, img, huberTH, acc);
  }
  const float tot = waveReduceTranspose(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s_part[wave][lane] = tot;
  __syncthreads();
  if (threadIdx.x < ACC_PAD) {
    float s = 0.0f;
#pragma unroll
    for (int wv = 0; wv < T / 64; wv++) s += s_part[wv][threadIdx.x];
    s_tot[threadIdx.x] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// single evaluation (host-driven LM / VIO mode hand-off / parity unit): G workgroups -> partials
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(T) k_eval_partial(const TrackerDev trk, const EvalP e, const float4* __restrict__ img, float* __restrict__ partials) {
  __shared__ float s_part[T / 64][ACC_PAD];
  __shared__ float s_tot[ACC_PAD];
  const int lvl = e.lvl;
  blockEval<T>(e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], blockIdx.x * T + threadIdx.x, gridDim.x * T, img, trk.huberTH, s_part, s_tot);
  if (threadIdx.x < ACC_PAD) partials[blockIdx.x * ACC_PAD + threadIdx.x] = s_tot[threadIdx.x];
}
__global__ void __launch_bounds__(64) k_eval_final(const float* __restrict__ partials, const int G, float* __restrict__ out) {
  float s = 0.0f;
  for (int g = 0; g < G; g++) s += partials[g * ACC_PAD + threadIdx.x];
  out[threadIdx.x] = s;
}

// H (8x8, double, SCALE_*-scaled) and b from the 45 sums, exactly as calcGSSSE's tail does (:340-355).
DMV_HD void systemFromSums(const float* tot, double* H, double* b) {
  const int nW = (int)tot[ACC_NW];
  const int n = (nW + 3) & ~3;  // buf_warped_n is padded to a multiple of 4 (CoarseTracker.cpp:486-498)
  const float invn = 1.0f / n;
  double M[9][9];
  int k = 0;
  for (int r = 0; r < 9; r++)
    for (int c = r; c < 9; c++) { M[r][c] = M[c][r] = (double)tot[ACC_H + k]; k++; }
  const float sc[8] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 10.0f, 1000.0f};  // SCALE_XI_ROT/TRANS, SCALE_A, SCALE_B
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H[r * 8 + c] = ((M[r][c] * invn) * sc[c]) * sc[r];
    b[r] = (M[r][8] * invn) * sc[r];
  }
}
DMV_HD void res6FromSums(const float* tot, double rs[6]) {
  rs[0] = tot[ACC_E];
  rs[1] = (int)tot[ACC_NE];
  rs[2] = tot[ACC_FT] / (tot[ACC_FN] + 0.1);
  rs[3] = 0;
  rs[4] = tot[ACC_FRT] / (tot[ACC_FN] + 0.1);
  rs[5] = (int)tot[ACC_NSAT] / (float)(int)tot[ACC_NE];
}

// Uniform parameters of an evaluation at (pose, aff) on level lvl (CoarseTracker.cpp:377-379, 385).
DMV_HD void makeEvalP(const TrackerDev& trk, int lvl, const Pose& T, double affA, double affB, float new_exposure, float cutoffTH, EvalP& e) {
  double Rd[9];
  quatToR(T.q, Rd);
  float Rf[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
  const float* Ki = trk.g[lvl].Ki;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) e.RKi[r * 3 + c] = Rf[r * 3 + 0] * Ki[0 * 3 + c] + Rf[r * 3 + 1] * Ki[1 * 3 + c] + Rf[r * 3 + 2] * Ki[2 * 3 + c];
  e.t[0] = (float)T.t[0]; e.t[1] = (float)T.t[1]; e.t[2] = (float)T.t[2];
  double aff[2];
  affFromTo(trk.ref_exposure, new_exposure, trk.ref_aff_a, trk.ref_aff_b, affA, affB, aff);
  e.aff0 = (float)aff[0]; e.aff1 = (float)aff[1];
  e.b0 = (float)trk.ref_aff_b;
  e.cutoff = cutoffTH;
  e.maxEnergy = 2 * trk.huberTH * cutoffTH - trk.huberTH * trk.huberTH;
  e.lvl = lvl;
}

// ---------------------------------------------------------------------------------------------
// device-resident trackNewestCoarse: one workgroup per alignment problem
// ---------------------------------------------------------------------------------------------
// The LM control step between two evaluations is executed by wave 0 of the workgroup:
//   * scalar decisions (accept/reject, lambda schedule, level logic) by lane 0,
//   * the 8x8 system: lane (r*8+c) owns H(r,c); H,b from the 45 sums, damping and the pivoted LDL^T solve
//     are wave-wide register + cross-lane operations (no scratch memory, no serial fp64 loops),
//   * SE3 exp / pose composition by lane 0 (static indices only -> registers).
enum { LM_LEVEL_BEGIN = 0, LM_INIT_EVAL, LM_ITER_BEGIN, LM_ITER_EVAL, LM_LEVEL_END };
enum { ACT_DONE = 0, ACT_EVAL_CUR = 1, ACT_SOLVE = 2 };

struct LMState {
  Pose cur, nxt;
  double affA, affB, affA_n, affB_n;
  double resOld[6];
  double incNorm;
  double lastRes[5];
  double flow[3];
  float lambda, cutoffRepeat;
  int lvl, iteration, st, totalIts, nEvals;
  long long nPointEvals;
  int haveRepeated;
};

// H(r,c) (SCALE_*-scaled, double) of lane = r*8+c from the 45 sums — calcGSSSE's tail (CoarseTracker.cpp:340-355).
__device__ __forceinline__ double systemEntryFromSums(const float* tot, const int r, const int c) {
  const int nW = (int)tot[ACC_NW];
  const int n = (nW + 3) & ~3;
  const float invn = 1.0f / n;
  const int rr = r < c ? r : c, cc = r < c ? c : r;
  const float scr = r == 6 ? 10.0f : (r == 7 ? 1000.0f : 1.0f);
  const float scc = c == 6 ? 10.0f : (c == 7 ? 1000.0f : 1.0f);
  return (((double)tot[accIdx(rr, cc)] * invn) * scc) * scr;
}
__device__ __forceinline__ double rhsEntryFromSums(const float* tot, const int r) {
  const int nW = (int)tot[ACC_NW];
  const int n = (nW + 3) & ~3;
  const float invn = 1.0f / n;
  const float scr = r == 6 ? 10.0f : (r == 7 ? 1000.0f : 1.0f);
  return ((double)tot[accIdx(r, 8)] * invn) * scr;
}

// Wave-cooperative LDL^T with symmetric diagonal pivoting (largest |d| first — the pivot rule of the
// decomposition CoarseTracker.cpp:639 calls).  lane = r*8+c holds m = A(r,c); dv = rhs(r) (replicated over c).
// Returns x(r) in every lane of row r.
__device__ __forceinline__ double waveLdltSolve8(double m, double dv, const int lane) {
  const int r = lane >> 3, c = lane & 7;
  int trk[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int p = k;
    double best = fabs(__shfl(m, k * 9, 64));
#pragma unroll
    for (int i = k + 1; i < 8; i++) {
      const double v = fabs(__shfl(m, i * 9, 64));
      if (v > best) { best = v; p = i; }
    }
    trk[k] = p;
    {
      const int pr = (r == k) ? p : ((r == p) ? k : r);
      const int pc = (c == k) ? p : ((c == p) ? k : c);
      m = __shfl(m, pr * 8 + pc, 64);
      dv = __shfl(dv, pr * 8 + c, 64);
    }
    const double dk = __shfl(m, k * 9, 64);
    const bool ok = fabs(dk) > 0;
    const double mrk = __shfl(m, r * 8 + k, 64);
    const double mck = __shfl(m, c * 8 + k, 64);
    const double Lrk = ok ? mrk / dk : mrk;
    const double Lck = ok ? mck / dk : mck;
    if (r > k && c > k) m = m - Lrk * (dk * Lck);
    if (c == k && r > k) m = Lrk;
  }
  // forward substitution  (L y = P b)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double dkv = __shfl(dv, k * 8, 64);
    const double Lrk = __shfl(m, r * 8 + k, 64);
    if (r > k) dv = dv - Lrk * dkv;
  }
  {
    const double D = __shfl(m, r * 9, 64);
    dv = (fabs(D) > 2.2250738585072014e-308) ? dv / D : 0.0;
  }
  // backward substitution  (L^T x = z)
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    const double dkv = __shfl(dv, k * 8, 64);
    const double Lkr = __shfl(m, k * 8 + r, 64);
    if (r < k) dv = dv - Lkr * dkv;
  }
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    const int p = trk[k];
    const int pr = (r == k) ? p : ((r == p) ? k : r);
    dv = __shfl(dv, pr * 8 + c, 64);
  }
  return dv;
}

// One LM control step, executed by all 64 lanes of wave 0.  Consumes the finished evaluation in s_tot, decides,
// and either prepares the next evaluation (s_e, returns true) or finishes the problem (returns false).
__device__ __forceinline__ bool lmWaveStep(LMState& S, const TrackerDev& trk, const LMProblemIn& in, LMProblemOut& out,
                                           const float* s_tot, double* s_H, double* s_b, double* s_x, EvalP& s_e, const int lane) {
  const int maxIterations[5] = {10, 20, 50, 50, 50};
  const float lambdaExtrapolationLimit = 0.001f;
  int takeH = 0, action = ACT_DONE;
  if (lane == 0) {
    // (1) consume the evaluation that just finished
    if (S.st == LM_INIT_EVAL) {
      res6FromSums(s_tot, S.resOld);
      if (S.resOld[5] > 0.6 && (S.cutoffRepeat < 50 || S.resOld[5] > 0.99)) {
        S.cutoffRepeat *= 2;
        action = ACT_EVAL_CUR;  // same pose, doubled cutoff; stay in LM_INIT_EVAL
      } else {
        takeH = 1;
        S.lambda = 0.01f;
        S.iteration = 0;
        S.st = LM_ITER_BEGIN;
      }
    } else if (S.st == LM_ITER_EVAL) {
      double resNew[6];
      res6FromSums(s_tot, resNew);
      const bool accept = (resNew[0] / resNew[1]) < (S.resOld[0] / S.resOld[1]);
      if (accept) {
        takeH = 1;
        for (int i = 0; i < 6; i++) S.resOld[i] = resNew[i];
        S.affA = S.affA_n; S.affB = S.affB_n;
        S.cur = S.nxt;
        S.lambda *= 0.5f;
      } else {
        S.lambda *= 4;
        if (S.lambda < lambdaExtrapolationLimit) S.lambda = lambdaExtrapolationLimit;
      }
      S.totalIts++;
      S.iteration++;
      S.st = (!(S.incNorm > 1e-3)) ? LM_LEVEL_END : LM_ITER_BEGIN;
    }
    // (2) bookkeeping until the next evaluation (or the end) is determined
    if (action != ACT_EVAL_CUR) {
      for (;;) {
        if (S.st == LM_ITER_BEGIN) {
          if (S.iteration >= maxIterations[S.lvl]) { S.st = LM_LEVEL_END; continue; }
          action = ACT_SOLVE;
          break;
        }
        if (S.st == LM_LEVEL_END) {
          S.lastRes[S.lvl] = sqrtf((float)(S.resOld[0] / S.resOld[1]));
          S.flow[0] = S.resOld[2]; S.flow[1] = S.resOld[3]; S.flow[2] = S.resOld[4];
          const bool failed = isnan(S.lastRes[S.lvl]) || (S.lastRes[S.lvl] > 1.5 * in.minRes[S.lvl]);
          if (failed) {
            // reference returns false without touching lastToNew_out / aff_g2l_out (CoarseTracker.cpp:731-732)
            for (int i = 0; i < 7; i++) out.pose7[i] = in.pose7[i];
            out.aff[0] = in.aff[0]; out.aff[1] = in.aff[1];
            out.good = 0;
            action = ACT_DONE;
            break;
          }
          if (S.cutoffRepeat > 1 && !S.haveRepeated) { S.lvl++; S.haveRepeated = 1; }
          S.lvl--;
          S.st = LM_LEVEL_BEGIN;
          continue;
        }
        // LM_LEVEL_BEGIN
        if (S.lvl < 0) {
          // success: write back (CoarseTracker.cpp:743-760)
          double aff[2] = {S.affA, S.affB};
          bool good = true;
          if ((trk.modeA != 0 && (fabsf((float)aff[0]) > 1.2f)) || (trk.modeB != 0 && (fabsf((float)aff[1]) > 200.0f))) good = false;
          double rel[2];
          affFromTo(trk.ref_exposure, in.new_exposure, trk.ref_aff_a, trk.ref_aff_b, aff[0], aff[1], rel);
          if ((trk.modeA == 0 && (fabsf(logf((float)rel[0])) > 1.5f)) || (trk.modeB == 0 && (fabsf((float)rel[1]) > 200.0f))) good = false;
          if (trk.modeA < 0) aff[0] = 0;
          if (trk.modeB < 0) aff[1] = 0;
          poseTo7(S.cur, out.pose7);
          out.aff[0] = aff[0]; out.aff[1] = aff[1];
          out.good = good ? 1 : 0;
          action = ACT_DONE;
          break;
        }
        S.cutoffRepeat = 1;
        S.st = LM_INIT_EVAL;
        action = ACT_EVAL_CUR;
        break;
      }
    }
  }
  takeH = __builtin_amdgcn_readfirstlane(takeH);
  action = __builtin_amdgcn_readfirstlane(action);

  const int r = lane >> 3, c = lane & 7;
  double hv, bv;
  if (takeH) {
    hv = systemEntryFromSums(s_tot, r, c);
    bv = rhsEntryFromSums(s_tot, r);
    s_H[lane] = hv;
    if (c == 0) s_b[r] = bv;
  } else {
    hv = s_H[lane];
    bv = s_b[r];
  }

  if (action == ACT_SOLVE) {
    const float lambda = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lane == 0 ? S.lambda : 0.0f)));
    const bool fixA = trk.modeA < 0, fixB = trk.modeB < 0;
    // damped system Hl = H, diag *= (1+lambda); rhs = -b          (CoarseTracker.cpp:601-602, 639)
    double m = (r == c) ? hv * (1 + lambda) : hv;
    double dv = -bv;
    if (fixA && !fixB) {
      // stitch b's row/col into slot 6 and drop slot 7 (CoarseTracker.cpp:653-664)
      const int sr = (r == 6) ? 7 : r, sc = (c == 6) ? 7 : c;
      m = __shfl(m, sr * 8 + sc, 64);
      dv = __shfl(dv, sr * 8 + c, 64);
    }
    const int nact = (fixA && fixB) ? 6 : ((fixA || fixB) ? 7 : 8);
    if (r >= nact || c >= nact) { m = (r == c) ? 1.0 : 0.0; }
    if (r >= nact) dv = 0.0;
    double x = waveLdltSolve8(m, dv, lane);
    if (fixA && !fixB) {
      // inc[7] = incStitch[6]; inc[6] = 0
      const double x6 = __shfl(x, 6 * 8, 64);
      x = (r == 7) ? x6 : ((r == 6) ? 0.0 : x);
    }
    if (c == 0) s_x[r] = x;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  if (lane == 0) {
    if (action == ACT_SOLVE) {
      float extrapFac = 1;
      if (S.lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / S.lambda));
      double inc[8], incScaled[8];
#pragma unroll
      for (int i = 0; i < 8; i++) inc[i] = s_x[i] * extrapFac;
#pragma unroll
      for (int i = 0; i < 6; i++) incScaled[i] = inc[i] * 1.0f;  // SCALE_XI_ROT / SCALE_XI_TRANS
      incScaled[6] = inc[6] * 10.0f;                               // SCALE_A
      incScaled[7] = inc[7] * 1000.0f;                             // SCALE_B
      double ssum = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) ssum += incScaled[i];
      if (!isfinite(ssum)) {
#pragma unroll
        for (int i = 0; i < 8; i++) incScaled[i] = 0;
      }
      S.nxt = poseMul(poseExp(incScaled), S.cur);
      S.affA_n = S.affA + incScaled[6];
      S.affB_n = S.affB + incScaled[7];
      double nn = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) nn += inc[i] * inc[i];
      S.incNorm = sqrt(nn);
      makeEvalP(trk, S.lvl, S.nxt, S.affA_n, S.affB_n, in.new_exposure, trk.coarseCutoffTH * S.cutoffRepeat, s_e);
      S.st = LM_ITER_EVAL;
    } else if (action == ACT_EVAL_CUR) {
      makeEvalP(trk, S.lvl, S.cur, S.affA, S.affB, in.new_exposure, trk.coarseCutoffTH * S.cutoffRepeat, s_e);
    }
    if (action != ACT_DONE) { S.nEvals++; S.nPointEvals += trk.pc_n[S.lvl]; }
  }
  return action != ACT_DONE;
}

template <int T>
__global__ void __launch_bounds__(T) k_track_lm(const TrackerDev trk, const FrameStore fs, const LMProblemIn* __restrict__ in,
                                                 LMProblemOut* __restrict__ out, const int coarsestLvl) {
  __shared__ float s_part[T / 64][ACC_PAD];
  __shared__ float s_tot[ACC_PAD];
  __shared__ EvalP s_e;
  __shared__ double s_H[64], s_b[8], s_x[8];
  __shared__ int s_go;
  __shared__ LMState S;  // written by lane 0 of wave 0 only
  const LMProblemIn& pin = in[blockIdx.x];
  LMProblemOut& pout = out[blockIdx.x];
  if (threadIdx.x == 0) {
    S.cur = poseFrom7(pin.pose7);
    S.affA = pin.aff[0]; S.affB = pin.aff[1];
    for (int i = 0; i < 5; i++) S.lastRes[i] = __builtin_nan("");
    for (int i = 0; i < 3; i++) S.flow[i] = 1000;
    S.lvl = coarsestLvl; S.st = LM_LEVEL_BEGIN; S.totalIts = 0; S.nEvals = 0; S.nPointEvals = 0; S.haveRepeated = 0;
    S.iteration = 0; S.lambda = 0.01f; S.cutoffRepeat = 1; S.incNorm = 0;
  }
  if (threadIdx.x < 64) { s_H[threadIdx.x] = 0; if (threadIdx.x < 8) { s_b[threadIdx.x] = 0; s_x[threadIdx.x] = 0; } }
  __syncthreads();
  const int slot = pin.new_slot;
  long long tStep = 0, tEval = 0;
  for (;;) {
    const long long t0 = wall_clock64();
    if (threadIdx.x < 64) {
      const bool go = lmWaveStep(S, trk, pin, pout, s_tot, s_H, s_b, s_x, s_e, threadIdx.x);
      if (threadIdx.x == 0) s_go = go ? 1 : 0;
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    tStep += t1 - t0;
    if (!s_go) break;
    const int lvl = s_e.lvl;
    blockEval<T>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], threadIdx.x, T, fs.level(slot, lvl), trk.huberTH, s_part, s_tot);
    tEval += wall_clock64() - t1;
  }
  if (threadIdx.x == 0) {
    for (int i = 0; i < 5; i++) pout.lastRes[i] = S.lastRes[i];
    for (int i = 0; i < 3; i++) pout.flow[i] = S.flow[i];
    pout.iterations = S.totalIts;
    pout.n_evals = S.nEvals;
    pout.n_point_evals = S.nPointEvals;
    pout.ticks_step = tStep;
    pout.ticks_eval = tEval;
  }
  if (threadIdx.x < 64) pout.H[threadIdx.x] = s_H[threadIdx.x];
  if (threadIdx.x < 8) pout.b[threadIdx.x] = s_b[threadIdx.x];
}

}  // namespace dmv
