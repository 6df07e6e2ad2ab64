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
//     point: 16 B point record + 4 taps x 12 B = 64 B (the kernel actually reads 16 B + a 48 B intensity
//     neighbourhood and rebuilds the gradients, see interp33).
//   * the 9x9 weighted outer-product reduction runs on the matrix cores (v_mfma_f32_16x16x4_f32, fp32 in /
//     fp32 accumulate): per wave the 64 Jacobian rows are staged in LDS and consumed as a (9 x 64)(64 x 9)
//     product, 4 accumulator registers per lane instead of 45.  The first version of this kernel kept 52
//     register accumulators per lane (230 VGPRs, 2 waves/SIMD, latency bound — profiles/r01_*); the MFMA
//     form is what buys the occupancy.  The 7 scalar statistics use a transposing wave64 butterfly.
//     Fixed summation order, bitwise reproducible run to run.
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
#include "interp.hpp"

namespace dmv {

#ifdef DMV_LM_TICKS
__device__ double g_lm_ticks[8];   // experiment only (profiles/r02_lm_control_step.md): LM control step — solve, lane-0 part 1, lane-0 part 2, whole step (100 MHz ticks), solves, steps; evaluation server — [6] seen -> evaluated, [7] evaluated -> stored
#endif


// Accumulator9 slot of H(r,c), r <= c: rows of the upper triangle back to back (MatrixAccumulators.h:1091-1166).
__host__ __device__ constexpr int accIdx(int r, int c) { return ACC_H + r * 9 - (r * (r - 1)) / 2 + (c - r); }

// per-lane running statistics of calcRes (everything except the 9x9 outer products)
struct EvalStats {
  float E, nE, nSat, nW, fT, fRT, fN;
};

// Wave-uniform copy of the evaluation parameters in scalar registers (the LDS / kernel-argument originals would be re-read every
// iteration: the compiler cannot prove they do not alias the staging writes).
__device__ __forceinline__ float uni(const float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
struct EvalU {
  float RKi[9], t[3], aff0, aff1, b0, cutoff, maxEnergy;
  float fx, fy, cx, cy, wM3, hM3, huberTH;
  int w;
};
__device__ __forceinline__ EvalU makeEvalU(const EvalP& e, const LevelGeom& g, const float huberTH) {
  EvalU u;
#pragma unroll
  for (int k = 0; k < 9; k++) u.RKi[k] = uni(e.RKi[k]);
#pragma unroll
  for (int k = 0; k < 3; k++) u.t[k] = uni(e.t[k]);
  u.aff0 = uni(e.aff0); u.aff1 = uni(e.aff1); u.b0 = uni(e.b0); u.cutoff = uni(e.cutoff); u.maxEnergy = uni(e.maxEnergy);
  u.fx = uni(g.fx); u.fy = uni(g.fy); u.cx = uni(g.cx); u.cy = uni(g.cy);
  u.w = __builtin_amdgcn_readfirstlane(g.w);
  u.wM3 = (float)(u.w - 3); u.hM3 = (float)(__builtin_amdgcn_readfirstlane(g.h) - 3);
  u.huberTH = uni(huberTH);
  return u;
}

// One template point: everything calcRes does for it (flow indicators excepted, see flowSamplePoint) plus its calcGS row.
// Returns the row J[0..8] = (J0..J7, r) and its Huber weight w (w = 0 and J = 0 when the point does not enter the system).
// Written without branches: lanes that fail a test keep computing on a safe in-bounds tap and are masked out at the end —
// the arithmetic of the surviving lanes is the reference's, operation for operation.
// Split in two so that the taps of the NEXT point are in flight while the current one is finished (software pipelining):
// projectPoint = warp + bounds test + the four tap loads; finishPoint = everything that consumes the taps.
struct PointProj { float u, v, Ku, Kv, new_idepth, refColor; bool inb; };

// IEEE fp32 division as the compiler expands it (LLVM AMDGPU LowerFDIV32: div_scale x2, rcp, fma x2, mul, fma x3, div_fmas, div_fixup),
// minus the range scaling and the special-case fix-up, with the refined reciprocal shared by all quotients of one denominator.
// For operands in the normal range without an extreme exponent gap (no scaling applied by v_div_scale) the operation sequence is the
// same, hence the same correctly rounded quotient; every surviving lane of the evaluation loop is in that range (pixel coordinates,
// inverse depths, residuals), lanes holding 0 / Inf / NaN are masked by the tests that follow.  11 -> 3 + 5 instructions per quotient.
struct Recip { float nb, r1; };
__device__ __forceinline__ Recip refinedRcp(const float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float e0 = __builtin_fmaf(-b, r, 1.0f);
  return Recip{-b, __builtin_fmaf(e0, r, r)};
}
__device__ __forceinline__ float divBy(const float a, const Recip& R) {
  const float q0 = a * R.r1;
  const float e1 = __builtin_fmaf(R.nb, q0, a);
  const float q1 = __builtin_fmaf(e1, R.r1, q0);
  const float e2 = __builtin_fmaf(R.nb, q1, a);
  return __builtin_fmaf(e2, R.r1, q1);
}
template <bool TILED = false>
__device__ __forceinline__ void projectPoint(const float4 P, const bool live, const EvalU& e, const float* __restrict__ img, PointProj& q, Taps33& taps) {
  const float x = P.x, y = P.y, id = P.z;
  const float pt0 = e.RKi[0] * x + e.RKi[1] * y + e.RKi[2] * 1.0f + e.t[0] * id;
  const float pt1 = e.RKi[3] * x + e.RKi[4] * y + e.RKi[5] * 1.0f + e.t[1] * id;
  const float pt2 = e.RKi[6] * x + e.RKi[7] * y + e.RKi[8] * 1.0f + e.t[2] * id;
  const Recip rz = refinedRcp(pt2);
  q.u = divBy(pt0, rz); q.v = divBy(pt1, rz);
  const float Ku = e.fx * q.u + e.cx, Kv = e.fy * q.v + e.cy;
  q.new_idepth = divBy(id, rz);
  q.refColor = P.w;
  q.inb = live && (Ku > 2 && Kv > 2 && Ku < e.wM3 && Kv < e.hM3 && q.new_idepth > 0);
  q.Ku = q.inb ? Ku : 2.5f; q.Kv = q.inb ? Kv : 2.5f;   // masked lanes tap a safe pixel
  if (TILED) interp33LoadTiled(img, q.Ku, q.Kv, e.w >> 3, taps);
  else interp33Load(img, q.Ku, q.Kv, e.w, taps);
}
template <bool GUARD>
__device__ __forceinline__ void finishPoint(const PointProj& q, const Taps33& taps, const EvalU& e, EvalStats& st, float (&J)[9], float& wOut) {
  const float u = q.u, v = q.v, new_idepth = q.new_idepth, refColor = q.refColor;
  const float3 hit = interp33Finish<GUARD>(taps, q.Ku, q.Kv);
  const bool fin = GUARD ? (q.inb && isfinite(hit.x)) : q.inb;
  const float residual = hit.x - (e.aff0 * refColor + e.aff1);
  const float ar = fabsf(residual);
  const float hw = ar < e.huberTH ? 1.0f : divBy(e.huberTH, refinedRcp(ar));
  const bool sat = ar > e.cutoff, ok = fin && !sat;
  st.nE += fin ? 1.0f : 0.0f;
  st.nSat += (fin && sat) ? 1.0f : 0.0f;
  st.nW += ok ? 1.0f : 0.0f;
  st.E += fin ? (sat ? e.maxEnergy : hw * residual * residual * (2 - hw)) : 0.0f;   // x + 0 == x: masked lanes leave E untouched
  // calcGSSSE row (CoarseTracker.cpp:314-338)
  const float dx = hit.y * e.fx, dy = hit.z * e.fy;
  J[0] = ok ? new_idepth * dx : 0.0f;
  J[1] = ok ? new_idepth * dy : 0.0f;
  J[2] = ok ? 0.0f - new_idepth * (u * dx + v * dy) : 0.0f;
  J[3] = ok ? 0.0f - ((u * v) * dx + dy * (1.0f + v * v)) : 0.0f;
  J[4] = ok ? (u * v) * dy + dx * (1.0f + u * u) : 0.0f;
  J[5] = ok ? u * dy - v * dx : 0.0f;
  J[6] = ok ? e.aff0 * (e.b0 - refColor) : 0.0f;
  J[7] = ok ? -1.0f : 0.0f;
  J[8] = ok ? residual : 0.0f;
  wOut = ok ? hw : 0.0f;
}

// Flow indicators of one sample point (CoarseTracker.cpp:416-447): every 32nd point of the reference's row-major list, level 0
// only.  Run as a separate dense pass over the flagged template entries (1/32 of level 0), not inside the evaluation loop.
__device__ __forceinline__ void flowSamplePoint(const float4 P, const EvalU& e, const float* __restrict__ Ki, EvalStats& st) {
  const float x = P.x, y = P.y, id = P.z;
  const float r0 = e.RKi[0] * x + e.RKi[1] * y + e.RKi[2] * 1.0f;
  const float r1 = e.RKi[3] * x + e.RKi[4] * y + e.RKi[5] * 1.0f;
  const float r2 = e.RKi[6] * x + e.RKi[7] * y + e.RKi[8] * 1.0f;
  const float pt0 = r0 + e.t[0] * id, pt1 = r1 + e.t[1] * id, pt2 = r2 + e.t[2] * id;
  const float Ku = e.fx * (pt0 / pt2) + e.cx, Kv = e.fy * (pt1 / pt2) + e.cy;
  const float k0 = Ki[0] * x + Ki[1] * y + Ki[2] * 1.0f;
  const float k1 = Ki[3] * x + Ki[4] * y + Ki[5] * 1.0f;
  const float k2 = Ki[6] * x + Ki[7] * y + Ki[8] * 1.0f;
  const float a0 = k0 + e.t[0] * id, a1 = k1 + e.t[1] * id, a2 = k2 + e.t[2] * id;
  const float KuT = e.fx * (a0 / a2) + e.cx, KvT = e.fy * (a1 / a2) + e.cy;
  const float b0 = k0 - e.t[0] * id, b1 = k1 - e.t[1] * id, b2 = k2 - e.t[2] * id;
  const float KuT2 = e.fx * (b0 / b2) + e.cx, KvT2 = e.fy * (b1 / b2) + e.cy;
  const float c0 = r0 - e.t[0] * id, c1 = r1 - e.t[1] * id, c2 = r2 - e.t[2] * id;
  const float Ku3 = e.fx * (c0 / c2) + e.cx, Kv3 = e.fy * (c1 / c2) + e.cy;
  float sT = (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
  sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
  float sRT = (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
  sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
  st.fT += sT;
  st.fRT += sRT;
  st.fN += 2.0f;
}

// Self-test of divBy: both quotients of every pair, so a test can compare them bit for bit (and against the host's IEEE division)
__global__ void __launch_bounds__(256) k_selftest_divide(const int n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ q_shared,
                                                         float* __restrict__ q_ieee) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  q_shared[i] = divBy(a[i], refinedRcp(b[i]));
  q_ieee[i] = a[i] / b[i];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS staging of one wave: the 64 rows J (transposed: component-major, padded stride) + the 64 weights.
// Stride 66 makes both the component-major writes (lane -> consecutive banks) and the MFMA operand reads
// (lane (i = l&15, k = l>>4) reads component i of point 4m+k: bank (2i + k + 4m) mod 32) conflict free.
enum { SJ_STRIDE = 66, SJ_ROWS = 16, SJ_WAVE_FLOATS = SJ_ROWS * SJ_STRIDE + 64 };

// 7 per-lane statistics -> wave totals: transposing butterfly over 8 slots (lane l ends with the total of slot l&7).
__device__ __forceinline__ float waveReduceStats(const EvalStats& st, const int lane) {
  float v[8] = {st.E, st.nE, st.nSat, st.nW, st.fT, st.fRT, st.fN, 0.0f};
  {
    const bool hi = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float keep = hi ? v[i + 4] : v[i], send = hi ? v[i] : v[i + 4];
      v[i] = keep + __shfl_xor(send, 4, 64);
    }
  }
  {
    const bool hi = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float keep = hi ? v[i + 2] : v[i], send = hi ? v[i] : v[i + 2];
      v[i] = keep + __shfl_xor(send, 2, 64);
    }
  }
  {
    const bool hi = (lane & 1) != 0;
    const float keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1];
    v[0] = keep + __shfl_xor(send, 1, 64);
  }
  float t = v[0];
  t += __shfl_xor(t, 8, 64);
  t += __shfl_xor(t, 16, 64);
  t += __shfl_xor(t, 32, 64);
  return t;  // total of slot (lane & 7)
}

// Workgroup-wide evaluation over the template points first, first+stride, ... < n (first = this THREAD's first index;
// stride = threads taking part).  The weighted 9x9 outer products  sum_p w_p J_p J_p^T  are accumulated on the matrix
// cores: per wave, 64 rows are staged component-major in LDS and consumed by 16 v_mfma_f32_16x16x4_f32 (A = J,
// B = w*J, K = 4 points per instruction, 9 of the 16 rows/columns used).  fp32 in, fp32 accumulate — same numerics
// class as the reference's fp32 sums; the accumulator costs 4 registers per lane instead of 45, which is what lets
// 4-8 waves per SIMD stay resident to hide the gather latency.  Fixed summation order (bitwise reproducible).
// Result: s_tot[0..63] (ACC_* slots) valid for all threads after return.
// GUARD = false: the new frame is stamped clean (FrameStore::bad_gen), the isfinite guards of the taps are compiled out
// TILED: `img` is a level-0 plane stored in 8x4 tiles (FrameStore::tiled0) — same twelve values per tap, other addresses
template <int T, bool GUARD = true, bool TILED = false>
__device__ __forceinline__ void blockEval(const EvalP& e, const LevelGeom& g, const float4* __restrict__ pc, const int n,
                                          const unsigned long long* __restrict__ flow_mask, const int first, const int stride,
                                          const float* __restrict__ img, const float huberTH, float* s_stage, float* s_partH,
                                          float (*s_partS)[8], float* s_tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* __restrict__ wJ = s_stage + wave * SJ_WAVE_FLOATS;
  float* __restrict__ wW = wJ + SJ_ROWS * SJ_STRIDE;
  // four independent accumulator chains: a 16x16x4 f32 MFMA has a 40-cycle dependent latency but a 32-cycle issue slot
  f32x4 accH0 = {0.0f, 0.0f, 0.0f, 0.0f}, accH1 = accH0, accH2 = accH0, accH3 = accH0;
  EvalStats st = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool lvl0 = (e.lvl == 0);
  const EvalU eu = makeEvalU(e, g, huberTH);
  const int mi = lane & 15, mk = lane >> 4;
  // wave-uniform trip count: lane 0 of the wave owns index first - lane.  The template record of the NEXT iteration is
  // requested before this iteration's taps, so the two dependent memory round trips of a point (record -> projection ->
  // taps) overlap across iterations.
  const int base0 = first - lane;
  // three-stage software pipeline per lane: template record two points ahead, warp + tap loads one point ahead, residual /
  // Jacobian row / MFMA of the current point — the gathers of point i+1 are in flight during the arithmetic of point i
  const int nm1 = max(n - 1, 0);
  float4 P1 = make_float4(0.f, 0.f, 0.f, 0.f);
  PointProj q0;
  Taps33 t0;
  q0.u = q0.v = q0.new_idepth = q0.refColor = 0.f; q0.Ku = q0.Kv = 2.5f; q0.inb = false;
  if (n > 0) {
    const float4 P0 = pc[min(base0 + lane, nm1)];
    P1 = pc[min(base0 + lane + stride, nm1)];
    projectPoint<TILED>(P0, base0 + lane < n, eu, img, q0, t0);
  }
  // one pipeline step: finishes the point held in (qc, tc) while the taps of the next one are requested into (qn, tn).  The loop below is
  // unrolled twice with the two register sets swapping roles, so the hand-over costs no register moves (19 per step otherwise).
  auto step = [&](const PointProj& qc, const Taps33& tc, PointProj& qn, Taps33& tn, const int base) __attribute__((always_inline)) {
    const int i = base + lane;
    projectPoint<TILED>(P1, i + stride < n, eu, img, qn, tn);     // next point: its taps are requested now, consumed next step
    P1 = pc[min(i + 2 * stride, nm1)];                     // unconditional (clamped) prefetch
    float J[9], w;
    finishPoint<GUARD>(qc, tc, eu, st, J, w);
#pragma unroll
    for (int k = 0; k < 9; k++) wJ[k * SJ_STRIDE + lane] = J[k];
    wW[lane] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 0; m < 16; m += 4) {
      const float a0 = wJ[mi * SJ_STRIDE + 4 * m + mk], a1 = wJ[mi * SJ_STRIDE + 4 * m + 4 + mk];
      const float a2 = wJ[mi * SJ_STRIDE + 4 * m + 8 + mk], a3 = wJ[mi * SJ_STRIDE + 4 * m + 12 + mk];
      const float b0 = a0 * wW[4 * m + mk], b1 = a1 * wW[4 * m + 4 + mk], b2 = a2 * wW[4 * m + 8 + mk], b3 = a3 * wW[4 * m + 12 + mk];
      accH0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, accH0, 0, 0, 0);
      accH1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accH1, 0, 0, 0);
      accH2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, accH2, 0, 0, 0);
      accH3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, accH3, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  };
  PointProj q1;
  Taps33 t1;
  for (int base = base0; base < n; base += 2 * stride) {
    step(q0, t0, q1, t1, base);
    if (base + stride >= n) break;   // wave-uniform
    step(q1, t1, q0, t0, base + stride);
  }
  if (lvl0) {
    // flow-indicator samples: dense pass over the flagged entries (bit j of word k <=> template entry 64k + j)
    const int nwords = (n + 63) >> 6;
    for (int wI = first; wI < nwords; wI += stride) {
      unsigned long long m = flow_mask[wI];
      while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        flowSamplePoint(pc[64 * wI + bit], eu, g.Ki, st);
      }
    }
  }
  // per-wave partials -> LDS.  accH[r] = D[row = mk*4 + r][col = mi]
#pragma unroll
  for (int r = 0; r < 4; r++) s_partH[wave * 256 + (mk * 4 + r) * 16 + mi] = (accH0[r] + accH1[r]) + (accH2[r] + accH3[r]);
  const float stot = waveReduceStats(st, lane);
  if (lane < 8) s_partS[wave][lane] = stot;
  __syncthreads();
  if (threadIdx.x < ACC_PAD) {
    // slot -> (r, c) of the upper triangle, or a statistic
    float s = 0.0f;
    const int k = threadIdx.x;
    if (k < 45) {
      int r = 0, off = 0;
      while (k >= off + (9 - r)) { off += 9 - r; r++; }
      const int c = r + (k - off);
#pragma unroll
      for (int wv = 0; wv < T / 64; wv++) s += s_partH[wv * 256 + r * 16 + c];
    } else if (k < ACC_N) {
#pragma unroll
      for (int wv = 0; wv < T / 64; wv++) s += s_partS[wv][k - 45];
    }
    s_tot[k] = s;
  }
  __syncthreads();
}

// blockEval with an UNEVEN split of the template over the wavefronts (k_track_lm_pp): the points are dealt out in chunks of 64; of every `period` consecutive chunks this wave
// takes the chunks [lo, hi).  (blockEval itself is the case period = stride / 64, one chunk per wave.)  Per point the same arithmetic; the per-wave partial sums group the
// points differently, the waves' partials are still combined in wave order — deterministic, reproducible run to run.
template <int T, bool GUARD = true>
__device__ __forceinline__ void blockEvalChunks(const EvalP& e, const LevelGeom& g, const float4* __restrict__ pc, const int n,
                                                const unsigned long long* __restrict__ flow_mask, const int period, const int lo, const int hi,
                                                const float* __restrict__ img, const float huberTH, float* s_stage, float* s_partH,
                                                float (*s_partS)[8], float* s_tot) {
  constexpr bool TILED = false;
  const int first = threadIdx.x, stride = T;   // (the flow-indicator pass below keeps the even split over the threads)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* __restrict__ wJ = s_stage + wave * SJ_WAVE_FLOATS;
  float* __restrict__ wW = wJ + SJ_ROWS * SJ_STRIDE;
  // four independent accumulator chains: a 16x16x4 f32 MFMA has a 40-cycle dependent latency but a 32-cycle issue slot
  f32x4 accH0 = {0.0f, 0.0f, 0.0f, 0.0f}, accH1 = accH0, accH2 = accH0, accH3 = accH0;
  EvalStats st = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool lvl0 = (e.lvl == 0);
  const EvalU eu = makeEvalU(e, g, huberTH);
  const int mi = lane & 15, mk = lane >> 4;
  // wave-uniform trip count: lane 0 of the wave owns index first - lane.  The template record of the NEXT iteration is
  // requested before this iteration's taps, so the two dependent memory round trips of a point (record -> projection ->
  // taps) overlap across iterations.
  const int m = hi - lo;                                   // chunks per period of this wave (wave-uniform; 0: the wave only takes part in the reduction)
  // chunk c -> the wave's next chunk: inside its run of the period, else the first of the next period
  auto nextBase = [&](const int base) -> int { const int c = base >> 6; return ((c % period) + 1 < hi ? c + 1 : c + period - (m - 1)) << 6; };
  const int base0 = m > 0 ? lo << 6 : n;
  // three-stage software pipeline per lane: template record two points ahead, warp + tap loads one point ahead, residual /
  // Jacobian row / MFMA of the current point — the gathers of point i+1 are in flight during the arithmetic of point i
  const int nm1 = max(n - 1, 0);
  float4 P1 = make_float4(0.f, 0.f, 0.f, 0.f);
  PointProj q0;
  Taps33 t0;
  q0.u = q0.v = q0.new_idepth = q0.refColor = 0.f; q0.Ku = q0.Kv = 2.5f; q0.inb = false;
  if (n > 0 && m > 0) {
    const float4 P0 = pc[min(base0 + lane, nm1)];
    P1 = pc[min(nextBase(base0) + lane, nm1)];
    projectPoint<TILED>(P0, base0 + lane < n, eu, img, q0, t0);
  }
  // one pipeline step: finishes the point held in (qc, tc) while the taps of the next one are requested into (qn, tn).  The loop below is
  // unrolled twice with the two register sets swapping roles, so the hand-over costs no register moves (19 per step otherwise).
  auto step = [&](const PointProj& qc, const Taps33& tc, PointProj& qn, Taps33& tn, const int base) __attribute__((always_inline)) {
    const int nb = nextBase(base);
    projectPoint<TILED>(P1, nb + lane < n, eu, img, qn, tn);      // next point: its taps are requested now, consumed next step
    P1 = pc[min(nextBase(nb) + lane, nm1)];                // unconditional (clamped) prefetch
    float J[9], w;
    finishPoint<GUARD>(qc, tc, eu, st, J, w);
#pragma unroll
    for (int k = 0; k < 9; k++) wJ[k * SJ_STRIDE + lane] = J[k];
    wW[lane] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 0; m < 16; m += 4) {
      const float a0 = wJ[mi * SJ_STRIDE + 4 * m + mk], a1 = wJ[mi * SJ_STRIDE + 4 * m + 4 + mk];
      const float a2 = wJ[mi * SJ_STRIDE + 4 * m + 8 + mk], a3 = wJ[mi * SJ_STRIDE + 4 * m + 12 + mk];
      const float b0 = a0 * wW[4 * m + mk], b1 = a1 * wW[4 * m + 4 + mk], b2 = a2 * wW[4 * m + 8 + mk], b3 = a3 * wW[4 * m + 12 + mk];
      accH0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, accH0, 0, 0, 0);
      accH1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accH1, 0, 0, 0);
      accH2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, accH2, 0, 0, 0);
      accH3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, accH3, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  };
  PointProj q1;
  Taps33 t1;
  for (int base = base0; base < n;) {
    step(q0, t0, q1, t1, base);
    base = nextBase(base);
    if (base >= n) break;   // wave-uniform
    step(q1, t1, q0, t0, base);
    base = nextBase(base);
  }
  if (lvl0) {
    // flow-indicator samples: dense pass over the flagged entries (bit j of word k <=> template entry 64k + j)
    const int nwords = (n + 63) >> 6;
    for (int wI = first; wI < nwords; wI += stride) {
      unsigned long long m = flow_mask[wI];
      while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        flowSamplePoint(pc[64 * wI + bit], eu, g.Ki, st);
      }
    }
  }
  // per-wave partials -> LDS.  accH[r] = D[row = mk*4 + r][col = mi]
#pragma unroll
  for (int r = 0; r < 4; r++) s_partH[wave * 256 + (mk * 4 + r) * 16 + mi] = (accH0[r] + accH1[r]) + (accH2[r] + accH3[r]);
  const float stot = waveReduceStats(st, lane);
  if (lane < 8) s_partS[wave][lane] = stot;
  __syncthreads();
  if (threadIdx.x < ACC_PAD) {
    // slot -> (r, c) of the upper triangle, or a statistic
    float s = 0.0f;
    const int k = threadIdx.x;
    if (k < 45) {
      int r = 0, off = 0;
      while (k >= off + (9 - r)) { off += 9 - r; r++; }
      const int c = r + (k - off);
#pragma unroll
      for (int wv = 0; wv < T / 64; wv++) s += s_partH[wv * 256 + r * 16 + c];
    } else if (k < ACC_N) {
#pragma unroll
      for (int wv = 0; wv < T / 64; wv++) s += s_partS[wv][k - 45];
    }
    s_tot[k] = s;
  }
  __syncthreads();
}

// zero the rows 9..15 of every wave's staging area once (they are never written afterwards)
template <int T>
__device__ __forceinline__ void initStage(float* s_stage) {
  for (int k = threadIdx.x; k < (T / 64) * SJ_WAVE_FLOATS; k += T) s_stage[k] = 0.0f;
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// single evaluation (host-driven LM / VIO mode hand-off / parity unit): G workgroups -> partials
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(T) k_eval_partial(const TrackerDev trk, const EvalP e, const float* __restrict__ img, float* __restrict__ partials) {
  __shared__ float s_stage[(T / 64) * SJ_WAVE_FLOATS];
  __shared__ float s_partH[(T / 64) * 256];
  __shared__ float s_partS[T / 64][8];
  __shared__ float s_tot[ACC_PAD];
  initStage<T>(s_stage);
  const int lvl = e.lvl;
  blockEval<T>(e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, blockIdx.x * T + threadIdx.x, gridDim.x * T, img, trk.huberTH,
               s_stage, s_partH, s_partS, s_tot);
  if (threadIdx.x < ACC_PAD) partials[blockIdx.x * ACC_PAD + threadIdx.x] = s_tot[threadIdx.x];
}
__global__ void __launch_bounds__(64) k_eval_final(const float* __restrict__ partials, const int G, float* __restrict__ out) {
  float s = 0.0f;
  for (int g = 0; g < G; g++) s += partials[g * ACC_PAD + threadIdx.x];
  out[threadIdx.x] = s;
}

// sum_{g < n} base[g * ACC_PAD] added in index order (the fixed order every multi-workgroup evaluation shares), with the loads of eight partials in flight at a time:
// a load per loop trip would wait for each one's full L2 latency in turn (measured: ≈0.65 µs per workgroup of the evaluation)
__device__ __forceinline__ float sumPartialsInOrder(const float* base, const int n) {
  float s = 0.0f;
  for (int g0 = 0; g0 < n; g0 += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = (g0 + q < n) ? __hip_atomic_load(base + (size_t)(g0 + q) * ACC_PAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
#pragma unroll
    for (int q = 0; q < 8; q++) if (g0 + q < n) s += v[q];
  }
  return s;
}

// The same evaluation as ONE launch whose result reaches the host without a stream synchronisation — the VIO hand-off path
// (CoarseTracker.cpp:612-637: every LM iteration hands H, b to IMUIntegration::computeCoarseUpdate on the host and waits for the
// pose it returns, so launch + wake-up latency is paid ~15 times per frame).  The last workgroup to arrive (one agent-scope counter)
// adds the partial sums in rank order — a fixed order, bit-identical to k_eval_partial + k_eval_final — stores them into pinned,
// host-coherent memory and then releases the launch's ticket next to them; the host spins on that word.
template <int T>
__global__ void __launch_bounds__(T) k_eval_fused(const TrackerDev trk, const FrameStore fs, const int slot, const EvalP e, float* __restrict__ partials,
                                                  unsigned int* __restrict__ arrive, float* __restrict__ out_host, const unsigned int ticket) {
  __shared__ float s_stage[(T / 64) * SJ_WAVE_FLOATS];
  __shared__ float s_partH[(T / 64) * 256];
  __shared__ float s_partS[T / 64][8];
  __shared__ float s_tot[ACC_PAD];
  __shared__ int s_last;
  initStage<T>(s_stage);
  const int lvl = e.lvl;
  const bool clean = __builtin_amdgcn_readfirstlane((int)(fs.bad_gen[slot] != fs.build_gen[slot])) != 0;
  const float* img = dmvUniformGlobal(fs.level(slot, lvl));
  if (clean)
    blockEval<T, false>(e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, blockIdx.x * T + threadIdx.x, gridDim.x * T, img, trk.huberTH,
                        s_stage, s_partH, s_partS, s_tot);
  else
    blockEval<T, true>(e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, blockIdx.x * T + threadIdx.x, gridDim.x * T, img, trk.huberTH,
                       s_stage, s_partH, s_partS, s_tot);
  if (gridDim.x == 1) {
    if (threadIdx.x < ACC_PAD) __hip_atomic_store(out_host + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    if (threadIdx.x < ACC_PAD) __hip_atomic_store(partials + blockIdx.x * ACC_PAD + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < ACC_PAD) {
      const float s = sumPartialsInOrder(partials + threadIdx.x, (int)gridDim.x);
      __hip_atomic_store(out_host + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned int*>(out_host) + ACC_PAD, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Evaluation server of the host-driven LM (dmvio_hip_tracker_track_vio and single-frame tracking): ONE launch per tracked frame instead of one per evaluation.  The host posts
// the parameters of an evaluation into a 128-byte mailbox in host-coherent memory (19 dwords of EvalP, then the request ticket — written last), every workgroup polls the ticket,
// evaluate exactly like k_eval_fused (same split of the template, same rank-order sum: bit-identical sums) and the last one to arrive stores the sums and the ticket into
// host-coherent memory, where the host polls them.  A request costs two PCIe round trips and the evaluation itself; no launch, no stream synchronisation.
// Termination: a ticket with the top bit set, or `idle_ticks` (100 MHz) without a new request — the host relaunches the server if it finds it gone.
#define EVAL_MAIL_DWORDS 32
#define EVAL_RECORD_FLOATS (ACC_PAD + 16)   // per-workgroup result record in host-coherent memory: ACC_PAD sums, then the ticket
#define EVAL_SERVER_MAX_BLOCKS 64
#define EVAL_QUIT_BIT 0x80000000u
// session: the identity of this launch's SESSION (serverStart .. serverStop on the host), also kept in dword EVAL_MAIL_SESSION of the mailbox.  serverStop only posts the quit
// ticket and returns; when the next session's first tickets overwrite it before every workgroup of this launch has polled it, those tickets carry another session
// number — a workgroup that reads one leaves instead of serving the next frame against this launch's slot and template.
#define EVAL_MAIL_SESSION (EVAL_MAIL_DWORDS - 2)
template <int T>
__global__ void __launch_bounds__(T) k_eval_server(const TrackerDev trk, const FrameStore fs, const int slot, const unsigned int* __restrict__ mail, unsigned int* __restrict__ leave,
                                                   const unsigned int first_seen, const long long idle_ticks, float* __restrict__ out_host, const unsigned int session) {
  __shared__ float s_stage[(T / 64) * SJ_WAVE_FLOATS];
  __shared__ float s_partH[(T / 64) * 256];
  __shared__ float s_partS[T / 64][8];
  __shared__ float s_tot[ACC_PAD];
  __shared__ EvalP s_e;
  __shared__ unsigned int s_tk;
  initStage<T>(s_stage);
  const bool clean = __builtin_amdgcn_readfirstlane((int)(fs.bad_gen[slot] != fs.build_gen[slot])) != 0;
  unsigned int seen = first_seen;
  constexpr int EP = (int)(sizeof(EvalP) / 4);
  for (;;) {
    // every workgroup polls the mailbox itself: its first wavefront reads the whole 128 bytes with one load per poll.  The host writes the parameters, then the ticket's copy
    // in the last dword, then the ticket in the first: a read that shows the same new ticket at both ends has the parameters that belong to it, however the 128 bytes were
    // fetched.  (Workgroup 0 polling alone and handing the request on through device memory: +0.7 us per request.)
    if (threadIdx.x < 64) {
      const long long t0 = wall_clock64();
      unsigned int v, front, back;
      bool timeout = false;
      for (;;) {
        v = threadIdx.x < EVAL_MAIL_DWORDS ? __hip_atomic_load(mail + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
        front = __builtin_amdgcn_readlane(v, 0); back = __builtin_amdgcn_readlane(v, EVAL_MAIL_DWORDS - 1);
        if (front == back && front != seen) { if (__builtin_amdgcn_readlane(v, EVAL_MAIL_SESSION) != session) timeout = true; break; }
        // leaving is collective: the first workgroup whose idle limit expires marks this launch (its first ticket is unique to it) and the others follow at their next
        // poll — a request that arrives at that very moment may be picked up by some workgroups only; the host then finds the kernel gone and posts it again
        if (__hip_atomic_load(leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == first_seen) { timeout = true; break; }
        if (wall_clock64() - t0 >= idle_ticks) { __hip_atomic_store(leave, first_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); timeout = true; break; }
      }
      if (threadIdx.x >= 1 && threadIdx.x <= EP) reinterpret_cast<unsigned int*>(&s_e)[threadIdx.x - 1] = v;
      if (threadIdx.x == 0) s_tk = timeout ? (seen | EVAL_QUIT_BIT) : front;
    }
    __syncthreads();
    const unsigned int tk = s_tk;
    if (tk & EVAL_QUIT_BIT) return;
    seen = tk;
#ifdef DMV_LM_TICKS
    const long long q_seen = wall_clock64();
#endif
    const int lvl = s_e.lvl;
    const float* img = dmvUniformGlobal(fs.level(slot, lvl));
    if (clean)
      blockEval<T, false>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, blockIdx.x * T + threadIdx.x, gridDim.x * T, img, trk.huberTH, s_stage, s_partH, s_partS,
                          s_tot);
    else
      blockEval<T, true>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, blockIdx.x * T + threadIdx.x, gridDim.x * T, img, trk.huberTH, s_stage, s_partH, s_partS,
                         s_tot);
    // every workgroup stores its own partial sums and then the request's ticket into ITS record of host-coherent memory; the host waits for all records and adds them in
    // rank order (the same fp32 additions in the same order as the device-side reduction of cluster mode / k_eval_fused: the same bits) — no arrive counter, no second pass
    // over the partials on the device
#ifdef DMV_LM_TICKS
    const long long q_eval = wall_clock64();
#endif
    float* mine = out_host + (size_t)blockIdx.x * EVAL_RECORD_FLOATS;
    if (threadIdx.x < ACC_PAD) __hip_atomic_store(mine + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned int*>(mine) + ACC_PAD, tk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef DMV_LM_TICKS
    if (threadIdx.x == 0) { const long long q_done = wall_clock64(); atomicAdd(&g_lm_ticks[6], (double)(q_eval - q_seen)); atomicAdd(&g_lm_ticks[7], (double)(q_done - q_eval)); atomicAdd(&g_lm_ticks[5], 1.0); }
#endif
  }
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

// value of `v` in lane `src` when `src` is wave-uniform: two v_readlane_b32 (a few cycles) instead of the LDS-crossbar shuffle (~100+)
__device__ __forceinline__ double readLaneD(const double v, const int src) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)b, src), hi = __builtin_amdgcn_readlane((unsigned int)(b >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Wave-cooperative LDL^T with symmetric diagonal pivoting, the decomposition CoarseTracker.cpp:639 calls (Eigen's ldlt_inplace<Lower>::unblocked as the
// host path's ldltSolveInPlace restate it).  lane = r*8+c holds m = A(r,c); dv = rhs(r) (replicated over c).  Returns x(r) in every lane of row r.
//
// Round 5: that algorithm is LEFT-looking — step k updates column k only, so its pivot search over the trailing diagonal always sees ORIGINAL diagonal entries: the whole
// pivot order follows from the diagonal of A (descending |a_ii|).  The order is therefore computed ONCE (two ballots), the system permuted once, and the factorisation runs
// unpivoted: no per-step search (it was a third of the ~1100 dependent instructions of the right-looking form this replaces, which also pivoted on the UPDATED diagonal — not
// the reference's rule), no per-step row / column swaps, no replay of the transpositions at the end.  The column updates are formed as the reference forms them
// (temp_j = D_j L_kj; a_rk -= sum_j L_rj temp_j, the sum added up in j order first).  Ties on the diagonal are broken by index (a STABLE order) — NOT Eigen's rule under
// ties: its selection-with-swaps takes the tied entry that stands first AFTER the earlier swaps (diag [5a, 5b, 9] -> 9, 5b, 5a; the stable order gives 9, 5a, 5b; k_ba_solve
// replays Eigen's order exactly, csrc/ba_batch_kernels.hpp).  Here the only ties that occur are the identity rows of fixed affine parameters, which are uncoupled from
// everything else: their order changes no operand of any other row, the solution is the same bit for bit; a tie between two live diagonal entries of an 8x8 photometric
// Hessian is a measure-zero event and this solve is not bit-pinned (third-party arithmetic).  The back substitution stays column-oriented (each new
// x_k is consumed by all rows at once).
// value of lane (l - k), k = 1..7, for a double: DPP row_shr on both halves (0.0 where the 16-lane DPP row ends)
__device__ __forceinline__ double dppShrD(const double v, const int k) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  int lo = (int)(unsigned int)b, hi = (int)(unsigned int)(b >> 32), rl = 0, rh = 0;
  switch (k) {
    case 1: rl = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true); break;
    case 2: rl = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xF, 0xF, true); break;
    case 3: rl = __builtin_amdgcn_update_dpp(0, lo, 0x113, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x113, 0xF, 0xF, true); break;
    case 4: rl = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xF, 0xF, true); break;
    case 5: rl = __builtin_amdgcn_update_dpp(0, lo, 0x115, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x115, 0xF, 0xF, true); break;
    case 6: rl = __builtin_amdgcn_update_dpp(0, lo, 0x116, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x116, 0xF, 0xF, true); break;
    default: rl = __builtin_amdgcn_update_dpp(0, lo, 0x117, 0xF, 0xF, true); rh = __builtin_amdgcn_update_dpp(0, hi, 0x117, 0xF, 0xF, true); break;
  }
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned int)rh << 32) | (unsigned int)rl);
}
__device__ __noinline__ double waveLdltSolve8(double m, double dv, const int lane, int* s_trk) {
  // loops are deliberately NOT unrolled: this runs once per LM iteration on one wave, while its register footprint is
  // charged to every wave of the kernel (the evaluation loop wants the occupancy).
  (void)s_trk;
  const int r = lane >> 3, c = lane & 7;
  // ---- pivot order: position of element i = number of elements with a larger |a_ii| (ties: lower index first)
  int pos_r;      // where original row r goes
  {
    const double ar = fabs(__shfl(m, r * 9, 64)), ac = fabs(__shfl(m, c * 9, 64));
    const bool before = (ac > ar) || (ac == ar && c < r);   // element c is picked before element r
    const unsigned long long mb = __ballot(before);
    pos_r = __popc((unsigned int)(mb >> (8 * r)) & 0xFFu);
    const int pos_c = __popc((unsigned int)(mb >> (8 * c)) & 0xFFu);
    const unsigned long long mp = __ballot(pos_c == r);      // bit (r, c): element c stands at position r
    const unsigned int br = (unsigned int)(mp >> (8 * r)) & 0xFFu, bc = (unsigned int)(mp >> (8 * c)) & 0xFFu;
    const int el_r = br ? __builtin_ctz(br) : r, el_c = bc ? __builtin_ctz(bc) : c;   // (a NaN diagonal leaves positions unfilled: identity there, the result is NaN anyway)
    m = __shfl(m, el_r * 8 + el_c, 64);
    dv = __shfl(dv, el_r * 8 + c, 64);
  }
  // ---- unpivoted left-looking LDL^T of the permuted matrix (lower triangle); zero matrix: x = 0 (Eigen: the decomposition stops, solve() returns zeros)
  const double a00 = readLaneD(m, 0);
  if (!(fabs(a00) > 0)) return 0.0;
  double Dc = a00;                                 // D_j of my column j = c, once it is final (j < k)
  {
    const double l = m / a00;
    if (c == 0 && r > 0) m = l;
  }
#pragma unroll 1
  for (int k = 1; k < 8; k++) {
    const double mk = __shfl(m, k * 8 + c, 64);    // L(k, j)
    const double temp = Dc * mk;                   // temp_j = D_j L(k, j)
    const double prod = c < k ? m * temp : 0.0;    // L(r, j) temp_j
    // sum_j prod(r, j) in j order, formed IN lane (r, k): the terms sit k, k-1, ..., 1 lanes to its left, so the shifts 7 ... 1 deliver j = k-7 ... k-1 — ascending j; a shift
    // that leaves the row lands on a lane with c >= k (its product is +0.0) or outside the 16-lane DPP row (0.0)
    double s = 0.0;
#pragma unroll
    for (int q = 7; q >= 1; q--) s = s + dppShrD(prod, q);
    if (c == k && r >= k) m = m - s;
    const double akk = readLaneD(m, k * 9);
    const bool ok = fabs(akk) > 0;
    const double l = ok ? m / akk : m;
    if (c == k) { if (r > k) m = l; Dc = akk; }
  }
  // row r and column r of L are gathered ONCE (16 independent cross-lane reads, pipelined); the substitutions then only need
  // wave-uniform reads of the running solution — no dependent shuffle per step
  double Lrow[8], Lcol[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { Lrow[k] = __shfl(m, r * 8 + k, 64); Lcol[k] = __shfl(m, k * 8 + r, 64); }
  // forward substitution  (L y = P b)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double dkv = readLaneD(dv, k * 8);
    if (r > k) dv = dv - Lrow[k] * dkv;
  }
  {
    const double D = __shfl(m, r * 9, 64);
    dv = (fabs(D) > 2.2250738585072014e-308) ? dv / D : 0.0;
  }
  // backward substitution  (L^T x = z)
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    const double dkv = readLaneD(dv, k * 8);
    if (r < k) dv = dv - Lcol[k] * dkv;
  }
  // x = P^T (...): original row r sits at position pos_r
  return __shfl(dv, pos_r * 8 + c, 64);
}

// One LM control step, executed by all 64 lanes of wave 0.  Consumes the finished evaluation in s_tot, decides,
// and either prepares the next evaluation (s_e, returns true) or finishes the problem (returns false).
__device__ __forceinline__ bool lmWaveStep(LMState& S, const TrackerDev& trk, const LMProblemIn& in, LMProblemOut& out,
                                           const float* s_tot, double* s_H, double* s_b, double* s_x, int* s_trk, EvalP& s_e, const int lane) {
  const int maxIterations[5] = {10, 20, 50, 50, 50};
  const float lambdaExtrapolationLimit = 0.001f;
  int takeH = 0, action = ACT_DONE;
#ifdef DMV_LM_TICKS
  const long long q0 = wall_clock64();
#endif
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
          if (S.cutoffRepeat > 1 && !S.haveRepeated) { out.repeated_lvl = S.lvl; out.first_pass_res = S.lastRes[S.lvl]; S.lvl++; S.haveRepeated = 1; }
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
#ifdef DMV_LM_TICKS
  const long long q1 = wall_clock64();
#endif

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
#ifdef DMV_LM_TICKS
    const long long q2 = wall_clock64();
#endif
    double x = waveLdltSolve8(m, dv, lane, s_trk);
#ifdef DMV_LM_TICKS
    if (lane == 0) { atomicAdd(&g_lm_ticks[0], (double)(wall_clock64() - q2)); atomicAdd(&g_lm_ticks[4], 1.0); }
#endif
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
#ifdef DMV_LM_TICKS
  const long long q3 = wall_clock64();
#endif

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
#ifdef DMV_LM_TICKS
  if (lane == 0 && action != ACT_DONE) { const long long q4 = wall_clock64(); atomicAdd(&g_lm_ticks[1], (double)(q1 - q0)); atomicAdd(&g_lm_ticks[2], (double)(q4 - q3)); atomicAdd(&g_lm_ticks[3], (double)(q4 - q0)); atomicAdd(&g_lm_ticks[5], 1.0); }
#endif
  return action != ACT_DONE;
}

// W = minimum waves per SIMD the register allocator must leave room for (launch-bounds hint): the LM control step
// (fp64) wants more registers than the evaluation loop; W trades its spills against the occupancy of the gathers.
// Cluster mode (few problems in flight): C workgroups share one alignment problem.  Every evaluation is split over the C
// workgroups (point index first = rank*T + thread, stride C*T); the per-workgroup sums are exchanged through global memory behind
// one device-scope arrive counter, every workgroup adds the C partials in rank order and runs the (deterministic) LM control
// step redundantly — one inter-workgroup barrier per evaluation, no second one.  Partials are double-buffered by evaluation
// parity; the launch guarantees B*C <= resident workgroups, so the spin cannot deadlock.
#define LM_LOG_EVALS 96   // evaluations per problem the diagnostic log holds (a track runs ~15; maxIterations sum to 180, the log is cut there)
struct ClusterArgs { int C; float* part; /* B x 2 x C x ACC_PAD */ unsigned int* cnt; /* B, zeroed before the launch */ LMProblemOut* discard; /* device scratch entry */
                     EvalP* log; /* diagnostics (dmvio_hip_tracker_debug_record_replay): B x LM_LOG_EVALS evaluation parameters, or NULL */ int* log_n; /* B */ };

__device__ __forceinline__ void clusterExchange(float* s_tot, const ClusterArgs& cl, const int prob, const int rank, const unsigned int phase) {
  float* __restrict__ mine = cl.part + (((size_t)prob * 2 + (phase & 1u)) * cl.C + rank) * ACC_PAD;
  if (threadIdx.x < ACC_PAD) __hip_atomic_store(mine + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cl.cnt + prob, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int target = (unsigned int)cl.C * (phase + 1u);
    while (__hip_atomic_load(cl.cnt + prob, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  if (threadIdx.x < ACC_PAD) {
    const float* __restrict__ all = cl.part + ((size_t)prob * 2 + (phase & 1u)) * cl.C * ACC_PAD;
    s_tot[threadIdx.x] = sumPartialsInOrder(all + threadIdx.x, cl.C);
  }
  __syncthreads();
}

// TL: the instantiation that can read level-0 planes stored in 8x4 tiles (FrameStore::tiled0, decided per problem at run time); launched only when a batch holds
// such a slot, so the plain instantiation's register budget is untouched
template <int T, int W, bool TL = false>
__global__ void __launch_bounds__(T, W) k_track_lm(const TrackerDev trk, const FrameStore fs, const LMProblemIn* __restrict__ in,
                                                 LMProblemOut* __restrict__ out, const int coarsestLvl, const ClusterArgs cl) {
  __shared__ float s_stage[(T / 64) * SJ_WAVE_FLOATS];
  __shared__ float s_partH[(T / 64) * 256];
  __shared__ float s_partS[T / 64][8];
  __shared__ float s_tot[ACC_PAD];
  __shared__ EvalP s_e;
  __shared__ double s_H[64], s_b[8], s_x[8];
  __shared__ int s_go, s_trk[8];
  __shared__ LMState S;  // written by lane 0 of wave 0 only
  const int prob = blockIdx.x / cl.C, rank = blockIdx.x % cl.C;
  // `in` is pinned host memory: one read of the 120-byte record per workgroup, kept in LDS
  __shared__ LMProblemIn s_in;
  static_assert(sizeof(LMProblemIn) % 4 == 0 && sizeof(LMProblemIn) <= 256, "LMProblemIn is copied as dwords by one wavefront");
  if (threadIdx.x < sizeof(LMProblemIn) / 4) reinterpret_cast<unsigned int*>(&s_in)[threadIdx.x] = reinterpret_cast<const unsigned int*>(in + prob)[threadIdx.x];
  __syncthreads();
  const LMProblemIn& pin = s_in;
  LMProblemOut& pout = rank == 0 ? out[prob] : *cl.discard;   // `out` is pinned host memory (written once, never read); non-leading workgroups write into device scratch
  unsigned int phase = 0;
  if (threadIdx.x == 0) {
    S.cur = poseFrom7(pin.pose7);
    S.affA = pin.aff[0]; S.affB = pin.aff[1];
    for (int i = 0; i < 5; i++) S.lastRes[i] = __builtin_nan("");
    for (int i = 0; i < 3; i++) S.flow[i] = 1000;
    S.lvl = coarsestLvl; S.st = LM_LEVEL_BEGIN; S.totalIts = 0; S.nEvals = 0; S.nPointEvals = 0; S.haveRepeated = 0;
    S.iteration = 0; S.lambda = 0.01f; S.cutoffRepeat = 1; S.incNorm = 0;
    pout.repeated_lvl = -1; pout.first_pass_res = __builtin_nan("");
  }
  if (threadIdx.x < 64) { s_H[threadIdx.x] = 0; if (threadIdx.x < 8) { s_b[threadIdx.x] = 0; s_x[threadIdx.x] = 0; } }
  initStage<T>(s_stage);
  const int slot = pin.new_slot;
  // every pixel of the new frame finite (stamped by its pyramid build): the evaluation loop without the isfinite guards gives the same values
  const bool clean = __builtin_amdgcn_readfirstlane((int)(fs.bad_gen[slot] != fs.build_gen[slot])) != 0;
  const bool tiled0 = TL && __builtin_amdgcn_readfirstlane((int)fs.tiled0[slot]) != 0;
  long long tStep = 0, tEval = 0;
  unsigned int phase_log = 0;
  for (;;) {
    const long long t0 = wall_clock64();
    if (threadIdx.x < 64) {
      // the control step is ONE dependent chain of ~1100 instructions on this wavefront while its three siblings wait: raised issue priority lets it through ahead of the
      // evaluation waves of the other workgroups that share the SIMD (they lose nothing they could not issue a few cycles later)
      __builtin_amdgcn_s_setprio(3);
      const bool go = lmWaveStep(S, trk, pin, pout, s_tot, s_H, s_b, s_x, s_trk, s_e, threadIdx.x);
      __builtin_amdgcn_s_setprio(0);
      if (threadIdx.x == 0) s_go = go ? 1 : 0;
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    tStep += t1 - t0;
    if (cl.log && rank == 0 && threadIdx.x == 0) {   // diagnostics: the schedule of evaluations this problem runs (k_track_replay runs it again without the control steps)
      const int k = (int)phase_log;
      if (s_go && k < LM_LOG_EVALS) cl.log[(size_t)prob * LM_LOG_EVALS + k] = s_e;
      if (!s_go) cl.log_n[prob] = k < LM_LOG_EVALS ? k : LM_LOG_EVALS;
    }
    phase_log++;
    if (!s_go) break;
    const int lvl = s_e.lvl;
    // the plane's address is wave-uniform (level 0 comes out of the pointer table): keep it in scalar registers
    const float* img = dmvUniformGlobal(fs.level(slot, lvl));
    if (TL && tiled0 && lvl == 0) {   // workgroup-uniform
      if (clean)
        blockEval<T, false, TL>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, rank * T + threadIdx.x, cl.C * T, img, trk.huberTH, s_stage, s_partH, s_partS,
                                s_tot);
      else
        blockEval<T, true, TL>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, rank * T + threadIdx.x, cl.C * T, img, trk.huberTH, s_stage, s_partH, s_partS,
                               s_tot);
    } else if (clean)
      blockEval<T, false>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, rank * T + threadIdx.x, cl.C * T, img, trk.huberTH, s_stage, s_partH, s_partS,
                          s_tot);
    else
      blockEval<T, true>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, rank * T + threadIdx.x, cl.C * T, img, trk.huberTH, s_stage, s_partH, s_partS,
                         s_tot);
    if (cl.C > 1) { clusterExchange(s_tot, cl, prob, rank, phase); phase++; }
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
  if (rank == 0) {
    if (threadIdx.x < 64) pout.H[threadIdx.x] = s_H[threadIdx.x];
    if (threadIdx.x < 8) pout.b[threadIdx.x] = s_b[threadIdx.x];
  }
}

// ---- full batches, round 5: the LM control step off the evaluation waves' critical path.  k_track_lm's four waves evaluate, then three of them wait while wave 0 solves
// the damped 8x8 system and steps the pose (13.7 us of every ~50 us round in a full batch; the evaluations alone run 17 % faster: k_track_replay, profiles/r05_tracker_floor.md).
// Here a workgroup is FIVE wavefronts and holds TWO alignment problems: waves 0-3 evaluate problem X = round & 1 while wave 4 runs the control step of the other problem on the
// sums its evaluation left in the previous round — the same blockEval over the same 256-thread grouping and the same lmWaveStep, so every problem's results are bit-identical
// to k_track_lm<256>'s.  A slot that finishes its problem takes the next one from a device-wide counter (the grid is persistent: any number of problems).
__global__ void __launch_bounds__(256, 4) k_track_lm_pp(const TrackerDev trk, const FrameStore fs, const LMProblemIn* __restrict__ in, LMProblemOut* __restrict__ out,
                                                       const int coarsestLvl, const int B, unsigned int* __restrict__ next) {
  constexpr int T = 256;
  __shared__ float s_stage[(T / 64) * SJ_WAVE_FLOATS];
  __shared__ float s_partH[(T / 64) * 256];
  __shared__ float s_partS[T / 64][8];
  __shared__ float s_tot[2][ACC_PAD];
  __shared__ EvalP s_e[2];
  __shared__ double s_H[2][64], s_b[2][8], s_x[2][8];
  __shared__ int s_go[2], s_gon[2], s_trk[2][8], s_prob[2];   // s_gon: the flag the control wave leaves for the NEXT round (s_go is read by every wave at the top of a round)
  __shared__ LMState S[2];
  __shared__ LMProblemIn s_in[2];
  __shared__ long long s_tstep[2], s_teval[2];
  const bool ctrl = threadIdx.x < 64;   // wave 0: the control steps, and a smaller share of every evaluation
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < (T / 64) * SJ_WAVE_FLOATS; k += T) s_stage[k] = 0.0f;
  // (control wave) slot <- the next problem of the batch, its first control step prepares its first evaluation; returns whether the slot has work
  auto fetch = [&](const int slot) -> bool {
    int prob = 0;
    if (lane == 0) prob = (int)atomicAdd(next, 1u);
    prob = __builtin_amdgcn_readfirstlane(prob);
    if (prob >= B) { if (lane == 0) { s_gon[slot] = 0; s_prob[slot] = -1; } return false; }
    if (lane < (int)(sizeof(LMProblemIn) / 4)) reinterpret_cast<unsigned int*>(&s_in[slot])[lane] = reinterpret_cast<const unsigned int*>(in + prob)[lane];
    s_H[slot][lane] = 0; if (lane < 8) { s_b[slot][lane] = 0; s_x[slot][lane] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
      LMState& Q = S[slot];
      Q.cur = poseFrom7(s_in[slot].pose7);
      Q.affA = s_in[slot].aff[0]; Q.affB = s_in[slot].aff[1];
      for (int i = 0; i < 5; i++) Q.lastRes[i] = __builtin_nan("");
      for (int i = 0; i < 3; i++) Q.flow[i] = 1000;
      Q.lvl = coarsestLvl; Q.st = LM_LEVEL_BEGIN; Q.totalIts = 0; Q.nEvals = 0; Q.nPointEvals = 0; Q.haveRepeated = 0;
      Q.iteration = 0; Q.lambda = 0.01f; Q.cutoffRepeat = 1; Q.incNorm = 0;
      out[prob].repeated_lvl = -1; out[prob].first_pass_res = __builtin_nan("");
      s_prob[slot] = prob; s_tstep[slot] = 0; s_teval[slot] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
  };
  // (control wave) one control step of the slot's problem; when the problem ends its results are stored and the slot refilled.  Leaves s_go[slot].
  auto control = [&](const int slot) {
    for (;;) {
      const int prob = __builtin_amdgcn_readfirstlane(s_prob[slot]);
      if (prob < 0) return;
      const long long t0 = wall_clock64();
      const bool go = lmWaveStep(S[slot], trk, s_in[slot], out[prob], s_tot[slot], s_H[slot], s_b[slot], s_x[slot], s_trk[slot], s_e[slot], lane);
      if (lane == 0) { s_tstep[slot] += wall_clock64() - t0; s_gon[slot] = go ? 1 : 0; }
      if (go) return;
      // the problem is finished: its remaining outputs (k_track_lm's tail), then the next problem's first step
      LMProblemOut& po = out[prob];
      if (lane == 0) {
        const LMState& Q = S[slot];
        for (int i = 0; i < 5; i++) po.lastRes[i] = Q.lastRes[i];
        for (int i = 0; i < 3; i++) po.flow[i] = Q.flow[i];
        po.iterations = Q.totalIts; po.n_evals = Q.nEvals; po.n_point_evals = Q.nPointEvals;
        po.ticks_step = s_tstep[slot]; po.ticks_eval = s_teval[slot];
      }
      po.H[lane] = s_H[slot][lane];
      if (lane < 8) po.b[lane] = s_b[slot][lane];
      if (!fetch(slot)) return;
    }
  };
  if (ctrl) {
    for (int slot = 0; slot < 2; slot++) if (fetch(slot)) control(slot);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { s_go[0] = s_gon[0]; s_go[1] = s_gon[1]; }
  }
  __syncthreads();
  for (int round = 0;; round++) {
    const int X = round & 1, Y = X ^ 1;
    const int goX = s_go[X], goY = s_go[Y];
    if (!goX && !goY) break;                      // workgroup-uniform: both slots drained, every evaluation consumed
    const bool stepped = round > 0 && goY;
    if (ctrl && stepped) control(Y);                // goY was set before slot Y's evaluation of the previous round: that evaluation ran, its sums wait in s_tot[Y]
    if (goX) {
      const long long t1 = wall_clock64();
      const int slot = s_in[X].new_slot;
      const int lvl = s_e[X].lvl;
      const bool clean = __builtin_amdgcn_readfirstlane((int)(fs.bad_gen[slot] != fs.build_gen[slot])) != 0;
      const float* img = dmvUniformGlobal(fs.level(slot, lvl));
      // the split of this level's points: of every `period` chunks of 64 wave 0 takes k0 (it also runs a control step of ~C per round), waves 1-3 three each;
      // balanced where k0 / period = 1/4 - 3 C / (16 E): large levels 2 of 11, middle ones 1 of 10, small ones none
      const int npt = trk.pc_n[lvl];
      const int k0 = (stepped || round == 0) ? (npt >= 7000 ? 2 : (npt >= 3500 ? 1 : 0)) : 3;   // no control step beside this evaluation: the even split
      const int period = 9 + k0;
      const int lo = wave == 0 ? 0 : k0 + 3 * (wave - 1), hi = wave == 0 ? k0 : k0 + 3 * wave;
      if (clean) blockEvalChunks<T, false>(s_e[X], trk.g[lvl], trk.pc[lvl], npt, trk.flow_mask, period, lo, hi, img, trk.huberTH, s_stage, s_partH, s_partS, s_tot[X]);
      else blockEvalChunks<T, true>(s_e[X], trk.g[lvl], trk.pc[lvl], npt, trk.flow_mask, period, lo, hi, img, trk.huberTH, s_stage, s_partH, s_partS, s_tot[X]);
      if (threadIdx.x == 64) s_teval[X] += wall_clock64() - t1;
    } else { __syncthreads(); __syncthreads(); }   // the two barriers of the evaluation
    if (ctrl && stepped && lane == 0) s_go[Y] = s_gon[Y];   // every wave has read this round's flags by now (they sit behind the two barriers above)
    __syncthreads();
  }
}

// Diagnostics (profiles/r05_tracker_floor.md): the evaluations of a recorded k_track_lm launch — the same template points, the same taps, the same fused reductions, in the
// same order per problem — WITHOUT the LM control steps between them (the parameters of evaluation k come out of the log instead of out of a solve).  The difference to the
// recorded launch is what the control step costs on the critical path of a full batch; the sums of every evaluation are still formed (and the last one stored), so nothing of
// the evaluation itself is optimised away.
template <int T, int W>
__global__ void __launch_bounds__(T, W) k_track_replay(const TrackerDev trk, const FrameStore fs, const LMProblemIn* __restrict__ in, const EvalP* __restrict__ log,
                                                     const int* __restrict__ log_n, float* __restrict__ sink) {
  __shared__ float s_stage[(T / 64) * SJ_WAVE_FLOATS];
  __shared__ float s_partH[(T / 64) * 256];
  __shared__ float s_partS[T / 64][8];
  __shared__ float s_tot[ACC_PAD];
  __shared__ EvalP s_e;
  const int prob = blockIdx.x;
  const int slot = in[prob].new_slot;
  initStage<T>(s_stage);
  const bool clean = __builtin_amdgcn_readfirstlane((int)(fs.bad_gen[slot] != fs.build_gen[slot])) != 0;
  const int n = log_n[prob];
  for (int k = 0; k < n; k++) {
    if (threadIdx.x < sizeof(EvalP) / 4) reinterpret_cast<unsigned int*>(&s_e)[threadIdx.x] = reinterpret_cast<const unsigned int*>(log + (size_t)prob * LM_LOG_EVALS + k)[threadIdx.x];
    __syncthreads();
    const int lvl = s_e.lvl;
    const float* img = dmvUniformGlobal(fs.level(slot, lvl));
    if (clean) blockEval<T, false>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, threadIdx.x, T, img, trk.huberTH, s_stage, s_partH, s_partS, s_tot);
    else blockEval<T, true>(s_e, trk.g[lvl], trk.pc[lvl], trk.pc_n[lvl], trk.flow_mask, threadIdx.x, T, img, trk.huberTH, s_stage, s_partH, s_partS, s_tot);
    __syncthreads();
  }
  if (threadIdx.x < ACC_PAD) sink[(size_t)prob * ACC_PAD + threadIdx.x] = s_tot[threadIdx.x];
}

}  // namespace dmv
