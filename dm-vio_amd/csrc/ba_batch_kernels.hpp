// CDNA4 (gfx950) kernels of the DEVICE-RESIDENT Gauss-Newton loop of the sliding-window bundle adjustment (round 5): the whole body of FullSystem::optimize's loop
// (src/dso/FullSystem/FullSystemOptimize.cpp:485-586) — solveSystemF (EnergyFunctional.cpp:841-996, the non-GTSAM branch), doStepFromBackup (FullSystemOptimize.cpp:224-317),
// setPrecalcValues (FullSystem.cpp:1670-1680, HessianBlocks.cpp:193-223), calcLEnergy / calcMEnergy (EnergyFunctional.cpp:322-431), the accept test (:553-586) — on the
// device, for W windows per launch:
//   * k_ba_solve: ONE workgroup per window — what csrc/ba_host.hpp does on the host between two kernel chains (assemble HFinal_top / bFinal_top from the stitched system, the
//     priors and the marginalisation prior; Jacobi scaling; the pivoted LDL^T of EnergyFunctional.cpp:971-973 with the arithmetic of BAHost::ldltSolveTransposed element for
//     element; orthogonalisation; the frame / calibration step; SE3 exponentials; the F (F - 1) frame-pair tables; E_L and E_M of the stepped state) — writes the arguments of
//     the window's next linearisation (calibration, pair tables, back-substitution inputs, the accept test's energies) into the window's record in device memory;
//   * k_ba_*_b: the kernels of csrc/ba_kernels.hpp with the window taken from blockIdx.y and their arguments read from that record (same bodies, same arithmetic).
// The loop then is a fixed sequence of launches enqueued up front (solve -> linearise + decide -> [rejected: restore + relinearise] -> [accepted: applyRes + per-point sums ->
// accumulate -> stitch -> gather]), each kernel gated on the window's own decision in device memory: no host round trip, no PCIe poll per iteration, and W windows cost W
// times the work but ONE launch sequence — windows of a batch never wait for each other inside a kernel (no cross-workgroup spinning: a grid that is only partly resident
// cannot deadlock).
#pragma once
#include "ba_kernels.hpp"
#include "lie_dev.h"

namespace dmv {

// ---- per-window record in device memory: the kernel arguments of the single-window path, plus what the solve kernel carries from iteration to iteration
struct BAFrameDev {
  Pose evalPT;
  double state[10], state_zero[10], state_backup[10];
  double prior[8];
  float ab_exposure, pad;
};
struct BASolveDev {
  int F, n, haveM, nBasis;
  int stepped;                 // a step is pending: the previous solve stepped the window and a decision pass has run since
  int iterations_done, n_accepted, exact_backsub;
  double lambda;
  double lastL, lastM, newL, newM;   // E_L / E_M of the state the window stands at / of the pending stepped state
  double c_value[4], c_value_zero[4], c_value_backup[4], cPrior[4];
  float cPriorF[4];
  BAFrameDev fr[BA_MAXF_CAP];
  const double* HM;            // n x n row-major (haveM) — the marginalisation prior
  const double* bM;            // n
  const double* basis;         // nBasis x n: orthonormal basis of the gauge nullspaces above the cut (BAHost::prepareOrthogonalize, host: a function of the evaluation points only)
  const float* adHostF;        // F*F x 64, index h + F*t (BAHost::setAdjointsF)
  const float* adTargetF;
  double* trace;               // 64 x 4: [E_A, E_L, E_M, accepted] per iteration, row 0 = the initial state
  double* x_last;              // n: the last solve's x (= MINUS the step), for tests
  int ticks[16];               // diagnostics: 100 MHz wall-clock stamps of the last k_ba_solve, relative to its start (dmvio_hip_ba_batch_last_solve_ticks)
};
struct BAWinDev {
  BAWindow W, Wb;              // Wb: calibration members of the backed-up state (the relinearisation after a rejected step)
  BAPoints P;
  BARes Rs;
  const BAPrecalc* pre;
  BADecide D;                  // per-window members; mode / update_th / publish / ticket are set per launch
  BAPreDyn T, Tb;              // step-dependent pair tables of the state the next linearisation evaluates / of the backed-up state
  ResubArgs X;
  AccumArgs A;
  StitchBufs SB;
  const double *adHost, *adTarget;
  double* sys;                 // device: [H_A | b_A | H_sc | b_sc | resInA]
  BACtl* ctl;
  int n_lin_blocks, n_pt8_blocks, n_acc_blocks, n_res_blocks, n_gather_blocks, n_stitch_blocks;
  int n_lin1_blocks;           // workgroups of the one-lane-per-residual linearisation (k_ba_linearize_b1): 256 residuals each
  float frameTH[BA_MAXF_CAP];  // FrameHessian::frameEnergyTH of the window's keyframes for the duration of a batch call (BADecide::frameTH points here)
  BASolveDev S;
};

// ------------------------------------------------------------------------------------------------ batched forms of the kernels of ba_kernels.hpp
// window = blockIdx.y; a workgroup beyond its window's own grid leaves at once (grid.x is the largest count of the batch)
enum { BA_LINB_INITIAL = 0, BA_LINB_STEPPED = 1, BA_LINB_RESTORE = 2, BA_LINB_FINAL = 3 };
// (three workgroups per CU instead of two: the batched grid is throughput-bound by resident waves; 170 -> <= 168 registers.  Four — 128 registers, 172 B of scratch per
// lane — was measured slower: 364 vs 250 us for 32 windows; k_ba_accumulate_b: six waves per SIMD, 88 -> 80 registers, 153 -> 88 us for 16 windows; seven spill)
__global__ void __launch_bounds__(LIN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) k_ba_linearize_b(const BAWinDev* __restrict__ wins, const FrameStore fs, const int kind) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_lin_blocks) return;
  BADecide D = V.D;
  D.publish = 0; D.update_th = 1; D.lastE0_from_ctl = 1;
  // INITIAL / FINAL: plain linearisation of the current state (energy + threshold); STEPPED: back-substitution + point step fused in front, accept test behind;
  // RESTORE: only after a rejected step — the points go back to their backup, the backed-up state is relinearised
  if (kind == BA_LINB_STEPPED) { D.mode = 1; baLinearizeBody(V.W, V.P, V.Rs, V.pre, fs, nullptr, nullptr, D, BA_GATE_ALWAYS, 0, V.T.v, 1, V.X.xc, V.X.xAd, 1, V.n_lin_blocks); }
  else if (kind == BA_LINB_RESTORE) { D.mode = 2; baLinearizeBody(V.Wb, V.P, V.Rs, V.pre, fs, nullptr, nullptr, D, BA_GATE_REJECTED, 1, V.Tb.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin_blocks); }
  else { D.mode = 0; baLinearizeBody(V.W, V.P, V.Rs, V.pre, fs, nullptr, nullptr, D, BA_GATE_ALWAYS, 0, V.T.v, kind == BA_LINB_INITIAL ? 1 : 0, V.X.xc, V.X.xAd, 0, V.n_lin_blocks); }
}
// the one-lane-per-residual form (baLinearizeBody1): what a grid that fills the device wants — an eighth of the lanes, no redundant geometry; same values, same energy partials
__global__ void __launch_bounds__(LIN_THREADS) k_ba_linearize_b1(const BAWinDev* __restrict__ wins, const FrameStore fs, const int kind) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_lin1_blocks) return;
  BADecide D = V.D;
  D.publish = 0; D.update_th = 1; D.lastE0_from_ctl = 1;
  if (kind == BA_LINB_STEPPED) { D.mode = 1; baLinearizeBody1(V.W, V.P, V.Rs, V.pre, fs, D, BA_GATE_ALWAYS, 0, V.T.v, 1, V.X.xc, V.X.xAd, 1, V.n_lin1_blocks, V.n_lin_blocks); }
  else if (kind == BA_LINB_RESTORE) { D.mode = 2; baLinearizeBody1(V.Wb, V.P, V.Rs, V.pre, fs, D, BA_GATE_REJECTED, 1, V.Tb.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin1_blocks, V.n_lin_blocks); }
  else { D.mode = 0; baLinearizeBody1(V.W, V.P, V.Rs, V.pre, fs, D, BA_GATE_ALWAYS, 0, V.T.v, kind == BA_LINB_INITIAL ? 1 : 0, V.X.xc, V.X.xAd, 0, V.n_lin1_blocks, V.n_lin_blocks); }
}
__global__ void __launch_bounds__(256) k_ba_reset_oob_b(const BAWinDev* __restrict__ wins) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_res_blocks) return;
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= V.W.R) return;
  const bool gone = V.Rs.removed[ri] != 0;   // PointFrameResidual::resetOOB of every residual still in the graph (k_ba_reset_oob)
  V.Rs.state[ri] = gone ? BA_OOB : BA_IN;
  V.Rs.newState[ri] = gone ? BA_OOB : BA_OUTLIER;
  V.Rs.energy[ri] = 0.f; V.Rs.newEnergy[ri] = 0.f;
}
__global__ void __launch_bounds__(256) k_ba_apply_b(const BAWinDev* __restrict__ wins, const int mark_removed, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_res_blocks || baGateClosed(V.ctl, gate)) return;
  baApplyBody(V.W.R, V.Rs, nullptr, mark_removed);
}
__global__ void __launch_bounds__(256) k_ba_point_sums_b(const BAWinDev* __restrict__ wins, const int backup, const int apply, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_pt8_blocks) return;
  baPointSumsBody(V.W, V.P, V.Rs, backup, apply, V.ctl, gate, nullptr);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) k_ba_accumulate_b(const BAWinDev* __restrict__ wins, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_acc_blocks) return;
  baAccumulateBody(V.A, V.Rs, V.P, V.ctl, gate);
}
// every window of a batch has the same F (the host groups them): blockDim = 64 F
__global__ void __launch_bounds__(64 * BA_MAXF_CAP) k_ba_stitch_b(const BAWinDev* __restrict__ wins, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  baStitchBody(V.A.F, V.A.nsTop, V.A.nsD, V.A.accTop, V.A.numTop, V.A.accD, V.A.numD, V.A.accE, V.adHost, V.adTarget, V.SB, V.ctl, gate);
}
template <int MF>
__global__ void __launch_bounds__(256) k_ba_stitch_gather_b(const BAWinDev* __restrict__ wins, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_gather_blocks || baGateClosed(V.ctl, gate)) return;
  const int F = V.A.F;
  gatherElement<MF>(F, V.A.nsC, V.A.accC, V.SB, V.A.numTop, F * F * V.A.nsTop, V.sys, blockIdx.x * blockDim.x + threadIdx.x, false);
}

// ------------------------------------------------------------------------------------------------ k_ba_solve
// lower triangle packed by rows: element (r, c), c <= r
__device__ __forceinline__ int triIdx(const int r, const int c) { return (r * (r + 1)) / 2 + c; }
#define BA_SOLVE_THREADS 256
template <int MF> struct BASolveDims {
  static constexpr int NMAX = 4 + 8 * MF;
  static constexpr int QMAX = (NMAX * (NMAX + 1) / 2 + BA_SOLVE_THREADS - 1) / BA_SOLVE_THREADS;
};
// dynamic LDS of k_ba_solve, in doubles: HM (n x n), the packed matrix, the nullspace basis (7 x n), 11 vectors, the frames' states
__host__ __device__ inline int baSolveHmStride(const int n) { return n | 1; }   // odd row stride (in doubles): thread i walking row i meets no LDS bank conflicts
__host__ __device__ inline size_t baSolveLdsDoubles(const int n, const int F) { return (size_t)n * baSolveHmStride(n) + (size_t)(n * (n + 1)) / 2 + 7 * (size_t)n + 11 * (size_t)n + 40 * (size_t)F + 64; }

// AffLight::fromToVecExposure as BAHost::affFromToHost evaluates it (exp in double, the exposure ratio in float-to-double promotion order)
__device__ __forceinline__ void baAffFromTo(float eF, float eT, const double aF, const double bF, const double aT, const double bT, double out[2]) {
  if (eF == 0 || eT == 0) { eT = eF = 1; }
  const double a = dexp(aT - aF) * eT / eF;
  out[0] = a; out[1] = bT - a * bF;
}

// s + a[0] b[0] + a[1] b[1] + ... in index order; the loads of eight terms go out together (a dependent chain over LDS costs an LDS latency per term otherwise)
__device__ __forceinline__ double baSeqDot(double s, const double* __restrict__ a, const double* __restrict__ b, const int n) {
  int j = 0;
  for (; j + 8 <= n; j += 8) {
    double x[8], y[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { x[q] = a[j + q]; y[q] = b[j + q]; }
#pragma unroll
    for (int q = 0; q < 8; q++) s += x[q] * y[q];
  }
  for (; j < n; j++) s += a[j] * b[j];
  return s;
}

// finish != 0: only settle the pending decision (behind the last iteration's chain); the host reads the final state from the window's record.
// Everything a sequential (order-preserving) loop reads is staged in LDS first: a dependent chain over global memory costs a memory latency per term.
template <int MF>
__global__ void __launch_bounds__(BA_SOLVE_THREADS) k_ba_solve(BAWinDev* __restrict__ wins, const int iteration, const int finish) {
  constexpr int QMAX = BASolveDims<MF>::QMAX;
  BAWinDev& V = wins[blockIdx.x];
  BASolveDev& S = V.S;
  extern __shared__ double s_mem[];
  const int tid = threadIdx.x, n = S.n, F = S.F;
  const int NP = (n * (n + 1)) / 2;
  const int hs = baSolveHmStride(n);
  double* const HMs = s_mem;          // n rows of stride hs: the marginalisation prior
  double* const M = HMs + (size_t)n * hs;   // NP: scaled, permuted matrix -> L (strict lower) and D (diagonal)
  double* const basis = M + NP;       // 7 x n
  double* const d = basis + 7 * n;    // n: stacked delta (calib | frames)
  double* const bP = d + n;           // n: bM + HM delta
  double* const HLd = bP + n;         // n
  double* const sv = HLd + n;         // n
  double* const rhs = sv + n;         // n: scaled right-hand side, permuted in place
  double* const xs = rhs + n;         // n
  double* const dg = xs + n;          // n: diagonal copy for the pivot search
  double* const tv = dg + n;          // n: calcMEnergy rows / projections
  double* const col = tv + n;         // n: the current column of L
  double* const bMs = col + n;        // n
  int* const perm = reinterpret_cast<int*>(bMs + n);   // n ints (<= n doubles reserved)
  double* const fst = bMs + 2 * n;    // F x 10 state | F x 10 state_zero | F x 10 state_backup | F x 8 prior (+ pad): 40 F
  double* const fzero = fst + 10 * F;
  double* const fbak = fzero + 10 * F;
  double* const fprior = fbak + 10 * F;
  __shared__ int s_flag[4];
  __shared__ double s_scal[8];
  __shared__ double s_cal[16];        // c_value, c_value_zero, c_value_backup, cPrior
  __shared__ Pose s_w2c[BA_MAXF_CAP], s_c2w[BA_MAXF_CAP];
  __shared__ double s_scaled[BA_MAXF_CAP][10];
  __shared__ float s_K[9], s_Ki[9];
  const long long t_begin = wall_clock64();
#define SOLVE_TICK(i) do { if (tid == 0) S.ticks[i] = (int)(wall_clock64() - t_begin); } while (0)

  // ---- stage the window's solve state (coalesced) while thread 0 settles the pending decision (FullSystemOptimize.cpp:556-583 behind the accept test)
  const int haveM = S.haveM;
  if (!finish) {
    if (haveM) {
      for (int i = tid; i < n * n; i += BA_SOLVE_THREADS) HMs[(i / n) * hs + (i % n)] = S.HM[i];
      for (int i = tid; i < n; i += BA_SOLVE_THREADS) bMs[i] = S.bM[i];
    }
    if (iteration >= 2) for (int i = tid; i < S.nBasis * n; i += BA_SOLVE_THREADS) basis[i] = S.basis[i];
  }
  for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) { const BAFrameDev& f = S.fr[i / 10]; fst[i] = f.state[i % 10]; fzero[i] = f.state_zero[i % 10]; fbak[i] = f.state_backup[i % 10]; }
  for (int i = tid; i < 8 * F; i += BA_SOLVE_THREADS) fprior[i] = S.fr[i >> 3].prior[i & 7];
  if (tid < 4) { s_cal[tid] = S.c_value[tid]; s_cal[4 + tid] = S.c_value_zero[tid]; s_cal[8 + tid] = S.c_value_backup[tid]; s_cal[12 + tid] = S.cPrior[tid]; }
  if (tid == 0) {
    int acc = 1;
    if (S.stepped) {
      acc = __hip_atomic_load(&V.ctl->accept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double E0 = __hip_atomic_load(&V.ctl->lastE0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (acc) { S.lastL = S.newL; S.lastM = S.newM; S.lambda = fmax(S.lambda * 0.25, 1e-5); S.n_accepted++; }
      else S.lambda *= 1e2;
      S.iterations_done++;
      const int row = S.iterations_done;
      if (row < 64) { S.trace[4 * row] = E0; S.trace[4 * row + 1] = S.lastL; S.trace[4 * row + 2] = S.lastM; S.trace[4 * row + 3] = acc ? 1.0 : 0.0; }
      S.stepped = 0;
    } else if (iteration == 0 && !finish) {
      // row 0 of the trace: the initial state (its photometric energy is what the initial linearisation's decision pass left in the control block)
      S.trace[0] = __hip_atomic_load(&V.ctl->lastE0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); S.trace[1] = S.lastL; S.trace[2] = S.lastM; S.trace[3] = 1.0;
    }
    s_flag[0] = acc; s_flag[1] = 0; s_flag[2] = 0; s_flag[3] = 0;
    s_scal[4] = S.lambda; s_scal[5] = S.lastL; s_scal[6] = S.lastM;
  }
  __syncthreads();
  SOLVE_TICK(0);   // staged + settled
  const int prevAccepted = s_flag[0];
  if (!prevAccepted) {
    // loadSateBackup, frame / calibration part: the state and its pair tables go back to the backup's
    for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) { fst[i] = fbak[i]; S.fr[i / 10].state[i % 10] = fbak[i]; }
    if (tid < 4) { s_cal[tid] = s_cal[8 + tid]; S.c_value[tid] = s_cal[8 + tid]; }
    const int np = F * (F - 1) * 14;
    float* Tcur = &V.T.v[0][0];
    const float* Tbk = &V.Tb.v[0][0];
    for (int i = tid; i < np; i += BA_SOLVE_THREADS) Tcur[i] = Tbk[i];
    if (tid == 0) { V.W.fx = V.Wb.fx; V.W.fy = V.Wb.fy; V.W.cx = V.Wb.cx; V.W.cy = V.Wb.cy; V.W.fxi = V.Wb.fxi; V.W.fyi = V.Wb.fyi; V.W.cxi = V.Wb.cxi; V.W.cyi = V.Wb.cyi; }
  }
  __syncthreads();
  if (finish) return;

  // ---- backupState (frames, calibration) + the stacked delta (EnergyFunctional::setDeltaF as BAHost::setPrecalcValues keeps it)
  for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) { fbak[i] = fst[i]; S.fr[i / 10].state_backup[i % 10] = fst[i]; }
  if (tid < 4) { s_cal[8 + tid] = s_cal[tid]; S.c_value_backup[tid] = s_cal[tid]; d[tid] = (double)(float)(s_cal[tid] - s_cal[4 + tid]); HLd[tid] = s_cal[12 + tid]; }
  for (int i = tid; i < 8 * F; i += BA_SOLVE_THREADS) {
    const int f = i >> 3, k = i & 7;
    d[4 + i] = fst[10 * f + k] - fzero[10 * f + k];
    HLd[4 + i] = fprior[i];
  }
  if (tid == 0) {   // the backed-up state's tables for a relinearisation after a rejected step
    V.Wb.fx = V.W.fx; V.Wb.fy = V.W.fy; V.Wb.cx = V.W.cx; V.Wb.cy = V.W.cy; V.Wb.fxi = V.W.fxi; V.Wb.fyi = V.W.fyi; V.Wb.cxi = V.W.cxi; V.Wb.cyi = V.W.cyi;
  }
  {
    const int np = F * (F - 1) * 14;
    const float* Tcur = &V.T.v[0][0];
    float* Tbk = &V.Tb.v[0][0];
    for (int i = tid; i < np; i += BA_SOLVE_THREADS) Tbk[i] = Tcur[i];
  }
  __syncthreads();
  // bM_top = bM + HM * delta (EnergyFunctional.cpp:864), row sums in index order
  for (int i = tid; i < n; i += BA_SOLVE_THREADS) {
    double s = haveM ? bMs[i] : 0.0;
    if (haveM) s = baSeqDot(s, HMs + (size_t)i * hs, d, n);
    bP[i] = s;
  }
  SOLVE_TICK(1);   // backup, delta, bM_top
  // ---- HFinal_top - H_sc / (1 + lambda), lower triangle (BAHost::solveSystem: (HL + HM) + HA, the diagonal times (1 + lambda), minus H_sc * fac)
  const double lambda = s_scal[4];
  const double fac = 1.0f / (1 + lambda);
  const double* __restrict__ HA = V.sys;
  const double* __restrict__ bA = V.sys + (size_t)n * n;
  const double* __restrict__ Hsc = bA + n;
  const double* __restrict__ bsc = Hsc + (size_t)n * n;
  for (int i = tid; i < n; i += BA_SOLVE_THREADS) {   // diagonal first: the Jacobi scaling needs it
    const size_t o = (size_t)i * n + i;
    double v = (HLd[i] + (haveM ? HMs[(size_t)i * hs + i] : 0.0)) + HA[o];
    v *= (1 + lambda);
    v = v - Hsc[o] * fac;
    sv[i] = 1.0 / sqrt(v + 10);
    dg[i] = v;
  }
  // the packed pairs this thread owns (tid, tid + 256, ...): (row << 8) | column
  int pr[QMAX];
  double acc[QMAX];                   // sum_{j < k} L(r, j) D(j) L(c, j) of the owned pairs, j ascending (ldltSolveTransposed's acc[r], one per column)
  double val[QMAX];
#pragma unroll
  for (int q = 0; q < QMAX; q++) {
    const int p = tid + q * BA_SOLVE_THREADS;
    pr[q] = -1; acc[q] = 0.0; val[q] = 0.0;
    if (p < NP) {
      int r = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
      while (triIdx(r + 1, 0) <= p) r++;
      while (triIdx(r, 0) > p) r--;
      pr[q] = (r << 8) | (p - triIdx(r, 0));
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < QMAX; q++) {
    if (pr[q] >= 0) {
      const int i = pr[q] >> 8, j = pr[q] & 255;
      double v;
      if (i == j) v = dg[i];
      else {
        const size_t o = (size_t)i * n + j;
        v = ((0.0 + (haveM ? HMs[(size_t)i * hs + j] : 0.0)) + HA[o]) - Hsc[o] * fac;
      }
      M[tid + q * BA_SOLVE_THREADS] = sv[i] * v * sv[j];   // (sv_i * H_ij) * sv_j
    }
  }
  for (int i = tid; i < n; i += BA_SOLVE_THREADS) {
    const double bL = i < 4 ? s_cal[12 + i] * d[i] : fprior[i - 4] * fst[10 * ((i - 4) >> 3) + ((i - 4) & 7)];   // prior * delta_prior (= state)
    const double bF = ((bL + bP[i]) + bA[i]) - bsc[i];
    rhs[i] = sv[i] * bF;
  }
  __syncthreads();
  SOLVE_TICK(2);   // system assembled and scaled
  // ---- pivot order: Eigen's LDLT (and BAHost::ldltSolveTransposed) picks the largest |diagonal| of the NOT YET UPDATED trailing diagonal (left-looking: step k only
  // touches column k), first one on ties, and swaps it to position k — the whole sequence follows from the diagonal alone.  Without ties it is the descending order of
  // |diagonal| (a rank count); with ties (or NaN) the selection-with-swaps is replayed step by step by wavefront 0.
  // Group id of an element = the number of elements with a larger |diagonal| (equal for tied ones, ascending with descending value); the selection then walks the groups in
  // order and inside a group always takes the member at the lowest CURRENT position >= k — two ballots per step on wavefront 0 (positions lane, lane + 64; n <= 128).
  if (tid < n) dg[tid] = M[triIdx(tid, tid)];
  __syncthreads();
  if (tid < n) {
    const double mine = fabs(dg[tid]);
    int rank = 0, same = 0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
      double o[8];
#pragma unroll
      for (int q = 0; q < 8; q++) o[q] = fabs(dg[j + q]);
#pragma unroll
      for (int q = 0; q < 8; q++) { rank += o[q] > mine ? 1 : 0; same += o[q] == mine ? 1 : 0; }
    }
    for (; j < n; j++) { const double o = fabs(dg[j]); rank += o > mine ? 1 : 0; same += o == mine ? 1 : 0; }
    if (!(mine == mine)) s_flag[2] = 1;   // NaN on the diagonal: the step-by-step replay below
    if (same > 1) s_flag[3] = 1;          // a tie: the selection order inside the group depends on the swaps before it
    perm[tid] = (rank << 8) | same;       // group id, group size
  }
  __syncthreads();
  const int needReplay = s_flag[2], haveTies = s_flag[3];
  if (!needReplay && !haveTies) {
    // all |diagonal| values distinct: step k selects the k-th largest whatever the swaps did to the others — the order is the rank itself
    int rk = -1;
    if (tid < n) rk = perm[tid] >> 8;
    __syncthreads();
    if (tid < n) perm[rk] = tid;
  } else if (tid < 64) {
    if (!needReplay) {
      int g0 = tid < n ? perm[tid] : 0x7fffff00, g1 = tid + 64 < n ? perm[tid + 64] : 0x7fffff00;   // (group << 8) | size of the element at position lane / lane + 64
      int p0 = tid, p1 = tid + 64;                                                                    // its original index
      __builtin_amdgcn_wave_barrier();
      int gcur = 0, left = 0, lastsize = 0;
      for (int k = 0; k < n; k++) {
        if (left == 0) gcur += lastsize;
        const unsigned long long m0 = __ballot((g0 >> 8) == gcur && tid >= k), m1 = __ballot((g1 >> 8) == gcur && tid + 64 >= k);
        const int big = m0 ? __builtin_ctzll(m0) : (m1 ? 64 + __builtin_ctzll(m1) : k);
        const int gb = big < 64 ? __builtin_amdgcn_readlane(g0, big) : __builtin_amdgcn_readlane(g1, big - 64);
        const int pb = big < 64 ? __builtin_amdgcn_readlane(p0, big) : __builtin_amdgcn_readlane(p1, big - 64);
        const int gk = k < 64 ? __builtin_amdgcn_readlane(g0, k) : __builtin_amdgcn_readlane(g1, k - 64);
        const int pk = k < 64 ? __builtin_amdgcn_readlane(p0, k) : __builtin_amdgcn_readlane(p1, k - 64);
        if (left == 0) { left = gb & 255; lastsize = left; }
        left--;
        if (big != k) {
          if (tid == (big & 63)) { if (big < 64) { g0 = gk; p0 = pk; } else { g1 = gk; p1 = pk; } }
          if (tid == (k & 63)) { if (k < 64) { g0 = gb; p0 = pb; } else { g1 = gb; p1 = pb; } }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (tid < n) perm[tid] = p0;
      if (tid + 64 < n) perm[tid + 64] = p1;
    } else {
      for (int i = tid; i < n; i += 64) perm[i] = i;
      __builtin_amdgcn_wave_barrier();
      for (int k = 0; k < n; k++) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int i = k + tid; i < n; i += 64) { const double v = fabs(dg[i]); if (v > best) { best = v; bi = i; } }   // ascending i per lane: strict > keeps the first
        for (int off = 32; off > 0; off >>= 1) {
          const double ov = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) bi = k;   // all NaN: no swap (fabs(NaN) > x is false in the host loop too)
        if (tid == 0 && bi != k) { const double t = dg[k]; dg[k] = dg[bi]; dg[bi] = t; const int q = perm[k]; perm[k] = perm[bi]; perm[bi] = q; }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  SOLVE_TICK(3);   // pivot order
  // ---- the permuted system: P A P^T in place of A (all swaps applied up front: the same operands meet in the same order as with swaps at every step); every thread
  // keeps its pairs' values in registers
  int rq[QMAX], cq[QMAX];
#pragma unroll
  for (int q = 0; q < QMAX; q++) {
    rq[q] = 0; cq[q] = -1;
    if (pr[q] >= 0) {
      rq[q] = pr[q] >> 8; cq[q] = pr[q] & 255;
      const int a = perm[rq[q]], b = perm[cq[q]];
      val[q] = M[a >= b ? triIdx(a, b) : triIdx(b, a)];
    }
  }
  {
    double rv = 0.0;
    if (tid < n) rv = rhs[perm[tid]];
    __syncthreads();
    if (tid < n) rhs[tid] = rv;
  }
  __syncthreads();
  SOLVE_TICK(4);   // permuted
  // ---- LDL^T, column by column; the forward substitution rides along (d[i] -= L(i, j) d[j], j ascending).  Per step: [owners publish column k = A - acc] barrier
  // [rows divide, forward-substitute, publish L(., k)] barrier [every pair (r, c > k) adds L(r, k) (D_k L(c, k)) to its sum].
  for (int k = 0; k < n; k++) {
#pragma unroll
    for (int q = 0; q < QMAX; q++) if (cq[q] == k) col[rq[q]] = k > 0 ? val[q] - acc[q] : val[q];
    __syncthreads();
    const double akk = col[k];
    const bool ok = fabs(akk) > 0;
    if (k == 0 && !ok) { if (tid == 0) s_flag[1] = 1; break; }
    if (tid >= k && tid < n) {
      double l = col[tid];
      if (tid > k) {
        if (ok) l /= akk;
        rhs[tid] -= l * rhs[k];
      }
      M[triIdx(tid, k)] = l;
      tv[tid] = tid > k ? l : 0.0;    // L(., k) for the trailing update (col[] is rewritten by the next step's column phase)
    }
    __syncthreads();
    // trailing update, branch-free: the loads of all owned pairs go out together; a pair outside the trailing block adds +0.0 (its sum is never -0.0: it starts at +0.0)
    double lr[QMAX], lc[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; q++) { lr[q] = tv[rq[q]]; lc[q] = tv[cq[q] > 0 ? cq[q] : 0]; }
#pragma unroll
    for (int q = 0; q < QMAX; q++) { const double pq = lr[q] * (akk * lc[q]); acc[q] += cq[q] > k ? pq : 0.0; }
    // (no barrier here: the next column phase writes col[], which the update above does not read; tv[] is rewritten only behind the next step's first barrier)
  }
  __syncthreads();
  SOLVE_TICK(5);   // factorised + forward substitution
  const bool zero = s_flag[1] != 0;
  // ---- diagonal solve, back substitution
  if (tid < n) {
    const double dd = M[triIdx(tid, tid)];
    double v = rhs[tid];
    if (fabs(dd) > 2.2250738585072014e-308) v /= dd; else v = 0;
    rhs[tid] = zero ? 0.0 : v;
  }
  __syncthreads();
  if (!zero) {
    if (S.exact_backsub) {
      // the order of ldltSolveTransposed (row i subtracts L(j, i) x_j for j = i + 1 .. n - 1, ascending): one dependent chain of n^2 / 2 subtractions
      if (tid == 0) for (int i = n - 1; i >= 0; i--) { double s = rhs[i]; for (int j = i + 1; j < n; j++) s -= M[triIdx(j, i)] * rhs[j]; rhs[i] = s; }
    } else if (tid < 64) {
      // column-oriented, one wavefront (no workgroup barrier per step): as soon as x_i stands, every row above subtracts its term (row r subtracts in the order
      // i = n - 1 .. r + 1: the same terms, another association)
      double x0 = tid < n ? rhs[tid] : 0.0, x1 = tid + 64 < n ? rhs[tid + 64] : 0.0;
      double m0n = M[triIdx(n - 1, min(tid, n - 2))], m1n = M[triIdx(n - 1, min(tid + 64, n - 2))];   // row i of L for the next step, fetched one step ahead
      for (int i = n - 1; i > 0; i--) {
        const double m0 = m0n, m1 = m1n;
        if (i > 1) { m0n = M[triIdx(i - 1, min(tid, i - 2))]; m1n = M[triIdx(i - 1, min(tid + 64, i - 2))]; }
        const double src = i >= 64 ? x1 : x0;
        const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(src) & 0xffffffffll), i & 63), hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(src) >> 32), i & 63);
        const double xi = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        if (tid < i) x0 -= m0 * xi;
        if (tid + 64 < i) x1 -= m1 * xi;
      }
      if (tid < n) rhs[tid] = x0;
      if (tid + 64 < n) rhs[tid + 64] = x1;
    }
  }
  __syncthreads();
  SOLVE_TICK(6);   // back substitution
  // undo the permutation and the scaling: x = S P^T x'
  if (tid < n) xs[perm[tid]] = rhs[tid];
  __syncthreads();
  if (tid < n) xs[tid] = sv[tid] * xs[tid];
  __syncthreads();
  // ---- orthogonalisation against the gauge nullspaces from iteration 2 on (SOLVER_ORTHOGONALIZE_X_LATER, EnergyFunctional.cpp:977-981; BAHost::orthogonalize)
  if (iteration >= 2 && S.nBasis > 0) {
    const int nB = S.nBasis;
    if (tid < nB) { double dot = 0; const double* u = basis + (size_t)tid * n; for (int k = 0; k < n; k++) dot += u[k] * xs[k]; tv[tid] = dot; }
    __syncthreads();
    if (tid < n) { double proj = 0; for (int b = 0; b < nB; b++) proj += basis[(size_t)b * n + tid] * tv[b]; xs[tid] -= proj; }
    __syncthreads();
  }
  if (tid < n) S.x_last[tid] = xs[tid];
  SOLVE_TICK(7);   // x
  // ---- resubstituteF_MT's inputs (BAHost::prepareResubstitute): xc, xAd[h F + t][c] = x_h . adHostF(:, c) + x_t . adTargetF(:, c) in fp32, sequential over r
  if (tid < 4) V.X.xc[tid] = (float)xs[tid];
  for (int o = tid; o < F * F * 8; o += BA_SOLVE_THREADS) {
    const int c = o & 7, t = (o >> 3) % F, hh = (o >> 3) / F;
    const size_t base = ((size_t)hh + (size_t)F * t) * 64;
    float ah[8], at[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { ah[r] = S.adHostF[base + r * 8 + c]; at[r] = S.adTargetF[base + r * 8 + c]; }   // sixteen loads in flight, then the ordered sums
    float s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) { s1 += (float)xs[4 + 8 * hh + r] * ah[r]; s2 += (float)xs[4 + 8 * t + r] * at[r]; }
    V.X.xAd[((size_t)F * hh + t) * 8 + c] = s1 + s2;
  }
  // ---- doStepFromBackup, frames and calibration (stepfac 1): value = backup + step, step = -x
  if (tid < 4) {
    const double stp = -xs[tid];
    const double nv = s_cal[8 + tid] + 1.0f * stp;
    S.c_value[tid] = nv;
    const double scaled = 50.0f * nv;
    s_scal[tid] = scaled;
    d[tid] = (double)(float)(nv - s_cal[4 + tid]);   // cDeltaF of the stepped state
  }
  for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) {
    const int f = i / 10, k = i % 10;
    const double stp = k < 8 ? -xs[4 + 8 * f + k] : 0.0;
    const double st = fbak[i] + (double)1.0f * stp;
    S.fr[f].state[k] = st;
    fst[i] = st;
    const float sc = (k < 6) ? 1.0f : ((k & 1) ? 1000.0f : 10.0f);   // SCALE_XI_*, SCALE_A (6, 8), SCALE_B (7, 9)
    s_scaled[f][k] = sc * st;
    if (k < 8) d[4 + 8 * f + k] = st - fzero[i];
  }
  __syncthreads();
  SOLVE_TICK(8);   // resubstitution inputs, stepped states
  if (tid == 0) {
    // CalibHessian::setValue (BAHost::calibSetValue) and the K / K^-1 of setPrecalcValues, in float like the host
    const float f0 = (float)s_scal[0], f1 = (float)s_scal[1], f2 = (float)s_scal[2], f3 = (float)s_scal[3];
    V.W.fx = f0; V.W.fy = f1; V.W.cx = f2; V.W.cy = f3;
    V.W.fxi = 1.0f / f0; V.W.fyi = 1.0f / f1; V.W.cxi = -f2 / f0; V.W.cyi = -f3 / f1;
    const float K[9] = {f0, 0, f2, 0, f1, f3, 0, 0, 1};
    const float a = K[0], e = K[4], c = K[2], ff = K[5];
    const float det = a * (e * 1.0f - ff * 0.0f), invdet = 1.0f / det;
    const float Ki[9] = {(e * 1.0f - ff * 0.0f) * invdet, (c * 0.0f - 0.0f * 1.0f) * invdet, (0.0f * ff - c * e) * invdet,
                         (ff * 0.0f - 0.0f * 1.0f) * invdet, (a * 1.0f - c * 0.0f) * invdet, (c * 0.0f - a * ff) * invdet,
                         (0.0f * 0.0f - e * 0.0f) * invdet, (0.0f * 0.0f - a * 0.0f) * invdet, (a * e - 0.0f * 0.0f) * invdet};
    for (int i = 0; i < 9; i++) { s_K[i] = K[i]; s_Ki[i] = Ki[i]; }
  }
  if (tid >= 64 && tid < 64 + F) {   // FrameHessian::setState: PRE_worldToCam = exp(state_scaled) * worldToCam_evalPT (HessianBlocks.h:199-214)
    const int f = tid - 64;
    const Pose w = poseMul(poseExp(s_scaled[f]), S.fr[f].evalPT);
    s_w2c[f] = w; s_c2w[f] = poseInv(w);
  }
  // E_M of the stepped state, rows in index order (independent of the poses: runs beside the exponentials)
  if (tid >= 128 && tid < 128 + n && tid < BA_SOLVE_THREADS) {
    const int i = tid - 128;
    double t = 0;
    if (haveM) t = baSeqDot(2 * bMs[i], HMs + (size_t)i * hs, d, n);
    tv[i] = t;
  }
  __syncthreads();
  SOLVE_TICK(9);   // exponentials, E_M rows
  // ---- FrameFramePrecalc::set for the F (F - 1) ordered pairs (HessianBlocks.cpp:193-223; BAHost::setPrecalcValues): the step-dependent members
  for (int o = tid; o < F * F; o += BA_SOLVE_THREADS) {
    const int hh = o % F, t = o / F;
    if (hh == t) continue;
    const Pose l = poseMul(s_w2c[t], s_c2w[hh]);
    double Rd[9];
    quatToR(l.q, Rd);
    float Rf[9], tf[3], KR[9], KRKi[9];
    for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
    for (int i = 0; i < 3; i++) tf[i] = (float)l.t[i];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) KR[r * 3 + c] = s_K[r * 3 + 0] * Rf[c] + s_K[r * 3 + 1] * Rf[3 + c] + s_K[r * 3 + 2] * Rf[6 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) KRKi[r * 3 + c] = KR[r * 3 + 0] * s_Ki[c] + KR[r * 3 + 1] * s_Ki[3 + c] + KR[r * 3 + 2] * s_Ki[6 + c];
    float* v = V.T.v[baPairIndex(hh, t, F)];
    for (int i = 0; i < 9; i++) v[i] = KRKi[i];
    for (int r = 0; r < 3; r++) v[9 + r] = s_K[r * 3 + 0] * tf[0] + s_K[r * 3 + 1] * tf[1] + s_K[r * 3 + 2] * tf[2];
    double aff[2];
    baAffFromTo(S.fr[hh].ab_exposure, S.fr[t].ab_exposure, s_scaled[hh][6], s_scaled[hh][7], s_scaled[t][6], s_scaled[t][7], aff);
    v[12] = (float)aff[0]; v[13] = (float)aff[1];
  }
  if (tid == BA_SOLVE_THREADS - 1) {
    // E_M = delta . (2 bM + HM delta) (EnergyFunctional.cpp:332; BAHost::calcMEnergy)
    double s = 0;
    if (haveM) for (int i = 0; i < n; i++) s += d[i] * tv[i];
    // E_L, frame / calibration part (EnergyFunctional.cpp:349-369; BAHost::calcLEnergyFrames): delta_prior = state
    double E = 0;
    for (int f = 0; f < F; f++) for (int i = 0; i < 8; i++) E += fst[10 * f + i] * fprior[8 * f + i] * fst[10 * f + i];
    float ec = 0;
    for (int i = 0; i < 4; i++) { const float cd = (float)d[i]; ec += cd * S.cPriorF[i] * cd; }
    const double newL = E + ec;
    S.newL = newL; S.newM = s;
    V.D.lastL = s_scal[5]; V.D.lastM = s_scal[6]; V.D.newL = newL; V.D.newM = s;
    S.stepped = 1;
    S.ticks[11] = (int)(wall_clock64() - t_begin);   // energies
  }
  SOLVE_TICK(10);  // pair tables (thread 0's share)
#undef SOLVE_TICK
}

}  // namespace dmv
