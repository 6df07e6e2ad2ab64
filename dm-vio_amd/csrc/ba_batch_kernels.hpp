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
  int pivot_branch, pad_;      // diagnostics: how the last solve found its pivot order (BA_PIVOT_*; dmvio_hip_ba_batch_last_pivot_branch)
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
  // residuals kept linearised outside a marginalisation (dmvio_hip_ba_fix_linearization; n_lin == 0: none, the pointers are NULL): what accumulateLF_MT / addPoint<1> and
  // calcLEnergyPt read (csrc/ba_kernels.hpp "residuals kept linearised"), the three-pass accumulation's views, and EnergyFunctional::setDeltaF's adHTdeltaF / cDeltaF of
  // the state the window stands at ([0]) and of its backup ([1]) — k_ba_solve keeps them like the pair tables
  int n_lin, n_lin_runs;
  unsigned int lin_cnt, pad_lin;
  const float* fullJ; const unsigned char* lin; const float* rtz;
  float* linRec; unsigned char* linActive; unsigned char* topActive;
  float* linE;                 // R x 8: every residual's terms of the linearised energy, in the reference's lane order
  double* sysL;                // [H_L | b_L] of the last accumulation's L pass
  float adHTdelta[2][BA_MAXF_CAP * BA_MAXF_CAP * 8];
  float cDeltaF[2][4];
  BASolveDev S;
};

// ---- the record's pointers as GLOBAL pointers.  A pointer that arrives as a kernel argument is known to point into global memory; one that a kernel loads from a
// structure in memory is a generic ("flat") pointer to the compiler: every access through it becomes a flat_load / flat_store, which counts on BOTH wait counters — an
// `s_waitcnt lgkmcnt(0)` in front of an LDS read then also waits for every outstanding global load, and the software pipelines of the accumulation / linearisation bodies
// (global loads issued tiles ahead of the LDS traffic that consumes them) collapse to load -> wait -> use.  gl() reads the pointer through an lvalue whose type carries the
// global address space; the address-space inference pass then turns every access derived from it into global_load / global_store.  The bl*() functions below build the
// local argument structures of a body from a window's record this way (they live in registers: no member is read with a dynamic index, see baRec()).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wincompatible-pointer-types-discards-qualifiers"
template <class T> __device__ __forceinline__ T* gl(T* const& member) {
  return (T*)(*reinterpret_cast<__attribute__((address_space(1))) T* const*>(&member));
}
#pragma clang diagnostic pop
__device__ __forceinline__ BAPoints blPoints(const BAPoints& g) {
  BAPoints v;
  v.host = gl(g.host); v.u = gl(g.u); v.v = gl(g.v);
  v.idepth = gl(g.idepth); v.idepth_zero = gl(g.idepth_zero); v.idepth_backup = gl(g.idepth_backup); v.step = gl(g.step);
  v.color = gl(g.color); v.weights = gl(g.weights); v.priorF = gl(g.priorF); v.res_begin = gl(g.res_begin);
  v.Hdd = gl(g.Hdd); v.bd = gl(g.bd); v.Hcd = gl(g.Hcd); v.HdiF = gl(g.HdiF); v.bdSumF = gl(g.bdSumF); v.idepth_hessian = gl(g.idepth_hessian);
  v.lHdd = gl(g.lHdd); v.lbd = gl(g.lbd); v.lHcd = gl(g.lHcd); v.HcdAF = gl(g.HcdAF);
  return v;
}
__device__ __forceinline__ BARes blRes(const BARes& g) {
  BARes v;
  v.point = gl(g.point); v.target = gl(g.target);
  v.state = gl(g.state); v.newState = gl(g.newState); v.active = gl(g.active); v.which = gl(g.which); v.removed = gl(g.removed);
  v.energy = gl(g.energy); v.newEnergy = gl(g.newEnergy); v.newEnergyWO = gl(g.newEnergyWO); v.center = gl(g.center);
  v.newestSlot = gl(g.newestSlot); v.newestE = gl(g.newestE);
  v.rec[0] = gl(g.rec[0]); v.rec[1] = gl(g.rec[1]);
  v.lin = gl(g.lin);
  return v;
}
__device__ __forceinline__ AccumArgs blAccum(const AccumArgs& g) {
  AccumArgs v;
  v.F = g.F; v.N = g.N; v.nsTop = g.nsTop; v.nsD = g.nsD; v.nsC = g.nsC;
  v.top_begin = gl(g.top_begin); v.top_members = gl(g.top_members); v.scd_begin = gl(g.scd_begin); v.scd_members = gl(g.scd_members);
  v.accTop = gl(g.accTop); v.accD = gl(g.accD); v.accE = gl(g.accE); v.accC = gl(g.accC); v.numTop = gl(g.numTop); v.numD = gl(g.numD);
  v.ticks = nullptr;
  return v;
}
__device__ __forceinline__ StitchBufs blStitch(const StitchBufs& g) {
  StitchBufs v;
  v.topHH = gl(g.topHH); v.topTT = gl(g.topTT); v.topHT = gl(g.topHT); v.topHC = gl(g.topHC); v.topTC = gl(g.topTC); v.topBH = gl(g.topBH); v.topBT = gl(g.topBT); v.topCC = gl(g.topCC);
  v.scHH = gl(g.scHH); v.scTH = gl(g.scTH); v.scTT = gl(g.scTT); v.scHT = gl(g.scHT); v.scHC = gl(g.scHC); v.scTC = gl(g.scTC); v.scBH = gl(g.scBH); v.scBT = gl(g.scBT);
  return v;
}
__device__ __forceinline__ BADecide blDecide(const BADecide& g) {
  BADecide v = g;
  v.newestE = gl(g.newestE); v.frameTH = gl(g.frameTH); v.epart = gl(g.epart); v.ctl = gl(g.ctl); v.host = gl(g.host);
  v.xchg_local = gl(g.xchg_local); v.xchg_all = gl(g.xchg_all);
  return v;
}

// ------------------------------------------------------------------------------------------------ batched forms of the kernels of ba_kernels.hpp
// window = blockIdx.y; a workgroup beyond its window's own grid leaves at once (grid.x is the largest count of the batch)
enum { BA_LINB_INITIAL = 0, BA_LINB_STEPPED = 1, BA_LINB_RESTORE = 2, BA_LINB_FINAL = 3, BA_LINB_STEPPED_DONE = 4 };   // _DONE: the step was taken by k_ba_resubstitute_b
// (three workgroups per CU instead of two: the batched grid is throughput-bound by resident waves; 170 -> <= 168 registers.  Four — 128 registers, 172 B of scratch per
// lane — was measured slower: 364 vs 250 us for 32 windows; k_ba_accumulate_b: six waves per SIMD, 88 -> 80 registers, 153 -> 88 us for 16 windows; seven spill)
__global__ void __launch_bounds__(LIN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) k_ba_linearize_b(const BAWinDev* __restrict__ wins, const FrameStore fs, const int kind) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_lin_blocks) return;
  BADecide D = blDecide(V.D);
  D.publish = 0; D.update_th = 1; D.lastE0_from_ctl = 1;
  const BAPoints P = blPoints(V.P); const BARes Rs = blRes(V.Rs); const BAPrecalc* const pre = gl(V.pre);
  // INITIAL / FINAL: plain linearisation of the current state (energy + threshold); STEPPED: back-substitution + point step fused in front, accept test behind;
  // RESTORE: only after a rejected step — the points go back to their backup, the backed-up state is relinearised
  if (kind == BA_LINB_STEPPED) { D.mode = 1; baLinearizeBody(V.W, P, Rs, pre, fs, nullptr, nullptr, D, BA_GATE_ALWAYS, 0, V.T.v, 1, V.X.xc, V.X.xAd, 1, V.n_lin_blocks); }
  else if (kind == BA_LINB_RESTORE) { D.mode = 2; baLinearizeBody(V.Wb, P, Rs, pre, fs, nullptr, nullptr, D, BA_GATE_REJECTED, 1, V.Tb.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin_blocks); }
  else { D.mode = 0; baLinearizeBody(V.W, P, Rs, pre, fs, nullptr, nullptr, D, BA_GATE_ALWAYS, 0, V.T.v, kind == BA_LINB_INITIAL ? 1 : 0, V.X.xc, V.X.xAd, 0, V.n_lin_blocks); }
}
// the one-lane-per-residual form (baLinearizeBody1): what a grid that fills the device wants — an eighth of the lanes, no redundant geometry; same values, same energy partials
__global__ void __launch_bounds__(LIN_THREADS) k_ba_linearize_b1(const BAWinDev* __restrict__ wins, const FrameStore fs, const int kind) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_lin1_blocks) return;
  BADecide D = blDecide(V.D);
  D.publish = 0; D.update_th = 1; D.lastE0_from_ctl = 1;
  const BAPoints P = blPoints(V.P); const BARes Rs = blRes(V.Rs); const BAPrecalc* const pre = gl(V.pre);
  extern __shared__ float s_patch[];   // LIN_THREADS x BA_PATCH_STRIDE floats: every lane's 8x8 window of its target image (then the decision pass's key staging)
  if (kind == BA_LINB_STEPPED) { D.mode = 1; baLinearizeBody1(V.W, P, Rs, pre, fs, D, BA_GATE_ALWAYS, 0, V.T.v, 1, V.X.xc, V.X.xAd, 1, V.n_lin1_blocks, V.n_lin_blocks, s_patch); }
  else if (kind == BA_LINB_STEPPED_DONE) { D.mode = 1; baLinearizeBody1(V.W, P, Rs, pre, fs, D, BA_GATE_ALWAYS, 0, V.T.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin1_blocks, V.n_lin_blocks, s_patch); }
  else if (kind == BA_LINB_RESTORE) { D.mode = 2; baLinearizeBody1(V.Wb, P, Rs, pre, fs, D, BA_GATE_REJECTED, 1, V.Tb.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin1_blocks, V.n_lin_blocks, s_patch); }
  else { D.mode = 0; baLinearizeBody1(V.W, P, Rs, pre, fs, D, BA_GATE_ALWAYS, 0, V.T.v, kind == BA_LINB_INITIAL ? 1 : 0, V.X.xc, V.X.xAd, 0, V.n_lin1_blocks, V.n_lin_blocks, s_patch); }
}
// What follows a stepped linearisation's decision, in ONE launch (a gated-off launch still costs ~4.5 us of dispatch): a window whose step was REJECTED restores its points
// and relinearises the backed-up state (BA_LINB_RESTORE above); a window whose step was ACCEPTED runs applyRes + the per-point sums of the new state (k_ba_point_sums_b;
// what = 0) or, behind the last iteration, applyRes alone (k_ba_apply_b; what = 1).  The decision does not change while this kernel runs (the restore's decision pass
// writes the energy only).  LIN1: the one-lane-per-residual form of the relinearisation.
template <bool LIN1>
__global__ void __launch_bounds__(LIN_THREADS) k_ba_post_decide_b(const BAWinDev* __restrict__ wins, const FrameStore fs, const int what) {
  const BAWinDev& V = wins[blockIdx.y];
  const BACtl* const ctl = gl(V.ctl);
  const BAPoints P = blPoints(V.P); const BARes Rs = blRes(V.Rs);
  if (!baGateClosed(ctl, BA_GATE_ACCEPTED)) {
    if (what == 0) { if ((int)blockIdx.x < V.n_pt8_blocks) baPointSumsBody(V.W, P, Rs, 1, 1, ctl, BA_GATE_ALWAYS, nullptr); }
    else if ((int)blockIdx.x < V.n_res_blocks) baApplyBody(V.W.R, Rs, nullptr, 0);
    return;
  }
  BADecide D = blDecide(V.D);
  D.publish = 0; D.update_th = 1; D.lastE0_from_ctl = 1; D.mode = 2;
  const BAPrecalc* const pre = gl(V.pre);
  extern __shared__ float s_patch[];
  if (LIN1) { if ((int)blockIdx.x < V.n_lin1_blocks) baLinearizeBody1(V.Wb, P, Rs, pre, fs, D, BA_GATE_ALWAYS, 1, V.Tb.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin1_blocks, V.n_lin_blocks, s_patch); }
  else if ((int)blockIdx.x < V.n_lin_blocks) baLinearizeBody(V.Wb, P, Rs, pre, fs, nullptr, nullptr, D, BA_GATE_ALWAYS, 1, V.Tb.v, 1, V.X.xc, V.X.xAd, 0, V.n_lin_blocks);
}
// resubstituteF_MT + the points' share of doStepFromBackup for every window of a launch (k_ba_resubstitute: eight lanes per point, the loads of a point's residuals side by
// side) — in front of the one-lane linearisation, whose own form of it walks a point's residuals one dependent load pair after the other
__global__ void __launch_bounds__(256) k_ba_resubstitute_b(const BAWinDev* __restrict__ wins) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_pt8_blocks) return;
  baResubstituteBody(V.W, blPoints(V.P), blRes(V.Rs), V.X.xc, V.X.xAd, 1);
}
__global__ void __launch_bounds__(256) k_ba_reset_oob_b(const BAWinDev* __restrict__ wins) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_res_blocks) return;
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= V.W.R) return;
  const BARes Rs = blRes(V.Rs);
  const bool gone = Rs.removed[ri] != 0;   // PointFrameResidual::resetOOB of every residual still in the graph (k_ba_reset_oob)
  Rs.state[ri] = gone ? BA_OOB : BA_IN;
  Rs.newState[ri] = gone ? BA_OOB : BA_OUTLIER;
  Rs.energy[ri] = 0.f; Rs.newEnergy[ri] = 0.f;
}
__global__ void __launch_bounds__(256) k_ba_apply_b(const BAWinDev* __restrict__ wins, const int mark_removed, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_res_blocks || baGateClosed(gl(V.ctl), gate)) return;
  baApplyBody(V.W.R, blRes(V.Rs), nullptr, mark_removed);
}
__global__ void __launch_bounds__(256) k_ba_point_sums_b(const BAWinDev* __restrict__ wins, const int backup, const int apply, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_pt8_blocks) return;
  baPointSumsBody(V.W, blPoints(V.P), blRes(V.Rs), backup, apply, gl(V.ctl), gate, nullptr);
}
// pass: 0 = the one accumulation of a graph without residuals kept linearised; 1 / 2 / 3 = the L / A / Schur pass of the three-pass accumulation (accumulateLin in
// capi_ba.hip: addPoint<1> over the linearised residuals' records, addPoint<0> over the others, the Schur side over every active one).  A window WITHOUT such residuals in a
// launch of pass 1 or 3 has nothing to do; in pass 2 it runs its one ordinary accumulation.
enum { BA_PASS_ALL = 0, BA_PASS_L = 1, BA_PASS_A = 2, BA_PASS_S = 3 };
__device__ __forceinline__ bool baPassIdle(const BAWinDev& V, const int pass) { return V.n_lin == 0 && (pass == BA_PASS_L || pass == BA_PASS_S); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) k_ba_accumulate_b(const BAWinDev* __restrict__ wins, const int gate, const int pass) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_acc_blocks || baPassIdle(V, pass)) return;
  const AccumArgs A = blAccum(V.A); const BAPoints P = blPoints(V.P); const BACtl* const ctl = gl(V.ctl);
  BARes Rv = blRes(V.Rs);
  if (V.n_lin == 0 || pass == BA_PASS_S || pass == BA_PASS_ALL) { baAccumulateBody(A, Rv, P, ctl, gate); return; }
  if (pass == BA_PASS_L) { Rv.rec[0] = Rv.rec[1] = gl(V.linRec); Rv.active = gl(V.linActive); Rv.lin = nullptr; }
  else Rv.active = gl(V.topActive);
  baAccumulateBody(A, Rv, P, ctl, gate);
}
// every window of a batch has the same F (the host groups them): blockDim = 64 F
__global__ void __launch_bounds__(64 * BA_MAXF_CAP) k_ba_stitch_b(const BAWinDev* __restrict__ wins, const int gate, const int pass) {
  const BAWinDev& V = wins[blockIdx.y];
  if (baPassIdle(V, pass)) return;
  const AccumArgs A = blAccum(V.A);
  baStitchBody(A.F, A.nsTop, A.nsD, A.accTop, A.numTop, A.accD, A.numD, A.accE, gl(V.adHost), gl(V.adTarget), blStitch(V.SB), gl(V.ctl), gate);
}
template <int MF>
__global__ void __launch_bounds__(256) k_ba_stitch_gather_b(const BAWinDev* __restrict__ wins, const int gate, const int pass) {
  const BAWinDev& V = wins[blockIdx.y];
  if ((int)blockIdx.x >= V.n_gather_blocks || baGateClosed(gl(V.ctl), gate) || baPassIdle(V, pass)) return;
  const int F = V.A.F;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const AccumArgs A = blAccum(V.A); const StitchBufs SB = blStitch(V.SB);
  if (V.n_lin == 0 || pass == BA_PASS_ALL) gatherElement<MF>(F, A.nsC, A.accC, SB, A.numTop, F * F * A.nsTop, gl(V.sys), t, false, 3, true);
  else if (pass == BA_PASS_L) gatherElement<MF>(F, A.nsC, A.accC, SB, A.numTop, F * F * A.nsTop, gl(V.sysL), t, false, 1, false);
  else gatherElement<MF>(F, A.nsC, A.accC, SB, A.numTop, F * F * A.nsTop, gl(V.sys), t, false, pass == BA_PASS_A ? 1 : 2, true);
}
// the records addPoint<1> consumes and the linearised residuals' per-point sums, at the deltas of the state the window stands at (k_ba_lin_records / k_ba_lin_point_sums)
__global__ void __launch_bounds__(256) k_ba_lin_records_b(const BAWinDev* __restrict__ wins, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if (V.n_lin == 0 || (int)blockIdx.x >= V.n_res_blocks || baGateClosed(gl(V.ctl), gate)) return;
  baLinRecordsBody(V.W, blPoints(V.P), blRes(V.Rs), gl(V.fullJ), gl(V.lin), gl(V.rtz), V.adHTdelta[0], make_float4(V.cDeltaF[0][0], V.cDeltaF[0][1], V.cDeltaF[0][2], V.cDeltaF[0][3]),
                   gl(V.linRec), gl(V.linActive), gl(V.topActive));
}
__global__ void __launch_bounds__(256) k_ba_lin_point_sums_b(const BAWinDev* __restrict__ wins, const int gate) {
  const BAWinDev& V = wins[blockIdx.y];
  if (V.n_lin == 0 || (int)blockIdx.x * 256 >= V.W.N || baGateClosed(gl(V.ctl), gate)) return;
  const BAPoints P = blPoints(V.P);
  baLinPointSumsBody(V.W, P, gl(V.linRec), gl(V.linActive), const_cast<float*>(P.lHdd), const_cast<float*>(P.lbd), const_cast<float*>(P.lHcd));
}
// calcLEnergyPt's term of the residuals kept linearised (EnergyFunctional.cpp:349-409), at the deltas of the STEPPED state k_ba_solve just left in the record: every residual's
// eight products (2 res_toZeroF + J delta) . (J delta) in parallel, then — last workgroup of the window — the reference's summation: an Accumulator11 (four fp32 lanes) per run of
// 50 points, two 4-lane updates per residual in point / residual order, the runs' totals added in double (capi_ba.hip: linEnergy).  The sum goes on top of the frame / calibration
// prior energy the solve stored (BADecide::newL, BASolveDev::newL): the stepped linearisation's accept test, next in the stream, reads it there.
__global__ void __launch_bounds__(256) k_ba_lin_energy_b(BAWinDev* __restrict__ wins) {
  BAWinDev& V = wins[blockIdx.y];
  if (V.n_lin == 0 || (int)blockIdx.x >= V.n_res_blocks) return;
  const int ri = blockIdx.x * 256 + threadIdx.x;
  const float* __restrict__ adHT = V.adHTdelta[0];
  float* const linE = gl(V.linE); const unsigned char* const lin = gl(V.lin); const unsigned char* const active = gl(V.Rs.active);
  const int* const res_begin = gl(V.P.res_begin);
  if (ri < V.W.R) {
    float* __restrict__ e = linE + (size_t)ri * 8;
    if (lin[ri] && active[ri]) {
      const int pi = gl(V.Rs.point)[ri];
      const float* __restrict__ J = gl(V.fullJ) + (size_t)ri * 74;
      const float* __restrict__ rtz = gl(V.rtz) + (size_t)ri * 8;
      const float* __restrict__ dp = adHT + (size_t)(gl(V.P.host)[pi] + V.W.F * gl(V.Rs.target)[ri]) * 8;
      const float dd = 0.0f;
      float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
      for (int i = 0; i < 6; i++) { sx += J[8 + i] * dp[i]; sy += J[14 + i] * dp[i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) { cx += J[20 + i] * V.cDeltaF[0][i]; cy += J[24 + i] * V.cDeltaF[0][i]; }
      const float Jp_delta_x = sx + cx + J[28] * dd, Jp_delta_y = sy + cy + J[29] * dd;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        float Jdelta = J[30 + q] * Jp_delta_x;
        Jdelta = Jdelta + J[38 + q] * Jp_delta_y;
        Jdelta = Jdelta + J[46 + q] * dp[6];
        Jdelta = Jdelta + J[54 + q] * dp[7];
        float r0 = rtz[q];
        r0 = r0 + r0;
        r0 = r0 + Jdelta;
        e[q] = Jdelta * r0;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) e[q] = 0.0f;   // (skipped below: the flag is tested again — a +0.0f would not change a sum that is never -0.0f either)
    }
  }
  __shared__ int s_last;
  __shared__ double s_run[256];
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&V.lin_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(V.n_res_blocks - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) __hip_atomic_store(&V.lin_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // runs of 50 points: thread = run (the four lanes of its Accumulator11 in registers); more than 256 runs: strided, the partial totals added in run order below
  const int nruns = (V.W.N + 49) / 50;
  double A = 0.0;
  for (int base = 0; base < nruns; base += 256) {
    const int run = base + threadIdx.x;
    double tot = 0.0;
    if (run < nruns) {
      float d1[4] = {0, 0, 0, 0};
      const int p1 = min(V.W.N, 50 * run + 50);
      const int r0 = res_begin[50 * run], r1 = res_begin[p1];
      for (int rj = r0; rj < r1; rj++) {
        if (!lin[rj] || !active[rj]) continue;
        const float* __restrict__ e = linE + (size_t)rj * 8;
#pragma unroll
        for (int k = 0; k < 4; k++) d1[k] = d1[k] + __hip_atomic_load(e + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 4; k++) d1[k] = d1[k] + __hip_atomic_load(e + 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tot = (double)(d1[0] + d1[1] + d1[2] + d1[3]);
    }
    s_run[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) for (int k = 0; k < min(256, nruns - base); k++) A += s_run[k];
    __syncthreads();
  }
  if (threadIdx.x == 0) { V.S.newL += A; V.D.newL += A; }
}

// ------------------------------------------------------------------------------------------------ k_ba_solve
// Row-packed lower triangle (the scaled, not yet permuted matrix): element (r, c), c <= r
__device__ __forceinline__ int triIdx(const int r, const int c) { return (r * (r + 1)) / 2 + c; }
// Column-packed lower triangle (the permuted matrix, then L and D in place): element (r, k), r >= k, at baColBase(k, n) + r — a column is contiguous, which is what the
// factorisation publishes and reads per step
__host__ __device__ inline int baColBase(const int k, const int n) { return k * (n - 1) - (k * (k - 1)) / 2; }
#define BA_SOLVE_THREADS 512
template <int MF> struct BASolveDims {
  static constexpr int NMAX = 4 + 8 * MF;
  static constexpr int QMAX = (NMAX * (NMAX + 1) / 2 + BA_SOLVE_THREADS - 1) / BA_SOLVE_THREADS;   // packed pairs per thread
  static constexpr int NCW = 2 * ((NMAX + 7) / 8); // columns per wavefront of the factorisation (the column PAIR c >> 1 belongs to wave (c >> 1) mod 4; wavefronts 4 .. 7 hold none)
  static constexpr bool ALIAS_HM = MF > 8;         // n = 100: the prior's LDS copy shares the room of the packed matrix and its D L copy (staged twice: bM_top, E_M)
};
// dynamic LDS of k_ba_solve, in doubles: HM (n x n; aliased: see ALIAS_HM), the packed matrix and its D L copy (+ 128 each: a wave reads a column with all its lanes), the
// nullspace basis (7 x n), 12 vectors, the frames' states
__host__ __device__ inline int baSolveHmStride(const int n) { return n | 1; }   // odd row stride (in doubles): thread i walking row i meets no LDS bank conflicts
__host__ __device__ inline size_t baSolveLdsDoubles(const int n, const int F, const bool aliasHM) {
  const size_t NP = (size_t)(n * (n + 1)) / 2, hm = (size_t)n * baSolveHmStride(n), lt = 2 * (NP + 128);
  return (aliasHM ? (hm > lt ? hm : lt) : hm + lt) + 7 * (size_t)n + 12 * (size_t)n + 40 * (size_t)F + 64;
}
// the debug kernel's share (no prior, no basis, no frame states)
__host__ __device__ inline size_t baSolveCoreLdsDoubles(const int n) { return 2 * ((size_t)(n * (n + 1)) / 2 + 128) + 7 * (size_t)n + 64; }

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
#define BA_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__device__ __forceinline__ double baReadLaneF64(const double v, const int l) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// the packed pairs a thread owns (tid, tid + 512, ...): (row << 8) | column of the row-packed triangle, -1 beyond it
template <int QMAX>
__device__ __forceinline__ void baOwnedPairs(const int n, int (&pr)[QMAX]) {
  const int NP = (n * (n + 1)) / 2;
#pragma unroll
  for (int q = 0; q < QMAX; q++) {
    const int p = (int)threadIdx.x + q * BA_SOLVE_THREADS;
    pr[q] = -1;
    if (p < NP) {
      int r = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
      while (triIdx(r + 1, 0) <= p) r++;
      while (triIdx(r, 0) > p) r--;
      pr[q] = (r << 8) | (p - triIdx(r, 0));
    }
  }
}

// ---- the pivoted LDL^T solve of EnergyFunctional.cpp:971-973 (Eigen's ldlt().solve() as BAHost::ldltSolveTransposed restates it), one 512-thread workgroup, every
// operation of the host's loop in the host's order per element:
//   in : Lc[triIdx(i, j)] = the scaled matrix (row-packed lower triangle), dgS[i] = its diagonal, rhsS[i] = the scaled right-hand side
//   out: xs[i] = the solution of the SCALED system in the ORIGINAL order (the caller multiplies by the scaling), perm[k] = the index Eigen's transpositions bring to position k,
//        s_flag[1] = the matrix was zero (x = 0), s_flag[2] = how the pivot order was found (BA_PIVOT_*)
// Pivot order: Eigen's unblocked LDL^T picks the largest |diagonal| of the NOT YET UPDATED trailing diagonal (left-looking: step k only touches column k), the first one
// on ties, and swaps it to position k — the whole sequence follows from the original diagonal alone.  Without ties it is the descending order of |diagonal| (a rank count);
// with ties the members of a tie group are taken in the order of their CURRENT positions when the group's first step comes (a member only moves when it is taken; the
// displaced occupant of position k goes where the taken element stood) — wavefront 0 reconstructs those positions by chasing the displacement links of the earlier steps
// (a few wave-wide rounds per tie group, see below) instead of replaying the n swaps one after the other; a NaN on the diagonal takes the literal loop on one lane.
// Factorisation: wave w of row group G (rows 64 G + lane) owns the columns c = w (mod 4) of its rows and keeps their running sums sum_j L(r, j) D_j L(c, j) (j ascending:
// the host's acc[r]) in registers, the next own column always in acc[0].  Step k: the owner of column k adds step k - 1's term to that column only, subtracts, takes the
// pivot from lane k, divides, forward-substitutes and PUBLISHES the column (L and D L) in LDS — then ONE workgroup barrier — and every other wave adds the published
// steps' terms to all its columns (the owner catches up in the next step: its trailing update is off the pivot chain).  Rows >= 64 (windows of more than 7 keyframes) form
// row group 1, which runs one step behind (its column k needs the pivot row group 0 published in step k) and carries the pivot chain itself from column 64 on.
#ifdef BA_SOLVE_PROBE
__device__ long long g_ba_probe[64];
#define BA_PROBE(base, i) do { if (k == BA_SOLVE_PROBE && lane == 0) g_ba_probe[(base) + (i)] = clock64(); } while (0)
#else
#define BA_PROBE(base, i) do { } while (0)
#endif
// One wavefront's share of the factorisation (see baLdltSolveCore): run<NL>(kg0, kg1) takes the groups of four slots (eight columns) kg0 .. kg1 - 1 with at most NL own
// columns left.  TWO: the matrix has more than 64 rows — a second register set holds rows 64 + lane.
template <int MF, bool TWO>
struct BAFactor {
  static constexpr int NCW = BASolveDims<MF>::NCW;
  int n, lane, w, cbase, done, cbK, cbJ;                     // cbase: the column of acc[0]; done: steps <= done are in every own column's sum; cbK / cbJ: baColBase of the
  double *__restrict__ Lc, *__restrict__ Tc, *__restrict__ rhs;   // slot's first column / of step done + 1, kept by additions
  int* s_flag;
  // element idx (a row index, uniform) of a column held one row per lane in (v, v2)
  __device__ __forceinline__ double pick(const double v, const double v2, const int idx) const {
    if (!TWO || idx < 64) return baReadLaneF64(v, idx & 63);
    return baReadLaneF64(v2, idx & 63);
  }
  // column k of the permuted matrix minus its finished sum -> pivot, L(., k) (the division), D_k L(., k); published.  Returns through l / l2 / t / t2.
  __device__ __forceinline__ void finish(const int k, const int cb, double a, double a2, double& l, double& l2, double& t, double& t2) {
    const int nx = n - 64;                                   // rows of the second set (TWO)
    const double akk = pick(a, a2, k);
    const bool ok = fabs(akk) > 0;
    if (k == 0 && !ok && lane == 0) s_flag[1] = 1;           // a zero matrix: x = 0 (everybody leaves behind the first group of slots)
    l = a; l2 = a2;
    if (ok) {
      if (TWO && k >= nx) {
        // rows 64 .. n - 1 ride in lanes 0 .. nx - 1 of the first set's division (rows below nx <= k are finished there)
        const double q = (lane < nx ? a2 : a) / akk;
        if (lane < nx) { if (64 + lane > k) l2 = q; } else if (lane > k) l = q;
      } else {
        const double q = a / akk;
        if (lane > k) l = q;
        if (TWO) { const double q2 = a2 / akk; if (64 + lane > k) l2 = q2; }
      }
    }
    t = akk * l; t2 = TWO ? akk * l2 : 0.0;
    if (lane >= k && lane < n) { Lc[cb + lane] = l; Tc[cb + lane] = t; }   // the diagonal slot of L keeps D_k
    if (TWO && lane < nx && 64 + lane >= k) { Lc[cb + 64 + lane] = l2; Tc[cb + 64 + lane] = t2; }
  }
  template <int NL>
  __device__ __forceinline__ void run(const int kg0, const int kg1, double (&acc)[NCW], double (&acc2)[NCW]) {
    for (int kg = kg0; kg < kg1; kg++) {
#pragma unroll 1
      for (int sub = 0; sub < 4; sub++) {
        const int k = 8 * kg + 2 * sub;                      // the slot's columns: k, k + 1
        const bool own = sub == w && k < n;
        if (own) {
          // owner of the column pair (k, k + 1) = (cbase, cbase + 1), in acc[0], acc[1]: their sums lack only the previous slot's two steps (added here for these two
          // columns alone: the other own columns catch up in the next slot), and column k + 1 also step k, which never leaves the wavefront
          BA_PROBE(0, 0);
          const int cb = cbK, cbn = cbK + n - 1 - k;         // baColBase(k), baColBase(k + 1)
          const bool has2 = k + 1 < n;
          double a = Lc[cb + lane], a2 = TWO ? Lc[cb + 64 + lane] : 0.0;
          double b = Lc[cbn + lane], b2 = TWO ? Lc[cbn + 64 + lane] : 0.0;
          double sa = acc[0], sa2 = acc2[0], sb = acc[1], sb2 = acc2[1];
          if (k > 0) {
            const int c1 = cbK - (n - k), c2 = c1 - (n - k + 1);   // baColBase(k - 1), baColBase(k - 2)
            const double lp = Lc[c2 + lane], lp2 = TWO ? Lc[c2 + 64 + lane] : 0.0;
            const double lq = Lc[c1 + lane], lq2 = TWO ? Lc[c1 + 64 + lane] : 0.0;
            const double tpa = Tc[c2 + k], tpb = Tc[c2 + k + 1], tqa = Tc[c1 + k], tqb = Tc[c1 + k + 1];   // D_j L(k, j), D_j L(k + 1, j) for j = k - 2, k - 1
            sa += lp * tpa; sa += lq * tqa;
            sb += lp * tpb; sb += lq * tqb;
            if (TWO) { sa2 += lp2 * tpa; sa2 += lq2 * tqa; sb2 += lp2 * tpb; sb2 += lq2 * tqb; }
          }
          a -= sa;
          if (TWO) a2 -= sa2;
          BA_PROBE(0, 1);
          double l, l2, t, t2;
          finish(k, cb, a, a2, l, l2, t, t2);
          BA_PROBE(0, 2);
          if (has2) {
            const double tk = pick(t, t2, k + 1);            // D_k L(k + 1, k)
            sb += l * tk;
            if (TWO) sb2 += l2 * tk;
            b -= sb;
            if (TWO) b2 -= sb2;
            double m, m2, u, u2;
            finish(k + 1, cbn, b, b2, m, m2, u, u2);
          }
          BA_PROBE(0, 4);
        }
        // every published step not yet added (the previous slot's two; four right after an own slot; none IN the own slot), to all own columns: L(., j) one row per lane,
        // D_j L(c, j) of the own columns as broadcast reads (the columns of acc[0], acc[1] once their own slot has passed, and columns beyond the matrix, collect values
        // nobody reads)
        const int jmax = (k < n && !own) ? k - 1 : done;
        BA_PROBE(16 + 16 * (w == ((sub + 1) & 3) ? 1 : 0), 0);
#pragma unroll 1
        for (int j = done + 1; j <= jmax; j++) {
          const double lr0 = Lc[cbJ + lane];
          const double lr0b = TWO ? Lc[cbJ + 64 + lane] : 0.0;
          const double* __restrict__ T0 = Tc + cbJ + cbase;
          double t[NL];
#pragma unroll
          for (int i = 0; i < NL; i++) t[i] = T0[8 * (i >> 1) + (i & 1)];
#pragma unroll
          for (int i = 0; i < NL; i++) {
            acc[i] += lr0 * t[i];
            if (TWO) acc2[i] += lr0b * t[i];
          }
          cbJ += n - 1 - j;
        }
        if (jmax > done) done = jmax;
        if (!own) BA_PROBE(16 + 16 * (w == ((sub + 1) & 3) ? 1 : 0), 1);
        cbK += 2 * (n - 1 - k) - 1;
        __syncthreads();
        if (k < n) BA_PROBE(own ? 0 : 16 + 16 * (w == ((sub + 1) & 3) ? 1 : 0), 5);
      }
      if (kg == 0 && s_flag[1]) return;                      // (a zero matrix)
#pragma unroll
      for (int i = 0; i + 2 < NL; i++) { acc[i] = acc[i + 2]; if (TWO) acc2[i] = acc2[i + 2]; }
      cbase += 8;
    }
  }
  __device__ __forceinline__ void factor() {
    const int ngroups = (n + 7) >> 3;
    if (w >= 4) {
      // wavefronts 4 .. 7 hold no columns: they keep the barrier count (and leave with the others on a zero matrix).  Wavefront 4 carries the forward substitution
      // (d[i] -= L(i, j) d[j], j ascending: the right-hand side one row per lane, a slot's two columns as soon as the barrier behind it has published them)
      double d1 = 0.0, d2 = 0.0;
      if (w == 4) { d1 = rhs[min(lane, n - 1)]; if (TWO) d2 = rhs[min(64 + lane, n - 1)]; }
      int cb = 0, j = 0;
      for (int s = 0; s < 4 * ngroups; s++) {
        __syncthreads();
        if (s == 3 && s_flag[1]) return;
        if (w == 4) {
          for (int e = 0; e < 2 && j < n - 1; e++, j++) {
            const double lj = Lc[cb + lane], lj2 = TWO ? Lc[cb + 64 + lane] : 0.0;
            const double dj = pick(d1, d2, j);
            if (lane > j) d1 -= lj * dj;
            if (TWO && 64 + lane > j) d2 -= lj2 * dj;
            cb += n - 1 - j;
          }
        }
      }
      if (w == 4) { if (lane < n) rhs[lane] = d1; if (TWO && 64 + lane < n) rhs[64 + lane] = d2; }
      return;
    }
    double acc[NCW], acc2[NCW];
#pragma unroll
    for (int i = 0; i < NCW; i++) { acc[i] = 0.0; acc2[i] = 0.0; }
    cbase = 2 * w; done = -1; cbK = 0; cbJ = 0;
    // the own columns still to come shrink by two per group of four slots: four copies of the loop, each sized for what is left when it starts
    constexpr int G = NCW / 2, Q1 = G / 4, Q2 = G / 2, Q3 = (3 * G) / 4;
    run<NCW>(0, min(Q1, ngroups), acc, acc2);
    if (s_flag[1]) return;
    run<NCW - 2 * Q1>(Q1, min(Q2, ngroups), acc, acc2);
    run<NCW - 2 * Q2>(Q2, min(Q3, ngroups), acc, acc2);
    run<NCW - 2 * Q3>(Q3, ngroups, acc, acc2);
  }
};
enum { BA_PIVOT_RANKS = 0, BA_PIVOT_TIES = 1, BA_PIVOT_NAN = 2 };
template <int MF>
__device__ __forceinline__ void baLdltSolveCore(const int n, double* __restrict__ Lc, double* __restrict__ Tc, double* __restrict__ dgS,
                                                const double* __restrict__ rhsS, double* __restrict__ rhs, double* __restrict__ xs, int* __restrict__ perm, int* s_flag,
                                                const int exact_backsub, const int (&pr)[BASolveDims<MF>::QMAX], int* ticks, const long long t_begin) {
  constexpr int QMAX = BASolveDims<MF>::QMAX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform by construction; said so, everything derived from it lives in scalar registers and branches)
  __shared__ int s_big[128];
  // ---- pivot order (wavefront 0; position p of the sequence = lane p & 63, register p >> 6)
  if (wave == 0) {
    const bool v0 = lane < n, v1 = lane + 64 < n;
    const double m0 = v0 ? fabs(dgS[lane]) : 0.0, m1 = v1 ? fabs(dgS[lane + 64]) : 0.0;
    int rank0 = 0, same0 = 0, rank1 = 0, same1 = 0;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
      double o[4];
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = fabs(dgS[j + q]);
#pragma unroll
      for (int q = 0; q < 4; q++) { rank0 += o[q] > m0 ? 1 : 0; same0 += o[q] == m0 ? 1 : 0; rank1 += o[q] > m1 ? 1 : 0; same1 += o[q] == m1 ? 1 : 0; }
    }
    for (; j < n; j++) { const double o = fabs(dgS[j]); rank0 += o > m0 ? 1 : 0; same0 += o == m0 ? 1 : 0; rank1 += o > m1 ? 1 : 0; same1 += o == m1 ? 1 : 0; }
    const bool anyNan = __ballot((v0 && !(m0 == m0)) || (v1 && !(m1 == m1))) != 0ull;
    const bool anyTie = __ballot((v0 && same0 > 1) || (v1 && same1 > 1)) != 0ull;
    if (anyNan) {
      // the literal selection loop (fabs(NaN) > x is false: a NaN is never selected, and nothing is selected over a NaN that stands at position k)
      if (lane == 0) {
        for (int i = 0; i < n; i++) perm[i] = i;
        for (int k = 0; k < n; k++) {
          int big = k; double bigv = fabs(dgS[k]);
          for (int i = k + 1; i < n; i++) { const double v = fabs(dgS[i]); if (v > bigv) { bigv = v; big = i; } }
          if (big != k) { const double t = dgS[k]; dgS[k] = dgS[big]; dgS[big] = t; const int q = perm[k]; perm[k] = perm[big]; perm[big] = q; }
        }
        s_flag[2] = BA_PIVOT_NAN;
      }
    } else if (!anyTie) {
      // all |diagonal| values distinct: step k selects the k-th largest whatever the swaps did to the others — the order is the rank itself
      if (v0) perm[rank0] = lane;
      if (v1) perm[rank1] = lane + 64;
      if (lane == 0) s_flag[2] = BA_PIVOT_RANKS;
    } else {
      // Ties.  Step t of the selection takes the element of rank t where that element is alone in its group (its rank = the number of larger elements = its step); a tie group
      // of m elements fills the steps rank .. rank + m - 1 in the order of the members' CURRENT positions when the group starts.  An element only moves when the step that
      // equals its position passes without taking it: it then goes to where that step's element stood (big[t], final once it is >= t).  So
      //   big[t] = chase(sel[t], t),   position of e at step s = chase(e, s),   chase(p, s): while (p < s) p = big[p]
      // and every big[u] a chase reads belongs to an earlier step: the steps before a tie group settle in a few wave-wide rounds (every unsettled step follows one link
      // per round; the lowest unsettled one always can), the members chase, rank themselves by position, and fill their steps.  perm[] doubles as sel[]; bigA in LDS.
      int* const bigA = s_big;
      const bool two = n > 64;
      if (v0) perm[lane] = -1;
      if (v1) perm[lane + 64] = -1;
      BA_WAVE_LDS_SYNC();
      if (v0 && same0 == 1) perm[rank0] = lane;
      if (v1 && same1 == 1) perm[rank1] = lane + 64;
      BA_WAVE_LDS_SYNC();
      int cur0 = v0 ? perm[lane] : 0, cur1 = v1 ? perm[lane + 64] : 0;          // step t = lane / lane + 64
      bool known0 = v0 && cur0 >= 0, known1 = v1 && cur1 >= 0;
      bool fin0 = known0 && cur0 >= lane, fin1 = known1 && cur1 >= lane + 64;
      if (v0) bigA[lane] = fin0 ? cur0 : -1;
      if (v1) bigA[lane + 64] = fin1 ? cur1 : -1;
      unsigned long long ts0 = __ballot(v0 && !known0), ts1 = two ? __ballot(v1 && !known1) : 0ull;   // the steps of tie groups
      while ((ts0 | ts1) != 0ull) {
        const int s0 = ts0 ? __builtin_ctzll(ts0) : 64 + __builtin_ctzll(ts1);
        const bool mem0 = v0 && rank0 == s0 && same0 > 1, mem1 = v1 && rank1 == s0 && same1 > 1;      // the group's members (elements lane / lane + 64)
        const unsigned long long mm0 = __ballot(mem0), mm1 = two ? __ballot(mem1) : 0ull;
        const int m = __popcll(mm0) + __popcll(mm1);
        // (a) settle every step before s0
        for (;;) {
          const bool w0 = known0 && !fin0 && lane < s0, w1 = two && known1 && !fin1 && lane + 64 < s0;
          if ((__ballot(w0) | __ballot(w1)) == 0ull) break;
          BA_WAVE_LDS_SYNC();
          if (w0) { const int bb = bigA[cur0]; if (bb >= 0) { cur0 = bb; if (cur0 >= lane) { fin0 = true; } } }
          if (w1) { const int bb = bigA[cur1]; if (bb >= 0) { cur1 = bb; if (cur1 >= lane + 64) { fin1 = true; } } }
          BA_WAVE_LDS_SYNC();
          if (w0 && fin0) bigA[lane] = cur0;
          if (w1 && fin1) bigA[lane + 64] = cur1;
        }
        BA_WAVE_LDS_SYNC();
        // (b) where the members stand when step s0 begins
        int p0 = lane, p1 = lane + 64;
        for (;;) {
          const bool c0 = mem0 && p0 < s0, c1 = mem1 && p1 < s0;
          if ((__ballot(c0) | __ballot(c1)) == 0ull) break;
          if (c0) p0 = bigA[p0];
          if (c1) p1 = bigA[p1];
        }
        // (c) the members in the order of their positions
        int o0 = 0, o1 = 0;
        for (unsigned long long q = mm0; q; q &= q - 1ull) { const int pb = __builtin_amdgcn_readlane(p0, __builtin_ctzll(q)); o0 += pb < p0 ? 1 : 0; o1 += pb < p1 ? 1 : 0; }
        for (unsigned long long q = mm1; q; q &= q - 1ull) { const int pb = __builtin_amdgcn_readlane(p1, __builtin_ctzll(q)); o0 += pb < p0 ? 1 : 0; o1 += pb < p1 ? 1 : 0; }
        if (mem0) { perm[s0 + o0] = lane; bigA[s0 + o0] = p0; }
        if (mem1) { perm[s0 + o1] = lane + 64; bigA[s0 + o1] = p1; }
        BA_WAVE_LDS_SYNC();
        // the group's steps are settled
        if (v0 && lane >= s0 && lane < s0 + m) { known0 = true; fin0 = true; cur0 = bigA[lane]; }
        if (v1 && lane + 64 >= s0 && lane + 64 < s0 + m) { known1 = true; fin1 = true; cur1 = bigA[lane + 64]; }
        // (clear the steps s0 .. s0 + m - 1)
        for (int t = s0; t < s0 + m; t++) { if (t < 64) ts0 &= ~(1ull << t); else ts1 &= ~(1ull << (t - 64)); }
      }
      BA_WAVE_LDS_SYNC();
      if (lane == 0) s_flag[2] = BA_PIVOT_TIES;
    }
  }
  __syncthreads();
  if (ticks && tid == 0) ticks[3] = (int)(wall_clock64() - t_begin);   // pivot order
  // ---- the permuted system P A P^T, column-packed in place of the row-packed A (through registers: all swaps applied up front — the same operands meet in the same
  // order as with a swap at every step)
  {
    double val[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; q++) {
      val[q] = 0.0;
      if (pr[q] >= 0) {
        const int a = perm[pr[q] >> 8], b = perm[pr[q] & 255];
        val[q] = Lc[a >= b ? triIdx(a, b) : triIdx(b, a)];
      }
    }
    if (tid < n) rhs[tid] = rhsS[perm[tid]];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QMAX; q++) if (pr[q] >= 0) Lc[baColBase(pr[q] & 255, n) + (pr[q] >> 8)] = val[q];
  }
  __syncthreads();
  if (ticks && tid == 0) ticks[4] = (int)(wall_clock64() - t_begin);   // permuted
  // ---- LDL^T with the forward substitution riding along (d[i] -= L(i, j) d[j], j ascending)
  if (n > 64) { BAFactor<MF, true> fc; fc.n = n; fc.lane = lane; fc.w = wave; fc.Lc = Lc; fc.Tc = Tc; fc.rhs = rhs; fc.s_flag = s_flag; fc.factor(); }
  else { BAFactor<MF, false> fc; fc.n = n; fc.lane = lane; fc.w = wave; fc.Lc = Lc; fc.Tc = Tc; fc.rhs = rhs; fc.s_flag = s_flag; fc.factor(); }
  __syncthreads();   // (wavefront 4's right-hand side)
  if (ticks && tid == 0) ticks[5] = (int)(wall_clock64() - t_begin);   // factorised + forward substitution
  const bool zero = s_flag[1] != 0;
  // ---- diagonal solve, back substitution (wavefront 0)
  if (wave == 0) {
    double x0 = 0.0, x1 = 0.0;
    if (lane < n) { const double dd = Lc[baColBase(lane, n) + lane]; x0 = rhs[lane]; if (fabs(dd) > 2.2250738585072014e-308) x0 /= dd; else x0 = 0; if (zero) x0 = 0.0; }
    if (lane + 64 < n) { const double dd = Lc[baColBase(lane + 64, n) + lane + 64]; x1 = rhs[lane + 64]; if (fabs(dd) > 2.2250738585072014e-308) x1 /= dd; else x1 = 0; if (zero) x1 = 0.0; }
    if (!zero) {
      if (exact_backsub) {
        // the order of ldltSolveTransposed (row i subtracts L(j, i) x_j for j = i + 1 .. n - 1, ascending): one dependent chain of n^2 / 2 subtractions
        if (lane < n) rhs[lane] = x0;
        if (lane + 64 < n) rhs[lane + 64] = x1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) for (int i = n - 1; i >= 0; i--) { double sacc = rhs[i]; const double* Li = Lc + baColBase(i, n); for (int j = i + 1; j < n; j++) sacc -= Li[j] * rhs[j]; rhs[i] = sacc; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < n) x0 = rhs[lane];
        if (lane + 64 < n) x1 = rhs[lane + 64];
      } else {
        // column-oriented (no barrier per step): as soon as x_i stands, every row above subtracts its term (row r subtracts in the order i = n - 1 .. r + 1: the same
        // terms, another association)
        const int b0 = baColBase(min(lane, n - 1), n), b1 = baColBase(min(lane + 64, n - 1), n);
        double m0n = Lc[b0 + n - 1], m1n = Lc[b1 + n - 1];   // L(n - 1, lane): row i of L for the next step, one step ahead (lanes at or beyond row i read a slot they do not use)
        for (int i = n - 1; i > 0; i--) {
          const double m0 = m0n, m1 = m1n;
          if (i > 1) { m0n = Lc[b0 + i - 1]; m1n = Lc[b1 + i - 1]; }
          const double xi = baReadLaneF64(i >= 64 ? x1 : x0, i & 63);
          if (lane < i) x0 -= m0 * xi;
          if (lane + 64 < i) x1 -= m1 * xi;
        }
      }
    }
    // undo the permutation
    if (lane < n) xs[perm[lane]] = x0;
    if (lane + 64 < n) xs[perm[lane + 64]] = x1;
  }
  __syncthreads();
  if (ticks && tid == 0) ticks[6] = (int)(wall_clock64() - t_begin);   // back substitution, x in the original order
}

// Diagnostics / tests (dmvio_hip_ba_debug_solve): the solve of a given system HPassed x = b exactly as k_ba_solve runs it — Jacobi scaling (H_ii + 10)^-1/2, pivot order,
// LDL^T, forward / back substitution — against dmvio_hip_ba_solve_ldlt on the host.  out: x[n] | perm[n] (as doubles) | branch | zero
template <int MF>
__global__ void __launch_bounds__(BA_SOLVE_THREADS) k_ba_solve_debug(const int n, const double* __restrict__ H, const double* __restrict__ b, const int exact_backsub, double* __restrict__ out) {
  constexpr int QMAX = BASolveDims<MF>::QMAX;
  extern __shared__ double s_mem[];
  const int tid = threadIdx.x;
  const int NP = (n * (n + 1)) / 2;
  double* const Lc = s_mem;
  double* const Tc = Lc + NP + 128;
  double* const sv = Tc + NP + 128;
  double* const dgS = sv + n;
  double* const rhsS = dgS + n;
  double* const rhs = rhsS + n;
  double* const xs = rhs + n;
  int* const perm = reinterpret_cast<int*>(xs + n);
  __shared__ int s_flag[4];
  int pr[QMAX];
  baOwnedPairs<QMAX>(n, pr);
#ifdef BA_SOLVE_PROBE
  for (int pass = 0; pass < 3; pass++) {   // (the same work three times in one launch: what a cold instruction cache costs the first pass)
  const long long tp0 = wall_clock64();
  __syncthreads();
#endif
  if (tid < 4) s_flag[tid] = 0;
  if (tid < n) { const double v = H[(size_t)tid * n + tid]; const double sc = 1.0 / sqrt(v + 10); sv[tid] = sc; dgS[tid] = sc * v * sc; rhsS[tid] = sc * b[tid]; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < QMAX; q++) if (pr[q] >= 0) { const int i = pr[q] >> 8, j = pr[q] & 255; Lc[tid + q * BA_SOLVE_THREADS] = sv[i] * H[(size_t)i * n + j] * sv[j]; }
  __syncthreads();
  baLdltSolveCore<MF>(n, Lc, Tc, dgS, rhsS, rhs, xs, perm, s_flag, exact_backsub, pr, nullptr, 0);
#ifdef BA_SOLVE_PROBE
  if (tid == 0) g_ba_probe[48 + pass] = wall_clock64() - tp0;
  }
#endif
  if (tid < n) { out[tid] = sv[tid] * xs[tid]; out[n + tid] = (double)perm[tid]; }
  if (tid == 0) { out[2 * n] = (double)s_flag[2]; out[2 * n + 1] = (double)s_flag[1]; }
}

// FINISH: only settle the pending decision (behind the last iteration's chain); the host reads the final state from the window's record.
// Everything a sequential (order-preserving) loop reads is staged in LDS first: a dependent chain over global memory costs a memory latency per term; the global loads the
// assembly needs (the stitched system's triangle, two values per owned pair) are issued at the top and land while the state is settled.
template <int MF, bool FINISH>
__global__ void __launch_bounds__(BA_SOLVE_THREADS) k_ba_solve(BAWinDev* __restrict__ wins, const int iteration) {
  constexpr bool finish = FINISH;
  constexpr int QMAX = BASolveDims<MF>::QMAX;
  constexpr bool ALIAS = BASolveDims<MF>::ALIAS_HM;
  BAWinDev& V = wins[blockIdx.x];
  BASolveDev& S = V.S;
  extern __shared__ double s_mem[];
  const int tid = threadIdx.x, n = S.n, F = S.F;
  const int NP = (n * (n + 1)) / 2;
  const int hs = baSolveHmStride(n);
  double* const HMs = s_mem;          // n rows of stride hs: the marginalisation prior
  double* const Lc = ALIAS ? s_mem : HMs + (size_t)n * hs;   // NP: the scaled matrix (row-packed), then the permuted one (column-packed) -> L (strict lower) and D (diagonal)
  double* const Tc = Lc + NP + 128;   // NP: D_k L(., k), column-packed like L
  double* const basis = ALIAS ? s_mem + max((size_t)n * hs, 2 * ((size_t)NP + 128)) : Tc + NP + 128;   // 7 x n
  double* const d = basis + 7 * n;    // n: stacked delta (calib | frames)
  double* const bP = d + n;           // n: bM + HM delta
  double* const HLd = bP + n;         // n
  double* const sv = HLd + n;         // n
  double* const rhsS = sv + n;        // n: scaled right-hand side
  double* const rhs = rhsS + n;       // n: ... permuted
  double* const xs = rhs + n;         // n
  double* const dgS = xs + n;         // n: scaled diagonal (the pivot search's input)
  double* const tv = dgS + n;         // n: calcMEnergy rows / projections
  double* const bMs = tv + n;         // n
  double* const dgV = bMs + n;        // n: the unscaled diagonal
  int* const perm = reinterpret_cast<int*>(dgV + n);   // n ints (<= n doubles reserved)
  double* const fst = dgV + 2 * n;    // F x 10 state | F x 10 state_zero | F x 10 state_backup | F x 8 prior (+ pad): 40 F
  double* const fzero = fst + 10 * F;
  double* const fbak = fzero + 10 * F;
  double* const fprior = fbak + 10 * F;
  __shared__ int s_flag[4];
  __shared__ double s_scal[8];
  __shared__ double s_cal[16];        // c_value, c_value_zero, c_value_backup, cPrior
  __shared__ Pose s_w2c[BA_MAXF_CAP], s_c2w[BA_MAXF_CAP], s_evalPT[BA_MAXF_CAP];
  __shared__ float s_abexp[BA_MAXF_CAP], s_cPriorF[4];
  __shared__ double s_scaled[BA_MAXF_CAP][10];
  __shared__ float s_K[9], s_Ki[9];
  const long long t_begin = wall_clock64();
#define SOLVE_TICK(i) do { if (tid == 0) S.ticks[i] = (int)(wall_clock64() - t_begin); } while (0)   // (measured: no effect on the kernel's duration)

  // ---- stage the window's solve state (coalesced) while thread 0 settles the pending decision (FullSystemOptimize.cpp:556-583 behind the accept test)
  const int haveM = S.haveM;
  // (gathering the system from the stitched blocks inside this kernel — no gather launch — was built and measured: one workgroup issuing the ~140k loads of
  // gatherValue costs 45 us against the 12 us of the 37-workgroup gather kernel it would replace)
  const double* const gHM = gl(S.HM); const double* const gbM = gl(S.bM); const double* const gBasis = gl(S.basis); double* const gTrace = gl(S.trace); BACtl* const gCtl = gl(V.ctl);
  const double* __restrict__ HA = gl(V.sys);
  const double* __restrict__ bA = gl(V.sys) + (size_t)n * n;
  const double* __restrict__ Hsc = bA + n;
  const double* __restrict__ bsc = Hsc + (size_t)n * n;
  int pr[QMAX];
  double hA[QMAX], hS[QMAX], hL[QMAX];   // the owned pairs' entries of H_A, H_sc and (residuals kept linearised: accumulateLF_MT's system) H_L
  double bAi = 0.0, bsci = 0.0, bLi = 0.0;
  const bool haveL = V.n_lin > 0;
  const double* __restrict__ HLr = gl(V.sysL);
  // the pair tables and the calibration members of the state / of the backup: one of the two sets is copied over the other below (loads up front, stores behind the decision)
  constexpr int TQ = (BA_MAXF_CAP * (BA_MAXF_CAP - 1) * 14 + BA_SOLVE_THREADS - 1) / BA_SOLVE_THREADS;
  const int npT = F * (F - 1) * 14;
  float tcur[TQ], tbk[TQ], wcur = 0.f, wbk = 0.f;
  {
    const float* Tcur = &V.T.v[0][0];
    const float* Tbk = &V.Tb.v[0][0];
#pragma unroll
    for (int q = 0; q < TQ; q++) { const int i = tid + q * BA_SOLVE_THREADS; tcur[q] = 0.f; tbk[q] = 0.f; if (i < npT) { tcur[q] = Tcur[i]; tbk[q] = Tbk[i]; } }
    if (tid < 8) { wcur = (&V.W.fx)[tid]; wbk = (&V.Wb.fx)[tid]; }
  }
  // pointers of the window's record the tail goes through (read once, up front: a pointer fetched where it is used costs a memory round trip in front of the access)
  double* const x_last = gl(S.x_last);
  float* const Xxc = V.X.xc;
  float* const XxAd = V.X.xAd;
  const float* const adHostF = gl(S.adHostF);
  const float* const adTargetF = gl(S.adTargetF);
  // resubstitution input o = tid (hh, t, c): its adjoint columns, constant over the call
  float ahP[8], atP[8];
  if (!finish && tid < F * F * 8) {
    const int c = tid & 7, t = (tid >> 3) % F, hh = (tid >> 3) / F;
    const size_t base = ((size_t)hh + (size_t)F * t) * 64;
#pragma unroll
    for (int r = 0; r < 8; r++) { ahP[r] = adHostF[base + r * 8 + c]; atP[r] = adTargetF[base + r * 8 + c]; }
  }
  if (!finish) {
    baOwnedPairs<QMAX>(n, pr);
#pragma unroll
    for (int q = 0; q < QMAX; q++) { hA[q] = 0.0; hS[q] = 0.0; hL[q] = 0.0; if (pr[q] >= 0) { const size_t o = (size_t)(pr[q] >> 8) * n + (pr[q] & 255); hA[q] = HA[o]; hS[q] = Hsc[o]; if (haveL) hL[q] = HLr[o]; } }
    if (tid < n) { bAi = bA[tid]; bsci = bsc[tid]; if (haveL) bLi = HLr[(size_t)n * n + tid]; }
    else if (tid >= 128 && tid < 128 + n) { const size_t o = (size_t)(tid - 128) * n + (tid - 128); bAi = HA[o]; bsci = Hsc[o]; if (haveL) bLi = HLr[o]; }   // (the diagonal, for the threads that scale it)
    if (haveM) {
      for (int i = tid; i < n * n; i += BA_SOLVE_THREADS) HMs[(i / n) * hs + (i % n)] = gHM[i];
      for (int i = tid; i < n; i += BA_SOLVE_THREADS) bMs[i] = gbM[i];
    }
    if (iteration >= 2) for (int i = tid; i < S.nBasis * n; i += BA_SOLVE_THREADS) basis[i] = gBasis[i];
  }
  for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) { const BAFrameDev& f = S.fr[i / 10]; fst[i] = f.state[i % 10]; fzero[i] = f.state_zero[i % 10]; fbak[i] = f.state_backup[i % 10]; }
  for (int i = tid; i < 8 * F; i += BA_SOLVE_THREADS) fprior[i] = S.fr[i >> 3].prior[i & 7];
  // (what the tail reads of the window's record: staged here so that no load stands between the tail's stores)
  for (int i = tid; i < 7 * F; i += BA_SOLVE_THREADS) reinterpret_cast<double*>(&s_evalPT[i / 7])[i % 7] = reinterpret_cast<const double*>(&S.fr[i / 7].evalPT)[i % 7];
  if (tid >= 64 && tid < 64 + F) s_abexp[tid - 64] = S.fr[tid - 64].ab_exposure;
  if (tid >= 128 && tid < 132) s_cPriorF[tid - 128] = S.cPriorF[tid - 128];
  if (tid < 4) { s_cal[tid] = S.c_value[tid]; s_cal[4 + tid] = S.c_value_zero[tid]; s_cal[8 + tid] = S.c_value_backup[tid]; s_cal[12 + tid] = S.cPrior[tid]; }
  if (tid == 0) {
    int acc = 1;
    if (S.stepped) {
      acc = __hip_atomic_load(&gCtl->accept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double E0 = __hip_atomic_load(&gCtl->lastE0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (acc) { S.lastL = S.newL; S.lastM = S.newM; S.lambda = fmax(S.lambda * 0.25, 1e-5); S.n_accepted++; }
      else S.lambda *= 1e2;
      S.iterations_done++;
      const int row = S.iterations_done;
      if (row < 64) { gTrace[4 * row] = E0; gTrace[4 * row + 1] = S.lastL; gTrace[4 * row + 2] = S.lastM; gTrace[4 * row + 3] = acc ? 1.0 : 0.0; }
      S.stepped = 0;
    } else if (iteration == 0 && !finish) {
      // row 0 of the trace: the initial state (its photometric energy is what the initial linearisation's decision pass left in the control block)
      gTrace[0] = __hip_atomic_load(&gCtl->lastE0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); gTrace[1] = S.lastL; gTrace[2] = S.lastM; gTrace[3] = 1.0;
    }
    s_flag[0] = acc; s_flag[1] = 0; s_flag[2] = 0; s_flag[3] = 0;
    s_scal[4] = S.lambda; s_scal[5] = S.lastL; s_scal[6] = S.lastM;
  }
  __syncthreads();
  // every load issued above has landed by now: say so once — the stores below carry loaded values, and behind the branches in between the compiler would otherwise drain the
  // memory pipeline in front of each of them (a store round trip apiece: 5 us)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  SOLVE_TICK(0);   // staged + settled
  const int prevAccepted = s_flag[0];
  {
    // accepted (or nothing pending): backupState — the backup takes the state; rejected: loadSateBackup — the state and its pair tables go back to the backup's
    // (after which the backup equals the state: one copy either way)
    float* Tcur = &V.T.v[0][0];
    float* Tbk = &V.Tb.v[0][0];
    if (prevAccepted) {
      if (!finish) {
        for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) fbak[i] = fst[i];   // (the record's copy of the backup is written at the very end: off the solve's path)
              if (tid < 4) s_cal[8 + tid] = s_cal[tid];
      #pragma unroll
        for (int q = 0; q < TQ; q++) { const int i = tid + q * BA_SOLVE_THREADS; if (i < npT) Tbk[i] = tcur[q]; }
        if (tid < 8) (&V.Wb.fx)[tid] = wcur;
        if (haveL) { for (int i = tid; i < F * F * 8; i += BA_SOLVE_THREADS) V.adHTdelta[1][i] = V.adHTdelta[0][i]; if (tid < 4) V.cDeltaF[1][tid] = V.cDeltaF[0][tid]; }
            }
    } else {
      for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) { fst[i] = fbak[i]; S.fr[i / 10].state[i % 10] = fbak[i]; }
      if (tid < 4) { s_cal[tid] = s_cal[8 + tid]; S.c_value[tid] = s_cal[8 + tid]; }
#pragma unroll
      for (int q = 0; q < TQ; q++) { const int i = tid + q * BA_SOLVE_THREADS; if (i < npT) Tcur[i] = tbk[q]; }
      if (tid < 8) (&V.W.fx)[tid] = wbk;
      if (haveL) { for (int i = tid; i < F * F * 8; i += BA_SOLVE_THREADS) V.adHTdelta[0][i] = V.adHTdelta[1][i]; if (tid < 4) V.cDeltaF[0][tid] = V.cDeltaF[1][tid]; }
    }
  }
  if (finish) return;
  __syncthreads();
  // ---- the stacked delta (EnergyFunctional::setDeltaF as BAHost::setPrecalcValues keeps it), the prior diagonal
  if (tid < 4) { d[tid] = (double)(float)(s_cal[tid] - s_cal[4 + tid]); HLd[tid] = s_cal[12 + tid]; }
  for (int i = tid; i < 8 * F; i += BA_SOLVE_THREADS) {
    const int f = i >> 3, k = i & 7;
    d[4 + i] = fst[10 * f + k] - fzero[10 * f + k];
    HLd[4 + i] = fprior[i];
  }
  __syncthreads();
  // bM_top = bM + HM * delta (EnergyFunctional.cpp:864), row sums in index order — beside it the diagonal of HFinal_top - H_sc / (1 + lambda) (BAHost::solveSystem:
  // (HL + HM) + HA, times (1 + lambda), minus H_sc * fac) and its Jacobi scaling
  const double lambda = s_scal[4];
  const double fac = 1.0f / (1 + lambda);
  if (tid < n) {
    double s = haveM ? bMs[tid] : 0.0;
    if (haveM) s = baSeqDot(s, HMs + (size_t)tid * hs, d, n);
    bP[tid] = s;
  } else if (tid >= 128 && tid < 128 + n) {
    const int i = tid - 128;
    double v = ((bLi + HLd[i]) + (haveM ? HMs[(size_t)i * hs + i] : 0.0)) + bAi;   // ((HL + prior) + HM) + HA (BAHost::solveSystem); bLi holds H_L's diagonal entry here
    v *= (1 + lambda);
    v = v - bsci * fac;
    const double sc = 1.0 / sqrt(v + 10);
    sv[i] = sc;
    dgV[i] = v;
    dgS[i] = sc * v * sc;
  }
  __syncthreads();
  SOLVE_TICK(1);   // backup, delta, bM_top, diagonal
  // ---- the scaled system, row-packed (ALIAS: the prior's entries leave the shared room before the first element is written)
  {
    double hm[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; q++) hm[q] = (pr[q] >= 0 && haveM) ? HMs[(size_t)(pr[q] >> 8) * hs + (pr[q] & 255)] : 0.0;
    if (ALIAS) __syncthreads();
#pragma unroll
    for (int q = 0; q < QMAX; q++) {
      if (pr[q] >= 0) {
        const int i = pr[q] >> 8, j = pr[q] & 255;
        double v;
        if (i == j) v = dgV[i];
        else v = (((hL[q] + 0.0) + hm[q]) + hA[q]) - hS[q] * fac;
        Lc[tid + q * BA_SOLVE_THREADS] = sv[i] * v * sv[j];   // (sv_i * H_ij) * sv_j
      }
    }
  }
  if (tid < n) {
    const int i = tid;
    const double bL = i < 4 ? s_cal[12 + i] * d[i] : fprior[i - 4] * fst[10 * ((i - 4) >> 3) + ((i - 4) & 7)];   // prior * delta_prior (= state)
    const double bF = (((bLi + bL) + bP[i]) + bAi) - bsci;   // bL_top = accumulateLF's b + prior * delta (BAHost::lfTop)
    rhsS[i] = sv[i] * bF;
  }
  __syncthreads();
  SOLVE_TICK(2);   // system assembled and scaled
  baLdltSolveCore<MF>(n, Lc, Tc, dgS, rhsS, rhs, xs, perm, s_flag, S.exact_backsub, pr, S.ticks, t_begin);
  if (ALIAS && haveM) for (int i = tid; i < n * n; i += BA_SOLVE_THREADS) HMs[(i / n) * hs + (i % n)] = gHM[i];   // (back into the shared room, for E_M below)
  if (tid == 0) S.pivot_branch = s_flag[2];
  // undo the scaling: x = S P^T x'
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
  if (tid < n) x_last[tid] = xs[tid];
  SOLVE_TICK(7);   // x
  // ---- resubstituteF_MT's inputs (BAHost::prepareResubstitute): xc, xAd[h F + t][c] = x_h . adHostF(:, c) + x_t . adTargetF(:, c) in fp32, sequential over r
  if (tid < 4) Xxc[tid] = (float)xs[tid];
  for (int o = tid; o < F * F * 8; o += BA_SOLVE_THREADS) {
    const int c = o & 7, t = (o >> 3) % F, hh = (o >> 3) / F;
    float ah[8], at[8];
    if (o == tid) {
#pragma unroll
      for (int r = 0; r < 8; r++) { ah[r] = ahP[r]; at[r] = atP[r]; }   // (fetched at the top)
    } else {
      const size_t base = ((size_t)hh + (size_t)F * t) * 64;
#pragma unroll
      for (int r = 0; r < 8; r++) { ah[r] = adHostF[base + r * 8 + c]; at[r] = adTargetF[base + r * 8 + c]; }   // sixteen loads in flight, then the ordered sums
    }
    float s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) { s1 += (float)xs[4 + 8 * hh + r] * ah[r]; s2 += (float)xs[4 + 8 * t + r] * at[r]; }
    XxAd[((size_t)F * hh + t) * 8 + c] = s1 + s2;
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
  if (haveL) {
    // EnergyFunctional::setDeltaF at the stepped state (BAHost::adHTdeltaF): adHTdeltaF[h + F t][c] = delta_h . adHostF(:, c) + delta_t . adTargetF(:, c) in fp32, sequential over r — what
    // the linearised residuals' energy (k_ba_lin_energy_b, next in the stream) and, once the step is accepted, their records (k_ba_lin_records_b) are evaluated at
    for (int o = tid; o < F * F * 8; o += BA_SOLVE_THREADS) {
      const int c = o & 7, t = (o >> 3) % F, hh = (o >> 3) / F;
      float ah[8], at[8];
      if (o == tid) {
#pragma unroll
        for (int r = 0; r < 8; r++) { ah[r] = ahP[r]; at[r] = atP[r]; }
      } else {
        const size_t base = ((size_t)hh + (size_t)F * t) * 64;
#pragma unroll
        for (int r = 0; r < 8; r++) { ah[r] = adHostF[base + r * 8 + c]; at[r] = adTargetF[base + r * 8 + c]; }
      }
      float s1 = 0, s2 = 0;
#pragma unroll
      for (int r = 0; r < 8; r++) { s1 += (float)d[4 + 8 * hh + r] * ah[r]; s2 += (float)d[4 + 8 * t + r] * at[r]; }
      V.adHTdelta[0][((size_t)hh + (size_t)F * t) * 8 + c] = s1 + s2;
    }
    if (tid < 4) V.cDeltaF[0][tid] = (float)d[tid];
  }
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
    const Pose w = poseMul(poseExp(s_scaled[f]), s_evalPT[f]);
    s_w2c[f] = w; s_c2w[f] = poseInv(w);
  }
  // E_M of the stepped state, rows in index order (independent of the poses: runs beside the exponentials)
  if (tid >= 128 && tid < 128 + n) {
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
    baAffFromTo(s_abexp[hh], s_abexp[t], s_scaled[hh][6], s_scaled[hh][7], s_scaled[t][6], s_scaled[t][7], aff);
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
    for (int i = 0; i < 4; i++) { const float cd = (float)d[i]; ec += cd * s_cPriorF[i] * cd; }
    const double newL = E + ec;
    S.newL = newL; S.newM = s;
    V.D.lastL = s_scal[5]; V.D.lastM = s_scal[6]; V.D.newL = newL; V.D.newM = s;
    S.stepped = 1;
    S.ticks[11] = (int)(wall_clock64() - t_begin);   // energies
  }
  SOLVE_TICK(10);  // pair tables (thread 0's share)
  // backupState's copy in the window's record (what a rejected step restores from, and what the host reads back)
  for (int i = tid; i < 10 * F; i += BA_SOLVE_THREADS) S.fr[i / 10].state_backup[i % 10] = fbak[i];
  if (tid < 4) S.c_value_backup[tid] = s_cal[8 + tid];
#undef SOLVE_TICK
}

}  // namespace dmv
