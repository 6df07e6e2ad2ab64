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

// Window size (the reference's setting_maxFrames, util/settings.cpp:100, a run-time setting registered in util/MainSettings.cpp:223,246): F is a run-time value of every
// kernel.  Two compiled sizes of the per-pair kernel ARGUMENTS (BAPreDyn, ResubArgs travel as kernel arguments so that no upload sits between the host's step and the
// launch): windows of up to BA_MAXF keyframes (the reference's default 7 + the newest) use the compact ones, larger windows up to BA_MAXF_CAP the wide ones.
#define BA_MAXF 8
#define BA_MAXF_CAP 12
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

// The members of BAPrecalc that change with every step of the GN loop (PRE_KRKiTll, PRE_KtTll, PRE_aff_mode) for the F*(F-1) ordered pairs
// h != t, passed as kernel arguments: the loop's linearisations then need no table upload between the host's step and the launch (R0, t0,
// b0 — functions of the evaluation point — stay in the device table of the last full upload).
template <int MF> struct BAPreDynT { float v[MF * (MF - 1)][14]; };
typedef BAPreDynT<BA_MAXF_CAP> BAPreDyn;   // the host keeps the wide form; baNarrow() cuts it down for a launch (pair index h (F-1) + t' < F (F-1): a prefix)
__host__ __device__ __forceinline__ int baPairIndex(const int h, const int t, const int F) { return h * (F - 1) + (t < h ? t : t - 1); }
// back-substitution inputs: xc (4 floats) and xAd (F*F x 8 floats, index h*F + t; EnergyFunctional.cpp:280-282), passed as kernel arguments
template <int MF> struct ResubArgsT { float xc[4]; float xAd[MF * MF * 8]; };
typedef ResubArgsT<BA_MAXF_CAP> ResubArgs;
template <class Narrow, class Wide> static inline Narrow baNarrow(const Wide& w) { Narrow n; static_assert(sizeof(Narrow) <= sizeof(Wide), "prefix"); __builtin_memcpy(&n, &w, sizeof(Narrow)); return n; }
struct BAWindow {
  int F, w, h, N, R;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;  // CalibHessian::fxl().. / fxli()..
  float wM3, hM3;
  float huberTH, outlierTHSum, modeA, modeB;
  int slot[BA_MAXF_CAP];
  float frameEnergyTH[BA_MAXF_CAP];
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
  float* idepth_hessian;                  // PointHessian::idepth_hessian (AccumulatedSCHessian.cpp:42,50)
  // residuals kept linearised outside a marginalisation (dmvio_hip_ba_fix_linearization; all NULL otherwise): Hdd_accLF, bd_accLF, Hcd_accLF of addPoint<1>
  // (AccumulatedTopHessian.cpp:84-98,131-143).  With them Hcd holds Hcd_accAF + Hcd_accLF — what every consumer reads — and HcdAF the A part alone.
  const float *lHdd, *lbd, *lHcd;
  float* HcdAF;
};

struct BARes {      // SoA, R entries
  const int *point, *target;
  unsigned char *state, *newState, *active, *which;  // which: buffer (0/1) holding the APPLIED record
  unsigned char* removed;   // the residual left the graph: FullSystem::linearizeAll(true) deletes every residual that is not active after its applyRes (FullSystemOptimize.cpp:176-212);
                            // it stays OOB / inactive for every later linearisation, resetOOB and marginalisation of this graph
  float *energy, *newEnergy, *newEnergyWO;
  float* center;   // R x 3 centerProjectedTo
  const int* newestSlot;   // R: position of the residual among those that target the newest keyframe, or -1
  float* newestE;          // state_NewEnergyWithOutlier of exactly those residuals, contiguous (what setNewFrameEnergyTH looks at)
  float* rec[2];   // R x REC_FLOATS
  const unsigned char* lin;   // EFResidual::isLinearized outside a marginalisation (NULL: none): such a residual is no member of activeResiduals (FullSystemOptimize.cpp:436-446) —
                              // not relinearised, not applied, not counted in the energy or the newest keyframe's threshold; its record and activity stay what they were
};

// the record buffer `w` of a residual: a select between the two pointers, never an indexed read of the array — a dynamically indexed member would force the whole
// structure into scratch memory wherever it is a local copy (the batched kernels, ba_batch_kernels.hpp)
__device__ __forceinline__ float* baRec(const BARes& Rs, const int w) { return w ? Rs.rec[1] : Rs.rec[0]; }

__constant__ int c_patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};  // settings.cpp:296, pattern 8

// ------------------------------------------------------------------------------------------------ device-side decisions
// The GN loop of FullSystem::optimize (FullSystemOptimize.cpp:485-586) needs three small decisions after every linearisation: the energy
// sum, the newest keyframe's outlier threshold (setNewFrameEnergyTH, :96-149: an nth_element over the residuals that target it) and
// accept / reject.  They are taken by the LAST workgroup of the linearisation kernel to finish, so that the kernels that follow in the
// stream (apply + per-point sums + accumulation + stitching when accepted, restore + relinearisation when rejected) can be enqueued
// without the host in between; each of them starts by reading BACtl::accept and returns at once when it is not its turn.
struct BACtl {
  unsigned int cnt_lin;      // arrive counters of the last-workgroup patterns (reset by the last arrival)
  unsigned int cnt_gather;
  int accept;                // decision of the last accept test
  int pad;
  double lastE0;             // photometric energy of the state the window stands at: set by every decision pass (an accepted step's energy, a plain or a
                             // restored state's), read by the next accept test — the host need not wait for a rejected step's relinearisation
};
#define BA_GATHER_MAX_BLOCKS 96   // (2 (n^2 + n) + 256) / 256 workgroups of k_ba_stitch_gather, n = 4 + 8 BA_MAXF_CAP = 100: 80
struct BAHostRes {           // host-coherent pinned memory, polled by the host
  double E[2];               // [0] energy of the last plain / stepped-state linearisation, [1] of the relinearisation after a rejected step
  float th[2];               // newest keyframe's threshold after each of them
  int accept;
  unsigned int ticket;       // published (release, system scope) by the final kernel of a chain — for an accept test as soon as the decision stands, BEFORE the threshold
                             // selection of the same pass (which nobody waits for: the next linearisation reads it in stream order); otherwise last
  unsigned int th_ticket;    // the same ticket once th[] of that pass is stored
  unsigned int gticket[BA_GATHER_MAX_BLOCKS];   // k_ba_stitch_gather: one slot per workgroup, the chain's ticket behind the workgroup's slice of the system
  int ticks[6];              // diagnostics: 100 MHz wall-clock ticks since the deciding workgroup started its own residuals: decision pass begin, energy
                             // summed, threshold keys loaded, threshold selected (dmvio_hip_ba_last_decide_ticks)
};
struct BADecide {
  const float* newestE;      // state_NewEnergyWithOutlier of the residuals that target the newest keyframe (BARes::newestE)
  int n_newest, newestFrame;
  float* frameTH;            // [BA_MAXF_CAP] FrameHessian::frameEnergyTH, device copy read by the linearisation
  double* epart;             // per-workgroup energy partials
  float thN, thFacMedian, thConstWeight, overallW, thCap;   // setting_frameEnergyTHN / FacMedian / ConstWeight, setting_overallEnergyTHWeight, IMU cap (<= 0: none)
  int mode;                  // -1 no decision pass at all, 0 energy + threshold, 1 + accept test of a stepped state, 2 relinearisation after a rejected step
  int update_th;
  double lastE0, lastL, lastM, newL, newM;
  int lastE0_from_ctl;       // the accept test compares with BACtl::lastE0 instead of the host's copy
  BACtl* ctl;
  BAHostRes* host;
  unsigned int ticket;
  int publish;               // this kernel is the last of its chain: store the ticket
  // points sharded over ranks (dmvio_hip_ba_set_comm): the linearisation's last workgroup only PACKS this rank's record
  // [fp64 local energy | n_newest | pad | n_newest energies | -1 padding] (mode 3); the records of all ranks are all-gathered and
  // k_ba_decide_global takes the decisions over their union, identically on every rank
  float* xchg_local;         // this rank's record (xchg_width floats)
  const float* xchg_all;     // world x xchg_width floats, rank order
  int xchg_width, world;
};
#define BA_XCHG_HEADER 4
enum { BA_GATE_ALWAYS = 0, BA_GATE_ACCEPTED = 1, BA_GATE_REJECTED = 2 };
__device__ __forceinline__ bool baGateClosed(const BACtl* ctl, const int gate) {
  if (gate == BA_GATE_ALWAYS) return false;
  const int a = __hip_atomic_load(&ctl->accept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return gate == BA_GATE_ACCEPTED ? a == 0 : a != 0;
}
// sum of the per-workgroup energy partials by all 256 threads of a workgroup: every thread adds a contiguous run, the 256 runs are combined by a
// fixed pairwise tree — one fixed order for every launch (a single thread walking hundreds of partials would cost tens of microseconds of
// dependent loads).  Result valid in thread 0.
__device__ __forceinline__ double baEnergyTree(const double* epart, const int nblocks, double* s_red /*[256]*/) {
  const int tid = threadIdx.x;
  const int chunk = (nblocks + 255) / 256;
  double e = 0;
  for (int i = tid * chunk; i < min((tid + 1) * chunk, nblocks); i++) e += __hip_atomic_load(epart + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s_red[tid] = e;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (tid < w) s_red[tid] += s_red[tid + w]; __syncthreads(); }
  return s_red[0];
}
// mode 3: this rank's record for the all-gather, written by the last workgroup of a linearisation
__device__ __forceinline__ void baPackBlock(const BADecide& D, const int nblocks) {
  __shared__ double s_red[256];
  const int tid = threadIdx.x;
  const double e = baEnergyTree(D.epart, nblocks, s_red);
  if (tid == 0) { __builtin_memcpy(D.xchg_local, &e, 8); D.xchg_local[2] = __int_as_float(D.n_newest); D.xchg_local[3] = 0.0f; }
  for (int i = tid; i < D.xchg_width - BA_XCHG_HEADER; i += 256)
    D.xchg_local[BA_XCHG_HEADER + i] = i < D.n_newest ? __hip_atomic_load(D.newestE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.0f;
}
#define BA_DECIDE_KEYS 4096
// executed by all 256 threads of the last workgroup of a linearisation
// EXT_KEYS: the radix select's key staging (16 KB) lives in LDS the calling kernel no longer needs (k_ba_linearize_b1: its pattern patches) instead of an array of its own
template <bool EXT_KEYS = false>
__device__ __forceinline__ void baDecideBlock(const BADecide& D, const int nblocks, const long long t_start, unsigned int* ext_keys = nullptr) {
  __shared__ unsigned int s_keys_own[EXT_KEYS ? 1 : BA_DECIDE_KEYS];
  __shared__ unsigned int s_scan_own[EXT_KEYS ? 1 : 256];
  __shared__ double s_red_own[EXT_KEYS ? 1 : 256];
  __shared__ unsigned int s_misc_own[EXT_KEYS ? 1 : 8];
  unsigned int* const s_keys = EXT_KEYS ? ext_keys : s_keys_own;                       // BA_DECIDE_KEYS
  unsigned int* const s_scan = EXT_KEYS ? ext_keys + BA_DECIDE_KEYS : s_scan_own;      // 256
  double* const s_red = EXT_KEYS ? reinterpret_cast<double*>(ext_keys + BA_DECIDE_KEYS + 256) : s_red_own;   // 256 doubles
  unsigned int* const s_misc = EXT_KEYS ? ext_keys + BA_DECIDE_KEYS + 256 + 512 : s_misc_own;               // 8
  unsigned int &s_cnt = s_misc[0], &s_prefix = s_misc[1], &s_k = s_misc[2];
  double& s_E = *reinterpret_cast<double*>(s_misc + 4);
  const int tid = threadIdx.x;
  const long long tk0 = wall_clock64();
  long long tk1 = tk0, tk2 = tk0, tk3 = tk0;
  // energy: every thread adds a contiguous run of the per-workgroup partials, the 256 runs are combined by a fixed pairwise tree — one fixed
  // order for every launch (a single thread walking hundreds of partials would cost tens of microseconds of dependent loads)
  const bool gathered = D.world > 0;   // decisions over the all-gathered records of all ranks (k_ba_decide_global)
  if (!gathered) {
    const double e = baEnergyTree(D.epart, nblocks, s_red);
    if (tid == 0) { s_E = e; s_cnt = 0; }
  } else if (tid == 0) {
    double e = 0;                        // rank order: the same sum on every rank
    for (int r = 0; r < D.world; r++) { double v; __builtin_memcpy(&v, D.xchg_all + (size_t)r * D.xchg_width, 8); e += v; }
    s_E = e; s_cnt = 0;
  }
  __syncthreads();
  tk1 = wall_clock64();
  if (D.mode == 1 && tid == 0) {
    // the accept test needs the energy only (FullSystemOptimize.cpp:553-554): decide and tell the host NOW; the threshold selection below (3 + 5 us) then runs while the host
    // reads the decision and enqueues the kernels of its branch, which start in stream order behind this one
    const double E = s_E;
    const double lastE0 = D.lastE0_from_ctl ? __hip_atomic_load(&D.ctl->lastE0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : D.lastE0;
    const int acc = (E + D.newL + D.newM < lastE0 + D.lastL + D.lastM) ? 1 : 0;   // energy[1] = 0, dynamic weight folded into lastM / newM by the host
    if (acc) __hip_atomic_store(&D.ctl->lastE0, E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&D.ctl->accept, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&D.host->accept, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&D.host->E[0], E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (D.publish) __hip_atomic_store(&D.host->ticket, D.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const float* const keys_src = gathered ? D.xchg_all : D.newestE;
  const int keys_n = gathered ? D.world * D.xchg_width : D.n_newest;
  const int rec_w = gathered ? D.xchg_width : 0;
  auto loadE = [&](const int i) -> float {   // record headers are not energies
    if (rec_w && (i % rec_w) < BA_XCHG_HEADER) return -1.0f;
    return __hip_atomic_load(keys_src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  float th = __hip_atomic_load(D.frameTH + D.newestFrame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (D.update_th) {
    const int n = keys_n;
    auto key = [&](const int i) -> unsigned int {
      if (i < BA_DECIDE_KEYS) return s_keys[i];
      const float v = loadE(i);
      return v >= 0 ? __float_as_uint(v) : 0xFFFFFFFFu;
    };
    unsigned int mine = 0;
    for (int base = 0; base < n; base += 8 * 256) {   // eight independent loads per thread in flight (an atomic load per loop trip would serialise on its latency)
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) { const int i = base + q * 256 + tid; v[q] = i < n ? loadE(i) : -1.0f; }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = base + q * 256 + tid;
        const unsigned int k = v[q] >= 0 ? __float_as_uint(v[q]) : 0xFFFFFFFFu;   // non-negative floats order like their bit patterns; skipped residuals sort last
        if (i < n && i < BA_DECIDE_KEYS) s_keys[i] = k;
        if (v[q] >= 0) mine++;
      }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    tk2 = wall_clock64();
    const unsigned int m = s_cnt;
    if (m == 0) th = 12 * 12 * 8;   // "should never happen, but lets make sure" (FullSystemOptimize.cpp:112-116)
    else {
      if (tid == 0) { s_prefix = 0; s_k = (unsigned int)(int)(D.thN * m); }   // nthIdx = setting_frameEnergyTHN * allResVec.size() (float product, truncated)
      unsigned int mask = 0;
      for (int shift = 24; shift >= 0; shift -= 8) {   // radix select: the nthIdx-th smallest key, 8 bits per pass
        s_scan[tid] = 0;
        __syncthreads();
        const unsigned int prefix = s_prefix, kk = s_k;
        for (int i = tid; i < n; i += 256) { const unsigned int k = key(i); if ((k & mask) == prefix) atomicAdd(&s_scan[(k >> shift) & 255u], 1u); }
        __syncthreads();
        // wavefront 0: lane l owns bins 4l..4l+3; inclusive scan over the lanes by six shuffles, then the bin whose range holds index kk
        if (tid < 64) {
          const unsigned int c0 = s_scan[4 * tid], c1 = s_scan[4 * tid + 1], c2 = s_scan[4 * tid + 2], c3 = s_scan[4 * tid + 3];
          const unsigned int tot = c0 + c1 + c2 + c3;
          unsigned int incl = tot;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) { const unsigned int up = __shfl_up(incl, off, 64); if (tid >= off) incl += up; }
          const unsigned int excl = incl - tot;
          if (kk >= excl && kk < incl) {
            unsigned int r = kk - excl, bin = 4 * tid;
            if (r >= c0) { r -= c0; bin++; if (r >= c1) { r -= c1; bin++; if (r >= c2) { r -= c2; bin++; } } }
            s_k = r; s_prefix = prefix | (bin << shift);
          }
        }
        mask |= 255u << shift;
        __syncthreads();
      }
      const float nthElement = sqrtf(__uint_as_float(s_prefix));
      th = nthElement * D.thFacMedian;
      th = 26.0f * D.thConstWeight + th * (1 - D.thConstWeight);
      th = th * th;
      th *= D.overallW * D.overallW;
      if (D.thCap > 0 && th > D.thCap) th = D.thCap;   // IMUIntegration::newFrameEnergyTH (FullSystemOptimize.cpp:136-140)
    }
  }
  tk3 = wall_clock64();
  if (tid == 0) {
    D.host->ticks[0] = (int)(tk0 - t_start); D.host->ticks[1] = (int)(tk1 - t_start); D.host->ticks[2] = (int)(tk2 - t_start); D.host->ticks[3] = (int)(tk3 - t_start);
    const double E = s_E;
    const int slot = D.mode == 2 ? 1 : 0;
    if (D.update_th) __hip_atomic_store(D.frameTH + D.newestFrame, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (D.mode == 0) {
      __hip_atomic_store(&D.ctl->accept, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&D.ctl->lastE0, E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (D.mode == 2) {
      __hip_atomic_store(&D.ctl->lastE0, E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (D.mode != 1) __hip_atomic_store(&D.host->E[slot], E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // mode 1: stored with the decision, above
    __hip_atomic_store(&D.host->th[slot], th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (D.publish) {
      __hip_atomic_store(&D.host->th_ticket, D.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (D.mode != 1) __hip_atomic_store(&D.host->ticket, D.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------------ linearize
// Eight lanes per residual: lane idx evaluates pattern pixel idx (projection, bilinear tap, weights, its products); the sums over
// the pattern are formed in pattern order — ((((0 + v0) + v1) + v2) ...) — by seven DPP row-shift adds per quantity, so every
// value is bit-identical to the reference's sequential loop while the eight gathers of a residual are in flight together.
#define LIN_THREADS 256
#define BA_PATCH_STRIDE 49   // floats per lane of the one-lane linearisation's 6x8 pattern patch (odd: lanes reading the same patch element meet no bank conflict)
#define LIN_RES_PER_BLOCK (LIN_THREADS / 8)
__device__ __forceinline__ float dppShl(const float v, const int k) {   // value of lane (l + k) of the same 16-lane row, k = 1..7
  int r = 0;
  const int x = __float_as_int(v);
  switch (k) {
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x101, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x102, 0xF, 0xF, true); break;
    case 3: r = __builtin_amdgcn_update_dpp(0, x, 0x103, 0xF, 0xF, true); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0xF, true); break;
    case 5: r = __builtin_amdgcn_update_dpp(0, x, 0x105, 0xF, 0xF, true); break;
    case 6: r = __builtin_amdgcn_update_dpp(0, x, 0x106, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x107, 0xF, 0xF, true); break;
  }
  return __int_as_float(r);
}
// in lane 0 of every 8-lane group: 0 + v(0) + v(1) + ... + v(7), added left to right
__device__ __forceinline__ float seqSum8(const float v) {
  float s = 0.0f + v;
#pragma unroll
  for (int k = 1; k < 8; k++) s = s + dppShl(v, k);
  return s;
}

// gate: BA_GATE_REJECTED = this launch is the relinearisation that follows a rejected step (loadSateBackup + linearizeAll,
// FullSystemOptimize.cpp:575-581): it returns at once when the step was accepted.  use_backup: the point part of loadSateBackup rides along
// (idepth = idepth_zero = idepth_backup, written back by the group of the point's first residual).
// The body is shared by the single-window kernel (arguments by value: no upload between the host's step and the launch) and the batched kernel of the device-resident
// loop (k_ba_linearize_b: arguments of window blockIdx.y read from device memory, where k_ba_solve wrote them).  Tv: the step-dependent precalc members per ordered pair,
// xc / xAd: the back-substitution inputs; nblocks: workgroups of THIS window's linearisation (the arrive count of the decision pass).
__device__ __forceinline__ void baLinearizeBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const BAPrecalc* __restrict__ pre,
                                                const FrameStore& fs, float* __restrict__ fullJ,
                                                const unsigned char* __restrict__ pt_mask, const BADecide& D, const int gate, const int use_backup,
                                                const float (*__restrict__ Tv)[14], const int use_dyn, const float* __restrict__ Xxc, const float* __restrict__ XxAd, const int do_resub,
                                                const int nblocks) {
  if (baGateClosed(D.ctl, gate)) {
    if (D.publish && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&D.host->ticket, D.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  double* __restrict__ energy_partials = D.epart;
  const long long t_kernel0 = wall_clock64();
  // pt_mask != NULL: only the residuals of the flagged points, after PointFrameResidual::resetOOB (Residuals.h:82-89) — the
  // relinearisation FullSystem::flagPointsForRemoval performs before a point is marginalised (FullSystem.cpp:836-849)
  const int ri = blockIdx.x * LIN_RES_PER_BLOCK + (threadIdx.x >> 3), idx = threadIdx.x & 7;
  const bool lead = idx == 0;
  double myE = 0.0;
  // all eight lanes of a residual take the same branches below (group-uniform conditions); stores come from the leading lane
  if (ri < W.R && (!pt_mask || pt_mask[Rs.point[ri]])) {
    float* __restrict__ rec = baRec(Rs, Rs.which[ri] ^ 1) + (size_t)ri * REC_FLOATS;  // write the NON-applied buffer
    int state = Rs.state[ri];
    const float oldEnergy = pt_mask ? 0.f : Rs.energy[ri];
    const bool gone = Rs.removed[ri] != 0;   // deleted by an earlier fix-linearisation: the reference has no such residual any more (state stays OOB below)
    if (pt_mask && !gone) state = BA_IN;
    if (lead) {
      Rs.newEnergyWO[ri] = -1.0f;
      { const int sl = Rs.newestSlot[ri]; if (sl >= 0) __hip_atomic_store(Rs.newestE + sl, -1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      if (pt_mask && !gone) { Rs.state[ri] = BA_IN; Rs.energy[ri] = 0.f; Rs.newEnergy[ri] = 0.f; Rs.newState[ri] = BA_OUTLIER; }
    }
    bool done = false;
    if (state == BA_OOB) { if (lead) Rs.newState[ri] = BA_OOB; myE = oldEnergy; done = true; }
    if (Rs.lin && !pt_mask && Rs.lin[ri]) { myE = 0.0; done = true; }   // kept linearised: outside activeResiduals (only the point's back-substitution below concerns it)
    const int pi = Rs.point[ri], ti = Rs.target[ri];
    const int hi = P.host[pi];
    BAPrecalc pc = pre[hi + W.F * ti];
    if (use_dyn) {
      const float* __restrict__ dv = Tv[baPairIndex(hi, ti, W.F)];
#pragma unroll
      for (int k = 0; k < 9; k++) pc.KRKi[k] = dv[k];
      pc.Kt[0] = dv[9]; pc.Kt[1] = dv[10]; pc.Kt[2] = dv[11]; pc.aff0 = dv[12]; pc.aff1 = dv[13];
    }
    const float pu = P.u[pi], pv = P.v[pi];
    // do_resub: EnergyFunctional::resubstituteFPt (EnergyFunctional.cpp:295-321) + the point part of doStepFromBackup for this residual's point,
    // fused in front of the linearisation of the stepped state (every group of a point's residuals computes the same step — lane q takes the
    // point's q-th residual, the products are subtracted in residual order like k_ba_resubstitute does; the group of the first residual
    // stores it).  Saves a kernel and its launch gap per Gauss-Newton iteration.
    float id_new = 0.0f;
    if (do_resub) {
      const int r0 = P.res_begin[pi], r1 = P.res_begin[pi + 1];
      const float bk = P.idepth_backup[pi];
      float bsum = P.bdSumF[pi];
      {
        float dotc = 0;
        dotc += Xxc[0] * (P.Hcd[4 * pi + 0] + 0.0f); dotc += Xxc[1] * (P.Hcd[4 * pi + 1] + 0.0f);
        dotc += Xxc[2] * (P.Hcd[4 * pi + 2] + 0.0f); dotc += Xxc[3] * (P.Hcd[4 * pi + 3] + 0.0f);
        bsum -= dotc;
      }
      const int grp = (threadIdx.x & 63) & ~7;
      int ngood = 0;
      for (int rb = r0; rb < r1; rb += 8) {
        const int rq = min(rb + idx, r1 - 1);
        const bool act = rb + idx < r1 && Rs.active[rq] != 0;
        const float* __restrict__ jp = baRec(Rs, Rs.which[rq]) + (size_t)rq * REC_FLOATS + REC_JPJD;
        const float* __restrict__ xa = XxAd + (size_t)(hi * W.F + Rs.target[rq]) * 8;
        float d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += xa[k] * jp[k];
        if (!act) d = 0.0f;
        bsum = bsum - d;
#pragma unroll
        for (int k = 1; k < 8; k++) bsum = bsum - dppShl(d, k);
        ngood += __popcll((__ballot(act) >> grp) & 0xFFull);
      }
      float st = ngood == 0 ? 0.0f : -bsum * P.HdiF[pi];
      // the leading lane holds the ordered sum: hand its step to the group's other lanes
      st = __shfl(st, (threadIdx.x & 63) & ~7, 64);
      id_new = bk + 1.0f * st;
      if (lead && r0 == ri) { P.step[pi] = st; P.idepth[pi] = id_new; P.idepth_zero[pi] = id_new; }
    }
    float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x = 0, d_d_y = 0;
    if (!done) {
      // centre pixel at the LINEARISATION point (idepth_zero, evalPT poses)
      const float idz = do_resub ? id_new : (use_backup ? P.idepth_backup[pi] : P.idepth_zero[pi]);
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
      if (!ok) { if (lead) Rs.newState[ri] = BA_OOB; myE = oldEnergy; done = true; }
      else {
        if (lead) { Rs.center[3 * ri + 0] = Ku; Rs.center[3 * ri + 1] = Kv; Rs.center[3 * ri + 2] = new_idepth; }
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
      const float ids = do_resub ? id_new : (use_backup ? P.idepth_backup[pi] : P.idepth[pi]);
      float* fj = fullJ ? fullJ + (size_t)ri * 74 : nullptr;
      // this lane's pattern pixel
      const float xu = pu + c_patternP[idx][0], xv = pv + c_patternP[idx][1];
      const float q0 = pc.KRKi[0] * xu + pc.KRKi[1] * xv + pc.KRKi[2] * 1.0f + pc.Kt[0] * ids;
      const float q1 = pc.KRKi[3] * xu + pc.KRKi[4] * xv + pc.KRKi[5] * 1.0f + pc.Kt[1] * ids;
      const float q2 = pc.KRKi[6] * xu + pc.KRKi[7] * xv + pc.KRKi[8] * 1.0f + pc.Kt[2] * ids;
      const float Ku = q0 / q2, Kv = q1 / q2;
      const bool inb = (Ku > 1.1f && Kv > 1.1f && Ku < W.wM3 && Kv < W.hM3);
      float3 hit = interp33(img, inb ? Ku : 2.5f, inb ? Kv : 2.5f, W.w);
      bool good = inb && isfinite(hit.x);
      // the residual goes OOB if ANY of its pattern pixels fails (the reference breaks out of the loop at the first one)
      {
        const unsigned long long bad = __ballot(!good);
        const int grp = (threadIdx.x & 63) & ~7;
        if ((bad >> grp) & 0xFFull) good = false;
      }
      if (!good) { if (lead) Rs.newState[ri] = BA_OOB; myE = oldEnergy; done = true; }
      else {
        const float color = P.color[pi * 8 + idx];
        const float residual = hit.x - (pc.aff0 * color + pc.aff1);
        const float drdA = (color - pc.b0);
        float wgt = sqrtf(W.outlierTHSum / (W.outlierTHSum + (hit.y * hit.y + hit.z * hit.z)));
        wgt = 0.5f * (wgt + P.weights[pi * 8 + idx]);
        float hw = fabsf(residual) < W.huberTH ? 1.0f : W.huberTH / fabsf(residual);
        const float eTerm = wgt * wgt * hw * residual * residual * (2 - hw);
        if (hw < 1) hw = sqrtf(hw);
        hw = hw * wgt;
        hit.y *= hw; hit.z *= hw;
        const float resF = residual * hw;
        float jab0 = drdA * hw, jab1 = hw;
        const float tJI00 = hit.y * hit.y, tJI11 = hit.z * hit.z, tJI10 = hit.y * hit.z;
        const float tJa00 = drdA * hw * hit.y, tJa01 = drdA * hw * hit.z, tJa10 = hw * hit.y, tJa11 = hw * hit.z;
        const float tJb00 = drdA * drdA * hw * hw, tJb01 = drdA * hw * hw, tJb11 = hw * hw;
        const float twJI2 = hw * hw * (hit.y * hit.y + hit.z * hit.z);
        if (W.modeA < 0) jab0 = 0;
        if (W.modeB < 0) jab1 = 0;
        // accumulation-side inner products of addPoint<0> (resApprox = resF)  (AccumulatedTopHessian.cpp:103-113)
        const float tJIr0 = resF * hit.y, tJIr1 = resF * hit.z, tJar0 = resF * jab0, tJar1 = resF * jab1, trr = resF * resF;
        if (fj) { fj[idx] = resF; fj[30 + idx] = hit.y; fj[38 + idx] = hit.z; fj[46 + idx] = jab0; fj[54 + idx] = jab1; }
        // sums over the pattern, in pattern order (valid in the leading lane)
        float energyLeft = seqSum8(eTerm);
        const float JI00 = seqSum8(tJI00), JI11 = seqSum8(tJI11), JI10 = seqSum8(tJI10);
        const float Ja00 = seqSum8(tJa00), Ja01 = seqSum8(tJa01), Ja10 = seqSum8(tJa10), Ja11 = seqSum8(tJa11);
        const float Jb00 = seqSum8(tJb00), Jb01 = seqSum8(tJb01), Jb11 = seqSum8(tJb11), wJI2 = seqSum8(twJI2);
        const float JIr0 = seqSum8(tJIr0), JIr1 = seqSum8(tJIr1), Jar0 = seqSum8(tJar0), Jar1 = seqSum8(tJar1), rr = seqSum8(trr);
        if (lead) {
          Rs.newEnergyWO[ri] = energyLeft;
          { const int sl = Rs.newestSlot[ri]; if (sl >= 0) __hip_atomic_store(Rs.newestE + sl, energyLeft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          const float th = fmaxf(D.frameTH[hi], D.frameTH[ti]);
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
  }
  // block energy partial, fixed order (only the leading lane of a residual contributes)
  __shared__ double s_e[LIN_RES_PER_BLOCK];
  if ((threadIdx.x & 7) == 0) s_e[threadIdx.x >> 3] = myE;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0;
    for (int k = 0; k < LIN_RES_PER_BLOCK; k++) sum += s_e[k];
    __hip_atomic_store(energy_partials + blockIdx.x, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // loadSateBackup, point part: the group of a point's first residual restores it (a point without residuals never left its backup: its step is 0)
  if (use_backup && ri < W.R && lead) {
    const int pi = Rs.point[ri];
    if (P.res_begin[pi] == ri) { const float bk = P.idepth_backup[pi]; P.idepth[pi] = bk; P.idepth_zero[pi] = bk; }
  }
  if (D.mode < 0) return;
  // the last workgroup to arrive takes the decisions (energy sum, threshold of the newest keyframe, accept / reject)
  // What the deciding workgroup reads (energy partials, per-residual energies) was stored with agent-scope atomic stores, i.e. written
  // through to the device's coherence point; waiting for those stores (workgroup-scope release = s_waitcnt) before a RELAXED arrive is enough.
  // An agent-scope release / acquire here would write back and invalidate the whole L2 of the XCD — once per workgroup, with megabytes of
  // freshly written records in it (measured: +40 us per launch).
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&D.ctl->cnt_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(nblocks - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (threadIdx.x == 0) __hip_atomic_store(&D.ctl->cnt_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (D.mode == 3) baPackBlock(D, nblocks);
  else baDecideBlock(D, nblocks, t_kernel0);
}
// The same linearisation with ONE lane per residual (the lane walks the eight pattern pixels itself): for launches that fill the device — the batched loop over many
// windows — where the eight-lane form's redundant geometry (every lane of a residual computes the same projection and Jacobian rows) costs throughput instead of hiding
// latency.  Every value is formed by the same operations in the same order: the pattern sums are 0 + v0 + v1 + ... in the lane, the point's back-substitution subtracts its
// residuals' products in residual order, and the energy partials keep the eight-lane kernel's partition (runs of LIN_RES_PER_BLOCK residuals, summed in order, one partial
// per run — nruns of them — so the decision pass adds the same numbers in the same tree).  nblocks: workgroups of this window's launch (256 residuals each).
// No pt_mask / fullJ form: the callers that need those use the eight-lane kernel.
__device__ __forceinline__ void baLinearizeBody1(const BAWindow& W, const BAPoints& P, const BARes& Rs, const BAPrecalc* __restrict__ pre, const FrameStore& fs,
                                                 const BADecide& D, const int gate, const int use_backup, const float (*__restrict__ Tv)[14], const int use_dyn,
                                                 const float* __restrict__ Xxc, const float* __restrict__ XxAd, const int do_resub, const int nblocks, const int nruns,
                                                 float* __restrict__ patch) {
  if (baGateClosed(D.ctl, gate)) return;
  const long long t_kernel0 = wall_clock64();
  const int ri = blockIdx.x * LIN_THREADS + threadIdx.x;
  double myE = 0.0;
  if (ri < W.R) {
    float* __restrict__ rec = baRec(Rs, Rs.which[ri] ^ 1) + (size_t)ri * REC_FLOATS;  // write the NON-applied buffer
    const int state = Rs.state[ri];
    const float oldEnergy = Rs.energy[ri];
    Rs.newEnergyWO[ri] = -1.0f;
    const int sl = Rs.newestSlot[ri];
    if (sl >= 0) __hip_atomic_store(Rs.newestE + sl, -1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool done = false;
    if (state == BA_OOB) { Rs.newState[ri] = BA_OOB; myE = oldEnergy; done = true; }
    if (Rs.lin && Rs.lin[ri]) { myE = 0.0; done = true; }
    const int pi = Rs.point[ri], ti = Rs.target[ri];
    const int hi = P.host[pi];
    BAPrecalc pc = pre[hi + W.F * ti];
    if (use_dyn) {
      const float* __restrict__ dv = Tv[baPairIndex(hi, ti, W.F)];
#pragma unroll
      for (int k = 0; k < 9; k++) pc.KRKi[k] = dv[k];
      pc.Kt[0] = dv[9]; pc.Kt[1] = dv[10]; pc.Kt[2] = dv[11]; pc.aff0 = dv[12]; pc.aff1 = dv[13];
    }
    const float pu = P.u[pi], pv = P.v[pi];
    float id_new = 0.0f;
    if (do_resub) {   // resubstituteFPt + the point part of doStepFromBackup (see baLinearizeBody)
      // the residuals of a point are consecutive lanes: the first of them IN THIS WAVEFRONT forms the point's step (the sums below, once per point and wavefront instead
      // of once per residual: a sixth of the scattered loads), the others take it from that lane
      const int r0 = P.res_begin[pi];
      const int lane = threadIdx.x & 63;
      const int lead = max(0, lane - (ri - r0));           // the lane of the point's first residual, or lane 0 where the point began in the previous wavefront
      if (lane == lead) {
        const int r1 = P.res_begin[pi + 1];
        const float bk = P.idepth_backup[pi];
        float bsum = P.bdSumF[pi];
        {
          float dotc = 0;
          dotc += Xxc[0] * (P.Hcd[4 * pi + 0] + 0.0f); dotc += Xxc[1] * (P.Hcd[4 * pi + 1] + 0.0f);
          dotc += Xxc[2] * (P.Hcd[4 * pi + 2] + 0.0f); dotc += Xxc[3] * (P.Hcd[4 * pi + 3] + 0.0f);
          bsum -= dotc;
        }
        int ngood = 0;
        for (int rq = r0; rq < r1; rq++) {
          if (Rs.active[rq] == 0) continue;   // the eight-lane form subtracts +0.0f for it
          const float* __restrict__ jp = baRec(Rs, Rs.which[rq]) + (size_t)rq * REC_FLOATS + REC_JPJD;
          const float* __restrict__ xa = XxAd + (size_t)(hi * W.F + Rs.target[rq]) * 8;
          float d = 0;
#pragma unroll
          for (int k = 0; k < 8; k++) d += xa[k] * jp[k];
          bsum = bsum - d;
          ngood++;
        }
        const float st = ngood == 0 ? 0.0f : -bsum * P.HdiF[pi];
        id_new = bk + 1.0f * st;
        if (r0 == ri) { P.step[pi] = st; P.idepth[pi] = id_new; P.idepth_zero[pi] = id_new; }
      }
      id_new = __shfl(id_new, lead, 64);
    }
    float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x = 0, d_d_y = 0;
    if (!done) {
      const float idz = do_resub ? id_new : (use_backup ? P.idepth_backup[pi] : P.idepth_zero[pi]);
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
      if (!ok) { Rs.newState[ri] = BA_OOB; myE = oldEnergy; done = true; }
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
      const float ids = do_resub ? id_new : (use_backup ? P.idepth_backup[pi] : P.idepth[pi]);
      float energyLeft = 0.0f, JI00 = 0.0f, JI11 = 0.0f, JI10 = 0.0f, Ja00 = 0.0f, Ja01 = 0.0f, Ja10 = 0.0f, Ja11 = 0.0f, Jb00 = 0.0f, Jb01 = 0.0f, Jb11 = 0.0f, wJI2 = 0.0f;
      float JIr0 = 0.0f, JIr1 = 0.0f, Jar0 = 0.0f, Jar1 = 0.0f, rr = 0.0f;
      bool allGood = true;
      // The pattern pixels' 4x4 tap neighbourhoods overlap — eight pixels inside 5x5 read the same image rows again and again, and with a different target image under
      // every lane none of those rows survives in the vector L1 (measured: 32 L2 requests per residual, one per tap row).  The pattern is sorted by rows (y = -2, -1, -1,
      // 0 | 0, 0, 1, 2): each half's neighbourhoods fit a window of 6 rows x 8 columns for any warp near unit scale.  Per half the lane fetches that window ONCE — six rows
      // of 32 bytes — into its private patch in LDS and interpolates from there: the same twelve values per tap, the same arithmetic (interp33Finish), 12 line requests
      // per residual instead of 32.  A half whose neighbourhoods do not fit (or an image too small) taps the image directly, as before.
      float* const mine = patch + (size_t)threadIdx.x * BA_PATCH_STRIDE;
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
        float Kus[4], Kvs[4];
        bool inbs[4];
        bool allIn = true;
        int ixmin = 0x7fffffff, ixmax = -1, iymin = 0x7fffffff, iymax = -1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int idx = 4 * half + k;
          const float xu = pu + c_patternP[idx][0], xv = pv + c_patternP[idx][1];
          const float q0 = pc.KRKi[0] * xu + pc.KRKi[1] * xv + pc.KRKi[2] * 1.0f + pc.Kt[0] * ids;
          const float q1 = pc.KRKi[3] * xu + pc.KRKi[4] * xv + pc.KRKi[5] * 1.0f + pc.Kt[1] * ids;
          const float q2 = pc.KRKi[6] * xu + pc.KRKi[7] * xv + pc.KRKi[8] * 1.0f + pc.Kt[2] * ids;
          const float Ku = q0 / q2, Kv = q1 / q2;
          const bool inb = (Ku > 1.1f && Kv > 1.1f && Ku < W.wM3 && Kv < W.hM3);
          Kus[k] = inb ? Ku : 2.5f; Kvs[k] = inb ? Kv : 2.5f; inbs[k] = inb;
          allIn = allIn && inb;
          const int ix = (int)Kus[k], iy = (int)Kvs[k];
          ixmin = min(ixmin, ix); ixmax = max(ixmax, ix); iymin = min(iymin, iy); iymax = max(iymax, iy);
        }
        const bool usePatch = allIn && ixmax - ixmin <= 4 && iymax - iymin <= 2 && W.w >= 8 && W.h >= 6;
        const int xs = min(ixmin - 1, W.w - 8), ys = min(iymin - 1, W.h - 6);   // window origin (kept inside the plane; the taps' columns / rows shift by the difference)
        if (usePatch) {
          const float* __restrict__ src = img + (size_t)ys * W.w + xs;
          float4 lo[6], hi4[6];
#pragma unroll
          for (int r = 0; r < 6; r++) { __builtin_memcpy(&lo[r], src + (size_t)r * W.w, 16); __builtin_memcpy(&hi4[r], src + (size_t)r * W.w + 4, 16); }
#pragma unroll
          for (int r = 0; r < 6; r++) {
            mine[8 * r + 0] = lo[r].x; mine[8 * r + 1] = lo[r].y; mine[8 * r + 2] = lo[r].z; mine[8 * r + 3] = lo[r].w;
            mine[8 * r + 4] = hi4[r].x; mine[8 * r + 5] = hi4[r].y; mine[8 * r + 6] = hi4[r].z; mine[8 * r + 7] = hi4[r].w;
          }
        }
#pragma unroll 2
      for (int k = 0; k < 4; k++) {
        const int idx = 4 * half + k;
        const float Ku = Kus[k], Kv = Kvs[k];
        const bool inb = inbs[k];
        Taps33 tp;
        if (usePatch) {
          const float* __restrict__ q = mine + ((int)Kv - 1 - ys) * 8 + ((int)Ku - 1 - xs);
          tp.A.x = q[1]; tp.A.y = q[2];
          tp.B.x = q[8]; tp.B.y = q[9]; tp.B.z = q[10]; tp.B.w = q[11];
          tp.C.x = q[16]; tp.C.y = q[17]; tp.C.z = q[18]; tp.C.w = q[19];
          tp.D.x = q[25]; tp.D.y = q[26];
        } else interp33Load(img, Ku, Kv, W.w, tp);
        float3 hit = interp33Finish<true>(tp, Ku, Kv);
        allGood = allGood && inb && isfinite(hit.x);
        const float color = P.color[pi * 8 + idx];
        const float residual = hit.x - (pc.aff0 * color + pc.aff1);
        const float drdA = (color - pc.b0);
        float wgt = sqrtf(W.outlierTHSum / (W.outlierTHSum + (hit.y * hit.y + hit.z * hit.z)));
        wgt = 0.5f * (wgt + P.weights[pi * 8 + idx]);
        float hw = fabsf(residual) < W.huberTH ? 1.0f : W.huberTH / fabsf(residual);
        const float eTerm = wgt * wgt * hw * residual * residual * (2 - hw);
        if (hw < 1) hw = sqrtf(hw);
        hw = hw * wgt;
        hit.y *= hw; hit.z *= hw;
        const float resF = residual * hw;
        float jab0 = drdA * hw, jab1 = hw;
        const float tJa00 = drdA * hw * hit.y, tJa01 = drdA * hw * hit.z, tJa10 = hw * hit.y, tJa11 = hw * hit.z;
        const float tJb00 = drdA * drdA * hw * hw, tJb01 = drdA * hw * hw, tJb11 = hw * hw;
        const float twJI2 = hw * hw * (hit.y * hit.y + hit.z * hit.z);
        if (W.modeA < 0) jab0 = 0;
        if (W.modeB < 0) jab1 = 0;
        energyLeft = energyLeft + eTerm;
        JI00 = JI00 + hit.y * hit.y; JI11 = JI11 + hit.z * hit.z; JI10 = JI10 + hit.y * hit.z;
        Ja00 = Ja00 + tJa00; Ja01 = Ja01 + tJa01; Ja10 = Ja10 + tJa10; Ja11 = Ja11 + tJa11;
        Jb00 = Jb00 + tJb00; Jb01 = Jb01 + tJb01; Jb11 = Jb11 + tJb11; wJI2 = wJI2 + twJI2;
        JIr0 = JIr0 + resF * hit.y; JIr1 = JIr1 + resF * hit.z; Jar0 = Jar0 + resF * jab0; Jar1 = Jar1 + resF * jab1; rr = rr + resF * resF;
      }
      }
      // the residual goes OOB if ANY of its pattern pixels fails (the reference breaks out of the loop at the first one)
      if (!allGood) { Rs.newState[ri] = BA_OOB; myE = oldEnergy; }
      else {
        Rs.newEnergyWO[ri] = energyLeft;
        if (sl >= 0) __hip_atomic_store(Rs.newestE + sl, energyLeft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float th = fmaxf(D.frameTH[hi], D.frameTH[ti]);
        if (energyLeft > th || wJI2 < 2) { energyLeft = th; Rs.newState[ri] = BA_OUTLIER; }
        else Rs.newState[ri] = BA_IN;
        Rs.newEnergy[ri] = energyLeft;
        myE = energyLeft;
        // compact record: lane l writes record l's floats 208 bytes apart.  Staging the records through LDS so that a store instruction covers consecutive floats was
        // measured SLOWER (318 vs 267 us for 64 windows: the staging traffic and its 28 KB of LDS cost more than the scattered stores, which L2 merges)
#pragma unroll
        for (int k = 0; k < 4; k++) { rec[REC_JPDC0 + k] = d_C_x[k]; rec[REC_JPDC1 + k] = d_C_y[k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) { rec[REC_JPDXI0 + k] = d_xi_x[k]; rec[REC_JPDXI1 + k] = d_xi_y[k]; }
        rec[REC_JIDX2 + 0] = JI00; rec[REC_JIDX2 + 1] = JI10; rec[REC_JIDX2 + 2] = JI11;
        rec[REC_JABJIDX + 0] = Ja00; rec[REC_JABJIDX + 1] = Ja01; rec[REC_JABJIDX + 2] = Ja10; rec[REC_JABJIDX + 3] = Ja11;
        rec[REC_JAB2 + 0] = Jb00; rec[REC_JAB2 + 1] = Jb01; rec[REC_JAB2 + 2] = Jb11;
        rec[REC_JI_R + 0] = JIr0; rec[REC_JI_R + 1] = JIr1; rec[REC_JAB_R + 0] = Jar0; rec[REC_JAB_R + 1] = Jar1; rec[REC_RR] = rr;
        rec[REC_JPDD + 0] = d_d_x; rec[REC_JPDD + 1] = d_d_y;
        const float v0 = JI00 * d_d_x + JI10 * d_d_y, v1 = JI10 * d_d_x + JI11 * d_d_y;   // takeDataF (EnergyFunctionalStructs.cpp:39-49)
#pragma unroll
        for (int k = 0; k < 6; k++) rec[REC_JPJD + k] = d_xi_x[k] * v0 + d_xi_y[k] * v1;
        rec[REC_JPJD + 6] = Ja00 * d_d_x + Ja01 * d_d_y;
        rec[REC_JPJD + 7] = Ja10 * d_d_x + Ja11 * d_d_y;
        rec[REC_BD] = JIr0 * d_d_x + JIr1 * d_d_y;
        rec[REC_HDD] = v0 * d_d_x + v1 * d_d_y;
#pragma unroll
        for (int k = 0; k < 4; k++) rec[REC_HCD + k] = d_C_x[k] * v0 + d_C_y[k] * v1;
      }
    }
  }
  // energy partials: one per run of LIN_RES_PER_BLOCK residuals, summed in order (the eight-lane kernel's workgroup partials)
  double* const s_e1 = reinterpret_cast<double*>(patch);   // LIN_THREADS doubles in the room of the patches (every lane is done with its own: the barrier)
  __syncthreads();
  s_e1[threadIdx.x] = myE;
  __syncthreads();
  if ((threadIdx.x & (LIN_RES_PER_BLOCK - 1)) == 0) {
    const int run = blockIdx.x * (LIN_THREADS / LIN_RES_PER_BLOCK) + threadIdx.x / LIN_RES_PER_BLOCK;
    if (run < nruns) {
      double sum = 0;
      for (int k = 0; k < LIN_RES_PER_BLOCK; k++) sum += s_e1[threadIdx.x + k];
      __hip_atomic_store(D.epart + run, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // loadSateBackup, point part: the lane of a point's first residual restores it
  if (use_backup && ri < W.R) {
    const int pi = Rs.point[ri];
    if (P.res_begin[pi] == ri) { const float bk = P.idepth_backup[pi]; P.idepth[pi] = bk; P.idepth_zero[pi] = bk; }
  }
  if (D.mode < 0) return;
  __shared__ int s_last1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) s_last1 = __hip_atomic_fetch_add(&D.ctl->cnt_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(nblocks - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (threadIdx.x == 0) __hip_atomic_store(&D.ctl->cnt_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (D.mode == 3) baPackBlock(D, nruns);
  else baDecideBlock<true>(D, nruns, t_kernel0, reinterpret_cast<unsigned int*>(patch));   // (every lane is done with its patch: behind the barrier above)
}
template <int MF>
__global__ void __launch_bounds__(LIN_THREADS) k_ba_linearize(const BAWindow W, const BAPoints P, const BARes Rs, const BAPrecalc* __restrict__ pre,
                                                               const FrameStore fs, float* __restrict__ fullJ,
                                                               const unsigned char* __restrict__ pt_mask, const BADecide D, const int gate, const int use_backup,
                                                               const BAPreDynT<MF> T, const int use_dyn, const ResubArgsT<MF> X, const int do_resub) {
  baLinearizeBody(W, P, Rs, pre, fs, fullJ, pt_mask, D, gate, use_backup, T.v, use_dyn, X.xc, X.xAd, do_resub, (int)gridDim.x);
}

// the decisions of a linearisation whose points are sharded over ranks: one workgroup over the all-gathered records
__global__ void __launch_bounds__(256) k_ba_decide_global(const BADecide D) { baDecideBlock(D, 0, wall_clock64()); }
// the all-reduced system -> host-coherent memory, then the chain's ticket
__global__ void __launch_bounds__(1024) k_ba_publish_sys(const double* __restrict__ src, double* __restrict__ dst, const int count, BAHostRes* __restrict__ host,
                                                          const unsigned int ticket) {
  for (int i = threadIdx.x; i < count; i += 1024) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&host->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// applyRes(true) for every active residual (Residuals.cpp:306-328): flips the applied-record selector.
// mark_removed: the tail of FullSystem::linearizeAll(true) (FullSystemOptimize.cpp:176-212) — a residual that is not active after this applyRes is deleted from the graph
__device__ __forceinline__ void baApplyBody(const int R, const BARes& Rs, const unsigned char* __restrict__ pt_mask, const int mark_removed);
__global__ void __launch_bounds__(256) k_ba_apply(const int R, const BARes Rs, const unsigned char* __restrict__ pt_mask, const int mark_removed) {
  baApplyBody(R, Rs, pt_mask, mark_removed);
}
__device__ __forceinline__ void baApplyBody(const int R, const BARes& Rs, const unsigned char* __restrict__ pt_mask, const int mark_removed) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= R) return;
  if (pt_mask && !pt_mask[Rs.point[ri]]) return;
  if (Rs.lin && !pt_mask && Rs.lin[ri]) return;   // not in activeResiduals: applyRes_Reductor and the removal of linearizeAll(true) never see it
  if (Rs.state[ri] != BA_OOB) {  // can never go back from OOB
    const int ns = Rs.newState[ri];
    if (ns == BA_IN) { Rs.active[ri] = 1; Rs.which[ri] ^= 1; }
    else Rs.active[ri] = 0;
    Rs.state[ri] = (unsigned char)ns;
    Rs.energy[ri] = Rs.newEnergy[ri];
  }
  if (mark_removed && !Rs.active[ri]) { Rs.removed[ri] = 1; Rs.state[ri] = BA_OOB; Rs.energy[ri] = 0.f; Rs.newEnergy[ri] = 0.f; }
}
// PointFrameResidual::resetOOB of every residual still in the graph (FullSystemOptimize.cpp:431-448)
__global__ void __launch_bounds__(256) k_ba_reset_oob(const int R, const BARes Rs) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= R) return;
  if (Rs.lin && Rs.lin[ri]) return;   // FullSystemOptimize.cpp:440-446: a linearised residual is not reset
  const bool gone = Rs.removed[ri] != 0;
  Rs.state[ri] = gone ? BA_OOB : BA_IN;
  Rs.newState[ri] = gone ? BA_OOB : BA_OUTLIER;
  Rs.energy[ri] = 0.f; Rs.newEnergy[ri] = 0.f;
}

// ------------------------------------------------------------------------------------------------ per-point sums
// Hdd_accAF, bd_accAF, Hcd_accAF (sequential over the point's residuals) and the head of AccumulatedSCHessianSSE::addPoint:
// HdiF, bdSumF (AccumulatedSCHessian.cpp:36-54).
// backup != 0 fuses the point part of backupState (FullSystemOptimize.cpp:320-352): idepth_backup = idepth
// Eight lanes per point: lane q fetches the six contributions of the point's q-th residual (all loads of a point in flight together — a point
// has at most F-1 <= 7 residuals), the sums are then formed in residual order by DPP row shifts in the leading lane, exactly like the
// reference's sequential loop (an inactive residual contributes +0.0f, which leaves an fp32 sum unchanged).  The kernel is latency-bound
// (2000 points): what matters is the length of the dependent load chain, which is res_begin -> which -> record here.
// running sum s (leading lane) + the values of lanes 0..7 of the group, left to right
__device__ __forceinline__ float seqAdd8(float s, const float v) {
  s = s + v;
#pragma unroll
  for (int k = 1; k < 8; k++) s = s + dppShl(v, k);
  return s;
}
#define PT_GROUPS_PER_BLOCK 32
// apply != 0 fuses applyRes_Reductor(true) (FullSystemOptimize.cpp:91-95, Residuals.cpp:306-328) for the point's residuals: every residual
// belongs to exactly one lane of one group.  gate: see BACtl.
// host_backup (may be NULL): host-coherent mirror of idepth_backup — doStepFromBackup's canbreak test sums |idepth_backup| over the points on the host, in
// the reference's order (FullSystemOptimize.cpp:269-291), when the GTSAM branch can end the loop early
__device__ __forceinline__ void baPointSumsBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const int backup, const int apply, const BACtl* __restrict__ ctl,
                                                const int gate, float* __restrict__ host_backup);
__global__ void __launch_bounds__(256) k_ba_point_sums(const BAWindow W, const BAPoints P, const BARes Rs, const int backup, const int apply, const BACtl* __restrict__ ctl,
                                                        const int gate, float* __restrict__ host_backup) {
  baPointSumsBody(W, P, Rs, backup, apply, ctl, gate, host_backup);
}
__device__ __forceinline__ void baPointSumsBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const int backup, const int apply, const BACtl* __restrict__ ctl,
                                                const int gate, float* __restrict__ host_backup) {
  if (baGateClosed(ctl, gate)) return;
  const int pi = blockIdx.x * PT_GROUPS_PER_BLOCK + (threadIdx.x >> 3), q = threadIdx.x & 7;
  if (pi >= W.N) return;   // group-uniform
  const bool lead = q == 0;
  const int r0 = P.res_begin[pi], r1 = P.res_begin[pi + 1];
  const float id = P.idepth[pi], idz = P.idepth_zero[pi], prior = P.priorF[pi];
  if (backup && lead) { P.idepth_backup[pi] = id; if (host_backup) __hip_atomic_store(host_backup + pi, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  float Hdd = 0, bd = 0, Hcd0 = 0, Hcd1 = 0, Hcd2 = 0, Hcd3 = 0;
  int ngood = 0;
  const int grp = (threadIdx.x & 63) & ~7;
  for (int rb = r0; rb < r1; rb += 8) {
    const int ri = min(rb + q, r1 - 1);
    const bool mine = rb + q < r1;
    int isAct = Rs.active[ri], wh = Rs.which[ri];
    const bool isLin = Rs.lin && Rs.lin[ri] != 0;   // addPoint<0> leaves it to addPoint<1> (AccumulatedTopHessian.cpp:52-53); the Schur side counts it (isActive())
    if (apply && mine && !isLin && Rs.state[ri] != BA_OOB) {   // can never go back from OOB
      const int ns = Rs.newState[ri];
      if (ns == BA_IN) { isAct = 1; wh ^= 1; Rs.which[ri] = (unsigned char)wh; } else isAct = 0;
      Rs.active[ri] = (unsigned char)isAct;
      Rs.state[ri] = (unsigned char)ns;
      Rs.energy[ri] = Rs.newEnergy[ri];
    }
    const bool good = mine && isAct != 0, act = good && !isLin;
    const float* __restrict__ rec = baRec(Rs, wh) + (size_t)ri * REC_FLOATS;
    const float v0 = act ? rec[REC_BD] : 0.0f, v1 = act ? rec[REC_HDD] : 0.0f;
    const float h0 = act ? rec[REC_HCD + 0] : 0.0f, h1 = act ? rec[REC_HCD + 1] : 0.0f, h2 = act ? rec[REC_HCD + 2] : 0.0f, h3 = act ? rec[REC_HCD + 3] : 0.0f;
    bd = seqAdd8(bd, v0); Hdd = seqAdd8(Hdd, v1);
    Hcd0 = seqAdd8(Hcd0, h0); Hcd1 = seqAdd8(Hcd1, h1); Hcd2 = seqAdd8(Hcd2, h2); Hcd3 = seqAdd8(Hcd3, h3);
    ngood += __popcll((__ballot(good) >> grp) & 0xFFull);
  }
  if (!lead) return;
  P.Hdd[pi] = Hdd; P.bd[pi] = bd;
  float HddL = 0.0f, bdL = 0.0f;
  if (P.lHdd) {   // Hdd_accLF, bd_accLF, Hcd_accLF of the residuals kept linearised (k_ba_lin_point_sums)
    HddL = P.lHdd[pi]; bdL = P.lbd[pi];
    P.HcdAF[4 * pi + 0] = Hcd0; P.HcdAF[4 * pi + 1] = Hcd1; P.HcdAF[4 * pi + 2] = Hcd2; P.HcdAF[4 * pi + 3] = Hcd3;
    Hcd0 = Hcd0 + P.lHcd[4 * pi + 0]; Hcd1 = Hcd1 + P.lHcd[4 * pi + 1]; Hcd2 = Hcd2 + P.lHcd[4 * pi + 2]; Hcd3 = Hcd3 + P.lHcd[4 * pi + 3];
  }
  P.Hcd[4 * pi + 0] = Hcd0; P.Hcd[4 * pi + 1] = Hcd1; P.Hcd[4 * pi + 2] = Hcd2; P.Hcd[4 * pi + 3] = Hcd3;
  if (ngood == 0) { P.HdiF[pi] = 0; P.bdSumF[pi] = 0; P.idepth_hessian[pi] = 0; return; }
  float H = Hdd + HddL + prior;
  if (H < 1e-10) H = 1e-10;
  P.idepth_hessian[pi] = H;
  P.HdiF[pi] = 1.0 / H;
  const float deltaF = id - idz;
  P.bdSumF[pi] = (bd + bdL) + prior * deltaF;  // shiftPriorToZero = true
}

// ------------------------------------------------------------------------------------------------ point marginalisation
// EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:76-106) for the active residuals of the points to marginalise, and the
// record AccumulatedTopHessianSSE::addPoint<2> consumes: identical to the applied record except that the inner products with the
// residual use res_toZeroF (AccumulatedTopHessian.cpp:70-71,103-113,131).  fullJ: the 74-float RawResidualJacobian of the applied
// linearisation ([0,8) resF, [8,20) Jpdxi, [20,28) Jpdc, 28-29 Jpdd, [30,46) JIdx, [46,62) JabF).
__global__ void __launch_bounds__(256) k_ba_fix_linearization(const BAWindow W, const BAPoints P, const BARes Rs, const float* __restrict__ fullJ,
                                                               const unsigned char* __restrict__ decision, const float* __restrict__ adHTdeltaF /* F*F x 8, h + F*t */,
                                                               const float4 cDeltaF, float* __restrict__ margRec, unsigned char* __restrict__ margActive,
                                                               float* __restrict__ res_toZeroF /* R x 8 or NULL */) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= W.R) return;
  const int pi = Rs.point[ri];
  const bool on = decision[pi] == 1 && Rs.active[ri] != 0;
  margActive[ri] = on ? 1 : 0;
  if (!on) return;
  const float* __restrict__ J = fullJ + (size_t)ri * 74;
  const float* __restrict__ dp = adHTdeltaF + (size_t)(P.host[pi] + W.F * Rs.target[ri]) * 8;
  const float dd = P.idepth[pi] - P.idepth_zero[pi];
  const float cd[4] = {cDeltaF.x, cDeltaF.y, cDeltaF.z, cDeltaF.w};
  float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) { sx += J[8 + i] * dp[i]; sy += J[14 + i] * dp[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { cx += J[20 + i] * cd[i]; cy += J[24 + i] * cd[i]; }
  const float Jp_delta_x = sx + cx + J[28] * dd, Jp_delta_y = sy + cy + J[29] * dd;
  float JIr0 = 0, JIr1 = 0, Jar0 = 0, Jar1 = 0, rr = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float rtz = J[i];
    rtz = rtz - J[30 + i] * Jp_delta_x; rtz = rtz - J[38 + i] * Jp_delta_y;
    rtz = rtz - J[46 + i] * dp[6]; rtz = rtz - J[54 + i] * dp[7];
    if (res_toZeroF) res_toZeroF[(size_t)ri * 8 + i] = rtz;
    JIr0 += rtz * J[30 + i]; JIr1 += rtz * J[38 + i]; Jar0 += rtz * J[46 + i]; Jar1 += rtz * J[54 + i]; rr += rtz * rtz;
  }
  const float* __restrict__ src = baRec(Rs, Rs.which[ri]) + (size_t)ri * REC_FLOATS;
  float* __restrict__ dst = margRec + (size_t)ri * REC_FLOATS;
  for (int k = 0; k < REC_FLOATS; k++) dst[k] = src[k];
  dst[REC_JI_R + 0] = JIr0; dst[REC_JI_R + 1] = JIr1; dst[REC_JAB_R + 0] = Jar0; dst[REC_JAB_R + 1] = Jar1; dst[REC_RR] = rr;
  dst[REC_BD] = JIr0 * J[28] + JIr1 * J[29];
}
// decision of flagPointsForRemoval for the candidates: marginalise (1) if idepth_hessian > setting_minIdepthH_marg, else drop (2)
__global__ void __launch_bounds__(256) k_ba_marg_decide(const int N, const unsigned char* __restrict__ cand, const float* __restrict__ idepth_hessian,
                                                         const float minIdepthH_marg, unsigned char* __restrict__ decision) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi < N) decision[pi] = cand[pi] ? (idepth_hessian[pi] > minIdepthH_marg ? 1 : 2) : 0;
}
// addPoint<2>'s per-point sums (Hdd_accLF, bd_accLF, Hcd_accLF; accAF = 0) and the head of AccumulatedSCHessianSSE::addPoint(p, false)
// with priorF * setting_idepthFixPriorMargFac; points that are not marginalised get HdiF = 0 (skipped by every accumulator)
__global__ void __launch_bounds__(256) k_ba_marg_point_sums(const BAWindow W, const BAPoints P, const BARes Rs, const float* __restrict__ margRec,
                                                             const unsigned char* __restrict__ margActive, const unsigned char* __restrict__ decision,
                                                             const float priorMargFac, float* __restrict__ HdiF, float* __restrict__ bdSumF, float* __restrict__ Hcd4) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= W.N) return;
  float Hdd = 0, bd = 0, Hcd[4] = {0, 0, 0, 0};
  int ngood = 0;
  if (decision[pi] == 1)
    for (int ri = P.res_begin[pi]; ri < P.res_begin[pi + 1]; ri++) {
      if (!margActive[ri]) continue;
      const float* __restrict__ rec = margRec + (size_t)ri * REC_FLOATS;
      bd += rec[REC_BD];
      Hdd += rec[REC_HDD];
#pragma unroll
      for (int k = 0; k < 4; k++) Hcd[k] += rec[REC_HCD + k];
      ngood++;
    }
#pragma unroll
  for (int k = 0; k < 4; k++) Hcd4[4 * pi + k] = 0.0f + Hcd[k];
  if (ngood == 0) { HdiF[pi] = 0; bdSumF[pi] = 0; return; }
  float H = Hdd + 0.0f + P.priorF[pi] * priorMargFac;
  if (H < 1e-10) H = 1e-10;
  HdiF[pi] = 1.0 / H;
  bdSumF[pi] = 0.0f + bd;   // shiftPriorToZero = false
}

// ------------------------------------------------------------------------------------------------ residuals kept linearised (addPoint<1>)
// The reference only ever linearises a residual on its way into the marginalisation prior (FullSystem.cpp:836-849), but EnergyFunctional carries the general case: a
// residual with isLinearized set stays out of activeResiduals, and its frozen Jacobian + res_toZeroF enter every system through accumulateLF_MT (addPoint<1>,
// AccumulatedTopHessian.cpp:84-98) and the energy through calcLEnergyPt (EnergyFunctional.cpp:349-409).  dmvio_hip_ba_fix_linearization builds that case on the resident
// graph; it is off every fast path (three accumulation passes per system instead of one).
// EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:85-113) for the active, not yet linearised residuals with mask != 0: res_toZeroF from the applied Jacobian
// (fullJ: the last linearisation, which the caller guarantees was applied) and the deltas of the current state; the applied record is frozen into linRec.
__global__ void __launch_bounds__(256) k_ba_lin_fix(const BAWindow W, const BAPoints P, const BARes Rs, const float* __restrict__ fullJ, const unsigned char* __restrict__ mask,
                                                     const float* __restrict__ adHTdeltaF /* F*F x 8, h + F*t */, const float4 cDeltaF, unsigned char* __restrict__ lin,
                                                     float* __restrict__ res_toZeroF /* R x 8 */, float* __restrict__ linRec) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= W.R) return;
  if (!mask[ri] || lin[ri] || !Rs.active[ri] || Rs.removed[ri]) return;
  const int pi = Rs.point[ri];
  const float* __restrict__ J = fullJ + (size_t)ri * 74;
  const float* __restrict__ dp = adHTdeltaF + (size_t)(P.host[pi] + W.F * Rs.target[ri]) * 8;
  const float dd = P.idepth[pi] - P.idepth_zero[pi];
  const float cd[4] = {cDeltaF.x, cDeltaF.y, cDeltaF.z, cDeltaF.w};
  float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) { sx += J[8 + i] * dp[i]; sy += J[14 + i] * dp[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { cx += J[20 + i] * cd[i]; cy += J[24 + i] * cd[i]; }
  const float Jp_delta_x = sx + cx + J[28] * dd, Jp_delta_y = sy + cy + J[29] * dd;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float rtz = J[i];
    rtz = rtz - J[30 + i] * Jp_delta_x; rtz = rtz - J[38 + i] * Jp_delta_y;
    rtz = rtz - J[46 + i] * dp[6]; rtz = rtz - J[54 + i] * dp[7];
    res_toZeroF[(size_t)ri * 8 + i] = rtz;
  }
  const float* __restrict__ src = baRec(Rs, Rs.which[ri]) + (size_t)ri * REC_FLOATS;
  float* __restrict__ dst = linRec + (size_t)ri * REC_FLOATS;
  for (int k = 0; k < REC_FLOATS; k++) dst[k] = src[k];
  lin[ri] = 1;
}
// A residual that ARRIVES linearised (dmvio_hip_ba_set_linearized_residuals: EFResidual::isLinearized with its RawResidualJacobian and res_toZeroF, EnergyFunctionalStructs.h:
// 63-87): what k_ba_lin_fix leaves behind for one linearised on the resident graph — the frozen Jacobian row, res_toZeroF, the Jacobian part of the applied record (formed
// from J by the linearisation's own expressions, takeDataF EnergyFunctionalStructs.cpp:39-49 included) — plus the state an active residual of the reference has
// (ResState::IN, in the energy functional).  idx: the n flagged residuals; packed: n x 82 floats [J (74) | res_toZeroF (8)].
__global__ void __launch_bounds__(256) k_ba_lin_import(const int n, const int* __restrict__ idx, const float* __restrict__ packed, const BARes Rs, float* __restrict__ fullJ,
                                                        float* __restrict__ res_toZeroF, float* __restrict__ linRec, unsigned char* __restrict__ lin) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int ri = idx[k];
  const float* __restrict__ J = packed + (size_t)k * 82;
  float* __restrict__ fj = fullJ + (size_t)ri * 74;
  for (int i = 0; i < 74; i++) fj[i] = J[i];
  for (int i = 0; i < 8; i++) res_toZeroF[(size_t)ri * 8 + i] = J[74 + i];
  float rec[REC_FLOATS];
  const float d_d_x = J[28], d_d_y = J[29];
  const float JI00 = J[62], JI10 = J[63], JI11 = J[65];
  const float Ja00 = J[66], Ja01 = J[67], Ja10 = J[68], Ja11 = J[69];
#pragma unroll
  for (int i = 0; i < 4; i++) { rec[REC_JPDC0 + i] = J[20 + i]; rec[REC_JPDC1 + i] = J[24 + i]; }
#pragma unroll
  for (int i = 0; i < 6; i++) { rec[REC_JPDXI0 + i] = J[8 + i]; rec[REC_JPDXI1 + i] = J[14 + i]; }
  rec[REC_JIDX2 + 0] = JI00; rec[REC_JIDX2 + 1] = JI10; rec[REC_JIDX2 + 2] = JI11;
  rec[REC_JABJIDX + 0] = Ja00; rec[REC_JABJIDX + 1] = Ja01; rec[REC_JABJIDX + 2] = Ja10; rec[REC_JABJIDX + 3] = Ja11;
  rec[REC_JAB2 + 0] = J[70]; rec[REC_JAB2 + 1] = J[71]; rec[REC_JAB2 + 2] = J[73];
  // the inner products with resF in pattern order, as the linearisation sums them (the applied record of the linearisation this residual was frozen at; the L pass works on
  // linRec, where k_ba_lin_records replaces them by those with resApprox)
  float JIr0 = 0.f, JIr1 = 0.f, Jar0 = 0.f, Jar1 = 0.f, rr = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const float resF = J[i];
    JIr0 = JIr0 + resF * J[30 + i]; JIr1 = JIr1 + resF * J[38 + i]; Jar0 = Jar0 + resF * J[46 + i]; Jar1 = Jar1 + resF * J[54 + i]; rr = rr + resF * resF;
  }
  rec[REC_JI_R + 0] = JIr0; rec[REC_JI_R + 1] = JIr1; rec[REC_JAB_R + 0] = Jar0; rec[REC_JAB_R + 1] = Jar1; rec[REC_RR] = rr;
  rec[REC_JPDD + 0] = d_d_x; rec[REC_JPDD + 1] = d_d_y;
  const float v0 = JI00 * d_d_x + JI10 * d_d_y, v1 = JI10 * d_d_x + JI11 * d_d_y;
#pragma unroll
  for (int i = 0; i < 6; i++) rec[REC_JPJD + i] = J[8 + i] * v0 + J[14 + i] * v1;
  rec[REC_JPJD + 6] = Ja00 * d_d_x + Ja01 * d_d_y;
  rec[REC_JPJD + 7] = Ja10 * d_d_x + Ja11 * d_d_y;
  rec[REC_BD] = JIr0 * d_d_x + JIr1 * d_d_y;
  rec[REC_HDD] = v0 * d_d_x + v1 * d_d_y;
#pragma unroll
  for (int i = 0; i < 4; i++) rec[REC_HCD + i] = J[20 + i] * v0 + J[24 + i] * v1;
  rec[REC_PAD] = 0.f;
  // the applied record (both buffers: whichever the selector points at) — the Schur pass and the points' back-substitution read JpJdF from it — and the frozen copy
  float* __restrict__ d0 = Rs.rec[0] + (size_t)ri * REC_FLOATS;
  float* __restrict__ d1 = Rs.rec[1] + (size_t)ri * REC_FLOATS;
  float* __restrict__ d2 = linRec + (size_t)ri * REC_FLOATS;
#pragma unroll
  for (int i = 0; i < REC_FLOATS; i++) { d0[i] = rec[i]; d1[i] = rec[i]; d2[i] = rec[i]; }
  lin[ri] = 1;
  Rs.active[ri] = 1; Rs.removed[ri] = 0; Rs.state[ri] = BA_IN; Rs.newState[ri] = BA_IN; Rs.energy[ri] = 0.f; Rs.newEnergy[ri] = 0.f;
}
// The record addPoint<1> consumes, at the deltas of the current state: resApprox = res_toZeroF + [JI Jp | Jab] delta (AccumulatedTopHessian.cpp:84-98), its inner products
// with the frozen Jacobian (:103-113) and the point's bd contribution (:132).  Also the two activity views of the three-pass accumulation: the A pass sees the active
// residuals that are NOT linearised (addPoint<0>), the L pass those that are.
__device__ __forceinline__ void baLinRecordsBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const float* __restrict__ fullJ, const unsigned char* __restrict__ lin,
                                                 const float* __restrict__ res_toZeroF, const float* __restrict__ adHTdeltaF, const float4 cDeltaF,
                                                 float* __restrict__ linRec, unsigned char* __restrict__ linActive, unsigned char* __restrict__ topActive);
__global__ void __launch_bounds__(256) k_ba_lin_records(const BAWindow W, const BAPoints P, const BARes Rs, const float* __restrict__ fullJ, const unsigned char* __restrict__ lin,
                                                         const float* __restrict__ res_toZeroF, const float* __restrict__ adHTdeltaF, const float4 cDeltaF,
                                                         float* __restrict__ linRec, unsigned char* __restrict__ linActive, unsigned char* __restrict__ topActive) {
  baLinRecordsBody(W, P, Rs, fullJ, lin, res_toZeroF, adHTdeltaF, cDeltaF, linRec, linActive, topActive);
}
__device__ __forceinline__ void baLinRecordsBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const float* __restrict__ fullJ, const unsigned char* __restrict__ lin,
                                                 const float* __restrict__ res_toZeroF, const float* __restrict__ adHTdeltaF, const float4 cDeltaF,
                                                 float* __restrict__ linRec, unsigned char* __restrict__ linActive, unsigned char* __restrict__ topActive) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= W.R) return;
  const bool act = Rs.active[ri] != 0, isLin = lin[ri] != 0;
  linActive[ri] = (act && isLin) ? 1 : 0;
  topActive[ri] = (act && !isLin) ? 1 : 0;
  if (!(act && isLin)) return;
  const int pi = Rs.point[ri];
  const float* __restrict__ J = fullJ + (size_t)ri * 74;
  const float* __restrict__ dp = adHTdeltaF + (size_t)(P.host[pi] + W.F * Rs.target[ri]) * 8;
  const float dd = P.idepth[pi] - P.idepth_zero[pi];
  const float cd[4] = {cDeltaF.x, cDeltaF.y, cDeltaF.z, cDeltaF.w};
  float sx = 0, sy = 0, cx = 0, cy = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) { sx += J[8 + i] * dp[i]; sy += J[14 + i] * dp[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { cx += J[20 + i] * cd[i]; cy += J[24 + i] * cd[i]; }
  const float Jp_delta_x = sx + cx + J[28] * dd, Jp_delta_y = sy + cy + J[29] * dd;
  float JIr0 = 0, JIr1 = 0, Jar0 = 0, Jar1 = 0, rr = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float ra = res_toZeroF[(size_t)ri * 8 + i];
    ra = ra + J[30 + i] * Jp_delta_x; ra = ra + J[38 + i] * Jp_delta_y;
    ra = ra + J[46 + i] * dp[6]; ra = ra + J[54 + i] * dp[7];
    JIr0 += ra * J[30 + i]; JIr1 += ra * J[38 + i]; Jar0 += ra * J[46 + i]; Jar1 += ra * J[54 + i]; rr += ra * ra;
  }
  float* __restrict__ dst = linRec + (size_t)ri * REC_FLOATS;
  dst[REC_JI_R + 0] = JIr0; dst[REC_JI_R + 1] = JIr1; dst[REC_JAB_R + 0] = Jar0; dst[REC_JAB_R + 1] = Jar1; dst[REC_RR] = rr;
  dst[REC_BD] = JIr0 * J[28] + JIr1 * J[29];
}
// Hdd_accLF, bd_accLF, Hcd_accLF (AccumulatedTopHessian.cpp:131-143, mode 1): sequential over the point's linearised active residuals
__device__ __forceinline__ void baLinPointSumsBody(const BAWindow& W, const BAPoints& P, const float* __restrict__ linRec, const unsigned char* __restrict__ linActive,
                                                   float* __restrict__ lHdd, float* __restrict__ lbd, float* __restrict__ lHcd4);
__global__ void __launch_bounds__(256) k_ba_lin_point_sums(const BAWindow W, const BAPoints P, const float* __restrict__ linRec, const unsigned char* __restrict__ linActive,
                                                            float* __restrict__ lHdd, float* __restrict__ lbd, float* __restrict__ lHcd4) {
  baLinPointSumsBody(W, P, linRec, linActive, lHdd, lbd, lHcd4);
}
__device__ __forceinline__ void baLinPointSumsBody(const BAWindow& W, const BAPoints& P, const float* __restrict__ linRec, const unsigned char* __restrict__ linActive,
                                                   float* __restrict__ lHdd, float* __restrict__ lbd, float* __restrict__ lHcd4) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= W.N) return;
  float Hdd = 0, bd = 0, Hcd[4] = {0, 0, 0, 0};
  for (int ri = P.res_begin[pi]; ri < P.res_begin[pi + 1]; ri++) {
    if (!linActive[ri]) continue;
    const float* __restrict__ rec = linRec + (size_t)ri * REC_FLOATS;
    bd += rec[REC_BD];
    Hdd += rec[REC_HDD];
#pragma unroll
    for (int k = 0; k < 4; k++) Hcd[k] += rec[REC_HCD + k];
  }
  lHdd[pi] = Hdd; lbd[pi] = bd;
#pragma unroll
  for (int k = 0; k < 4; k++) lHcd4[4 * pi + k] = Hcd[k];
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

// Tile addition in member order.  Rows of inactive members (and rows past the end of the tile) hold exact zeros and x + 0 == x,
// so the owner adds every row of a range back to back: 16 independent LDS reads, then 16 dependent adds (a taken branch costs a
// lone wavefront ~15 ns, a dependent add ~2 ns — measured, tools/chaintest.hip).
template <class F>
__device__ __forceinline__ void addRange(float& d, int lo, const int hi, F contrib) {
  for (; lo + 16 <= hi; lo += 16) {
    float c[16];
#pragma unroll
    for (int q = 0; q < 16; q++) c[q] = contrib(lo + q);
#pragma unroll
    for (int q = 0; q < 16; q++) d += c[q];
  }
  for (; lo < hi; lo++) d += contrib(lo);
}
// n1u: the accumulator's numIn1 counter (identical for every element, so every thread mirrors it).  A tile either fits below the
// 1000-entry shift-up threshold, or the (1001 - n1u)-th ACTIVE member of the tile triggers shiftUp: `cross` is its row.
// ADD_FIRST: the element is added before the counters advance (AccumulatorApprox::update's 10x10 part, AccumulatorXX/X::update);
// otherwise after (updateTopRight / updateBotRight run after update() has advanced them).  rows = cnt rounded up to 16 (<= tile).
template <bool ADD_FIRST, class F>
__device__ __forceinline__ void addTile(Acc3& acc, const int cnt, const int nact, const int cross, F contrib) {
  if (acc.tileFits(nact)) {
    addRange(acc.d, 0, (cnt + 15) & ~15, contrib);
    acc.n1 += nact;
  } else {
    const int split = ADD_FIRST ? cross + 1 : cross, used = 1001 - acc.n1;
    addRange(acc.d, 0, split, contrib);
    acc.n1 = 1000; acc.bump();   // numIn1 reaches 1001 -> shiftUp
    addRange(acc.d, split, cnt, contrib);
    acc.n1 = nact - used;
  }
}
// row of the target-th (1-based) active member of a workgroup-wide tile with one flag per thread (stride = threads per member);
// s_w: 5 ints of LDS.  Called by all threads (contains barriers).
__device__ __forceinline__ int blockFindActive(const bool flag, const int row, const int target, int* s_w) {
  const unsigned long long mask = __ballot(flag);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_w[wave] = __popcll(mask);
  if (threadIdx.x == 0) s_w[4] = 0;
  __syncthreads();
  int before = __popcll(mask & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) before += s_w[w];
  if (flag && before + 1 == target) s_w[4] = row;
  __syncthreads();
  return s_w[4];
}

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
// Roles of the owner threads of a (host,target) workgroup — one role per wavefront, so the tile loops do not diverge:
//   role 0 (wave 0, lanes 0..54): element of the upper triangle of the 10x10 [calib4 | pose6] block,
//   role 1 (wave 1, lanes 0..29): TopRight 10x3,   role 2 (wave 3, lanes 0..5): BotRight,   role 3 (wave 2, lanes 0..39): accE / accEB.
// Every contribution is a fixed expression of <= 4 thread-specific columns of the staged member row (offsets o0..o3).
struct HTRole { int role, o0, o1, o2, o3, out; };
__device__ __forceinline__ int topX(const int i) { return i < 4 ? REC_JPDC0 + i : REC_JPDXI0 + (i - 4); }   // x[i] = (i < 4) ? Jpdc0[i] : Jpdxi0[i-4]
__device__ __forceinline__ int topY(const int i) { return i < 4 ? REC_JPDC1 + i : REC_JPDXI1 + (i - 4); }
__device__ __forceinline__ HTRole htRole(const int tid) {
  HTRole R; R.role = -1; R.o0 = R.o1 = R.o2 = R.o3 = 0; R.out = 0;
  const int wave = tid >> 6, lane = tid & 63;
  if (wave == 0 && lane < 55) {
    int off = 0, r = 0;
    while (lane >= off + (10 - r)) { off += 10 - r; r++; }
    const int c = r + (lane - off);
    R.role = 0; R.o0 = topX(r); R.o1 = topX(c); R.o2 = topY(r); R.o3 = topY(c); R.out = lane;
  } else if (wave == 1 && lane < 30) {
    const int r = lane / 3, c = lane % 3;
    // TR col 0: (JabJIdx00, JabJIdx01), col 1: (JabJIdx10, JabJIdx11), col 2: (JI_r0, JI_r1)
    R.role = 1; R.o0 = topX(r); R.o1 = topY(r);
    R.o2 = c == 0 ? REC_JABJIDX + 0 : (c == 1 ? REC_JABJIDX + 2 : REC_JI_R + 0);
    R.o3 = c == 0 ? REC_JABJIDX + 1 : (c == 1 ? REC_JABJIDX + 3 : REC_JI_R + 1);
    R.out = 55 + lane;
  } else if (wave == 3 && lane < 6) {
    // a00 = Jab2_00, a01 = Jab2_01, a02 = Jab_r0, a11 = Jab2_11, a12 = Jab_r1, a22 = rr
    const int col[6] = {REC_JAB2 + 0, REC_JAB2 + 1, REC_JAB_R + 0, REC_JAB2 + 2, REC_JAB_R + 1, REC_RR};
    R.role = 2; R.o0 = col[lane]; R.out = 85 + lane;
  } else if (wave == 2 && lane < 40) {
    // accE (8x4): (HdiF * JpJd[i]) * Hcd[k];  accEB (8): (HdiF*bdSumF) * JpJd[i] — written as (..)*1 with the active flag (exact)
    R.role = 3; R.out = lane;
    if (lane < 32) { R.o0 = 49; R.o1 = REC_JPJD + (lane >> 2); R.o2 = 45 + (lane & 3); }
    else { R.o0 = 50; R.o1 = REC_JPJD + (lane - 32); R.o2 = 51; }
  }
  return R;
}
template <int ROLE>
__device__ __forceinline__ float htContribution(const float* q, const HTRole& R) {
  if (ROLE == 0) {
    const float xr = q[R.o0], xc = q[R.o1], yr = q[R.o2], yc = q[R.o3];
    const float a = q[REC_JIDX2], bb = q[REC_JIDX2 + 1], cc = q[REC_JIDX2 + 2];
    return a * xc * xr + cc * yc * yr + bb * (xc * yr + yc * xr);
  } else if (ROLE == 1) {
    return q[R.o0] * q[R.o2] + q[R.o1] * q[R.o3];
  } else if (ROLE == 2) {
    return q[R.o0];
  } else {
    return (q[R.o0] * q[R.o1]) * q[R.o2];
  }
}

// wave-level LDS hand-off: the 64 lanes of a wavefront run in lockstep, a fence orders the LDS traffic (no workgroup barrier)
__device__ __forceinline__ void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup per (host,target) bucket accumulates BOTH the 13x13 top block (owners: threads 0..90) and the Schur terms
// accE (8x4) / accEB (8) (owners: threads 128..167) — the two walk the same member list in the same order
// (AccumulatedTopHessian.cpp:34-146, AccumulatedSCHessian.cpp:56-78), so the records are staged once.
// Staging is software-pipelined three tiles deep (member index -> activity / buffer / point -> record) so that no global
// load is waited for in the iteration that issued it; the owners add a tile while the next one is in flight.
#define HT_TILE 64
#define HT_STRIDE 53   // [0,45) record, [45,49) Hcd, 49 HdiF, 50 HdiF*bdSumF, 51 active flag
struct HTMeta { int act, which, pi; };
struct HTFetch { float r[12]; float hc, hdi, bds; };

// all staging loads are UNCONDITIONAL (indices clamped into the bucket, validity applied when the tile is written to LDS): no
// branch, hence no s_waitcnt, sits between issuing a load and the iteration that consumes it
__device__ __forceinline__ HTMeta htLoadMeta(const BARes& Rs, const int ri) {
  HTMeta m; m.act = Rs.active[ri] != 0; m.which = Rs.which[ri]; m.pi = Rs.point[ri];
  return m;
}
__device__ __forceinline__ void htLoadRec(const BARes& Rs, const BAPoints& P, const int ri, const HTMeta m, const int part, HTFetch& f) {
  // three 16-byte loads per lane (floats [16 k + 4 part, + 4), k = 0..2: columns 0..47 of the record; the 208-byte records are 16-byte aligned) — the four lanes of a member
  // read 64 contiguous bytes per instruction.  Twelve 4-byte loads per lane (column part + 4 k) cost four times the instructions and vector-L1 tag look-ups for the same
  // lines, and the look-ups are what bounds this kernel (profiles/r06_counters_ba.md)
  const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(baRec(Rs, m.which) + (size_t)ri * REC_FLOATS) + part;
#pragma unroll
  for (int k = 0; k < 3; k++) { const float4 q = rec4[4 * k]; f.r[4 * k] = q.x; f.r[4 * k + 1] = q.y; f.r[4 * k + 2] = q.z; f.r[4 * k + 3] = q.w; }   // columns >= 45 are loaded, not staged
  f.hc = P.Hcd[4 * m.pi + part] + 0.0f;
  f.hdi = P.HdiF[m.pi];
  f.bds = P.bdSumF[m.pi];
}

__device__ __forceinline__ void accumHTBlock(float* __restrict__ s_buf, const int b, const int sp, const int nsp, const BARes& Rs, const BAPoints& P,
                                             const int* __restrict__ bucket_begin, const int* __restrict__ bucket_members, float* __restrict__ outTop,
                                             int* __restrict__ outNum, float* __restrict__ outE) {
  float (*s_rec)[HT_STRIDE] = reinterpret_cast<float (*)[HT_STRIDE]>(s_buf);
  int* s_w = reinterpret_cast<int*>(s_buf + HT_TILE * HT_STRIDE);
  const int tid = threadIdx.x, j = tid >> 2, part = tid & 3;
  const int mb = bucket_begin[b], mcnt = bucket_begin[b + 1] - mb;
  const int m0 = mb + (int)(((long long)mcnt * sp) / nsp), m1 = mb + (int)(((long long)mcnt * (sp + 1)) / nsp);
  const HTRole R = htRole(tid);
  Acc3 acc; acc.init();
  int num = 0, n1u = 0;
  if (m1 <= m0) {   // empty bucket (host == target)
    if (R.role >= 0 && R.role < 3) outTop[(b * nsp + sp) * 96 + R.out] = 0.0f;
    if (R.role == 3) outE[(b * nsp + sp) * 40 + R.out] = 0.0f;
    if (tid == 255) outNum[b * nsp + sp] = 0;
    return;
  }
  // pipeline prologue
  int ri1 = bucket_members[min(m0 + j, m1 - 1)];
  int ri2 = bucket_members[min(m0 + HT_TILE + j, m1 - 1)];
  int ri3 = bucket_members[min(m0 + 2 * HT_TILE + j, m1 - 1)];
  HTMeta me1 = htLoadMeta(Rs, ri1), me2 = htLoadMeta(Rs, ri2);
  HTFetch pf;
  htLoadRec(Rs, P, ri1, me1, part, pf);
  for (int base = m0; base < m1; base += HT_TILE) {
    const int cnt = min(HT_TILE, m1 - base);
    __syncthreads();   // the owners are done with the previous tile
    {
      const bool act = base + j < m1 && me1.act;
#pragma unroll
      for (int k = 0; k < 12; k++) { const int idx = 16 * (k >> 2) + 4 * part + (k & 3); if (idx < 45) s_rec[j][idx] = act ? pf.r[k] : 0.0f; }
      s_rec[j][45 + part] = act ? pf.hc : 0.0f;
      if (part == 0) { s_rec[j][49] = act ? pf.hdi : 0.0f; s_rec[j][50] = act ? pf.hdi * pf.bds : 0.0f; s_rec[j][51] = act ? 1.0f : 0.0f; }
    }
    const bool flag = part == 0 && base + j < m1 && me1.act;
    const int nact = __syncthreads_count(flag);
    int cross = 0;
    if (n1u + nact > 1000) cross = blockFindActive(flag, j, 1001 - n1u, s_w);   // block-uniform condition
    n1u = n1u + nact > 1000 ? nact - (1001 - n1u) : n1u + nact;
    // issue the loads of the following tiles (consumed one / two / three iterations from now)
    ri1 = ri2; me1 = me2;
    htLoadRec(Rs, P, ri1, me1, part, pf);
    ri2 = ri3; me2 = htLoadMeta(Rs, ri2);
    ri3 = bucket_members[min(base + 3 * HT_TILE + j, m1 - 1)];
    switch (R.role) {   // wave-uniform
      case 0: addTile<true>(acc, cnt, nact, cross, [&](const int m) { return htContribution<0>(s_rec[m], R); }); break;
      case 1: addTile<false>(acc, cnt, nact, cross, [&](const int m) { return htContribution<1>(s_rec[m], R); }); break;
      case 2: addTile<false>(acc, cnt, nact, cross, [&](const int m) { return htContribution<2>(s_rec[m], R); }); break;
      case 3: addTile<true>(acc, cnt, nact, cross, [&](const int m) { return htContribution<3>(s_rec[m], R); }); break;
      default: break;
    }
    num += nact;
  }
  if (R.role >= 0 && R.role < 3) outTop[(b * nsp + sp) * 96 + R.out] = acc.finish();
  if (R.role == 3) outE[(b * nsp + sp) * 40 + R.out] = acc.finish();
  if (tid == 255) outNum[b * nsp + sp] = num;
}

// ------------------------------------------------------------------------------------------------ Schur accumulation
// accD[h,t1,t2] (8x8) += (HdiF * JpJd(r1)) JpJd(r2)^T : one WAVEFRONT per bucket (four buckets per workgroup), lane (i,j) owns
// element (i,j); member = (r1, r2, point).  Same three-deep staging pipeline, wave-level synchronisation only.
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte access at a 4-byte boundary
#define SCD_STRIDE 19   // [0,8) HdiF-side JpJd(r1), [8,16) JpJd(r2), 16 HdiF, 17 active flag
struct SCDMeta { int act, w1, w2; float hdi; };
__device__ __forceinline__ SCDMeta scdLoadMeta(const BARes& Rs, const BAPoints& P, const int r1, const int r2, const int pi) {
  SCDMeta m;
  const int ac1 = Rs.active[r1], ac2 = Rs.active[r2];
  m.act = (ac1 != 0) & (ac2 != 0); m.w1 = Rs.which[r1]; m.w2 = Rs.which[r2]; m.hdi = P.HdiF[pi];
  return m;
}
__device__ __forceinline__ void accumScDWave(float* __restrict__ s_buf, const int bucket, const int nsp, const BARes& Rs, const BAPoints& P,
                                             const int* __restrict__ bucket_begin, const int* __restrict__ members /* 3 ints each */,
                                             float* __restrict__ outD, int* __restrict__ outNum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane >> 3, jj = lane & 7;
  float (*s_m)[SCD_STRIDE] = reinterpret_cast<float (*)[SCD_STRIDE]>(s_buf + wave * 64 * SCD_STRIDE);
  const int b = bucket / nsp, sp = bucket % nsp;
  const int mb = bucket_begin[b], mcnt = bucket_begin[b + 1] - mb;
  const int m0 = mb + (int)(((long long)mcnt * sp) / nsp), m1 = mb + (int)(((long long)mcnt * (sp + 1)) / nsp);
  Acc3 acc; acc.init();
  int num = 0;
  int a1[3], a2[3], a3[3];
  if (m1 <= m0) { outD[bucket * 64 + lane] = 0.0f; if (lane == 0) outNum[bucket] = 0; return; }
  auto ldIdx = [&](const int m, int* a) {   // unconditional, clamped (see htLoadMeta)
    const int mc = min(m, m1 - 1);
    a[0] = members[3 * mc]; a[1] = members[3 * mc + 1]; a[2] = members[3 * mc + 2];
  };
  ldIdx(m0 + lane, a1); ldIdx(m0 + 64 + lane, a2); ldIdx(m0 + 128 + lane, a3);
  SCDMeta me1 = scdLoadMeta(Rs, P, a1[0], a1[1], a1[2]), me2 = scdLoadMeta(Rs, P, a2[0], a2[1], a2[2]);
  float q1[8], q2[8];
  auto ldRec = [&](const int* a, const SCDMeta& m) {
    const float* __restrict__ p1 = baRec(Rs, m.w1) + (size_t)a[0] * REC_FLOATS + REC_JPJD;
    const float* __restrict__ p2 = baRec(Rs, m.w2) + (size_t)a[1] * REC_FLOATS + REC_JPJD;
    // (JpJdF sits at float 37 of the record: 4-byte aligned only; global memory takes 16-byte accesses at any 4-byte boundary)
    const float4u a0 = *reinterpret_cast<const float4u*>(p1), a1_ = *reinterpret_cast<const float4u*>(p1 + 4);
    const float4u b0 = *reinterpret_cast<const float4u*>(p2), b1_ = *reinterpret_cast<const float4u*>(p2 + 4);
    q1[0] = a0.x; q1[1] = a0.y; q1[2] = a0.z; q1[3] = a0.w; q1[4] = a1_.x; q1[5] = a1_.y; q1[6] = a1_.z; q1[7] = a1_.w;
    q2[0] = b0.x; q2[1] = b0.y; q2[2] = b0.z; q2[3] = b0.w; q2[4] = b1_.x; q2[5] = b1_.y; q2[6] = b1_.z; q2[7] = b1_.w;
  };
  ldRec(a1, me1);
  for (int base = m0; base < m1; base += 64) {
    const int cnt = min(64, m1 - base);
    waveSync();
    const bool act = base + lane < m1 && me1.act;
#pragma unroll
    for (int k = 0; k < 8; k++) { s_m[lane][k] = act ? q1[k] : 0.0f; s_m[lane][8 + k] = act ? q2[k] : 0.0f; }
    s_m[lane][16] = act ? me1.hdi : 0.0f;
    s_m[lane][17] = act ? 1.0f : 0.0f;
    const unsigned long long amask = __ballot(act);
    const int nact = __popcll(amask);
    waveSync();
#pragma unroll
    for (int k = 0; k < 3; k++) { a1[k] = a2[k]; a2[k] = a3[k]; }
    me1 = me2;
    ldRec(a1, me1);
    me2 = scdLoadMeta(Rs, P, a2[0], a2[1], a2[2]);
    ldIdx(base + 192 + lane, a3);
    int cross = 0;
    if (!acc.tileFits(nact)) {   // wave-uniform: row of the (1001 - n1)-th active member
      const bool isCross = act && __popcll(amask & ((1ull << lane) - 1ull)) + 1 == 1001 - acc.n1;
      cross = __ffsll((long long)__ballot(isCross)) - 1;
    }
    addTile<true>(acc, cnt, nact, cross, [&](const int m) { return (s_m[m][16] * s_m[m][i]) * s_m[m][8 + jj]; });
    num += nact;
  }
  outD[bucket * 64 + lane] = acc.finish();
  if (lane == 0) outNum[bucket] = num;
}

// accHcc (4x4) += HdiF Hcd Hcd^T ; accbc (4) += (bdSumF*HdiF) Hcd over all points with an active residual: 20 owners per workgroup,
// nsp partial accumulators over contiguous point ranges (1 = the reference's single-threaded order)
#define SCC_STRIDE 260   // element-major tile [20][256 (+4)]: the owner of element e reads four consecutive members per ds_read_b128
__device__ __forceinline__ void accumScCBlock(float* __restrict__ s_buf, const int sp, const int nsp, const int N, const BAPoints& P, float* __restrict__ outC /* 20 */) {
  float (*s_c)[SCC_STRIDE] = reinterpret_cast<float (*)[SCC_STRIDE]>(s_buf);
  int* s_w = reinterpret_cast<int*>(s_buf + 20 * SCC_STRIDE);
  const int e = threadIdx.x;
  const int p0 = (int)(((long long)N * sp) / nsp), p1 = (int)(((long long)N * (sp + 1)) / nsp);
  if (p1 <= p0) { if (e < 20) outC[sp * 20 + e] = 0.0f; return; }
  Acc3 acc; acc.init();
  int n1u = 0;
  float hdi, bds, hc[4];
  auto ld = [&](const int pi_) {   // unconditional, clamped (see htLoadMeta)
    const int pi = min(pi_, p1 - 1);
    hdi = P.HdiF[pi]; bds = P.bdSumF[pi];
#pragma unroll
    for (int k = 0; k < 4; k++) hc[k] = P.Hcd[4 * pi + k] + 0.0f;
  };
  ld(p0 + e);
  for (int base = p0; base < p1; base += 256) {
    const int cnt = min(256, p1 - base);
    __syncthreads();
    const bool act = base + e < p1 && hdi > 0;   // HdiF == 0 <=> no active residual (point skipped by addPoint)
    {
      const float wb = bds * hdi;
#pragma unroll
      for (int q = 0; q < 16; q++) s_c[q][e] = act ? (hdi * hc[q >> 2]) * hc[q & 3] : 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) s_c[16 + q][e] = act ? wb * hc[q] : 0.0f;
    }
    const int nact = __syncthreads_count(act);
    int cross = 0;
    if (n1u + nact > 1000) cross = blockFindActive(act, e, 1001 - n1u, s_w);   // block-uniform condition
    const bool fits = n1u + nact <= 1000;
    n1u = fits ? n1u + nact : nact - (1001 - n1u);
    ld(base + 256 + e);
    if (e < 20) {
      if (fits) {
        for (int m0 = 0; m0 < cnt; m0 += 16) {
          float4 c[4];
#pragma unroll
          for (int m = 0; m < 4; m++) c[m] = *reinterpret_cast<const float4*>(&s_c[e][m0 + 4 * m]);
#pragma unroll
          for (int m = 0; m < 4; m++) { acc.d += c[m].x; acc.d += c[m].y; acc.d += c[m].z; acc.d += c[m].w; }
        }
        acc.n1 += nact;
      } else {
        addTile<true>(acc, cnt, nact, cross, [&](const int m) { return s_c[e][m]; });
      }
    }
  }
  if (e < 20) outC[sp * 20 + e] = acc.finish();
}

// All accumulations of solveSystemF in ONE launch (they only depend on the applied records and the per-point sums).  Longest
// sequential chains first: blocks [0, nsC) calibration partials, then the (host,target) buckets (top + accE), then accD (4 per block).
struct AccumArgs {
  int F, N, nsTop, nsD, nsC;
  const int *top_begin, *top_members, *scd_begin, *scd_members;
  float *accTop, *accD, *accE, *accC;
  int *numTop, *numD;
  long long* ticks;   // optional [gridDim.x][2] start / end wall-clock stamps per block (DMVIO_HIP_BA_TIMING), else null
};
#define ACC_LDS_FLOATS (20 * SCC_STRIDE + 8)
__device__ __forceinline__ void baAccumulateBody(const AccumArgs& A, const BARes& Rs, const BAPoints& P, const BACtl* __restrict__ ctl, const int gate);
__global__ void __launch_bounds__(256) k_ba_accumulate(const AccumArgs A, const BARes Rs, const BAPoints P, const BACtl* __restrict__ ctl, const int gate) {
  baAccumulateBody(A, Rs, P, ctl, gate);
}
__device__ __forceinline__ void baAccumulateBody(const AccumArgs& A, const BARes& Rs, const BAPoints& P, const BACtl* __restrict__ ctl, const int gate) {
  if (baGateClosed(ctl, gate)) return;
  __shared__ float s_buf[ACC_LDS_FLOATS];
  static_assert(ACC_LDS_FLOATS >= HT_TILE * HT_STRIDE + 8 && ACC_LDS_FLOATS >= 4 * 64 * SCD_STRIDE, "LDS carve-up");
  const int F2 = A.F * A.F;
  const int nHT = F2 * A.nsTop, nD = F2 * A.F * A.nsD;
  int blk = blockIdx.x;
  if (A.ticks && threadIdx.x == 0) A.ticks[2 * blockIdx.x] = wall_clock64();
  if (blk < A.nsC) accumScCBlock(s_buf, blk, A.nsC, A.N, P, A.accC);
  else if (blk < A.nsC + nHT) {
    blk -= A.nsC;
    accumHTBlock(s_buf, blk / A.nsTop, blk % A.nsTop, A.nsTop, Rs, P, A.top_begin, A.top_members, A.accTop, A.numTop, A.accE);
  } else {
    blk -= A.nsC + nHT;
    const int bucket = blk * 4 + (threadIdx.x >> 6);
    if (bucket < nD) accumScDWave(s_buf, bucket, A.nsD, Rs, P, A.scd_begin, A.scd_members, A.accD, A.numD);
  }
  if (A.ticks) { __syncthreads(); if (threadIdx.x == 0) A.ticks[2 * blockIdx.x + 1] = wall_clock64(); }
}

// ------------------------------------------------------------------------------------------------ member lists of the Schur buckets, built on the device
// AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:56-100) adds, for every point, one term per PAIR of its residuals (r1, r2) into accD[host, target(r1), target(r2)];
// k_ba_accumulate walks each bucket's members [r1, r2, p] in the reference's traversal order (points ascending; inside a point r1, then r2).  A point has at most one residual
// per target keyframe, so a bucket (h, t1, t2) receives at most ONE member per point, and the traversal order inside a bucket is simply "points ascending".  That makes the
// lists cheap to build where they are used (the host built and uploaded 86k triples = 1 MB per keyframe before): ridx[p][t] = the residual of point p that targets t,
// one thread per bucket scans the points of its host keyframe in order — first to count, then, behind a scan over the F^3 counts, to write.
__global__ void __launch_bounds__(256) k_ba_scd_ridx(const int R, const int F, const int* __restrict__ point, const int* __restrict__ target, int* __restrict__ ridx) {
  const int ri = blockIdx.x * 256 + threadIdx.x;
  if (ri < R) ridx[point[ri] * F + target[ri]] = ri;
}
// mode 0: counts[k] = members of bucket k = h + F*t1 + F*F*t2;  mode 1: members written at begin[k]...   One WAVEFRONT per bucket (grid: F*F / 4 blocks of four waves x F
// host keyframes): the lanes take 64 consecutive points of the host keyframe at a time, a ballot ranks the ones that observe both targets — points ascending, as the
// reference's traversal leaves them.
__global__ void __launch_bounds__(256) k_ba_scd_lists(const int F, const int* __restrict__ pt_first, const int* __restrict__ pt_last, const int* __restrict__ host,
                                                        const int* __restrict__ ridx, const int mode, int* __restrict__ counts, const int* __restrict__ begin,
                                                        int* __restrict__ members) {
  const int h = blockIdx.y, F2 = F * F, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= F2) return;
  const int t1 = b % F, t2 = b / F;
  const int k = h + F * t1 + F2 * t2;
  int n = 0;
  const int base = mode ? begin[k] : 0;
  if (t1 != h && t2 != h)
    for (int p0 = pt_first[h]; p0 <= pt_last[h]; p0 += 64) {
      const int p = p0 + lane;
      int r1 = -1, r2 = -1;
      if (p <= pt_last[h] && host[p] == h) { r1 = ridx[p * F + t1]; r2 = ridx[p * F + t2]; }
      const bool in = r1 >= 0 && r2 >= 0;
      const unsigned long long m = __ballot(in);
      if (mode && in) {
        int* o = members + 3 * (size_t)(base + n + __popcll(m & ((1ull << lane) - 1ull)));
        o[0] = r1; o[1] = r2; o[2] = p;
      }
      n += __popcll(m);
    }
  if (!mode && lane == 0) counts[k] = n;
}
// begin[0 .. n] = exclusive prefix sums of counts[0 .. n)   (n = F^3 <= 1728: one workgroup, 1024 threads x 2 entries, Hillis-Steele in LDS)
__global__ void __launch_bounds__(1024) k_ba_scd_scan(const int n, const int* __restrict__ counts, int* __restrict__ begin) {
  __shared__ int s[2048];
  for (int i = threadIdx.x; i < 2048; i += 1024) s[i] = i < n ? counts[i] : 0;
  __syncthreads();
  for (int d = 1; d < 2048; d <<= 1) {
    int v[2];
    for (int q = 0; q < 2; q++) { const int i = threadIdx.x + 1024 * q; v[q] = i >= d ? s[i - d] : 0; }
    __syncthreads();
    for (int q = 0; q < 2; q++) s[threadIdx.x + 1024 * q] += v[q];
    __syncthreads();
  }
  for (int i = threadIdx.x; i <= n; i += 1024) begin[i] = i == 0 ? 0 : s[i - 1];
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

// Step 1 in ONE launch of F + F*F workgroups, F wavefronts each (one wavefront per inner index, results summed over the inner
// index in index order by wavefront 0 — the same fp64 sums as a sequential loop, computed F-way in parallel):
//   blocks [0, F):      host h, wavefront t: bucket k = h + F*t, B = acc.H (13x13)
//   blocks [F, F+F*F):  pair (i,j), wavefront k: accD index (i + F*j) + k*F*F; wavefront 0 also stitches accE / accEB
struct StitchWave { double B[13][13]; double AH[64], AT[64], T1[64], T2[64]; double pHH[64], pHC[32], pBH[8], pCC[20]; };

__device__ __forceinline__ void stitchTopWave(StitchWave& W, const int F, const int hI, const int t, const int nsplit, const float* __restrict__ acc,
                                              const int* __restrict__ num, const double* __restrict__ adHost, const double* __restrict__ adTarget,
                                              const StitchBufs& S, const int e) {
  const int r = e >> 3, c = e & 7;
  const int k = hI + F * t;
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
    W.B[q / 13][q % 13] = v;
  }
  W.AH[e] = adHost[k * 64 + e]; W.AT[e] = adTarget[k * 64 + e];
  waveSync();
  double t1 = 0, t2 = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) { t1 += W.AH[r * 8 + q] * W.B[4 + q][4 + c]; t2 += W.AT[r * 8 + q] * W.B[4 + q][4 + c]; }
  W.T1[e] = t1; W.T2[e] = t2;
  waveSync();
  W.pHH[e] = rowDotT(W.T1, W.AH, r, c);
  S.topTT[k * 64 + e] = rowDotT(W.T2, W.AT, r, c);
  S.topHT[k * 64 + e] = rowDotT(W.T1, W.AT, r, c);
  if (c < 4) {
    double h1 = 0, h2 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { h1 += W.AH[r * 8 + q] * W.B[4 + q][c]; h2 += W.AT[r * 8 + q] * W.B[4 + q][c]; }
    W.pHC[r * 4 + c] = h1; S.topTC[k * 32 + r * 4 + c] = h2;
  }
  if (c == 4) {
    double b1 = 0, b2 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { b1 += W.AH[r * 8 + q] * W.B[4 + q][12]; b2 += W.AT[r * 8 + q] * W.B[4 + q][12]; }
    W.pBH[r] = b1; S.topBT[k * 8 + r] = b2;
  }
  if (e < 16) W.pCC[e] = W.B[e >> 2][e & 3];
  else if (e < 20) W.pCC[e] = W.B[e - 16][12];
}

__device__ __forceinline__ void stitchScWave(StitchWave& W, const int F, const int ij, const int kk, const int nsplit, const float* __restrict__ accD,
                                             const int* __restrict__ numD, const double* __restrict__ adHost, const double* __restrict__ adTarget,
                                             const StitchBufs& S, const int e) {
  const int r = e >> 3, c = e & 7;
  const int F2 = F * F, i = ij % F;
  const int ijk = ij + kk * F2, ik = i + F * kk;
  // LDS reuse inside the wavefront's slot: B (169 doubles) holds D in [0,64) and AH_ik in [64,128); AT_ik lives in pHH
  double* sD = &W.B[0][0];
  double* AHik = &W.B[0][0] + 64;
  double* ATik = W.pHH;
  {
    double v = 0.0;
    for (int sp = 0; sp < nsplit; sp++) if (numD[ijk * nsplit + sp] > 0) v += (double)accD[(ijk * nsplit + sp) * 64 + e];
    sD[e] = v;
  }
  W.AH[e] = adHost[ij * 64 + e]; W.AT[e] = adTarget[ij * 64 + e];
  AHik[e] = adHost[ik * 64 + e]; ATik[e] = adTarget[ik * 64 + e];
  waveSync();
  double t1 = 0, t2 = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) { t1 += W.AH[r * 8 + q] * sD[q * 8 + c]; t2 += W.AT[r * 8 + q] * sD[q * 8 + c]; }
  W.T1[e] = t1; W.T2[e] = t2;
  waveSync();
  const double hh = rowDotT(W.T1, AHik, r, c), th = rowDotT(W.T2, AHik, r, c);
  S.scTT[ijk * 64 + e] = rowDotT(W.T2, ATik, r, c);
  S.scHT[ijk * 64 + e] = rowDotT(W.T1, ATik, r, c);
  waveSync();
  // partials for the ordered sum over k: reuse T1 / T2
  W.T1[e] = hh; W.T2[e] = th;
}

__device__ __forceinline__ void baStitchBody(const int F, const int nsTop, const int nsD, const float* __restrict__ accTop, const int* __restrict__ numTop,
                                             const float* __restrict__ accD, const int* __restrict__ numD, const float* __restrict__ accE,
                                             const double* __restrict__ adHost, const double* __restrict__ adTarget, const StitchBufs& S,
                                             const BACtl* __restrict__ ctl, const int gate);
template <int MF>
__global__ void __launch_bounds__(64 * MF) k_ba_stitch(const int F, const int nsTop, const int nsD, const float* __restrict__ accTop, const int* __restrict__ numTop,
                                                    const float* __restrict__ accD, const int* __restrict__ numD, const float* __restrict__ accE /* F*F x nsTop x 40 */,
                                                    const double* __restrict__ adHost, const double* __restrict__ adTarget, const StitchBufs S,
                                                    const BACtl* __restrict__ ctl, const int gate) {
  baStitchBody(F, nsTop, nsD, accTop, numTop, accD, numD, accE, adHost, adTarget, S, ctl, gate);
}
__device__ __forceinline__ void baStitchBody(const int F, const int nsTop, const int nsD, const float* __restrict__ accTop, const int* __restrict__ numTop,
                                             const float* __restrict__ accD, const int* __restrict__ numD, const float* __restrict__ accE,
                                             const double* __restrict__ adHost, const double* __restrict__ adTarget, const StitchBufs& S,
                                             const BACtl* __restrict__ ctl, const int gate) {
  if (baGateClosed(ctl, gate)) return;
  extern __shared__ double s_dyn[];
  StitchWave* Ws = reinterpret_cast<StitchWave*>(s_dyn);
  const int wave = threadIdx.x >> 6, e = threadIdx.x & 63, r = e >> 3, c = e & 7;
  if ((int)blockIdx.x < F) {
    const int hI = blockIdx.x;
    stitchTopWave(Ws[wave], F, hI, wave, nsTop, accTop, numTop, adHost, adTarget, S, e);
    __syncthreads();
    if (wave == 0) {
      double hh = 0, hc = 0, bh = 0, cc = 0;
      for (int t = 0; t < F; t++) {
        hh += Ws[t].pHH[e];
        if (c < 4) hc += Ws[t].pHC[r * 4 + c];
        if (c == 4) bh += Ws[t].pBH[r];
        if (e < 20) cc += Ws[t].pCC[e];
      }
      S.topHH[hI * 64 + e] = hh;
      if (c < 4) S.topHC[hI * 32 + r * 4 + c] = hc;
      if (c == 4) S.topBH[hI * 8 + r] = bh;
      if (e < 20) S.topCC[hI * 20 + e] = cc;
    }
  } else {
    const int ij = blockIdx.x - F;
    stitchScWave(Ws[wave], F, ij, wave, nsD, accD, numD, adHost, adTarget, S, e);
    __syncthreads();
    if (wave == 0) {
      double hh = 0, th = 0;
      for (int kk = 0; kk < F; kk++) { hh += Ws[kk].T1[e]; th += Ws[kk].T2[e]; }
      S.scHH[ij * 64 + e] = hh;
      S.scTH[ij * 64 + e] = th;
      // accE (8x4) and accEB (8): AH_ij / AT_ij are still in this wavefront's AH / AT
      StitchWave& W = Ws[0];
      double* sE = &W.B[0][0];
      waveSync();
      if (e < 40) {
        double v = 0.0;
        for (int sp = 0; sp < nsTop; sp++) v += (double)accE[(ij * nsTop + sp) * 40 + e];
        sE[e] = v;
      }
      waveSync();
      if (c < 4) {
        double h1 = 0, h2 = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { h1 += W.AH[r * 8 + q] * sE[q * 4 + c]; h2 += W.AT[r * 8 + q] * sE[q * 4 + c]; }
        S.scHC[ij * 32 + r * 4 + c] = h1; S.scTC[ij * 32 + r * 4 + c] = h2;
      }
      if (c == 4) {
        double b1 = 0, b2 = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { b1 += W.AH[r * 8 + q] * sE[32 + q]; b2 += W.AT[r * 8 + q] * sE[32 + q]; }
        S.scBH[ij * 8 + r] = b1; S.scBT[ij * 8 + r] = b2;
      }
    }
  }
}

// step 2: one thread per element of H_A, H_sc (n x n, n = 4+8F) and b_A, b_sc; fixed summation order.
// Output layout: out[0 .. n*n) = H_A, then b_A (n), then H_sc (n*n), then b_sc (n).
// `out` is host-coherent pinned memory.  The last workgroup to finish publishes the chain's ticket behind the data (system-scope release); a
// gated-off launch (rejected step: the system of the restored state is the one the host already holds) publishes at once.
template <int MF>
__device__ __forceinline__ void gatherElement(const int F, const int nsC, const float* __restrict__ accC, const StitchBufs& S, const int* __restrict__ numTop,
                                              const int nNum, double* __restrict__ out, const int tid, const bool sys, const int parts, const bool with_count);
template <int MF>
__global__ void __launch_bounds__(256) k_ba_stitch_gather(const int F, const int nsC, const float* __restrict__ accC, const StitchBufs S,
                                                           const int* __restrict__ numTop, const int nNum, double* __restrict__ out, BACtl* __restrict__ ctl, const int gate,
                                                           BAHostRes* __restrict__ host, const unsigned int ticket) {
  if (baGateClosed(ctl, gate)) {
    if (host && threadIdx.x == 0) __hip_atomic_store(&host->gticket[blockIdx.x], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  gatherElement<MF>(F, nsC, accC, S, numTop, nNum, out, blockIdx.x * blockDim.x + threadIdx.x, host != nullptr, 3, true);
  if (!host) return;
  // every workgroup publishes its own slice: write-through stores, a workgroup-scope release (its stores are acknowledged), then the chain's ticket into the
  // workgroup's slot of host-coherent memory; the host waits for all slots.  (A last-workgroup pattern needed a system-scope fence per thread and an agent-scope
  // acquire-release per workgroup: 10 us for this kernel instead of 6.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&host->gticket[blockIdx.x], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// element t of one stitched system — t < n^2: H[t / n][t % n], then b[t - n^2] — sc: the Schur-complement system (H_sc, b_sc) instead of the top system (H_A, b_A)
template <int MF>
__device__ __forceinline__ double gatherValue(const int F, const int nsC, const float* __restrict__ accC, const StitchBufs& S, const int t, const bool sc) {
  const int n = 4 + 8 * F, F2 = F * F;
  // sum_{q < F} base[q * stride] in index order; the (up to 8) loads are issued together, only the adds are sequential
  auto sumF = [&](const double* __restrict__ base, const int stride) {
    double v[MF];
#pragma unroll
    for (int q = 0; q < MF; q++) v[q] = q < F ? base[(size_t)q * stride] : 0.0;
    double acc = 0;
#pragma unroll
    for (int q = 0; q < MF; q++) if (q < F) acc += v[q];
    return acc;
  };
  double val = 0;
  if (t < n * n) {
    const int row = t / n, col = t % n;
    if (row < 4 && col < 4) {
      if (!sc) val = sumF(S.topCC + row * 4 + col, 20);
      else for (int sp = 0; sp < nsC; sp++) val += (double)accC[sp * 20 + row * 4 + col];
    } else if (row < 4 || col < 4) {
      // calib cross terms: H[fIdx.., 0..4) accumulated, mirrored into H[0..4), fIdx..)
      const int cc = row < 4 ? row : col, fr = (row < 4 ? col : row) - 4;
      const int f = fr >> 3, r = fr & 7;
      if (!sc) { val = S.topHC[f * 32 + r * 4 + cc]; val += sumF(S.topTC + (F * f) * 32 + r * 4 + cc, 32); }
      else { val = sumF(S.scHC + f * 32 + r * 4 + cc, F * 32); val += sumF(S.scTC + (F * f) * 32 + r * 4 + cc, 32); }
    } else {
      const int bi = (row - 4) >> 3, bj = (col - 4) >> 3, r = (row - 4) & 7, c = (col - 4) & 7;
      if (!sc) {
        // H[h,h] += HH[h,t], H[t,t] += TT[h,t], H[h,t] += HT[h,t]; then (h<t): H[h,t] += H[t,h]^T, H[t,h] = H[h,t]^T
        if (bi == bj) { val = S.topHH[bi * 64 + r * 8 + c]; val += sumF(S.topTT + (F * bi) * 64 + r * 8 + c, 64); val += S.topHT[(bi + F * bi) * 64 + r * 8 + c]; }
        else val = S.topHT[(bi + F * bj) * 64 + r * 8 + c] + S.topHT[(bj + F * bi) * 64 + c * 8 + r];
      } else {
        // H[i,i] += HH[ij] (all j); H[j,k] += TT[ijk] (all i); H[j,i] += TH[ij]; H[i,k] += HT[ijk] (all j)
        if (bi == bj) val = sumF(S.scHH + bi * 64 + r * 8 + c, F * 64);
        val += sumF(S.scTT + ((F * bi) + bj * F2) * 64 + r * 8 + c, 64);
        val += S.scTH[(bj + F * bi) * 64 + r * 8 + c];
        val += sumF(S.scHT + (bi + bj * F2) * 64 + r * 8 + c, F * 64);
      }
    }
  } else {
    const int row = t - n * n;
    if (row < 4) {
      if (!sc) val = sumF(S.topCC + 16 + row, 20);
      else for (int sp = 0; sp < nsC; sp++) val += (double)accC[sp * 20 + 16 + row];
    } else {
      const int f = (row - 4) >> 3, r = (row - 4) & 7;
      if (!sc) { val = S.topBH[f * 8 + r]; val += sumF(S.topBT + (F * f) * 8 + r, 8); }
      else { val = sumF(S.scBH + f * 8 + r, F * 8); val += sumF(S.scBT + (F * f) * 8 + r, 8); }
    }
  }
  return val;
}
// parts: 1 = the top system (+ resInA), 2 = the Schur system, 3 = both (one accumulation serves both); the three-pass accumulation of a graph with residuals kept
// linearised takes the top system of its L pass (into a buffer of its own) and of its A pass, the Schur system of its third pass
template <int MF>
__device__ __forceinline__ void gatherElement(const int F, const int nsC, const float* __restrict__ accC, const StitchBufs& S, const int* __restrict__ numTop,
                                              const int nNum, double* __restrict__ out, const int tid, const bool sys, const int parts, const bool with_count) {
  // sys: `out` is host-coherent memory the host polls — write through (system-scope stores) instead of a system-scope fence per thread
  auto put = [&](const int i, const double v) { if (sys) __hip_atomic_store(out + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else out[i] = v; };
  const int n = 4 + 8 * F;
  const int per = n * n + n;
  if (tid == 2 * per) {   // resInA: number of active residuals that entered the top accumulation, appended to the system
    if (!(parts & 1) || !with_count) return;
    int cnt = 0;
    for (int k = 0; k < nNum; k++) cnt += numTop[k];
    put(2 * per, (double)cnt);
    return;
  }
  if (tid >= 2 * per) return;
  const bool sc = tid >= per;
  if (!(parts & (sc ? 2 : 1))) return;
  const int t = sc ? tid - per : tid;
  put(tid, gatherValue<MF>(F, nsC, accC, S, t, sc));
}

// ------------------------------------------------------------------------------------------------ back-substitution / stepping
// xAd: F*F x 8 floats, index h*F + t (EnergyFunctional.cpp:280-282), xc: 4 floats
// apply_step != 0 fuses doStepFromBackup for stepfac = 1 (FullSystemOptimize.cpp:224-317): idepth = idepth_zero = backup + step
// xc and xAd travel as kernel arguments (2 KB): no separate upload, no staging buffer
// Eight lanes per point like k_ba_point_sums: lane q forms xAd[h,t_q] . JpJdF of the point's q-th residual (sequential over the 8 entries), the
// leading lane subtracts the products in residual order (an inactive residual subtracts +0.0f: no change).
__device__ __forceinline__ void baResubstituteBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const float* __restrict__ xc, const float* __restrict__ xAd, const int apply_step);
template <int MF>
__global__ void __launch_bounds__(256) k_ba_resubstitute(const BAWindow W, const BAPoints P, const BARes Rs, const ResubArgsT<MF> X, const int apply_step) {
  baResubstituteBody(W, P, Rs, X.xc, X.xAd, apply_step);
}
__device__ __forceinline__ void baResubstituteBody(const BAWindow& W, const BAPoints& P, const BARes& Rs, const float* __restrict__ xc, const float* __restrict__ xAd, const int apply_step) {
  const int pi = blockIdx.x * PT_GROUPS_PER_BLOCK + (threadIdx.x >> 3), q = threadIdx.x & 7;
  if (pi >= W.N) return;   // group-uniform
  const bool lead = q == 0;
  const int r0 = P.res_begin[pi], r1 = P.res_begin[pi + 1];
  const int hi = P.host[pi];
  const float bdSum = P.bdSumF[pi], HdiF = P.HdiF[pi], bk = P.idepth_backup[pi];
  const float hc0 = P.Hcd[4 * pi + 0], hc1 = P.Hcd[4 * pi + 1], hc2 = P.Hcd[4 * pi + 2], hc3 = P.Hcd[4 * pi + 3];
  const int grp = (threadIdx.x & 63) & ~7;
  float b = bdSum;
  {
    float dotc = 0;
    dotc += xc[0] * (hc0 + 0.0f); dotc += xc[1] * (hc1 + 0.0f); dotc += xc[2] * (hc2 + 0.0f); dotc += xc[3] * (hc3 + 0.0f);
    b -= dotc;
  }
  int ngood = 0;
  for (int rb = r0; rb < r1; rb += 8) {
    const int ri = min(rb + q, r1 - 1);
    const bool act = rb + q < r1 && Rs.active[ri] != 0;
    const float* __restrict__ jp = baRec(Rs, Rs.which[ri]) + (size_t)ri * REC_FLOATS + REC_JPJD;
    const float* __restrict__ xa = xAd + (size_t)(hi * W.F + Rs.target[ri]) * 8;
    float d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += xa[k] * jp[k];
    if (!act) d = 0.0f;
    // b -= d_0; b -= d_1; ... in residual order (leading lane)
    b = b - d;
#pragma unroll
    for (int k = 1; k < 8; k++) b = b - dppShl(d, k);
    ngood += __popcll((__ballot(act) >> grp) & 0xFFull);
  }
  if (!lead) return;
  if (ngood == 0) {
    P.step[pi] = 0;
    if (apply_step) { const float v = bk + 1.0f * 0.0f; P.idepth[pi] = v; P.idepth_zero[pi] = v; }
    return;
  }
  const float st = -b * HdiF;
  P.step[pi] = st;
  if (apply_step) { const float v = bk + 1.0f * st; P.idepth[pi] = v; P.idepth_zero[pi] = v; }
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
  if (mode == 1 && partials) {
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
