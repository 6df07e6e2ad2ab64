// libdmvio_hip.so — C ABI of the bundle-adjustment path (include/dmvio_hip.h, "sliding-window BA" section).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is loaded on first use (below)
#include <dlfcn.h>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <atomic>
#include <string>

#include "../../include/dmvio_hip.h"
#include "internal.h"
#include "ba_kernels.hpp"
#include "ba_host.hpp"
#include "ba_batch_kernels.hpp"

using namespace dmv;

// RCCL is bound at RUN time, on the first call that needs a communicator: libdmvio_hip.so itself has no link dependency on librccl.so, so hosts without RCCL (a single-GPU
// workstation, a CPU-only build box) load the library and run everything but the multi-GPU entry points, which then fail with a message instead of a loader error.
namespace {
struct RcclApi {
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclAllGather) allGather = nullptr;
  decltype(&ncclCommCount) commCount = nullptr;
  decltype(&ncclCommUserRank) commUserRank = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
  bool ok = false;
  std::string why;
};
RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { api.why = std::string("librccl.so cannot be loaded (") + (dlerror() ? dlerror() : "?") + ")"; return; }
    bool all = true;
    auto get = [&](const char* sym) { void* p = dlsym(h, sym); if (!p) { all = false; api.why = std::string("librccl.so lacks ") + sym; } return p; };
    api.allReduce = (decltype(api.allReduce))get("ncclAllReduce"); api.allGather = (decltype(api.allGather))get("ncclAllGather");
    api.commCount = (decltype(api.commCount))get("ncclCommCount"); api.commUserRank = (decltype(api.commUserRank))get("ncclCommUserRank");
    api.getUniqueId = (decltype(api.getUniqueId))get("ncclGetUniqueId"); api.commInitRank = (decltype(api.commInitRank))get("ncclCommInitRank");
    api.commDestroy = (decltype(api.commDestroy))get("ncclCommDestroy"); api.getErrorString = (decltype(api.getErrorString))get("ncclGetErrorString");
    api.ok = all;
  });
  return api;
}
}  // namespace
#define RCCL_READY() do { if (!rccl().ok) return failmsg("RCCL is not available: " + rccl().why); } while (0)

#include <chrono>
// optional host-side time split of the GN iteration (DMVIO_HIP_BA_TIMING=1 prints it when the handle is destroyed)
struct BATimes { double t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long n = 0; };
static inline double nowUs() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct dmvio_hip_ba {
  // The mapping side owns a HIP stream and a lock of its own: the tracking thread (context stream, context lock) and the mapping
  // thread (this stream, this lock) overlap on the device like coarseTracker / mapping do in the reference (FullSystem.cpp:980-985).
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::recursive_mutex mu;   // every entry point that takes the handle holds it from its first line (entry points may call each other: recursive)
  dmvio_hip_ctx* ctx = nullptr;
  DmvBounce bounce;   // caller-owned arrays (and this file's short-lived host vectors) cross PCIe through the library's pinned memory (internal.h), on `stream`, under `mu`
  BAHost H;
  BAWindow W{};
  BAPoints P{};
  BARes Rs{};
  // host copies of the graph
  std::vector<int> h_host, h_point, h_target, h_res_begin;
  std::vector<int> h_newest;   // residuals that target the newest keyframe (inputs of setNewFrameEnergyTH), ascending
  std::vector<unsigned char> h_prior_flag;
  // device storage.  The window is rebuilt for every keyframe (dmvio_hip_ba_set_graph): its ~75 device arrays are carved out of a few large chunks that stay with the
  // handle and are cleared with one memset each — not allocated, cleared and freed one by one (that cost milliseconds per keyframe, more than optimize(6) itself)
  std::vector<void*> allocs;
  struct Arena { std::vector<std::pair<char*, size_t>> chunks; std::vector<size_t> used; size_t cur = 0, off = 0; bool on = false; } arena;   // used[k]: bytes of chunk k handed out since it was last cleared
  size_t cap_spart = 0, cap_idepth_backup = 0;   // capacities of the grow-only pinned host buffers
  BAPrecalc* d_pre = nullptr;        // the precalc table the kernels read: one of the two halves of d_pre2
  BAPrecalc* d_pre2 = nullptr;       // [2][F*F]: the table of the backed-up state stays resident, a rejected step switches back to it
  int pre_half = 0;
  double *d_adHost = nullptr, *d_adTarget = nullptr;
  int *d_top_begin = nullptr, *d_top_members = nullptr, *d_scd_begin = nullptr, *d_scd_members = nullptr;
  float *d_accTop = nullptr, *d_accD = nullptr, *d_accE = nullptr, *d_accC = nullptr;
  int *d_numTop = nullptr, *d_numD = nullptr;
  StitchBufs SB{};
  double *h_sys = nullptr;     // [H_A | b_A | H_sc | b_sc | resInA]: pinned host memory, written by k_ba_stitch_gather
  float *d_spart = nullptr, *h_spart = nullptr;  // point-step partial sums
  // pinned staging for the per-linearisation precalc upload (no pageable copy, no sync before the kernel that consumes it)
  BAPrecalc* h_pre[2] = {nullptr, nullptr};
  int pre_toggle = 0;
  float* d_fullJ = nullptr;
  // device-side decisions (ba_kernels.hpp, BACtl): control block, host-coherent result block, device copies of what the decisions read
  BACtl* d_ctl = nullptr;
  BAHostRes* h_res = nullptr;
  bool th_pending = false;            // the newest keyframe's threshold of the last accept-test pass is stored a few microseconds behind its decision (BAHostRes::th_ticket)
  unsigned int th_pending_ticket = 0;
  float *d_frameTH = nullptr, *h_frameTH = nullptr;   // FrameHessian::frameEnergyTH of every keyframe (the newest one is updated on the device)
  bool th_dirty = true;        // the host changed a threshold: upload before the next linearisation
  double* d_epart = nullptr;
  int* d_newestSlot = nullptr;   // per residual: its position among the residuals that target the newest keyframe, or -1
  float* d_newestE = nullptr;    // their state_NewEnergyWithOutlier, contiguous
  ResubArgs x_none{};            // placeholder argument of linearisations without the fused back-substitution
  BAPreDyn dyn_cur;              // step-dependent precalc members of the CURRENT state (kernel argument of the GN loop's linearisations)
  bool pre_static_valid = false; // the device table holds the evaluation-point members (R0, t0, b0) of the current window
  float* d_newEnergyWO = nullptr;
  unsigned int ticket = 0, acc_ticket = 0;
  float th_cap = -1.0f;        // IMUIntegration::newFrameEnergyTH cap (<= 0: none)
  // a rejected step whose relinearisation the host has not waited for (the loop inside dmvio_hip_ba_optimize): its energy / threshold are picked up at the next wait
  bool pending_reject = false;
  unsigned int pending_ticket = 0;
  int pending_trace = -1;
  bool sys_ready = false;      // h_sys holds the stitched system of the CURRENT state (left behind by the previous GN iteration's chain)
  int n_lin_blocks = 0, n_pt_blocks = 0, n_pt8_blocks = 0, n_epart = 0;   // n_pt8: kernels with eight lanes per point
  bool keep_fullJ = false;   // the 74-float RawResidualJacobian is only materialised on request (dmvio_hip_ba_keep_jacobians) and for marginalisation
  // partial accumulators per bucket: 1 (default) replays the single-threaded reference order bit for bit; DMVIO_HIP_BA_SPLIT=k uses k
  // partial accumulators per bucket: k > 1 = the structure of the reference's multi-threaded accumulation (per-worker fp32 accumulators summed
  // in double, AccumulatedTopHessian.h:91-139) with a FIXED assignment of members to partials; 1 = the reference's single-threaded order, bit for bit
  int nsTop = 4, nsD = 4, nsC = 16;
  bool graph_ready = false;
  // energies of the last optimize
  double trace[64][4];
  int iterations_done = 0;
  double final_energy = 0;
  BATimes tm;
  bool timing = false;
  double tm_graph[6] = {0, 0, 0, 0, 0, 0}; long tm_graph_n = 0;   // dmvio_hip_ba_set_graph: drain + arena memset, host lists, allocation, uploads, pinned buffers + slot table, adjoints + final wait
  // true only between a REJECTED step of gnIteration and the next gnIteration: the state was restored to the one the per-point sums (and the
  // point backup) were computed at, so k_ba_point_sums would reproduce what is already there.  Every other entry point clears it.
  bool sums_fresh = false;
  // point marginalisation scratch (dmvio_hip_ba_marginalize_points)
  unsigned char *d_cand = nullptr, *d_decision = nullptr, *d_margActive = nullptr;
  float *d_mHdiF = nullptr, *d_mbdSumF = nullptr, *d_mHcd = nullptr, *d_margRec = nullptr, *d_adHTdelta = nullptr;
  long long* d_accTicks = nullptr;   // per-block stamps of k_ba_accumulate (timing mode only)
  int accTicksBlocks = 0;
  // flat arrays of dmvio_hip_ba_set_graph_from (kept between keyframes: no allocation in the steady state)
  struct GraphScratch { std::vector<int> host, res_point, res_target; std::vector<float> u, v, idepth, color, weights, linJ, linRtz; std::vector<unsigned char> prior, lin; } gscratch;
  // ---- points sharded over ranks (dmvio_hip_ba_set_comm): every rank holds all keyframes and ITS points; the stitched system is summed by
  // an all-reduce in HBM on this handle's stream, the accept / threshold decisions are taken over the all-gathered per-rank records
  int rank = 0, world = 0;           // world == 0: no communicator
  ncclComm_t nccl = nullptr;         // RCCL communicator (not owned)
  dmvio_hip_comm_callbacks comm_cb{};   // host-staged transport (MPI, gloo, ...) when nccl == NULL
  double* d_sys = nullptr;           // [H_A | b_A | H_sc | b_sc | resInA] of this rank's points, all-reduced in place
  float *d_xchg_local = nullptr, *d_xchg_all = nullptr;
  int xchg_width = 0;                // floats per rank record: BA_XCHG_HEADER + the largest per-rank count of residuals that target the newest keyframe
  std::vector<double> h_stage;       // callback transport only
  // ---- the reference's DEFAULT solver branch (setting_useGTSAMIntegration, dmvio_hip_ba_optimize_vio): hooks of the running call, the dynamic weight,
  // PointHessian::idepth_backup mirrored into host-coherent memory by the per-point sums (the |idepth_backup| sum of doStepFromBackup's canbreak test)
  const dmvio_hip_ba_callbacks* vio = nullptr;
  const dmvio_hip_ba_vio_options* vio_opt = nullptr;
  double dynW = 1.0;
  int resInA_solve = 0;              // ef->resInA as the reference holds it: set by the accumulation of the last solveSystemF
  float* h_idepth_backup = nullptr;
  std::vector<dmvio_hip_ba_frame_view> vio_frames;
  hipEvent_t* prof = nullptr;        // dmvio_hip_ba_profile_chain: six events recorded between the launches of linearise -> per-point sums -> accumulate -> stitch -> gather
  // dmvio_hip_ba_set_device_loop: dmvio_hip_ba_optimize runs the device-resident loop (a batch of one window) instead of the host-driven one
  bool device_loop = false;
  struct dmvio_hip_ba_batch* own_batch = nullptr;
  // dmvio_hip_ba_comm_timing: HIP events around the collectives of the sharded iteration (RCCL transport), kind 0 = all-reduce of the packed system, 1 = all-gather of the
  // decision records; up to COMM_EVS of each are kept and summed when asked for
  enum { COMM_EVS = 64 };
  bool comm_timing = false;
  hipEvent_t comm_ev[2][COMM_EVS][2] = {};
  int comm_n[2] = {0, 0};
  long comm_total[2] = {0, 0};
  // ---- residuals kept linearised outside a marginalisation (dmvio_hip_ba_fix_linearization; ba_kernels.hpp "residuals kept linearised"): flags, res_toZeroF, the record
  // addPoint<1> consumes, the activity views of the three accumulation passes, the per-point LF sums; host copies of what calcLEnergyPt reads
  int n_lin = 0;
  long long n_lin_global = 0;   // sharded window: the ranks' n_lin summed (dmvio_hip_ba_fix_linearization is collective there) — every rank takes the three-pass accumulation or none
  double* d_red1 = nullptr;     // one double for small all-reduces (the linearised energy of a sharded window)
  unsigned char *d_lin = nullptr, *d_linMask = nullptr, *d_linActive = nullptr, *d_topActive = nullptr;
  float *d_rtz = nullptr, *d_linRec = nullptr, *d_lHdd = nullptr, *d_lbd = nullptr, *d_lHcd = nullptr, *d_HcdAF = nullptr, *d_linE = nullptr;
  std::vector<unsigned char> h_lin, h_linAct;
  std::vector<float> h_linJ, h_rtz;
  bool fullJ_applied = false;   // d_fullJ holds the Jacobians of the APPLIED linearisation (the last linearisation was followed by its applyRes)
  bool adj_dirty = false;   // the host's adjoint tables (H.adHost / adTarget) are newer than the device copy: uploaded by the next consumer (accumulateViews, a batch call)
};
#define BA_LOCK(b) std::lock_guard<std::recursive_mutex> lk_(b->mu)
#define BA_PROF(b, k) do { if ((b)->prof) hipEventRecord((b)->prof[k], (b)->stream); } while (0)
// The kernels whose ARGUMENTS are sized by the window (ba_kernels.hpp: BAPreDynT / ResubArgsT, the stitch workgroup of 64 F threads): windows of up to BA_MAXF keyframes
// take the compact instantiation (the host's wide structs cut down to their prefix), larger ones (up to BA_MAXF_CAP) the wide one.
#define BA_LAUNCH_LINEARIZE(b, W_, fullJ_, mask_, D_, gate_, use_backup_, T_, use_dyn_, X_, do_resub_) do { \
    (b)->fullJ_applied = false; \
    if ((b)->H.F <= BA_MAXF) hipLaunchKernelGGL((k_ba_linearize<BA_MAXF>), dim3((b)->n_lin_blocks), dim3(LIN_THREADS), 0, (b)->stream, W_, (b)->P, (b)->Rs, (const BAPrecalc*)(b)->d_pre, \
        (b)->ctx->fs, fullJ_, mask_, D_, (int)(gate_), (int)(use_backup_), baNarrow<BAPreDynT<BA_MAXF>>(T_), (int)(use_dyn_), baNarrow<ResubArgsT<BA_MAXF>>(X_), (int)(do_resub_)); \
    else hipLaunchKernelGGL((k_ba_linearize<BA_MAXF_CAP>), dim3((b)->n_lin_blocks), dim3(LIN_THREADS), 0, (b)->stream, W_, (b)->P, (b)->Rs, (const BAPrecalc*)(b)->d_pre, \
        (b)->ctx->fs, fullJ_, mask_, D_, (int)(gate_), (int)(use_backup_), T_, (int)(use_dyn_), X_, (int)(do_resub_)); \
  } while (0)
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return failmsg((std::string("RCCL: ") + rccl().getErrorString(r_) + " in " #x).c_str()); } while (0)

// one chunk holds the ~75 arrays of a window of 8 keyframes / 4000 points / 30k residuals; allocated (and the pinned host buffers with it) when the handle is created, so that
// no keyframe of a live system pays for hipMalloc / hipHostMalloc (seen in tests/dropin: 1.2 ms per keyframe on average over the first ten, 0.4 ms in the steady state)
static constexpr size_t BA_ARENA_CHUNK = (size_t)48 << 20;
template <class T>
static int dalloc(dmvio_hip_ba* b, T** p, size_t n) {
  if (b->arena.on) {   // inside dmvio_hip_ba_set_graph: a zeroed piece of the handle's arena
    dmvio_hip_ba::Arena& A = b->arena;
    const size_t bytes = (sizeof(T) * std::max<size_t>(n, 1) + 255) & ~(size_t)255;
    while (A.cur < A.chunks.size() && A.off + bytes > A.chunks[A.cur].second) { A.cur++; A.off = 0; }
    if (A.cur == A.chunks.size()) {
      const size_t size = std::max<size_t>(bytes, BA_ARENA_CHUNK);
      char* base = nullptr;
      HIPCHK(hipMalloc((void**)&base, size));
      HIPCHK(hipMemset(base, 0, size));
      HIPCHK(hipStreamSynchronize(nullptr));
      A.chunks.push_back(std::make_pair(base, size)); A.used.push_back(0);
      A.off = 0;
    }
    *p = reinterpret_cast<T*>(A.chunks[A.cur].first + A.off);
    A.off += bytes;
    A.used[A.cur] = std::max(A.used[A.cur], A.off);
    return 0;
  }
  HIPCHK(hipMalloc((void**)p, sizeof(T) * std::max<size_t>(n, 1)));
  HIPCHK(hipMemset(*p, 0, sizeof(T) * std::max<size_t>(n, 1)));
  // hipMemset clears on the NULL stream without blocking the host, and the handle's stream is non-blocking: without this wait an upload enqueued next could
  // land before the clear does (seen with two processes sharing a GPU)
  HIPCHK(hipStreamSynchronize(nullptr));
  b->allocs.push_back((void*)*p);
  return 0;
}
// what a new graph replaces: the arrays allocated outside the arena (exchange buffers of a sharded window); the arena's chunks and the pinned host buffers stay
static void freeDevice(dmvio_hip_ba* b) {
  for (void* p : b->allocs) hipFree(p);
  b->allocs.clear();
  b->graph_ready = false;
}
static void freeAll(dmvio_hip_ba* b) {
  freeDevice(b);
  for (auto& ch : b->arena.chunks) hipFree(ch.first);
  b->arena.chunks.clear(); b->arena.used.clear(); b->arena.cur = b->arena.off = 0;
  b->bounce.release();
  if (b->h_sys) { hipHostFree(b->h_sys); b->h_sys = nullptr; }
  if (b->h_spart) { hipHostFree(b->h_spart); b->h_spart = nullptr; }
  if (b->h_res) { hipHostFree(b->h_res); b->h_res = nullptr; }
  if (b->h_frameTH) { hipHostFree(b->h_frameTH); b->h_frameTH = nullptr; }
  if (b->h_idepth_backup) { hipHostFree(b->h_idepth_backup); b->h_idepth_backup = nullptr; }
  for (int k = 0; k < 2; k++) if (b->h_pre[k]) { hipHostFree(b->h_pre[k]); b->h_pre[k] = nullptr; }
  b->cap_spart = b->cap_idepth_backup = 0;
}

// switch_back: the state was restored to the one whose table is still in the other half (loadSateBackup after a rejected step)
static void fillWindow(dmvio_hip_ba* b) {
  BAHost& H = b->H;
  BAWindow& W = b->W;
  W.F = H.F; W.w = H.w; W.h = H.h; W.N = H.N; W.R = H.R;
  W.fx = H.c_f[0]; W.fy = H.c_f[1]; W.cx = H.c_f[2]; W.cy = H.c_f[3];
  W.fxi = H.c_i[0]; W.fyi = H.c_i[1]; W.cxi = H.c_i[2]; W.cyi = H.c_i[3];
  W.wM3 = H.w - 3; W.hM3 = H.h - 3;
  W.huberTH = H.S.huberTH; W.outlierTHSum = H.S.outlierTHSumComponent; W.modeA = H.S.affineOptModeA; W.modeB = H.S.affineOptModeB;
  for (int f = 0; f < H.F; f++) { W.slot[f] = H.fr[f].slot; W.frameEnergyTH[f] = H.fr[f].frameEnergyTH; }
}
static void dynFromHost(const BAHost& H, BAPreDyn& T) {
  for (int hh = 0; hh < H.F; hh++)
    for (int t = 0; t < H.F; t++) {
      if (hh == t) continue;
      const BAPrecalc& pc = H.pre[hh + H.F * t];
      float* v = T.v[baPairIndex(hh, t, H.F)];
      for (int k = 0; k < 9; k++) v[k] = pc.KRKi[k];
      v[9] = pc.Kt[0]; v[10] = pc.Kt[1]; v[11] = pc.Kt[2]; v[12] = pc.aff0; v[13] = pc.aff1;
    }
}
static int uploadWindowTables(dmvio_hip_ba* b, bool new_state = false, bool switch_back = false) {
  BAHost& H = b->H;
  BAWindow& W = b->W;
  b->pre_static_valid = true;
  W.F = H.F; W.w = H.w; W.h = H.h; W.N = H.N; W.R = H.R;
  W.fx = H.c_f[0]; W.fy = H.c_f[1]; W.cx = H.c_f[2]; W.cy = H.c_f[3];
  W.fxi = H.c_i[0]; W.fyi = H.c_i[1]; W.cxi = H.c_i[2]; W.cyi = H.c_i[3];
  W.wM3 = H.w - 3; W.hM3 = H.h - 3;
  W.huberTH = H.S.huberTH; W.outlierTHSum = H.S.outlierTHSumComponent; W.modeA = H.S.affineOptModeA; W.modeB = H.S.affineOptModeB;
  for (int f = 0; f < H.F; f++) { W.slot[f] = H.fr[f].slot; W.frameEnergyTH[f] = H.fr[f].frameEnergyTH; }
  // two pinned staging copies: at most one earlier upload can still be in flight (every linearize ends with a stream sync)
  if (switch_back) { b->pre_half ^= 1; b->d_pre = b->d_pre2 + (size_t)b->pre_half * H.F * H.F; return 0; }
  if (new_state) b->pre_half ^= 1;   // keep the previous state's table in the other half
  b->d_pre = b->d_pre2 + (size_t)b->pre_half * H.F * H.F;
  BAPrecalc* stage = b->h_pre[b->pre_toggle ^= 1];
  memcpy(stage, H.pre.data(), sizeof(BAPrecalc) * H.F * H.F);
  HIPCHK(hipMemcpyAsync(b->d_pre, stage, sizeof(BAPrecalc) * H.F * H.F, hipMemcpyHostToDevice, b->stream));
  return 0;
}
static int uploadAdjoints(dmvio_hip_ba* b) {
  b->adj_dirty = false;
  // staged in the handle's pinned area: the host tables may change (setAdjointsF of the next state) while the copy is still in flight
  HIPCHK(b->bounce.h2d(b->d_adHost, b->H.adHost.data(), sizeof(double) * b->H.adHost.size(), b->stream));
  HIPCHK(b->bounce.h2d(b->d_adTarget, b->H.adTarget.data(), sizeof(double) * b->H.adTarget.size(), b->stream));
  return 0;
}

// Completion of a chain of kernels: its last kernel stores the chain's ticket into host-coherent memory behind its results; the host spins
// on that word instead of paying a stream synchronisation (wake-up latency) per Gauss-Newton iteration.
static int waitTicket(dmvio_hip_ba* b, const unsigned int ticket) {
  volatile unsigned int* flag = &b->h_res->ticket;
  unsigned long long spins = 0;
  while (*flag != ticket) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {
      const hipError_t q = hipStreamQuery(b->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail("BA kernel chain", __FILE__, __LINE__, q);
      if (q == hipSuccess && *flag != ticket) return failmsg("BA kernel chain finished without publishing its ticket");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}
// the host's mirror of the newest keyframe's threshold after an accepted step: the decision pass publishes its decision first and the threshold behind it
static int resolveTh(dmvio_hip_ba* b) {
  if (!b->th_pending) return 0;
  volatile unsigned int* flag = &b->h_res->th_ticket;
  unsigned long long spins = 0;
  while ((int)(*flag - b->th_pending_ticket) < 0) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {
      const hipError_t q = hipStreamQuery(b->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail("BA decision pass", __FILE__, __LINE__, q);
      if (q == hipSuccess && (int)(*flag - b->th_pending_ticket) < 0) return failmsg("BA decision pass finished without publishing its threshold");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  b->H.fr[b->H.F - 1].frameEnergyTH = b->h_res->th[0];
  b->th_pending = false;
  return 0;
}
static int uploadThresholds(dmvio_hip_ba* b) {
  if (int r = resolveTh(b)) return r;
  if (!b->th_dirty) return 0;
  for (int f = 0; f < BA_MAXF_CAP; f++) b->h_frameTH[f] = f < b->H.F ? b->H.fr[f].frameEnergyTH : 0.0f;
  HIPCHK(hipMemcpyAsync(b->d_frameTH, b->h_frameTH, sizeof(float) * BA_MAXF_CAP, hipMemcpyHostToDevice, b->stream));
  b->th_dirty = false;
  return 0;
}
static BADecide makeDecide(dmvio_hip_ba* b, int mode, bool update_th, bool publish) {
  BADecide D;
  memset(&D, 0, sizeof(D));
  const BAHost& H = b->H;
  D.newestE = b->d_newestE; D.n_newest = (int)b->h_newest.size(); D.newestFrame = H.F - 1;
  D.frameTH = b->d_frameTH; D.epart = b->d_epart;
  D.thN = H.S.frameEnergyTHN; D.thFacMedian = H.S.frameEnergyTHFacMedian; D.thConstWeight = H.S.frameEnergyTHConstWeight; D.overallW = H.S.overallEnergyTHWeight;
  D.thCap = b->th_cap;
  D.mode = mode; D.update_th = update_th ? 1 : 0;
  D.ctl = b->d_ctl; D.host = b->h_res;
  D.publish = publish ? 1 : 0;
  if (publish) D.ticket = ++b->ticket;
  return D;
}

// ---- exchange steps of the sharded iteration.  With an RCCL communicator they are enqueued on the handle's stream between the kernels that produce
// and consume the buffers (no host hop); the callback transport stages them through host memory.
static bool sharded(const dmvio_hip_ba* b) { return b->world > 0; }
static bool linAny(const dmvio_hip_ba* b) { return sharded(b) ? b->n_lin_global > 0 : b->n_lin > 0; }   // any rank holds a residual kept linearised
static hipEvent_t* commEvents(dmvio_hip_ba* b, const int kind) {
  b->comm_total[kind]++;
  if (!b->comm_timing || b->comm_n[kind] >= dmvio_hip_ba::COMM_EVS) return nullptr;
  hipEvent_t* e = b->comm_ev[kind][b->comm_n[kind]];
  if (!e[0] && (hipEventCreate(&e[0]) != hipSuccess || hipEventCreate(&e[1]) != hipSuccess)) return nullptr;
  b->comm_n[kind]++;
  return e;
}
static int commAllReduceSum(dmvio_hip_ba* b, double* d_buf, size_t count) {
  if (b->nccl) {
    hipEvent_t* e = commEvents(b, 0);
    if (e) HIPCHK(hipEventRecord(e[0], b->stream));
    NCCLCHK(rccl().allReduce(d_buf, d_buf, count, ncclDouble, ncclSum, b->nccl, b->stream));
    if (e) HIPCHK(hipEventRecord(e[1], b->stream));
    return 0;
  }
  b->h_stage.resize(count);
  HIPCHK(hipMemcpyAsync(b->h_stage.data(), d_buf, sizeof(double) * count, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->comm_cb.allreduce_sum_f64(b->comm_cb.user, b->h_stage.data(), count) != 0) return failmsg("comm callback allreduce_sum_f64 failed");
  HIPCHK(hipMemcpyAsync(d_buf, b->h_stage.data(), sizeof(double) * count, hipMemcpyHostToDevice, b->stream));
  return 0;
}
static int commAllGather(dmvio_hip_ba* b, const float* d_in, float* d_out, size_t count_per_rank) {
  if (b->nccl) {
    hipEvent_t* e = commEvents(b, 1);
    if (e) HIPCHK(hipEventRecord(e[0], b->stream));
    NCCLCHK(rccl().allGather(d_in, d_out, count_per_rank, ncclFloat, b->nccl, b->stream));
    if (e) HIPCHK(hipEventRecord(e[1], b->stream));
    return 0;
  }
  const size_t bytes = sizeof(float) * count_per_rank;
  b->h_stage.resize((bytes * (b->world + 1) + 7) / 8);
  char* in = (char*)b->h_stage.data(); char* out = in + bytes;
  HIPCHK(hipMemcpyAsync(in, d_in, bytes, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->comm_cb.allgather(b->comm_cb.user, in, out, bytes) != 0) return failmsg("comm callback allgather failed");
  HIPCHK(hipMemcpyAsync(d_out, out, bytes * b->world, hipMemcpyHostToDevice, b->stream));
  return 0;
}
// record width of the decision exchange: agreed on once per graph (the largest per-rank number of residuals that target the newest keyframe)
static int ensureExchange(dmvio_hip_ba* b) {
  if (b->xchg_width) return 0;
  int mine = (int)b->h_newest.size(), widest = mine;
  if (b->world > 1) {
    float* d_tmp = nullptr;
    if (dalloc(b, &d_tmp, (size_t)b->world + 1)) return -1;
    const float v = (float)mine;   // exact below 2^24 residuals
    HIPCHK(hipMemcpyAsync(d_tmp, &v, sizeof(float), hipMemcpyHostToDevice, b->stream));
    if (int r = commAllGather(b, d_tmp, d_tmp + 1, 1)) return r;
    std::vector<float> all(b->world);
    HIPCHK(hipMemcpyAsync(all.data(), d_tmp + 1, sizeof(float) * b->world, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (float a : all) widest = std::max(widest, (int)a);
  }
  b->xchg_width = BA_XCHG_HEADER + ((widest + 1) & ~1);   // even: the fp64 energy of every record stays 8-byte aligned
  if (dalloc(b, &b->d_xchg_local, (size_t)b->xchg_width) || dalloc(b, &b->d_xchg_all, (size_t)b->xchg_width * b->world)) return -1;
  return 0;
}
// tail of a sharded linearisation: the linearisation's last workgroup packed this rank's record (mode 3); gather the records of all ranks and take
// the decisions of `D` over their union
static int decideGlobal(dmvio_hip_ba* b, BADecide D) {
  if (int r = commAllGather(b, b->d_xchg_local, b->d_xchg_all, (size_t)b->xchg_width)) return r;
  D.xchg_all = b->d_xchg_all; D.xchg_width = b->xchg_width; D.world = b->world;
  hipLaunchKernelGGL(k_ba_decide_global, dim3(1), dim3(256), 0, b->stream, D);
  HIPCHK(hipGetLastError());
  return 0;
}
// the linearisation kernel's view of a decision block when the points are sharded: pack only
static BADecide packOnly(dmvio_hip_ba* b, const BADecide& D) {
  BADecide Dp = D;
  Dp.mode = 3; Dp.publish = 0; Dp.xchg_local = b->d_xchg_local; Dp.xchg_width = b->xchg_width;
  return Dp;
}

// FullSystem::linearizeAll (FullSystemOptimize.cpp:150-218) — returns the energy sum; updates the newest frame's energy threshold
// (setNewFrameEnergyTH, on the device by the kernel's last workgroup) unless keep_threshold.
// defer: enqueue only — the caller waits for a LATER ticket of the same stream (kernels it enqueues behind this one) and then picks the results up with linearizePickUp
static void linearizePickUp(dmvio_hip_ba* b, double* energy, bool keep_threshold) {
  *energy = b->h_res->E[0];
  if (!keep_threshold) { b->H.fr[b->H.F - 1].frameEnergyTH = b->h_res->th[0]; b->th_pending = false; }
}
static int linearizeAll(dmvio_hip_ba* b, bool fix, double* energy, int table_mode = 0 /* 0 re-upload in place, 1 new state, 2 switch back */, bool keep_threshold = false,
                        bool defer = false) {
  BAHost& H = b->H;
  if (int r = uploadWindowTables(b, table_mode == 1, table_mode == 2)) return r;  // precalc of the current state
  if (int r = uploadThresholds(b)) return r;
  const bool shard = sharded(b);
  if (shard) { if (int r = ensureExchange(b)) return r; }
  // the decision pass (last workgroup of the linearisation) publishes the ticket the host polls; the applyRes kernel of a fix-linearisation runs behind it on the
  // same stream and nothing it writes is read by the host, so no stream synchronisation is needed for it either
  const BADecide D = makeDecide(b, 0, !keep_threshold, true);
  BA_PROF(b, 0);
  BA_LAUNCH_LINEARIZE(b, b->W, b->keep_fullJ ? b->d_fullJ : (float*)nullptr, (const unsigned char*)nullptr, shard ? packOnly(b, D) : D, BA_GATE_ALWAYS, 0, b->dyn_cur, 0, b->x_none, 0);
  BA_PROF(b, 1);
  if (fix) { hipLaunchKernelGGL(k_ba_apply, dim3((H.R + 255) / 256), dim3(256), 0, b->stream, H.R, b->Rs, (const unsigned char*)nullptr, 1); b->fullJ_applied = b->keep_fullJ; }   // + linearizeAll(true)'s removal of inactive residuals
  HIPCHK(hipGetLastError());
  if (shard) { if (int r = decideGlobal(b, D)) return r; }
  if (defer) return 0;
  if (int r = waitTicket(b, D.ticket)) return r;
  linearizePickUp(b, energy, keep_threshold);
  return 0;
}
static int applyRes(dmvio_hip_ba* b) {
  hipLaunchKernelGGL(k_ba_apply, dim3((b->H.R + 255) / 256), dim3(256), 0, b->stream, b->H.R, b->Rs, (const unsigned char*)nullptr, 0);
  HIPCHK(hipGetLastError());
  b->fullJ_applied = b->keep_fullJ;
  return 0;
}
// accumulateAF + accumulateSCF + adjoint stitching on the device; result in h_sys
static int accumulateViews(dmvio_hip_ba* b, const BARes& Rs, const BAPoints& P, bool wait = true, int gate = BA_GATE_ALWAYS);
static int accumulateWait(dmvio_hip_ba* b);
// apply_first: applyRes_Reductor(true) fused into the per-point sums; gate: the whole chain only runs when the last accept test says so
static int accumulateLin(dmvio_hip_ba* b, bool backup_points, bool apply_first);
static int accumulate(dmvio_hip_ba* b, bool backup_points = false, bool wait = true, bool sums_fresh = false, bool apply_first = false, int gate = BA_GATE_ALWAYS) {
  if (linAny(b)) {
    if (!wait || gate != BA_GATE_ALWAYS) return failmsg("ba: a graph with residuals kept linearised is accumulated synchronously");
    return accumulateLin(b, backup_points, apply_first);
  }
  // (the flag describes the DEVICE's state: a gated chain may not run its applyRes, and a chain without one leaves the buffer as the last linearisation wrote it)
  if (apply_first) b->fullJ_applied = gate == BA_GATE_ALWAYS ? b->keep_fullJ : false;
  if (!sums_fresh) hipLaunchKernelGGL(k_ba_point_sums, dim3(b->n_pt8_blocks), dim3(256), 0, b->stream, b->W, b->P, b->Rs, backup_points ? 1 : 0, apply_first ? 1 : 0,
                                      (const BACtl*)b->d_ctl, gate, (backup_points && b->vio) ? b->h_idepth_backup : (float*)nullptr);
  BA_PROF(b, 2);
  return accumulateViews(b, b->Rs, b->P, wait, gate);
}
// the accumulation + stitching launches over an arbitrary (records, activity, per-point sums) view of the graph
static int accumulateViews(dmvio_hip_ba* b, const BARes& RsV, const BAPoints& PV, bool wait, int gate) {
  BAHost& H = b->H;
  if (b->adj_dirty) { if (int r = uploadAdjoints(b)) return r; }
  const int F = H.F, F2 = F * F, n = H.n();
  hipStream_t s = b->stream;
  {
    AccumArgs A;
    A.F = F; A.N = H.N; A.nsTop = b->nsTop; A.nsD = b->nsD; A.nsC = b->nsC;
    A.top_begin = b->d_top_begin; A.top_members = b->d_top_members; A.scd_begin = b->d_scd_begin; A.scd_members = b->d_scd_members;
    A.accTop = b->d_accTop; A.accD = b->d_accD; A.accE = b->d_accE; A.accC = b->d_accC; A.numTop = b->d_numTop; A.numD = b->d_numD;
    const int nblk = b->nsC + F2 * b->nsTop + (F2 * F * b->nsD + 3) / 4;
    A.ticks = nullptr;
    if (b->timing) {
      if (b->accTicksBlocks != nblk) { if (b->d_accTicks) hipFree(b->d_accTicks); HIPCHK(hipMalloc((void**)&b->d_accTicks, sizeof(long long) * 2 * nblk)); b->accTicksBlocks = nblk; }
      A.ticks = b->d_accTicks;
    }
    hipLaunchKernelGGL(k_ba_accumulate, dim3(nblk), dim3(256), 0, s, A, RsV, PV, (const BACtl*)b->d_ctl, gate);
  }
  BA_PROF(b, 3);
  if (F <= BA_MAXF) hipLaunchKernelGGL((k_ba_stitch<BA_MAXF>), dim3(F + F2), dim3(64 * F), sizeof(StitchWave) * F, s, F, b->nsTop, b->nsD, b->d_accTop, b->d_numTop, b->d_accD, b->d_numD,
                                       b->d_accE, b->d_adHost, b->d_adTarget, b->SB, (const BACtl*)b->d_ctl, gate);
  else hipLaunchKernelGGL((k_ba_stitch<BA_MAXF_CAP>), dim3(F + F2), dim3(64 * F), sizeof(StitchWave) * F, s, F, b->nsTop, b->nsD, b->d_accTop, b->d_numTop, b->d_accD, b->d_numD,
                          b->d_accE, b->d_adHost, b->d_adTarget, b->SB, (const BACtl*)b->d_ctl, gate);
  BA_PROF(b, 4);
  const int tot = 2 * (n * n + n);
  if (!sharded(b)) {
    b->acc_ticket = ++b->ticket;
    if (F <= BA_MAXF) hipLaunchKernelGGL((k_ba_stitch_gather<BA_MAXF>), dim3((tot + 256) / 256), dim3(256), 0, s, F, b->nsC, b->d_accC, b->SB, b->d_numTop, F2 * b->nsTop, b->h_sys,
                                         b->d_ctl, gate, b->h_res, b->acc_ticket);
    else hipLaunchKernelGGL((k_ba_stitch_gather<BA_MAXF_CAP>), dim3((tot + 256) / 256), dim3(256), 0, s, F, b->nsC, b->d_accC, b->SB, b->d_numTop, F2 * b->nsTop, b->h_sys,
                            b->d_ctl, gate, b->h_res, b->acc_ticket);
  } else {
    // this rank's part of the system stays in HBM, is summed over the ranks in place and only then published to the host
    if (gate != BA_GATE_ALWAYS) return failmsg("sharded accumulation cannot be gated: every rank must enter the collective");
    if (F <= BA_MAXF) hipLaunchKernelGGL((k_ba_stitch_gather<BA_MAXF>), dim3((tot + 256) / 256), dim3(256), 0, s, F, b->nsC, b->d_accC, b->SB, b->d_numTop, F2 * b->nsTop, b->d_sys,
                                         b->d_ctl, gate, (BAHostRes*)nullptr, 0u);
    else hipLaunchKernelGGL((k_ba_stitch_gather<BA_MAXF_CAP>), dim3((tot + 256) / 256), dim3(256), 0, s, F, b->nsC, b->d_accC, b->SB, b->d_numTop, F2 * b->nsTop, b->d_sys,
                            b->d_ctl, gate, (BAHostRes*)nullptr, 0u);
    HIPCHK(hipGetLastError());
    if (int r = commAllReduceSum(b, b->d_sys, (size_t)tot + 1)) return r;
    b->acc_ticket = ++b->ticket;
    hipLaunchKernelGGL(k_ba_publish_sys, dim3(1), dim3(1024), 0, s, (const double*)b->d_sys, b->h_sys, tot + 1, b->h_res, b->acc_ticket);
  }
  HIPCHK(hipGetLastError());
  BA_PROF(b, 5);
  return wait ? accumulateWait(b) : 0;
}
// solveSystemF's three accumulations for a graph with residuals kept linearised (EnergyFunctional.cpp:846-852: accumulateAF_MT, accumulateLF_MT, accumulateSCF_MT).  One
// accumulation kernel serves a (host,target) bucket's top block and its Schur terms from ONE activity flag; here the three reference passes see different sets — addPoint<0>
// the active residuals that are not linearised, addPoint<1> those that are (with the res_toZeroF + J delta record), the Schur side every active one — so the kernel chain
// runs three times over three views and each pass contributes its part: [HL | bL] -> BAHost::HLraw / bLraw, [HA | bA | resInA] and [Hsc | bsc] -> h_sys as always.
static int accumulateLin(dmvio_hip_ba* b, bool backup_points, bool apply_first) {
  BAHost& H = b->H;
  hipStream_t s = b->stream;
  const int R = H.R, N = H.N, n = H.n(), tot = 2 * (n * n + n);
  if (apply_first) { if (int r = applyRes(b)) return r; }
  std::vector<float> adHT;
  H.adHTdeltaF(adHT);   // EnergyFunctional::setDeltaF at the current state
  HIPCHK(b->bounce.h2d(b->d_adHTdelta, adHT.data(), sizeof(float) * adHT.size(), s));
  const float4 cd = make_float4(H.cDeltaF[0], H.cDeltaF[1], H.cDeltaF[2], H.cDeltaF[3]);
  hipLaunchKernelGGL(k_ba_lin_records, dim3((R + 255) / 256), dim3(256), 0, s, b->W, b->P, b->Rs, (const float*)b->d_fullJ, (const unsigned char*)b->d_lin, (const float*)b->d_rtz,
                     (const float*)b->d_adHTdelta, cd, b->d_linRec, b->d_linActive, b->d_topActive);
  hipLaunchKernelGGL(k_ba_lin_point_sums, dim3((N + 255) / 256), dim3(256), 0, s, b->W, b->P, (const float*)b->d_linRec, (const unsigned char*)b->d_linActive, b->d_lHdd, b->d_lbd, b->d_lHcd);
  hipLaunchKernelGGL(k_ba_point_sums, dim3(b->n_pt8_blocks), dim3(256), 0, s, b->W, b->P, b->Rs, backup_points ? 1 : 0, 0, (const BACtl*)b->d_ctl, (int)BA_GATE_ALWAYS,
                     (backup_points && b->vio) ? b->h_idepth_backup : (float*)nullptr);
  HIPCHK(hipGetLastError());
  // L pass
  BARes RsL = b->Rs; RsL.rec[0] = RsL.rec[1] = b->d_linRec; RsL.active = b->d_linActive; RsL.lin = nullptr;
  if (int r = accumulateViews(b, RsL, b->P, true)) return r;
  H.HLraw.assign(b->h_sys, b->h_sys + (size_t)n * n); H.bLraw.assign(b->h_sys + (size_t)n * n, b->h_sys + (size_t)n * n + n);
  // A pass
  BARes RsA = b->Rs; RsA.active = b->d_topActive;
  if (int r = accumulateViews(b, RsA, b->P, true)) return r;
  std::vector<double> top(b->h_sys, b->h_sys + (size_t)n * n + n);
  const int resInA = H.resInA;
  // Schur pass
  if (int r = accumulateViews(b, b->Rs, b->P, true)) return r;
  memcpy(b->h_sys, top.data(), sizeof(double) * top.size());
  b->h_sys[tot] = (double)resInA; H.resInA = resInA;
  return 0;
}
// calcLEnergyPt (EnergyFunctional.cpp:349-409) for the residuals kept linearised: (2 res_toZeroF + J delta) . (J delta), summed the way the reference does — an Accumulator11
// (four fp32 lanes; its 1k / 1M levels never fill within a run) per run of 50 points (IndexThreadReduce::reduce with step 50, EnergyFunctional.cpp:427-428), two 4-lane
// updates per residual in point / residual order, the runs' fp32 totals added in double.  The points' own prior term deltaF^2 priorF is zero: idepth_zero follows idepth.
static double linEnergy(dmvio_hip_ba* b) {
  const BAHost& H = b->H;
  std::vector<float> adHT;
  H.adHTdeltaF(adHT);
  double A = 0;
  for (int p0 = 0; p0 < H.N; p0 += 50) {
    float d1[4] = {0, 0, 0, 0};
    for (int pi = p0; pi < std::min(H.N, p0 + 50); pi++)
      for (int ri = b->h_res_begin[pi]; ri < b->h_res_begin[pi + 1]; ri++) {
        if (!b->h_lin[ri] || !b->h_linAct[ri]) continue;
        const float* J = &b->h_linJ[(size_t)ri * 74];
        const float* rtz = &b->h_rtz[(size_t)ri * 8];
        const float* dp = &adHT[(size_t)(b->h_host[pi] + H.F * b->h_target[ri]) * 8];
        const float dd = 0.0f;
        float sx = 0, sy = 0, cx = 0, cy = 0;
        for (int i = 0; i < 6; i++) { sx += J[8 + i] * dp[i]; sy += J[14 + i] * dp[i]; }
        for (int i = 0; i < 4; i++) { cx += J[20 + i] * H.cDeltaF[i]; cy += J[24 + i] * H.cDeltaF[i]; }
        const float Jp_delta_x = sx + cx + J[28] * dd, Jp_delta_y = sy + cy + J[29] * dd;
        for (int i = 0; i < 8; i += 4)
          for (int k = 0; k < 4; k++) {
            float Jdelta = J[30 + i + k] * Jp_delta_x;
            Jdelta = Jdelta + J[38 + i + k] * Jp_delta_y;
            Jdelta = Jdelta + J[46 + i + k] * dp[6];
            Jdelta = Jdelta + J[54 + i + k] * dp[7];
            float r0 = rtz[i + k];
            r0 = r0 + r0;
            r0 = r0 + Jdelta;
            d1[k] = d1[k] + Jdelta * r0;
          }
      }
    A += (double)(d1[0] + d1[1] + d1[2] + d1[3]);
  }
  return A;
}
// EnergyFunctional::calcLEnergyF_MT (EnergyFunctional.cpp:414-431)
static double calcLEnergy(dmvio_hip_ba* b) { return b->n_lin > 0 ? b->H.calcLEnergyFrames() + linEnergy(b) : b->H.calcLEnergyFrames(); }
// ... of a window whose points are sharded over ranks: the frame / calibration part is the same on every rank, the linearised residuals' term is this rank's points' share —
// summed over the ranks by one more (one-double) all-reduce, which every rank enters (n_lin_global decides, not the rank's own count)
static int calcLEnergyR(dmvio_hip_ba* b, double* out) {
  if (!sharded(b) || b->n_lin_global == 0) { *out = calcLEnergy(b); return 0; }
  double l = b->n_lin > 0 ? linEnergy(b) : 0.0;
  if (!b->d_red1) { if (dalloc(b, &b->d_red1, 1)) return -1; }
  HIPCHK(b->bounce.h2d(b->d_red1, &l, sizeof(double), b->stream));
  if (int r = commAllReduceSum(b, b->d_red1, 1)) return r;
  HIPCHK(b->bounce.d2h(&l, b->d_red1, sizeof(double), b->stream));
  HIPCHK(b->bounce.finish(b->stream));
  *out = b->H.calcLEnergyFrames() + l;
  return 0;
}
// the gather kernel publishes per workgroup (BAHostRes::gticket): wait until every slot shows the chain's ticket
static int waitGather(dmvio_hip_ba* b, const unsigned int ticket, const int nblk) {
  volatile unsigned int* slots = b->h_res->gticket;
  unsigned long long spins = 0;
  for (;;) {
    bool all = true;
    for (int k = nblk - 1; k >= 0; k--) if (slots[k] != ticket) { all = false; break; }
    if (all) break;
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {
      const hipError_t q = hipStreamQuery(b->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail("BA kernel chain", __FILE__, __LINE__, q);
      if (q == hipSuccess) {
        for (int k = 0; k < nblk; k++) if (slots[k] != ticket) return failmsg("BA accumulation chain finished without publishing its system");
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}
static int accumulateWait(dmvio_hip_ba* b) {
  const int n = b->H.n(), tot = 2 * (n * n + n);
  if (!sharded(b)) { if (int r = waitGather(b, b->acc_ticket, (tot + 256) / 256)) return r; }
  else if (int r = waitTicket(b, b->acc_ticket)) return r;   // h_sys = [H_A | b_A | H_sc | b_sc | resInA]
  b->H.resInA = (int)b->h_sys[tot];
  return 0;
}
static int resubstitute(dmvio_hip_ba* b, const std::vector<double>& x, bool apply_step = false) {
  float xc[4];
  std::vector<float> xAd;
  b->H.prepareResubstitute(x, xc, xAd);
  ResubArgs X;
  memcpy(X.xc, xc, sizeof(xc));
  memset(X.xAd, 0, sizeof(X.xAd));
  memcpy(X.xAd, xAd.data(), sizeof(float) * xAd.size());
  if (b->H.F <= BA_MAXF) hipLaunchKernelGGL((k_ba_resubstitute<BA_MAXF>), dim3(b->n_pt8_blocks), dim3(256), 0, b->stream, b->W, b->P, b->Rs, baNarrow<ResubArgsT<BA_MAXF>>(X), apply_step ? 1 : 0);
  else hipLaunchKernelGGL((k_ba_resubstitute<BA_MAXF_CAP>), dim3(b->n_pt8_blocks), dim3(256), 0, b->stream, b->W, b->P, b->Rs, X, apply_step ? 1 : 0);
  HIPCHK(hipGetLastError());
  return 0;
}
// mode 0 backup, 1 step from backup, 2 restore; the step-norm sums (mode 1) are only fetched when the caller asks for them
static int pointStep(dmvio_hip_ba* b, int mode, float fac, float* sumID, float* sumNID) {
  hipLaunchKernelGGL(k_ba_point_step, dim3(b->n_pt_blocks), dim3(256), 0, b->stream, b->H.N, b->P, mode, fac, (mode == 1 && sumID && sumNID) ? b->d_spart : (float*)nullptr);
  HIPCHK(hipGetLastError());
  if (mode == 1 && sumID && sumNID) {
    HIPCHK(hipMemcpyAsync(b->h_spart, b->d_spart, sizeof(float) * 2 * b->n_pt_blocks, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    float a = 0, d = 0;
    for (int i = 0; i < b->n_pt_blocks; i++) { a += b->h_spart[2 * i]; d += b->h_spart[2 * i + 1]; }
    *sumID = a / b->H.N; *sumNID = d / b->H.N;
  }
  return 0;
}

extern "C" {

dmvio_hip_ba* dmvio_hip_ba_create(dmvio_hip_ctx* ctx) {
  if (!ctx) { failmsg("ba_create: null ctx"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { failmsg("ba_create: hipSetDevice failed"); return nullptr; }
  dmvio_hip_ba* b = new dmvio_hip_ba();
  b->ctx = ctx;
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { failmsg("ba_create: stream creation failed"); delete b; return nullptr; }
  b->own_stream = true;
  b->H.w = ctx->w; b->H.h = ctx->h;
  {
    // everything a first keyframe would otherwise allocate: the first arena chunk, the pinned buffers the kernels store into, the staging area of the uploads
    constexpr int NMAXF = 4 + 8 * BA_MAXF_CAP;
    char* base = nullptr;
    bool ok = hipMalloc((void**)&base, BA_ARENA_CHUNK) == hipSuccess && hipMemset(base, 0, BA_ARENA_CHUNK) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
    if (ok) { b->arena.chunks.push_back(std::make_pair(base, BA_ARENA_CHUNK)); b->arena.used.push_back(0); }
    ok = ok && hipHostMalloc((void**)&b->h_sys, sizeof(double) * (2 * (NMAXF * NMAXF + NMAXF) + 1), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&b->h_res, sizeof(BAHostRes), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&b->h_frameTH, sizeof(float) * BA_MAXF_CAP, hipHostMallocDefault) == hipSuccess;
    for (int k = 0; k < 2 && ok; k++) ok = hipHostMalloc((void**)&b->h_pre[k], sizeof(BAPrecalc) * BA_MAXF_CAP * BA_MAXF_CAP, hipHostMallocDefault) == hipSuccess;
    b->cap_idepth_backup = 8192;
    ok = ok && hipHostMalloc((void**)&b->h_idepth_backup, sizeof(float) * b->cap_idepth_backup, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
    b->cap_spart = 4096;
    ok = ok && hipHostMalloc((void**)&b->h_spart, sizeof(float) * b->cap_spart, hipHostMallocDefault) == hipSuccess;
    size_t off = 0;
    ok = ok && b->bounce.reserve((size_t)4 << 20, b->stream, &off) == hipSuccess;
    b->bounce.used = 0;
    if (!ok) { failmsg("ba_create: device / pinned allocation failed"); freeAll(b); hipStreamDestroy(b->stream); delete b; return nullptr; }
  }
  // the only environment variable the library reads: a host-side timing printout of the GN loop (stderr), no effect on what is computed.  The accumulation order is
  // chosen with dmvio_hip_ba_set_accumulators alone (tests/test_capi_cpu.py greps csrc/ for other getenv calls)
  if (const char* e = getenv("DMVIO_HIP_BA_TIMING")) b->timing = atoi(e) != 0;
  return b;
}
void dmvio_hip_ba_batch_destroy(struct dmvio_hip_ba_batch* B);
void dmvio_hip_ba_destroy(dmvio_hip_ba* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  if (b->own_batch) { dmvio_hip_ba_batch_destroy(b->own_batch); b->own_batch = nullptr; }
  for (int k = 0; k < 2; k++) for (int i = 0; i < dmvio_hip_ba::COMM_EVS; i++) for (int j = 0; j < 2; j++) if (b->comm_ev[k][i][j]) hipEventDestroy(b->comm_ev[k][i][j]);
  hipStreamSynchronize(b->stream);
  if (b->timing && b->tm_graph_n > 0)
    fprintf(stderr, "[dmvio_hip_ba] set_graph over %ld calls (us/call): drain + arena memset=%.1f host lists=%.1f allocation=%.1f uploads=%.1f pinned buffers + slot table=%.1f adjoints + wait=%.1f\n", b->tm_graph_n,
            b->tm_graph[0] / b->tm_graph_n, b->tm_graph[1] / b->tm_graph_n, b->tm_graph[2] / b->tm_graph_n, b->tm_graph[3] / b->tm_graph_n, b->tm_graph[4] / b->tm_graph_n, b->tm_graph[5] / b->tm_graph_n);
  if (b->timing && b->tm.n > 0) {
    const char* names[8] = {"backup+prepare", "host solve", "step+precalc+energies+args", "launch", "wait for the decision", "accepted: sums+accumulate+stitch / rejected: relinearise", "-", "-"};
    fprintf(stderr, "[dmvio_hip_ba] GN iteration, host clock between phases (no synchronisation added) over %ld iterations (us/iter):", b->tm.n);
    for (int i = 0; i < 8; i++) fprintf(stderr, " %s=%.1f", names[i], b->tm.t[i] / b->tm.n);
    fprintf(stderr, "\n");
    if (b->d_accTicks) {   // block timeline of the LAST k_ba_accumulate launch (wall_clock64 = 100 MHz)
      std::vector<long long> tk(2 * (size_t)b->accTicksBlocks);
      hipMemcpy(tk.data(), b->d_accTicks, sizeof(long long) * tk.size(), hipMemcpyDeviceToHost);
      long long t0 = tk[0];
      for (int i = 0; i < b->accTicksBlocks; i++) t0 = std::min(t0, tk[2 * i]);
      const int F2 = b->H.F * b->H.F, lim[3] = {b->nsC, b->nsC + F2 * b->nsTop, b->accTicksBlocks};
      const char* cls[3] = {"calib", "top+E", "accD x4"};
      int i0 = 0;
      for (int c = 0; c < 3; c++) {
        double dmax = 0, dsum = 0, smax = 0, emax = 0; int argmax = -1;
        for (int i = i0; i < lim[c]; i++) {
          const double d = (tk[2 * i + 1] - tk[2 * i]) * 0.01, st = (tk[2 * i] - t0) * 0.01, en = (tk[2 * i + 1] - t0) * 0.01;
          dsum += d; if (d > dmax) { dmax = d; argmax = i - i0; } smax = std::max(smax, st); emax = std::max(emax, en);
        }
        fprintf(stderr, "[dmvio_hip_ba]   k_ba_accumulate %s blocks=%d: duration mean %.1f max %.1f us (block %d), latest start %.1f, latest end %.1f us\n", cls[c],
                lim[c] - i0, dsum / std::max(1, lim[c] - i0), dmax, argmax, smax, emax);
        i0 = lim[c];
      }
    }
  }
  if (b->d_accTicks) hipFree(b->d_accTicks);
  freeAll(b);
  if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
  delete b;
}

// Accumulation order (takes effect with the next dmvio_hip_ba_set_graph): k partial accumulators per bucket, 1 <= k <= 8.
int dmvio_hip_ba_set_accumulators(dmvio_hip_ba* b, int k) {
  if (!b || k < 1 || k > 8) return failmsg("ba_set_accumulators: 1 <= k <= 8");
  BA_LOCK(b);
  b->nsTop = k; b->nsD = std::min(k, 4); b->nsC = k == 1 ? 1 : 4 * k;
  b->graph_ready = false;
  return 0;
}
// The 74-float RawResidualJacobian per residual (dmvio_hip_ba_get_jacobians) is written by the linearisation only while this is on.
int dmvio_hip_ba_keep_jacobians(dmvio_hip_ba* b, int on) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  b->keep_fullJ = on != 0;
  return 0;
}

// ---- communicator of a window whose points are sharded over ranks (include/dmvio_hip.h)
static int setComm(dmvio_hip_ba* b, ncclComm_t comm, const dmvio_hip_comm_callbacks* cb, int rank, int world) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  if (world == 0 || (!comm && !cb)) { b->world = 0; b->rank = 0; b->nccl = nullptr; b->comm_cb = dmvio_hip_comm_callbacks{}; b->sys_ready = false; b->sums_fresh = false; return 0; }
  if (world < 1 || rank < 0 || rank >= world) return failmsg("ba_set_comm: 0 <= rank < world");
  if (b->n_lin > 0) return failmsg("ba_set_comm: the graph already carries residuals kept linearised — set the communicator first, dmvio_hip_ba_fix_linearization is collective on a sharded window");
  if (cb && (!cb->allreduce_sum_f64 || !cb->allgather)) return failmsg("ba_set_comm_callbacks: both callbacks are required");
  if (comm) {
    RCCL_READY();
    int n = 0, r = -1;
    NCCLCHK(rccl().commCount(comm, &n));
    NCCLCHK(rccl().commUserRank(comm, &r));
    if (n != world || r != rank) return failmsg("ba_set_comm: rank / world do not match the communicator");
  }
  b->rank = rank; b->world = world; b->nccl = comm;
  b->comm_cb = cb ? *cb : dmvio_hip_comm_callbacks{};
  b->xchg_width = 0;            // agreed on at the next linearisation
  b->sys_ready = false; b->sums_fresh = false;
  return 0;
}
int dmvio_hip_ba_set_comm(dmvio_hip_ba* b, void* nccl_comm, int rank, int world) { return setComm(b, (ncclComm_t)nccl_comm, nullptr, rank, world); }
// Measurement of the sharded iteration's two collectives (RCCL transport): on = 1 starts collecting HIP events around them on the BA stream (and clears the counters); the
// query waits for the stream and returns mean microseconds of [all-reduce of the packed system, all-gather of the decision records] over the first 64 of each since then,
// and how many of each were issued in total.  What a multi-GPU scaling line needs to explain itself (a window sharded over N GPUs pays both per iteration).
int dmvio_hip_ba_comm_timing(dmvio_hip_ba* b, int on) {
  if (!b) return failmsg("ba: null handle");
  BA_LOCK(b);
  b->comm_timing = on != 0; b->comm_n[0] = b->comm_n[1] = 0; b->comm_total[0] = b->comm_total[1] = 0;
  return 0;
}
int dmvio_hip_ba_comm_times(dmvio_hip_ba* b, double mean_us2[2], long issued2[2]) {
  if (!b || !mean_us2 || !issued2) return failmsg("ba_comm_times: null argument");
  BA_LOCK(b);
  HIPCHK(hipSetDevice(b->ctx->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (int k = 0; k < 2; k++) {
    double sum = 0;
    for (int i = 0; i < b->comm_n[k]; i++) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, b->comm_ev[k][i][0], b->comm_ev[k][i][1])); sum += 1e3 * ms; }
    mean_us2[k] = b->comm_n[k] ? sum / b->comm_n[k] : 0.0;
    issued2[k] = b->comm_total[k];
  }
  return 0;
}
int dmvio_hip_ba_partition_points(const int* host, int N, int world, double max_imbalance, int* owner_out) {
  if (N < 0 || world < 1 || (N > 0 && (!host || !owner_out))) return failmsg("dmvio_hip_ba_partition_points: bad argument");
  if (max_imbalance <= 0) max_imbalance = 1.25;
  int nkf = 0;
  for (int i = 0; i < N; i++) {
    if (host[i] < 0) return failmsg("dmvio_hip_ba_partition_points: negative host keyframe index");
    nkf = std::max(nkf, host[i] + 1);
  }
  if (world == 1) { for (int i = 0; i < N; i++) owner_out[i] = 0; return 0; }
  std::vector<long long> counts(nkf, 0), load(world, 0);
  for (int i = 0; i < N; i++) counts[host[i]]++;
  std::vector<int> order(nkf), owner(nkf, 0);
  for (int k = 0; k < nkf; k++) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return counts[a] > counts[b]; });   // largest keyframe first, ties in index order
  for (int kf : order) {
    int r = 0;
    for (int q = 1; q < world; q++) if (load[q] < load[r]) r = q;                                       // the lightest rank, the lowest one among equals
    owner[kf] = r; load[r] += counts[kf];
  }
  const long long heaviest = *std::max_element(load.begin(), load.end());
  if ((double)heaviest > max_imbalance * std::max((double)N / world, 1.0)) {
    for (int r = 0; r < world; r++) {
      const long long lo = ((long long)N * r) / world, hi = ((long long)N * (r + 1)) / world;
      for (long long i = lo; i < hi; i++) owner_out[i] = r;
    }
    return 1;
  }
  for (int i = 0; i < N; i++) owner_out[i] = owner[host[i]];
  return 0;
}
int dmvio_hip_ba_set_comm_callbacks(dmvio_hip_ba* b, const dmvio_hip_comm_callbacks* cb, int rank, int world) { return setComm(b, nullptr, cb, rank, world); }
int dmvio_hip_comm_unique_id(unsigned char id128[128]) {
  if (!id128) return failmsg("null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  RCCL_READY();
  ncclUniqueId id;
  NCCLCHK(rccl().getUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}
int dmvio_hip_comm_init_rank(dmvio_hip_ctx* ctx, const unsigned char id128[128], int rank, int world, void** out) {
  if (!ctx || !id128 || !out) return failmsg("null argument");
  HIPCHK(hipSetDevice(ctx->device));
  RCCL_READY();
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t comm = nullptr;
  NCCLCHK(rccl().commInitRank(&comm, world, id, rank));
  *out = (void*)comm;
  return 0;
}
// ---- hypothesis-parallel FullSystem::trackNewCoarse (include/dmvio_hip.h): the per-try records of dmvio_hip_tracker_track_new_coarse summed over the ranks
int dmvio_hip_tracker_set_comm(dmvio_hip_tracker* t, void* nccl_comm, int rank, int world) {
  dmvio_hip_ctx* c = dmv_tracker_ctx(t);
  if (!c) return failmsg("null tracker");
  const bool force1 = world == 1 && nccl_comm && dmv_tracker_debug_split1(t);   // test hook (dmvio_hip_tracker_debug_split_single_rank), see dmv_tracker_set_exchange
  if (!nccl_comm || (world <= 1 && !force1)) return dmv_tracker_set_exchange(t, nullptr, 0, 0);
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  RCCL_READY();
  int n = 0, r = -1;
  NCCLCHK(rccl().commCount(comm, &n));
  NCCLCHK(rccl().commUserRank(comm, &r));
  if (n != world || r != rank) return failmsg("tracker_set_comm: rank / world do not match the communicator");
  // 20 doubles per hypothesis: a few KB, staged through a device buffer that stays with the exchange (and through the context's pinned staging area) for RCCL on the context's stream
  struct XchgBuf { double* d = nullptr; size_t cap = 0; int device = 0; ~XchgBuf() { if (d) { hipSetDevice(device); hipFree(d); } } };
  std::shared_ptr<XchgBuf> st = std::make_shared<XchgBuf>();
  st->device = c->device;
  return dmv_tracker_set_exchange(t, [c, comm, st](double* buf, size_t count) -> int {
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (count > st->cap) {
      if (st->d) { HIPCHK(hipFree(st->d)); st->d = nullptr; st->cap = 0; }
      HIPCHK(hipMalloc((void**)&st->d, sizeof(double) * 2 * count));
      st->cap = 2 * count;
    }
    HIPCHK(c->bounce.h2d(st->d, buf, sizeof(double) * count, c->stream));
    const ncclResult_t nr = rccl().allReduce(st->d, st->d, count, ncclDouble, ncclSum, comm, c->stream);
    if (nr != ncclSuccess) return failmsg(std::string("RCCL: ") + rccl().getErrorString(nr) + " in the hypothesis exchange");
    HIPCHK(c->bounce.d2h(buf, st->d, sizeof(double) * count, c->stream));
    HIPCHK(c->bounce.finish(c->stream));
    return 0;
  }, rank, world);
}
int dmvio_hip_tracker_set_comm_callbacks(dmvio_hip_tracker* t, const dmvio_hip_comm_callbacks* cb, int rank, int world) {
  if (!dmv_tracker_ctx(t)) return failmsg("null tracker");
  if (!cb || world <= 1) return dmv_tracker_set_exchange(t, nullptr, 0, 0);
  if (!cb->allreduce_sum_f64) return failmsg("tracker_set_comm_callbacks: allreduce_sum_f64 is required");
  const dmvio_hip_comm_callbacks k = *cb;
  return dmv_tracker_set_exchange(t, [k](double* buf, size_t count) -> int {
    return k.allreduce_sum_f64(k.user, buf, count) == 0 ? 0 : failmsg("comm callback allreduce_sum_f64 failed");
  }, rank, world);
}
// ncclCommCount / ncclCommUserRank of a communicator: what RCCL itself says about the group (bench.py prints it in the N > 1 line)
int dmvio_hip_comm_info(void* comm, int* n_ranks, int* rank) {
  if (!comm) return failmsg("null communicator");
  RCCL_READY();
  int n = 0, r = -1;
  NCCLCHK(rccl().commCount((ncclComm_t)comm, &n));
  NCCLCHK(rccl().commUserRank((ncclComm_t)comm, &r));
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  return 0;
}
int dmvio_hip_comm_destroy(void* comm) {
  if (!comm) return 0;
  RCCL_READY();
  NCCLCHK(rccl().commDestroy((ncclComm_t)comm));
  return 0;
}

// The stream the mapping side enqueues on (default: a stream owned by the handle).  NULL restores an own stream.
int dmvio_hip_ba_set_stream(dmvio_hip_ba* b, void* stream) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  HIPCHK(hipSetDevice(b->ctx->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->own_stream) { HIPCHK(hipStreamDestroy(b->stream)); b->own_stream = false; }
  if (stream) b->stream = (hipStream_t)stream;
  else { HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)); b->own_stream = true; }
  return 0;
}

int dmvio_hip_ba_max_frames(void) { return BA_MAXF_CAP; }
int dmvio_hip_ba_set_window(dmvio_hip_ba* b, int F, const int* slots, const double* pose7_w2c, const double* aff_ab, const float* exposures,
                            const int* frameIDs, const double fxfycxcy[4]) {
  if (!b || !slots || !pose7_w2c || !fxfycxcy) return failmsg("ba_set_window: null argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  if (F < 1 || F > BA_MAXF_CAP) return failmsg("ba_set_window: 1 <= F <= " + std::to_string(BA_MAXF_CAP) + " keyframes (dmvio_hip_ba_max_frames)");
  // a threshold still on its way from the previous window's last accepted step (th_ticket follows the decision) belongs to THAT window: it must neither be waited for
  // after the host-coherent record is cleared (set_graph) nor land in the new window's newest keyframe
  b->th_pending = false;
  BAHost& H = b->H;
  H.F = F;
  H.calibInitScaled(fxfycxcy);
  for (int f = 0; f < F; f++) {
    if (slots[f] < 0 || slots[f] >= b->ctx->n_slots) return failmsg("ba_set_window: frame slot out of range");
    if (int r = dmv_ensure_row_major(b->ctx, slots[f])) return r;   // the linearisation taps row-major level-0 planes (a slot the batched raw build left in tiles is converted once)
    BAFrameHost& fr = H.fr[f];
    fr = BAFrameHost();
    fr.slot = slots[f];
    fr.ab_exposure = exposures ? exposures[f] : 1.0f;
    fr.frameID = frameIDs ? frameIDs[f] : f;
    fr.evalPT = poseFrom7(pose7_w2c + 7 * f);
    const double a = aff_ab ? aff_ab[2 * f] : 0.0, bb = aff_ab ? aff_ab[2 * f + 1] : 0.0;
    double st[10] = {0, 0, 0, 0, 0, 0, (1.0f / 10.0f) * a, (1.0f / 1000.0f) * bb, 0, 0};   // setEvalPT_scaled (HessianBlocks.h:220-227)
    for (int i = 0; i < 10; i++) { fr.step[i] = 0; fr.state_backup[i] = 0; }
    BAHost::frameSetState(fr, st);
    BAHost::frameSetStateZero(fr, fr.state);
  }
  const int n = H.n();
  H.HM.assign((size_t)n * n, 0.0); H.bM.assign(n, 0.0);
  for (int f = 0; f < F; f++) H.frameTakeData(H.fr[f]);
  H.setAdjointsF();
  H.setPrecalcValues();
  b->graph_ready = false;
  b->th_dirty = true;
  return 0;
}

int dmvio_hip_ba_set_marg_prior(dmvio_hip_ba* b, const double* HM, const double* bM) {
  if (!b || !HM || !bM) return failmsg("ba_set_marg_prior: null argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  const int n = b->H.n();
  b->H.HM.assign(HM, HM + (size_t)n * n); b->H.bM.assign(bM, bM + n);
  return 0;
}

// FullSystem::flagPointsForRemoval's relinearisation (FullSystem.cpp:829-859) + EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:678-742)
int dmvio_hip_ba_marginalize_points(dmvio_hip_ba* b, const unsigned char* candidates, unsigned char* decision, double* Hadd, double* badd, int* resInM, int update_prior) {
  if (!b || !b->graph_ready) return failmsg("ba_marginalize_points: window / graph not set");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  if (!candidates || !decision) return failmsg("ba_marginalize_points: null argument");
  dmvio_hip_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  BAHost& H = b->H;
  const int N = H.N, R = H.R, n = H.n();
  hipStream_t s = b->stream;
  const float setting_minIdepthH_marg = 50, setting_idepthFixPriorMargFac = 600 * 600;
  const double setting_margWeightFac = 0.5 * 0.5;
  // deltas at the current state (EnergyFunctional::setDeltaF, EnergyFunctional.cpp:175-198)
  std::vector<float> adHT;
  H.adHTdeltaF(adHT);
  HIPCHK(b->bounce.h2d(b->d_cand, candidates, N, s));
  HIPCHK(b->bounce.h2d(b->d_adHTdelta, adHT.data(), sizeof(float) * adHT.size(), s));
  // (adHT is staged in the pinned bounce: no wait)
  if (int r = uploadWindowTables(b)) return r;
  if (int r = uploadThresholds(b)) return r;
  {
    const bool fullJ_applied = b->fullJ_applied;
    const BADecide D = makeDecide(b, -1, false, false);   // masked relinearisation: no energy / threshold / accept pass
    BA_LAUNCH_LINEARIZE(b, b->W, b->d_fullJ, (const unsigned char*)b->d_cand, D, BA_GATE_ALWAYS, 0, b->dyn_cur, 0, b->x_none, 0);
    b->fullJ_applied = fullJ_applied;   // the candidates' rows are rewritten and applied right below, the others untouched
  }
  hipLaunchKernelGGL(k_ba_apply, dim3((R + 255) / 256), dim3(256), 0, s, R, b->Rs, (const unsigned char*)b->d_cand, 0);
  if (b->n_lin > 0) {   // FullSystem.cpp:840-843: a candidate point's residuals are relinearised with isLinearized = false
    for (int ri = 0; ri < R; ri++) if (candidates[b->h_point[ri]] && b->h_lin[ri]) { b->h_lin[ri] = 0; b->n_lin--; }
    HIPCHK(b->bounce.h2d(b->d_lin, b->h_lin.data(), R, s));
    if (!sharded(b)) b->n_lin_global = b->n_lin;
    if (b->n_lin == 0 && !sharded(b)) { b->Rs.lin = nullptr; b->P.lHdd = b->P.lbd = b->P.lHcd = nullptr; b->P.HcdAF = nullptr; b->H.HLraw.clear(); b->H.bLraw.clear(); }
  }
  hipLaunchKernelGGL(k_ba_marg_decide, dim3((N + 255) / 256), dim3(256), 0, s, N, b->d_cand, b->P.idepth_hessian, setting_minIdepthH_marg, b->d_decision);
  const float4 cd = make_float4(H.cDeltaF[0], H.cDeltaF[1], H.cDeltaF[2], H.cDeltaF[3]);
  hipLaunchKernelGGL(k_ba_fix_linearization, dim3((R + 255) / 256), dim3(256), 0, s, b->W, b->P, b->Rs, b->d_fullJ, b->d_decision, b->d_adHTdelta, cd, b->d_margRec,
                     b->d_margActive, (float*)nullptr);
  hipLaunchKernelGGL(k_ba_marg_point_sums, dim3(b->n_pt_blocks), dim3(256), 0, s, b->W, b->P, b->Rs, b->d_margRec, b->d_margActive, b->d_decision,
                     setting_idepthFixPriorMargFac, b->d_mHdiF, b->d_mbdSumF, b->d_mHcd);
  HIPCHK(hipGetLastError());
  BARes RsV = b->Rs; RsV.rec[0] = RsV.rec[1] = b->d_margRec; RsV.active = b->d_margActive;
  BAPoints PV = b->P; PV.HdiF = b->d_mHdiF; PV.bdSumF = b->d_mbdSumF; PV.Hcd = b->d_mHcd;
  const int resInA_keep = H.resInA;
  if (int r = accumulateViews(b, RsV, PV)) return r;
  const int nres = H.resInA;
  H.resInA = resInA_keep;
  HIPCHK(b->bounce.d2h(decision, b->d_decision, N, s));
  HIPCHK(b->bounce.finish(s));
  const double* M = b->h_sys; const double* Mb = M + (size_t)n * n; const double* Msc = Mb + n; const double* Mbsc = Msc + (size_t)n * n;
  if (H.HM.size() != (size_t)n * n) { H.HM.assign((size_t)n * n, 0.0); H.bM.assign(n, 0.0); }
  for (size_t k = 0; k < (size_t)n * n; k++) { const double v = setting_margWeightFac * (M[k] - Msc[k]); if (Hadd) Hadd[k] = v; if (update_prior) H.HM[k] += v; }
  for (int k = 0; k < n; k++) { const double v = setting_margWeightFac * (Mb[k] - Mbsc[k]); if (badd) badd[k] = v; if (update_prior) H.bM[k] += v; }
  if (resInM) *resInM = nres;
  return 0;
}

static int setGraphImpl(dmvio_hip_ba* b, int N, const int* host, const float* u, const float* v, const float* idepth, const float* color8,
                        const float* weights8, const unsigned char* hasDepthPrior, int R, const int* res_point, const int* res_target, bool wait);
// synchronous on return, like every entry point that takes caller arrays; dmvio_hip_ba_set_graph_from is the stream-ordered form
int dmvio_hip_ba_set_graph(dmvio_hip_ba* b, int N, const int* host, const float* u, const float* v, const float* idepth, const float* color8,
                           const float* weights8, const unsigned char* hasDepthPrior, int R, const int* res_point, const int* res_target) {
  return setGraphImpl(b, N, host, u, v, idepth, color8, weights8, hasDepthPrior, R, res_point, res_target, true);
}
static int setGraphImpl(dmvio_hip_ba* b, int N, const int* host, const float* u, const float* v, const float* idepth, const float* color8,
                        const float* weights8, const unsigned char* hasDepthPrior, int R, const int* res_point, const int* res_target, bool wait) {
  if (!b || !host || !u || !v || !idepth || !color8 || !weights8 || !res_point || !res_target) return failmsg("ba_set_graph: null argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  BAHost& H = b->H;
  if (H.F < 1) return failmsg("ba_set_graph: set_window first");
  if (N < 1 || R < 1) return failmsg("ba_set_graph: empty graph");
  dmvio_hip_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  double tg0 = b->timing ? nowUs() : 0, tg1;
#define SG_PH(i) do { if (b->timing) { tg1 = nowUs(); b->tm_graph[i] += tg1 - tg0; tg0 = tg1; } } while (0)
  HIPCHK(hipStreamSynchronize(b->stream));
  b->th_pending = false;   // the stream is drained and h_res is cleared below (th_ticket restarts at 0 while b->ticket keeps counting): nothing of the old graph may be awaited
  freeDevice(b);
  // residuals kept linearised belong to the graph they were linearised in
  b->n_lin = 0; b->n_lin_global = 0; b->Rs.lin = nullptr; b->P.lHdd = b->P.lbd = b->P.lHcd = nullptr; b->P.HcdAF = nullptr; b->d_lin = nullptr; b->d_red1 = nullptr; b->H.HLraw.clear(); b->H.bLraw.clear();
  b->h_lin.clear(); b->h_linAct.clear(); b->h_linJ.clear(); b->h_rtz.clear(); b->fullJ_applied = false;
  // the arena of the previous graph, cleared for this one on the handle's stream; the uploads and kernels below follow on the same stream: no wait
  for (size_t k = 0; k < b->arena.chunks.size(); k++) if (b->arena.used[k]) { HIPCHK(hipMemsetAsync(b->arena.chunks[k].first, 0, b->arena.used[k], b->stream)); b->arena.used[k] = 0; }   // what the previous graph used, not the whole chunk
  SG_PH(0);
  b->arena.cur = 0; b->arena.off = 0;
  struct ArenaScope { dmvio_hip_ba* b; ~ArenaScope() { b->arena.on = false; } } arenaScope{b};
  b->arena.on = true;
  const int F = H.F, F2 = F * F;
  H.N = N; H.R = R;
  b->h_host.assign(host, host + N); b->h_point.assign(res_point, res_point + R); b->h_target.assign(res_target, res_target + R);
  b->h_newest.clear();
  for (int ri = 0; ri < R; ri++) if (res_target[ri] == H.F - 1) b->h_newest.push_back(ri);
  // residuals must be grouped by point, points in window order
  b->h_res_begin.assign(N + 1, 0);
  for (int ri = 0; ri < R; ri++) {
    const int p = res_point[ri];
    if (p < 0 || p >= N || res_target[ri] < 0 || res_target[ri] >= F) return failmsg("ba_set_graph: residual index out of range");
    if (ri > 0 && p < res_point[ri - 1]) return failmsg("ba_set_graph: residuals must be sorted by point");
    if (host[p] < 0 || host[p] >= F || host[p] == res_target[ri]) return failmsg("ba_set_graph: bad host / target");
    b->h_res_begin[p + 1]++;
  }
  for (int p = 0; p < N; p++) b->h_res_begin[p + 1] += b->h_res_begin[p];
  // bucket member lists in the reference's traversal order (points, then residuals of the point)
  std::vector<int> top_begin(F2 + 1, 0), top_members(R), scd_begin(F2 * F + 1, 0);
  for (int ri = 0; ri < R; ri++) top_begin[host[res_point[ri]] + F * res_target[ri] + 1]++;
  for (int k = 0; k < F2; k++) top_begin[k + 1] += top_begin[k];
  { std::vector<int> cur(top_begin.begin(), top_begin.end() - 1); for (int ri = 0; ri < R; ri++) top_members[cur[host[res_point[ri]] + F * res_target[ri]]++] = ri; }
  size_t npairs = 0;
  for (int p = 0; p < N; p++) { const size_t k = b->h_res_begin[p + 1] - b->h_res_begin[p]; npairs += k * k; }
  // The Schur buckets' member lists are built on the device (k_ba_scd_*: ba_kernels.hpp) when no point observes a keyframe twice — always, in the reference's graphs (one
  // PointFrameResidual per point and target, FullSystem.cpp:1248-1262); a graph that does is served by the host loops below, as every graph was before.
  std::vector<int> pt_first(F, N), pt_last(F, -1);
  bool scd_on_device = true;
  for (int p = 0; p < N && scd_on_device; p++) {
    pt_first[host[p]] = std::min(pt_first[host[p]], p); pt_last[host[p]] = std::max(pt_last[host[p]], p);
    unsigned int seen = 0;
    for (int ri = b->h_res_begin[p]; ri < b->h_res_begin[p + 1]; ri++) { const unsigned int bit = 1u << res_target[ri]; if (seen & bit) scd_on_device = false; seen |= bit; }
  }
  std::vector<int> scd_members;
  if (!scd_on_device) {
    scd_members.resize(3 * npairs);
    for (int p = 0; p < N; p++)
      for (int r1 = b->h_res_begin[p]; r1 < b->h_res_begin[p + 1]; r1++)
        for (int r2 = b->h_res_begin[p]; r2 < b->h_res_begin[p + 1]; r2++) scd_begin[(host[p] + F * res_target[r1]) + res_target[r2] * F2 + 1]++;
    for (int k = 0; k < F2 * F; k++) scd_begin[k + 1] += scd_begin[k];
    std::vector<int> cur(scd_begin.begin(), scd_begin.end() - 1);
    for (int p = 0; p < N; p++)
      for (int r1 = b->h_res_begin[p]; r1 < b->h_res_begin[p + 1]; r1++)
        for (int r2 = b->h_res_begin[p]; r2 < b->h_res_begin[p + 1]; r2++) {
          const int k = (host[p] + F * res_target[r1]) + res_target[r2] * F2;
          const int o = cur[k]++;
          scd_members[3 * o] = r1; scd_members[3 * o + 1] = r2; scd_members[3 * o + 2] = p;
        }
  }
  SG_PH(1);
  // ---- device arrays
  int *d_host, *d_res_begin, *d_point, *d_target;
  float *d_u, *d_v, *d_color, *d_weights, *d_prior;
  if (dalloc(b, &d_host, N) || dalloc(b, &d_res_begin, N + 1) || dalloc(b, &d_point, R) || dalloc(b, &d_target, R) || dalloc(b, &d_u, N) || dalloc(b, &d_v, N) ||
      dalloc(b, &d_color, (size_t)N * 8) || dalloc(b, &d_weights, (size_t)N * 8) || dalloc(b, &d_prior, N)) return -1;
  BAPoints& P = b->P; BARes& Rs = b->Rs;
  if (dalloc(b, &P.idepth, N) || dalloc(b, &P.idepth_zero, N) || dalloc(b, &P.idepth_backup, N) || dalloc(b, &P.step, N) || dalloc(b, &P.Hdd, N) || dalloc(b, &P.bd, N) ||
      dalloc(b, &P.Hcd, (size_t)N * 4) || dalloc(b, &P.HdiF, N) || dalloc(b, &P.bdSumF, N) || dalloc(b, &P.idepth_hessian, N) ||
      dalloc(b, &b->d_cand, N) || dalloc(b, &b->d_decision, N) || dalloc(b, &b->d_mHdiF, N) || dalloc(b, &b->d_mbdSumF, N) || dalloc(b, &b->d_mHcd, (size_t)N * 4) ||
      dalloc(b, &b->d_margRec, (size_t)R * REC_FLOATS) || dalloc(b, &b->d_margActive, R) || dalloc(b, &b->d_adHTdelta, (size_t)F2 * 8)) return -1;
  if (dalloc(b, &Rs.removed, R) || dalloc(b, &Rs.state, R) || dalloc(b, &Rs.newState, R) || dalloc(b, &Rs.active, R) || dalloc(b, &Rs.which, R) || dalloc(b, &Rs.energy, R) || dalloc(b, &Rs.newEnergy, R) ||
      dalloc(b, &Rs.center, (size_t)R * 3) || dalloc(b, &Rs.rec[0], (size_t)R * REC_FLOATS) || dalloc(b, &Rs.rec[1], (size_t)R * REC_FLOATS)) return -1;
  P.host = d_host; P.u = d_u; P.v = d_v; P.color = d_color; P.weights = d_weights; P.priorF = d_prior; P.res_begin = d_res_begin;
  Rs.point = d_point; Rs.target = d_target;
  std::vector<float> prior(N, 0.0f);
  if (hasDepthPrior) for (int p = 0; p < N; p++) prior[p] = hasDepthPrior[p] ? H.S.idepthFixPrior : 0.0f;   // EFPoint::takeData
  hipStream_t s = b->stream;
  SG_PH(2);
  // the uploads of this call are staged and leave merged (DmvBounce::begin_group): the nine point / residual tables and the two copies of idepth are one copy, ...
  struct GroupScope { DmvBounce& bn; ~GroupScope() { bn.grouping = false; bn.group.clear(); } } groupScope{b->bounce};   // an error return drops what was only staged
  b->bounce.begin_group();
  HIPCHK(b->bounce.h2d(d_host, host, sizeof(int) * N, s));
  HIPCHK(b->bounce.h2d(d_res_begin, b->h_res_begin.data(), sizeof(int) * (N + 1), s));
  HIPCHK(b->bounce.h2d(d_point, res_point, sizeof(int) * R, s));
  HIPCHK(b->bounce.h2d(d_target, res_target, sizeof(int) * R, s));
  HIPCHK(b->bounce.h2d(d_u, u, sizeof(float) * N, s));
  HIPCHK(b->bounce.h2d(d_v, v, sizeof(float) * N, s));
  HIPCHK(b->bounce.h2d(d_color, color8, sizeof(float) * N * 8, s));
  HIPCHK(b->bounce.h2d(d_weights, weights8, sizeof(float) * N * 8, s));
  HIPCHK(b->bounce.h2d(d_prior, prior.data(), sizeof(float) * N, s));
  HIPCHK(b->bounce.h2d(P.idepth, idepth, sizeof(float) * N, s));
  HIPCHK(b->bounce.h2d(P.idepth_zero, idepth, sizeof(float) * N, s));
  b->pre_half = 0;
  if (dalloc(b, &b->d_pre2, 2 * (size_t)F2) || dalloc(b, &b->d_adHost, (size_t)F2 * 64) || dalloc(b, &b->d_adTarget, (size_t)F2 * 64) || dalloc(b, &b->d_top_begin, F2 + 1) ||
      dalloc(b, &b->d_top_members, R) || dalloc(b, &b->d_scd_begin, F2 * F + 1) || dalloc(b, &b->d_scd_members, 3 * npairs) || dalloc(b, &b->d_accTop, (size_t)F2 * 96 * b->nsTop) ||
      dalloc(b, &b->d_accD, (size_t)F2 * F * 64 * b->nsD) || dalloc(b, &b->d_accE, (size_t)F2 * 40 * b->nsTop) || dalloc(b, &b->d_accC, 20 * b->nsC) || dalloc(b, &b->d_numTop, F2 * b->nsTop) || dalloc(b, &b->d_numD, F2 * F * b->nsD)) return -1;
  HIPCHK(b->bounce.h2d(b->d_top_begin, top_begin.data(), sizeof(int) * (F2 + 1), s));
  HIPCHK(b->bounce.h2d(b->d_top_members, top_members.data(), sizeof(int) * R, s));
  if (!scd_on_device) {
    HIPCHK(b->bounce.h2d(b->d_scd_begin, scd_begin.data(), sizeof(int) * (F2 * F + 1), s));
    HIPCHK(b->bounce.h2d(b->d_scd_members, scd_members.data(), sizeof(int) * 3 * npairs, s));
    HIPCHK(b->bounce.end_group(s));
  } else {
    // behind the uploads of host / point / target on the same stream: residual table, counts, scan, members
    int *d_ridx, *d_cnt, *d_first, *d_last;
    if (dalloc(b, &d_ridx, (size_t)N * F) || dalloc(b, &d_cnt, (size_t)F2 * F) || dalloc(b, &d_first, F) || dalloc(b, &d_last, F)) return -1;
    HIPCHK(b->bounce.h2d(d_first, pt_first.data(), sizeof(int) * F, s));
    HIPCHK(b->bounce.h2d(d_last, pt_last.data(), sizeof(int) * F, s));
    HIPCHK(b->bounce.end_group(s));   // the kernels below read what was staged so far
    HIPCHK(hipMemsetAsync(d_ridx, 0xff, sizeof(int) * (size_t)N * F, s));
    hipLaunchKernelGGL(k_ba_scd_ridx, dim3((R + 255) / 256), dim3(256), 0, s, R, F, (const int*)d_point, (const int*)d_target, d_ridx);
    hipLaunchKernelGGL(k_ba_scd_lists, dim3((F2 + 3) / 4, F), dim3(256), 0, s, F, (const int*)d_first, (const int*)d_last, (const int*)d_host, (const int*)d_ridx, 0, d_cnt, (const int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(k_ba_scd_scan, dim3(1), dim3(1024), 0, s, F2 * F, (const int*)d_cnt, b->d_scd_begin);
    hipLaunchKernelGGL(k_ba_scd_lists, dim3((F2 + 3) / 4, F), dim3(256), 0, s, F, (const int*)d_first, (const int*)d_last, (const int*)d_host, (const int*)d_ridx, 1, d_cnt, (const int*)b->d_scd_begin, b->d_scd_members);
    HIPCHK(hipGetLastError());
  }
  SG_PH(3);
  b->bounce.begin_group();   // ... the slot table and the adjoint tables another
  StitchBufs& SB = b->SB;
  if (dalloc(b, &SB.topHH, (size_t)F * 64) || dalloc(b, &SB.topTT, (size_t)F2 * 64) || dalloc(b, &SB.topHT, (size_t)F2 * 64) || dalloc(b, &SB.topHC, (size_t)F * 32) ||
      dalloc(b, &SB.topTC, (size_t)F2 * 32) || dalloc(b, &SB.topBH, (size_t)F * 8) || dalloc(b, &SB.topBT, (size_t)F2 * 8) || dalloc(b, &SB.topCC, (size_t)F * 20) ||
      dalloc(b, &SB.scHH, (size_t)F2 * 64) || dalloc(b, &SB.scTT, (size_t)F2 * F * 64) || dalloc(b, &SB.scTH, (size_t)F2 * 64) || dalloc(b, &SB.scHT, (size_t)F2 * F * 64) ||
      dalloc(b, &SB.scHC, (size_t)F2 * 32) ||
      dalloc(b, &SB.scTC, (size_t)F2 * 32) || dalloc(b, &SB.scBH, (size_t)F2 * 8) || dalloc(b, &SB.scBT, (size_t)F2 * 8)) return -1;
  const int n = H.n(), tot = 2 * (n * n + n);
  b->n_lin_blocks = (R + LIN_RES_PER_BLOCK - 1) / LIN_RES_PER_BLOCK; b->n_pt_blocks = (N + 255) / 256; b->n_pt8_blocks = (N + PT_GROUPS_PER_BLOCK - 1) / PT_GROUPS_PER_BLOCK;
  // energy partials (doubles) and the per-residual energies with outliers (floats) share one pinned allocation the kernel stores into
  b->n_epart = std::max(b->n_lin_blocks, F2 * 8);
  if (dalloc(b, &b->d_spart, 2 * b->n_pt_blocks) || dalloc(b, &b->d_fullJ, (size_t)R * 74)) return -1;
  // what the host reads back every iteration (the stitched system, the energy partials, the per-residual energies) is written by the
  // kernels straight into pinned host memory: no copy engine between the last kernel and the host's wait
  constexpr int NMAXF = 4 + 8 * BA_MAXF_CAP;
  if (!b->h_sys) HIPCHK(hipHostMalloc((void**)&b->h_sys, sizeof(double) * (2 * (NMAXF * NMAXF + NMAXF) + 1), hipHostMallocCoherent | hipHostMallocMapped));   // polled: host-coherent; sized for BA_MAXF_CAP keyframes once
  if (dalloc(b, &b->d_sys, (size_t)tot + 1)) return -1;
  b->xchg_width = 0; b->d_xchg_local = b->d_xchg_all = nullptr;
  b->pending_reject = false; b->pending_trace = -1;
  if (!b->h_res) HIPCHK(hipHostMalloc((void**)&b->h_res, sizeof(BAHostRes), hipHostMallocCoherent | hipHostMallocMapped));
  memset(b->h_res, 0, sizeof(BAHostRes));
  if (!b->h_frameTH) HIPCHK(hipHostMalloc((void**)&b->h_frameTH, sizeof(float) * BA_MAXF_CAP, hipHostMallocDefault));
  if ((size_t)N > b->cap_idepth_backup) {
    if (b->h_idepth_backup) HIPCHK(hipHostFree(b->h_idepth_backup));
    b->cap_idepth_backup = (size_t)N + (size_t)N / 2 + 256;
    HIPCHK(hipHostMalloc((void**)&b->h_idepth_backup, sizeof(float) * b->cap_idepth_backup, hipHostMallocCoherent | hipHostMallocMapped));
  }
  if (dalloc(b, &b->d_ctl, 1) || dalloc(b, &b->d_frameTH, BA_MAXF_CAP) || dalloc(b, &b->d_epart, (size_t)b->n_epart) || dalloc(b, &b->d_newestSlot, (size_t)R) || dalloc(b, &b->d_newestE, b->h_newest.size()) ||
      dalloc(b, &b->d_newEnergyWO, (size_t)R)) return -1;
  {
    std::vector<int> slot(R, -1);
    for (size_t k = 0; k < b->h_newest.size(); k++) slot[b->h_newest[k]] = (int)k;
    if (R) HIPCHK(b->bounce.h2d(b->d_newestSlot, slot.data(), sizeof(int) * R, s));   // staged (the vector may go): asynchronous like the other uploads
    Rs.newestSlot = b->d_newestSlot; Rs.newestE = b->d_newestE;
  }
  b->pre_static_valid = false;
  b->th_dirty = true; b->sys_ready = false;
  for (int k = 0; k < 2; k++) if (!b->h_pre[k]) HIPCHK(hipHostMalloc((void**)&b->h_pre[k], sizeof(BAPrecalc) * BA_MAXF_CAP * BA_MAXF_CAP, hipHostMallocDefault));
  Rs.newEnergyWO = b->d_newEnergyWO;
  if ((size_t)2 * b->n_pt_blocks > b->cap_spart) {
    if (b->h_spart) HIPCHK(hipHostFree(b->h_spart));
    b->cap_spart = (size_t)2 * b->n_pt_blocks + 64;
    HIPCHK(hipHostMalloc((void**)&b->h_spart, sizeof(float) * b->cap_spart, hipHostMallocDefault));
  }
  SG_PH(4);
  if (int r = uploadAdjoints(b)) return r;
  HIPCHK(b->bounce.end_group(s));
  // everything above is staged in pinned memory and enqueued on the handle's stream, and so is whatever uses it: dmvio_hip_ba_set_graph_from does not wait (the 0.1 ms the
  // uploads take overlap the caller's next calls — frame states, prior, the first host-side steps of optimize)
  HIPCHK(hipGetLastError());
  if (wait) HIPCHK(hipStreamSynchronize(s));
  SG_PH(5);
#undef SG_PH
  b->tm_graph_n++;
  b->graph_ready = true;
  return 0;
}

// dmvio_hip_ba_set_graph from a resident graph (capi_graph.hip): the mirror flattened in makeIDX order — compact records, a few tens of microseconds — instead of arrays
// the caller rebuilt from its pointer graph.  The window (dmvio_hip_ba_set_window) must have as many keyframes as the graph.  The scratch arrays stay with the handle.
static int linImport(dmvio_hip_ba* b, const unsigned char* flags, const float* J74, const float* rtz, int* n_linearized);
int dmvio_hip_ba_set_graph_from(dmvio_hip_ba* b, dmvio_hip_graph* g) {
  if (!b || !g) return failmsg("ba_set_graph_from: null argument");
  BA_LOCK(b);
  int N = 0, R = 0;
  unsigned long long flat_version = 0;
  {
    std::lock_guard<std::mutex> lg(g->mu);
    if ((int)g->frames.size() != b->H.F) return failmsg("ba_set_graph_from: the graph has " + std::to_string(g->frames.size()) + " keyframes, the window " + std::to_string(b->H.F));
    if (g->nDangling) return failmsg("ba_set_graph_from: " + std::to_string(g->nDangling) + " residuals still target a removed keyframe (their dropResidual has not been forwarded)");
    N = g->nPoints; R = g->nRes;
    if (N < 1 || R < 1) return failmsg("ba_set_graph: empty graph");
    flat_version = g->version;
    auto& S = b->gscratch;
    S.host.resize(N); S.u.resize(N); S.v.resize(N); S.idepth.resize(N); S.color.resize(8 * (size_t)N); S.weights.resize(8 * (size_t)N); S.prior.resize(N);
    S.res_point.resize(R); S.res_target.resize(R);
    const bool anyLin = g->nLin > 0;   // residuals that arrive linearised: their flags / Jacobians / res_toZeroF in flat order (only then)
    if (anyLin) { S.lin.assign(R, 0); S.linJ.resize((size_t)R * 74); S.linRtz.resize((size_t)R * 8); } else S.lin.clear();
    int pi = 0, ri = 0;
    for (int f = 0; f < (int)g->frames.size(); f++)
      for (const DmvGraphPoint& P : g->frames[f]) {
        S.host[pi] = f; S.u[pi] = P.u; S.v[pi] = P.v; S.idepth[pi] = P.idepth; S.prior[pi] = P.prior;
        memcpy(&S.color[8 * (size_t)pi], P.color, sizeof(P.color)); memcpy(&S.weights[8 * (size_t)pi], P.weights, sizeof(P.weights));
        for (int k = 0; k < P.nres; k++, ri++) {
          S.res_point[ri] = pi; S.res_target[ri] = P.target[k];
          if (anyLin && P.lin[k] >= 0) {
            S.lin[ri] = 1;
            memcpy(&S.linJ[(size_t)ri * 74], g->linPool[P.lin[k]].J, sizeof(float) * 74); memcpy(&S.linRtz[(size_t)ri * 8], g->linPool[P.lin[k]].res_toZeroF, sizeof(float) * 8);
          }
        }
        pi++;
      }
  }
  const auto& S = b->gscratch;
  if (int r = setGraphImpl(b, N, S.host.data(), S.u.data(), S.v.data(), S.idepth.data(), S.color.data(), S.weights.data(), S.prior.data(), R, S.res_point.data(), S.res_target.data(), false)) return r;
  if (!S.lin.empty()) { if (int r = linImport(b, S.lin.data(), S.linJ.data(), S.linRtz.data(), nullptr)) { b->graph_ready = false; return r; } }
  // only a window that WAS built makes its flat order the one dmvio_hip_graph_set_idepths accepts values in (until the structure changes).  The uploads are stream-ordered:
  // an asynchronous copy error of this call is reported by the next BA call that synchronises
  std::lock_guard<std::mutex> lg(g->mu);
  g->flat_version = flat_version;
  return 0;
}

#define BA_READY_LOCKED(b) do { if (!(b)->graph_ready) return failmsg("ba: set_window + set_graph first"); (b)->sums_fresh = false; (b)->sys_ready = false; HIPCHK(hipSetDevice((b)->ctx->device)); } while (0)
// first statement of an entry point: null check, the handle's lock for the whole call (declares a guard in the function's scope), then the state checks
#define BA_READY(b) if (!(b)) return failmsg("ba: null handle"); BA_LOCK(b); BA_READY_LOCKED(b)

// buffers of the residuals kept linearised (first use on a graph)
static int linAlloc(dmvio_hip_ba* b) {
  const int R = b->H.R, N = b->H.N;
  if (b->d_lin) return 0;
  if (dalloc(b, &b->d_lin, R) || dalloc(b, &b->d_linMask, R) || dalloc(b, &b->d_linActive, R) || dalloc(b, &b->d_topActive, R) || dalloc(b, &b->d_rtz, (size_t)R * 8) ||
      dalloc(b, &b->d_linRec, (size_t)R * REC_FLOATS) || dalloc(b, &b->d_lHdd, N) || dalloc(b, &b->d_lbd, N) || dalloc(b, &b->d_lHcd, (size_t)N * 4) ||
      dalloc(b, &b->d_HcdAF, (size_t)N * 4) || dalloc(b, &b->d_linE, (size_t)R * 8)) return -1;
  b->h_lin.assign(R, 0);
  HIPCHK(hipMemsetAsync(b->d_lin, 0, R, b->stream));   // (the arena hands out zeroed memory after set_graph; a second graph's first use must not depend on it)
  return 0;
}
// after the device flags changed: the host copies calcLEnergyPt reads (the energy terms are host arithmetic in this library), the counts, the kernels' views
static int linFinish(dmvio_hip_ba* b, int* n_linearized) {
  const int R = b->H.R;
  hipStream_t s = b->stream;
  b->h_linAct.resize(R); b->h_linJ.resize((size_t)R * 74); b->h_rtz.resize((size_t)R * 8);
  HIPCHK(b->bounce.d2h(b->h_lin.data(), b->d_lin, R, s));
  HIPCHK(b->bounce.d2h(b->h_linAct.data(), b->Rs.active, R, s));
  HIPCHK(b->bounce.d2h(b->h_linJ.data(), b->d_fullJ, sizeof(float) * 74 * R, s));
  HIPCHK(b->bounce.d2h(b->h_rtz.data(), b->d_rtz, sizeof(float) * 8 * R, s));
  HIPCHK(b->bounce.finish(s));
  b->n_lin = 0;
  for (int ri = 0; ri < R; ri++) if (b->h_lin[ri]) b->n_lin++;
  b->n_lin_global = b->n_lin;
  if (sharded(b)) {   // collective: every rank of the window calls this (with the flags of its own residuals); the ranks' counts are summed
    double cnt = (double)b->n_lin;
    if (!b->d_red1) { if (dalloc(b, &b->d_red1, 1)) return -1; }
    HIPCHK(b->bounce.h2d(b->d_red1, &cnt, sizeof(double), s));
    if (int r = commAllReduceSum(b, b->d_red1, 1)) return r;
    HIPCHK(b->bounce.d2h(&cnt, b->d_red1, sizeof(double), s));
    HIPCHK(b->bounce.finish(s));
    b->n_lin_global = (long long)(cnt + 0.5);
  }
  if (linAny(b)) { b->Rs.lin = b->d_lin; b->P.lHdd = b->d_lHdd; b->P.lbd = b->d_lbd; b->P.lHcd = b->d_lHcd; b->P.HcdAF = b->d_HcdAF; }
  if (n_linearized) *n_linearized = b->n_lin;
  return 0;
}
// residuals that arrive linearised: flags (R), J74 (R x 74) and res_toZeroF (R x 8) indexed by residual; only the flagged rows travel
static int linImport(dmvio_hip_ba* b, const unsigned char* flags, const float* J74, const float* rtz, int* n_linearized) {
  const int R = b->H.R;
  hipStream_t s = b->stream;
  std::vector<int> idx;
  for (int ri = 0; ri < R; ri++) if (flags[ri]) idx.push_back(ri);
  if (idx.empty() && !sharded(b)) { if (n_linearized) *n_linearized = b->n_lin; return 0; }
  if (int r = linAlloc(b)) return r;
  const int n = (int)idx.size();
  if (n > 0) {
    std::vector<float> packed((size_t)n * 82);
    for (int k = 0; k < n; k++) {
      memcpy(&packed[(size_t)k * 82], J74 + (size_t)idx[k] * 74, sizeof(float) * 74);
      memcpy(&packed[(size_t)k * 82 + 74], rtz + (size_t)idx[k] * 8, sizeof(float) * 8);
    }
    int* d_idx = nullptr; float* d_packed = nullptr;
    if (dalloc(b, &d_idx, n) || dalloc(b, &d_packed, (size_t)n * 82)) return -1;
    HIPCHK(b->bounce.h2d(d_idx, idx.data(), sizeof(int) * n, s));
    HIPCHK(b->bounce.h2d(d_packed, packed.data(), sizeof(float) * packed.size(), s));
    hipLaunchKernelGGL(k_ba_lin_import, dim3((n + 255) / 256), dim3(256), 0, s, n, (const int*)d_idx, (const float*)d_packed, b->Rs, b->d_fullJ, b->d_rtz, b->d_linRec, b->d_lin);
    HIPCHK(hipGetLastError());
  }
  b->sums_fresh = false; b->sys_ready = false;
  return linFinish(b, n_linearized);
}

int dmvio_hip_ba_set_residual_flags(dmvio_hip_ba* b, int R, const unsigned char* isLinearized) {
  if (!b || !isLinearized) return failmsg("ba_set_residual_flags: null argument");
  BA_LOCK(b);
  if (!b->graph_ready) return failmsg("ba: set_window + set_graph first");
  if (R != b->H.R) return failmsg("ba_set_residual_flags: R differs from the graph's residual count");
  for (int ri = 0; ri < R; ri++)
    if (isLinearized[ri]) {
      b->graph_ready = false;   // refused as a whole: nothing may run on a graph whose linearised energy term (accumulateLF_MT) would be missing
      return failmsg("ba_set_residual_flags: residual " + std::to_string(ri) + " arrives linearised (EFResidual::isLinearized): its frozen Jacobian and res_toZeroF, which "
                     "accumulateLF_MT / addPoint<1> (EnergyFunctional.cpp:223-233, AccumulatedTopHessian.cpp:84-98) need, do not come with a bare flag — hand them over with "
                     "dmvio_hip_ba_set_linearized_residuals (or dmvio_hip_graph_set_residual_linearized + dmvio_hip_ba_set_graph_from); the graph is refused");
    }
  return 0;
}

// EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:85-113) for the active residuals with res_mask != 0, at the current state, on the resident graph: from then on
// they stay out of activeResiduals (FullSystemOptimize.cpp:436-446) and enter every system through accumulateLF_MT / addPoint<1> and the energy through calcLEnergyPt, until
// the next dmvio_hip_ba_set_graph (or until their point is marginalised).  Needs the Jacobians of the applied linearisation: dmvio_hip_ba_keep_jacobians(ba, 1) before the
// dmvio_hip_ba_optimize / dmvio_hip_ba_linearize(fix) that precedes this call.  n_linearized: residuals of the graph that are linearised after the call.
int dmvio_hip_ba_fix_linearization(dmvio_hip_ba* b, int R, const unsigned char* res_mask, int* n_linearized) {
  if (!b || !res_mask) return failmsg("ba_fix_linearization: null argument");
  BA_LOCK(b);
  BA_READY_LOCKED(b);
  BAHost& H = b->H;
  if (R != H.R) return failmsg("ba_fix_linearization: R differs from the graph's residual count");
  if (!b->keep_fullJ || !b->fullJ_applied)
    return failmsg("ba_fix_linearization: the Jacobians of the applied linearisation are not resident — call dmvio_hip_ba_keep_jacobians(ba, 1) before the optimize / "
                   "linearize(fix) + apply that precedes this call");
  hipStream_t s = b->stream;
  if (int r = linAlloc(b)) return r;
  H.setPrecalcValues();
  std::vector<float> adHT;
  H.adHTdeltaF(adHT);
  HIPCHK(b->bounce.h2d(b->d_linMask, res_mask, R, s));
  HIPCHK(b->bounce.h2d(b->d_adHTdelta, adHT.data(), sizeof(float) * adHT.size(), s));
  const float4 cd = make_float4(H.cDeltaF[0], H.cDeltaF[1], H.cDeltaF[2], H.cDeltaF[3]);
  hipLaunchKernelGGL(k_ba_lin_fix, dim3((R + 255) / 256), dim3(256), 0, s, b->W, b->P, b->Rs, (const float*)b->d_fullJ, (const unsigned char*)b->d_linMask, (const float*)b->d_adHTdelta, cd,
                     b->d_lin, b->d_rtz, b->d_linRec);
  HIPCHK(hipGetLastError());
  return linFinish(b, n_linearized);
}
// EFResidual::isLinearized / J / res_toZeroF of residuals that were linearised ELSEWHERE (a graph handed over with dmvio_hip_ba_set_graph whose residuals carry the flag:
// EnergyFunctionalStructs.h:63-87), for the flagged residuals: rows of J74 (R x 74, the RawResidualJacobian in dmvio_hip_ba_get_full_jacobians' layout) and res_toZeroF
// (R x 8); the rows of the others are not read.  They become what dmvio_hip_ba_fix_linearization leaves behind.
int dmvio_hip_ba_set_linearized_residuals(dmvio_hip_ba* b, int R, const unsigned char* isLinearized, const float* J74, const float* res_toZeroF, int* n_linearized) {
  if (!b || !isLinearized || !J74 || !res_toZeroF) return failmsg("ba_set_linearized_residuals: null argument");
  BA_LOCK(b);
  BA_READY_LOCKED(b);
  if (R != b->H.R) return failmsg("ba_set_linearized_residuals: R differs from the graph's residual count");
  return linImport(b, isLinearized, J74, res_toZeroF, n_linearized);
}
// ... and back: EFResidual::isLinearized / J / res_toZeroF of the graph's residuals as they stand (linearised here or handed over) — what an adapter writes back into the
// reference's objects, or hands to the next window's dmvio_hip_graph_set_residual_linearized.  Rows of residuals that are not linearised are zeroed.  Any output may be NULL.
int dmvio_hip_ba_get_linearized_residuals(dmvio_hip_ba* b, int R, unsigned char* isLinearized, float* J74, float* res_toZeroF) {
  if (!b) return failmsg("ba: null handle");
  BA_LOCK(b);
  if (!b->graph_ready) return failmsg("ba: set_window + set_graph first");
  if (R != b->H.R) return failmsg("ba_get_linearized_residuals: R differs from the graph's residual count");
  const bool have = b->d_lin != nullptr && (int)b->h_lin.size() == R && b->h_linJ.size() == (size_t)R * 74;
  for (int ri = 0; ri < R; ri++) {
    const bool l = have && b->h_lin[ri];
    if (isLinearized) isLinearized[ri] = l ? 1 : 0;
    if (J74) { if (l) memcpy(J74 + (size_t)ri * 74, &b->h_linJ[(size_t)ri * 74], sizeof(float) * 74); else memset(J74 + (size_t)ri * 74, 0, sizeof(float) * 74); }
    if (res_toZeroF) { if (l) memcpy(res_toZeroF + (size_t)ri * 8, &b->h_rtz[(size_t)ri * 8], sizeof(float) * 8); else memset(res_toZeroF + (size_t)ri * 8, 0, sizeof(float) * 8); }
  }
  return 0;
}
// accumulateLF_MT's system as the reference returns it (stitched linearised residuals + priors, EnergyFunctional.cpp:223-233) for the state of the LAST accumulation
// (dmvio_hip_ba_accumulate / _solve / _optimize); without residuals kept linearised it holds the priors only
int dmvio_hip_ba_get_lf_system(dmvio_hip_ba* b, double* HL, double* bL) {
  if (!b) return failmsg("ba: null handle");
  BA_LOCK(b);
  if (!b->graph_ready) return failmsg("ba: set_window + set_graph first");
  const BAHost& H = b->H;
  const int n = H.n();
  double HLd[4 + 8 * BA_MAXF_CAP], bLv[4 + 8 * BA_MAXF_CAP];
  const bool haveL = H.lfTop(HLd, bLv);
  if (HL) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) HL[(size_t)i * n + j] = (haveL ? H.HLraw[(size_t)i * n + j] : 0.0) + (i == j ? HLd[i] : 0.0);
  if (bL) memcpy(bL, bLv, sizeof(double) * n);
  return 0;
}

// activeResiduals of FullSystem::optimize: every residual is (re)activated: resetOOB (FullSystemOptimize.cpp:431-448)
int dmvio_hip_ba_activate_all(dmvio_hip_ba* b) {
  BA_READY(b);
  hipStream_t s = b->stream;
  // every residual still in the graph; those an earlier fix-linearisation deleted (FullSystemOptimize.cpp:195-212) stay out
  hipLaunchKernelGGL(k_ba_reset_oob, dim3((b->H.R + 255) / 256), dim3(256), 0, s, b->H.R, b->Rs);
  HIPCHK(hipGetLastError());
  return 0;
}
int dmvio_hip_ba_linearize(dmvio_hip_ba* b, int fix, double* energy) {
  BA_READY(b);
  double e = 0;
  if (int r = linearizeAll(b, fix != 0, &e)) return r;
  if (energy) *energy = e;
  return 0;
}
int dmvio_hip_ba_apply(dmvio_hip_ba* b) {
  BA_READY(b);
  return applyRes(b);
}
int dmvio_hip_ba_get_res_state(dmvio_hip_ba* b, unsigned char* newState, float* newEnergy, float* newEnergyWO, unsigned char* active, float* center3) {
  BA_READY(b);
  hipStream_t s = b->stream;
  const int R = b->H.R;
  if (newState) HIPCHK(b->bounce.d2h(newState, b->Rs.newState, R, s));
  if (newEnergy) HIPCHK(b->bounce.d2h(newEnergy, b->Rs.newEnergy, sizeof(float) * R, s));
  if (active) HIPCHK(b->bounce.d2h(active, b->Rs.active, R, s));
  if (center3) HIPCHK(b->bounce.d2h(center3, b->Rs.center, sizeof(float) * 3 * R, s));
  if (newEnergyWO) HIPCHK(b->bounce.d2h(newEnergyWO, b->d_newEnergyWO, sizeof(float) * R, s));
  HIPCHK(b->bounce.finish(s));
  return 0;
}
// RawResidualJacobian of the LAST linearisation (74 floats per residual, RawResidualJacobian.h:32-61 order) — parity/debug
int dmvio_hip_ba_get_jacobians(dmvio_hip_ba* b, float* J74) {
  BA_READY(b);
  if (!b->keep_fullJ) return failmsg("ba_get_jacobians: call dmvio_hip_ba_keep_jacobians(ba, 1) before the linearisation");
  HIPCHK(b->bounce.d2h(J74, b->d_fullJ, sizeof(float) * 74 * b->H.R, b->stream));
  HIPCHK(b->bounce.finish(b->stream));
  return 0;
}
int dmvio_hip_ba_get_frame_energy_th(dmvio_hip_ba* b, float* th) {
  if (!b || !th) return failmsg("null argument");
  BA_LOCK(b);
  if (int r = resolveTh(b)) return r;
  for (int f = 0; f < b->H.F; f++) th[f] = b->H.fr[f].frameEnergyTH;
  return 0;
}
int dmvio_hip_ba_accumulate(dmvio_hip_ba* b, double* HA, double* bA, double* Hsc, double* bsc, int* resInA) {
  BA_READY(b);
  if (int r = accumulate(b)) return r;
  const int n = b->H.n();
  const double* p = b->h_sys;
  if (HA) memcpy(HA, p, sizeof(double) * n * n);
  if (bA) memcpy(bA, p + n * n, sizeof(double) * n);
  if (Hsc) memcpy(Hsc, p + n * n + n, sizeof(double) * n * n);
  if (bsc) memcpy(bsc, p + 2 * n * n + n, sizeof(double) * n);
  if (resInA) *resInA = b->H.resInA;
  return 0;
}
int dmvio_hip_ba_get_point_acc(dmvio_hip_ba* b, float* Hdd, float* bd, float* Hcd4, float* HdiF, float* bdSumF) {
  BA_READY(b);
  hipStream_t s = b->stream;
  const int N = b->H.N;
  if (Hdd) HIPCHK(b->bounce.d2h(Hdd, b->P.Hdd, sizeof(float) * N, s));
  if (bd) HIPCHK(b->bounce.d2h(bd, b->P.bd, sizeof(float) * N, s));
  if (Hcd4) HIPCHK(b->bounce.d2h(Hcd4, linAny(b) ? b->P.HcdAF : b->P.Hcd, sizeof(float) * 4 * N, s));   // Hcd_accAF
  if (HdiF) HIPCHK(b->bounce.d2h(HdiF, b->P.HdiF, sizeof(float) * N, s));
  if (bdSumF) HIPCHK(b->bounce.d2h(bdSumF, b->P.bdSumF, sizeof(float) * N, s));
  HIPCHK(b->bounce.finish(s));
  return 0;
}
// EnergyFunctional::solveSystemF: accumulate on the device, solve on the host (the hand-off point of
// BAGTSAMIntegration::computeBAUpdate in VIO mode, EnergyFunctional.cpp:958-969), back-substitute on the device.
int dmvio_hip_ba_solve(dmvio_hip_ba* b, int iteration, double lambda, double* x_out) {
  BA_READY(b);
  if (int r = accumulate(b)) return r;
  const int n = b->H.n();
  const double* p = b->h_sys;
  std::vector<double> x;
  b->H.solveSystem(iteration, lambda, p, p + n * n, p + n * n + n, p + 2 * n * n + n, x);
  if (x_out) memcpy(x_out, x.data(), sizeof(double) * n);
  return resubstitute(b, x);
}
int dmvio_hip_ba_resubstitute(dmvio_hip_ba* b, const double* x) {
  BA_READY(b);
  std::vector<double> xv(x, x + b->H.n());
  return resubstitute(b, xv);
}
int dmvio_hip_ba_get_point_hessian(dmvio_hip_ba* b, float* idepth_hessian) {
  if (!b || !b->graph_ready || !idepth_hessian) return failmsg("ba_get_point_hessian: bad argument");   // a pure read: the loop state (sys_ready) is left alone
  BA_LOCK(b);
  HIPCHK(hipSetDevice(b->ctx->device));
  HIPCHK(b->bounce.d2h(idepth_hessian, b->P.idepth_hessian, sizeof(float) * b->H.N, b->stream));
  HIPCHK(b->bounce.finish(b->stream));
  return 0;
}
int dmvio_hip_ba_get_points(dmvio_hip_ba* b, float* idepth, float* step) {
  BA_READY(b);
  hipStream_t s = b->stream;
  if (idepth) HIPCHK(b->bounce.d2h(idepth, b->P.idepth, sizeof(float) * b->H.N, s));
  if (step) HIPCHK(b->bounce.d2h(step, b->P.step, sizeof(float) * b->H.N, s));
  HIPCHK(b->bounce.finish(s));
  return 0;
}
int dmvio_hip_ba_get_frame(dmvio_hip_ba* b, int f, double pose7_w2c[7], double aff[2], double state10[10]) {
  if (!b || f < 0 || f >= b->H.F) return failmsg("ba_get_frame: bad argument");
  BA_LOCK(b);
  const BAFrameHost& fr = b->H.fr[f];
  if (pose7_w2c) poseTo7(fr.w2c, pose7_w2c);
  if (aff) { aff[0] = fr.state_scaled[6]; aff[1] = fr.state_scaled[7]; }
  if (state10) memcpy(state10, fr.state, sizeof(double) * 10);
  return 0;
}
// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:520-660), visual-only branch: the marginalisation prior of the window
// without keyframe `frame` ((n-8) x (n-8), n-8); also returns the current prior when HM_cur / bM_cur are given.
int dmvio_hip_ba_marginalize_frame(dmvio_hip_ba* b, int frame, double* HM_new, double* bM_new) {
  if (!b || !HM_new || !bM_new || frame < 0 || frame >= b->H.F) return failmsg("ba_marginalize_frame: bad argument");
  BA_LOCK(b);
  std::vector<double> Hn, bn;
  b->H.marginalizeFrame(frame, Hn, bn);
  memcpy(HM_new, Hn.data(), sizeof(double) * Hn.size());
  memcpy(bM_new, bn.data(), sizeof(double) * bn.size());
  return 0;
}
int dmvio_hip_ba_get_marg_prior(dmvio_hip_ba* b, double* HM, double* bM) {
  if (!b || !HM || !bM) return failmsg("ba_get_marg_prior: null argument");
  BA_LOCK(b);
  const int n = b->H.n();
  if (b->H.HM.size() != (size_t)n * n) { memset(HM, 0, sizeof(double) * n * n); memset(bM, 0, sizeof(double) * n); return 0; }
  memcpy(HM, b->H.HM.data(), sizeof(double) * n * n); memcpy(bM, b->H.bM.data(), sizeof(double) * n);
  return 0;
}
// FrameHessian::setState (HessianBlocks.h:179-199) for one keyframe of the window, followed by FullSystem::setPrecalcValues
int dmvio_hip_ba_set_frame_state(dmvio_hip_ba* b, int f, const double state10[10]) {
  if (!b || !state10 || f < 0 || f >= b->H.F) return failmsg("ba_set_frame_state: bad argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  BAHost::frameSetState(b->H.fr[f], state10);
  b->H.setPrecalcValues();
  return 0;
}
// Windows taken over from a running system (rather than built fresh): FrameHessian::state_zero (HessianBlocks.cpp:74-107), frameEnergyTH
// of every keyframe and CalibHessian::value_zero (unscaled units) where they differ from what dmvio_hip_ba_set_window starts with.
int dmvio_hip_ba_set_frame_zero(dmvio_hip_ba* b, int f, const double state_zero10[10]) {
  if (!b || !state_zero10 || f < 0 || f >= b->H.F) return failmsg("ba_set_frame_zero: bad argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  BAHost::frameSetStateZero(b->H.fr[f], state_zero10);
  b->pre_static_valid = false;
  b->H.frameTakeData(b->H.fr[f]);
  b->H.setPrecalcValues();
  return 0;
}
int dmvio_hip_ba_set_frame_states(dmvio_hip_ba* b, const double* state_zero10, const double* state10) {
  if (!b || b->H.F < 1) return failmsg("ba_set_frame_states: bad argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  for (int f = 0; f < b->H.F; f++) {
    if (state_zero10) { BAHost::frameSetStateZero(b->H.fr[f], state_zero10 + 10 * f); b->H.frameTakeData(b->H.fr[f]); }
    if (state10) BAHost::frameSetState(b->H.fr[f], state10 + 10 * f);
  }
  if (state_zero10) b->pre_static_valid = false;
  b->H.setPrecalcValues();
  return 0;
}
int dmvio_hip_ba_set_frame_energy_th(dmvio_hip_ba* b, const float* th) {
  if (!b || !th) return failmsg("ba_set_frame_energy_th: null argument");
  BA_LOCK(b);
  b->th_pending = false;
  for (int f = 0; f < b->H.F; f++) b->H.fr[f].frameEnergyTH = th[f];
  b->th_dirty = true;
  return 0;
}
// IMUIntegration::newFrameEnergyTH (src/IMU/IMUIntegration.cpp:365-373, called from FullSystem::setNewFrameEnergyTH, FullSystemOptimize.cpp:136-140, when setting_useIMU):
// IMUSettings::maxFrameEnergyThreshold caps the newest keyframe's threshold; <= 0 = no cap (the reference's default)
int dmvio_hip_ba_set_frame_energy_th_cap(dmvio_hip_ba* b, float maxFrameEnergyThreshold) {
  if (!b) return failmsg("ba_set_frame_energy_th_cap: null handle");
  BA_LOCK(b);
  b->th_cap = maxFrameEnergyThreshold;
  return 0;
}
int dmvio_hip_ba_set_calib_values(dmvio_hip_ba* b, const double value[4], const double value_zero[4]) {
  if (!b || !value || !value_zero) return failmsg("ba_set_calib_values: null argument");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  for (int i = 0; i < 4; i++) b->H.c_value_zero[i] = value_zero[i];
  b->H.calibSetValue(value);
  b->H.setPrecalcValues();
  return 0;
}
int dmvio_hip_ba_get_calib_values(dmvio_hip_ba* b, double value[4], double value_zero[4]) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  if (value) memcpy(value, b->H.c_value, sizeof(double) * 4);
  if (value_zero) memcpy(value_zero, b->H.c_value_zero, sizeof(double) * 4);
  return 0;
}
int dmvio_hip_ba_get_res_in_a(dmvio_hip_ba* b, int* resInA) {
  if (!b || !resInA) return failmsg("null argument");
  BA_LOCK(b);
  *resInA = b->H.resInA;
  return 0;
}
int dmvio_hip_ba_get_calib(dmvio_hip_ba* b, double fxfycxcy[4]) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  memcpy(fxfycxcy, b->H.c_value_scaled, sizeof(double) * 4);
  return 0;
}

// One Gauss-Newton iteration = the loop body of FullSystem::optimize (FullSystemOptimize.cpp:485-586).
//
// The host solves the 68x68 system (the hand-off point of the reference's GTSAM branch, EnergyFunctional.cpp:958-969) and steps the
// frames; everything else is ONE chain of kernels enqueued without waiting in between:
//   resubstitute + point step  ->  linearise the stepped state, its last workgroup sums the energy, sets the newest keyframe's threshold
//   and takes the accept / reject decision  ->  [rejected only] restore the points and relinearise the backed-up state  ->  [accepted only]
//   applyRes + per-point sums + point backup -> accumulation -> adjoint stitching -> the stitched system of the NEW state in host memory.
// The host waits once per iteration, by polling the ticket the chain's last kernel stores behind its results.  After a rejected step the
// system in host memory is still the one of the (restored) state, so the next iteration goes straight to its solve with the larger lambda.
// the energy / threshold of a relinearisation the host did not wait for, once a later ticket of the same stream has been seen (or after waiting for its own)
static int settleReject(dmvio_hip_ba* b, double lastE[3], bool wait) {
  if (!b->pending_reject) return 0;
  if (wait) { if (int r = waitTicket(b, b->pending_ticket)) return r; }
  lastE[0] = b->h_res->E[1];
  b->H.fr[b->H.F - 1].frameEnergyTH = b->h_res->th[1]; b->th_pending = false;
  if (b->pending_trace >= 0 && b->pending_trace < 64) b->trace[b->pending_trace][0] = lastE[0];
  b->pending_reject = false; b->pending_trace = -1;
  return 0;
}
// ---- the reference's default solver branch (setting_useGTSAMIntegration): views of the window for the hooks, calcMEnergyF with the GTSAM term
static void fillFrameViews(dmvio_hip_ba* b) {
  const BAHost& H = b->H;
  b->vio_frames.resize(H.F);
  for (int f = 0; f < H.F; f++) {
    dmvio_hip_ba_frame_view& v = b->vio_frames[f];
    const BAFrameHost& fr = H.fr[f];
    v.frameID = fr.frameID; v.index = f;
    poseTo7(fr.w2c, v.PRE_worldToCam7); poseTo7(fr.evalPT, v.worldToCam_evalPT7);
    memcpy(v.state10, fr.state, sizeof(v.state10)); memcpy(v.state_zero10, fr.state_zero, sizeof(v.state_zero10));
  }
}
// EnergyFunctional::calcMEnergyF (EnergyFunctional.cpp:322-345): without hooks delta.dot(2 bM + HM delta); with them updateBAValues(frames) when the current values are
// asked for, then getBAEnergy(useNewValues) + delta.dot(2 bMForGTSAM + HMForGTSAM delta)
static double calcMEnergy(dmvio_hip_ba* b, bool useNewValues) {
  if (!b->vio) return b->H.calcMEnergy();
  if (!useNewValues && b->vio->updateBAValues) { fillFrameViews(b); b->vio->updateBAValues(b->vio->user, b->H.F, b->vio_frames.data(), b->H.c_value); }
  const double g = b->vio->getBAEnergy ? b->vio->getBAEnergy(b->vio->user, useNewValues ? 1 : 0) : 0.0;
  return g + b->H.calcMEnergy();
}
static double vioDynamicWeight(dmvio_hip_ba* b, double E0, int resInA) {
  if (!b->vio || !b->vio->updateDynamicWeight) return 1.0;
  const float rmse = sqrtf((float)(E0 / (8 * resInA)));   // patternNum * ef->resInA (FullSystemOptimize.cpp:495-498)
  return b->vio->updateDynamicWeight(b->vio->user, E0, (double)rmse, b->vio_opt ? b->vio_opt->coarseTrackingWasGood : 1);
}
// defer: after a rejected step return without waiting for the relinearisation of the restored state (its energy stays on the device for the next accept
// test, BACtl::lastE0); lastE[0] is then filled in by the next call's wait or by settleReject.
// last: the caller will not iterate again — an accepted step is applied, but the system of the new state (which only the next solve would read) is not
// accumulated: the per-point sums (HdiF, idepth_hessian) and resInA then stay those of the last solveSystemF, as in the reference.
// canbreak_out: doStepFromBackup's return value && baIntegration->canBreak() (FullSystemOptimize.cpp:520-523); only evaluated with hooks (false otherwise).
static int gnIteration(dmvio_hip_ba* b, int iteration, double& lambda, double lastE[3], bool& accepted, bool defer = false, int trace_slot = -1, bool last = false,
                       bool* canbreak_out = nullptr) {
  BAHost& H = b->H;
  const int n = H.n(), tot = 2 * (n * n + n);
  const bool shard = sharded(b);
  const dmvio_hip_ba_callbacks* vio = b->vio;
  const bool dynDuring = vio && b->vio_opt && b->vio_opt->updateDynamicWeightDuringOptimization;
  if (shard) { if (int r = ensureExchange(b)) return r; }
  // backupState (the point part rode in the per-point sums that produced the system at hand)
  double tq0 = b->timing ? nowUs() : 0, tq1;
#define BA_PH(i) do { if (b->timing) { tq1 = nowUs(); b->tm.t[i] += tq1 - tq0; tq0 = tq1; } } while (0)
  H.backupFrames();
  H.prepareSolve();   // nullspaces, orthogonalisation basis, prior right-hand side
  if (!b->sys_ready) {
    const bool sums_fresh = b->sums_fresh;
    b->sums_fresh = false;
    if (int r = accumulate(b, true, true, sums_fresh)) return r;
  }
  b->sys_ready = false;
  if (vio && (dynDuring || iteration == 0)) {
    // the weight of the photometric energy against the GTSAM factors, updated before solving (FullSystemOptimize.cpp:491-503); ef->resInA is the count the LAST
    // solveSystemF left behind — before the first solve of this call the caller's (vio_options::resInA_at_entry) or, without one, the current system's
    if (b->pending_reject) { if (int r = settleReject(b, lastE, true)) return r; }
    const int resInA_ref = iteration == 0 ? ((b->vio_opt && b->vio_opt->resInA_at_entry >= 0) ? b->vio_opt->resInA_at_entry : (int)b->h_sys[tot]) : b->resInA_solve;
    b->dynW = vioDynamicWeight(b, lastE[0], resInA_ref);
  }
  BA_PH(0);
  // solveSystem
  const double* p = b->h_sys;
  b->resInA_solve = (int)p[tot];
  std::vector<double> x;
  if (!vio) H.solveSystem(iteration, lambda, p, p + n * n, p + n * n + n, p + 2 * n * n + n, x);
  else {
    // hand-over of EnergyFunctional.cpp:958-969: the damped Schur system, its right-hand side and the undamped system go to the caller's solver
    std::vector<double> HP((size_t)n * n), bP(n), HN((size_t)n * n);
    H.buildGtsamSystem(lambda, p, p + n * n, p + n * n + n, p + 2 * n * n + n, HP.data(), bP.data(), HN.data());
    fillFrameViews(b);
    x.assign(n, 0.0);
    if (!vio->computeBAUpdate) return failmsg("ba_optimize_vio: computeBAUpdate hook missing");
    if (vio->computeBAUpdate(vio->user, n, HP.data(), bP.data(), lambda, HN.data(), H.F, b->vio_frames.data(), H.c_value, x.data()) != 0)
      return failmsg("ba_optimize_vio: computeBAUpdate hook reported an error");
    H.finishExternalSolve(iteration, x);
  }
  BA_PH(1);
  // resubstituteF_MT + the point part of doStepFromBackup (stepfac 1) ride in front of the linearisation of the stepped state
  ResubArgs X;
  {
    float xc[4];
    std::vector<float> xAd;
    H.prepareResubstitute(x, xc, xAd);
    memcpy(X.xc, xc, sizeof(xc));
    memset(X.xAd, 0, sizeof(X.xAd));
    memcpy(X.xAd, xAd.data(), sizeof(float) * xAd.size());
  }
  // doStepFromBackup, frames and calibration; the step norms only feed canbreak, which stays false without the GTSAM hooks (FullSystemOptimize.cpp:387,523,583)
  if (!b->pre_static_valid) { if (int r = uploadWindowTables(b)) return r; }   // evaluation-point members of the table (R0, t0, b0): once per window
  fillWindow(b);
  dynFromHost(H, b->dyn_cur);         // backed-up state: calibration (W) and step-dependent precalc members, for a relinearisation after a rejected step
  const BAWindow W_backup = b->W;
  const BAPreDyn dyn_backup = b->dyn_cur;
  float fs[4];
  H.stepFrames(1.0f, fs);
  H.setPrecalcValues();
  bool canbreak = false;
  if (vio) {
    // FullSystemOptimize.cpp:269-313: sumNID = mean |idepth_backup| over the points in window order (the per-point sums mirrored idepth_backup into host memory)
    float sumNID = 0, numID = 0;
    for (int pi = 0; pi < H.N; pi++) { sumNID += fabsf(b->h_idepth_backup[pi]); numID++; }
    sumNID /= numID;
    const float thOpt = H.S.thOptIterations;
    canbreak = sqrtf(fs[0]) < 0.0005 * thOpt && sqrtf(fs[1]) < 0.00005 * thOpt && sqrtf(fs[3]) < 0.00005 * thOpt && sqrtf(fs[2]) * sumNID < 0.00005 * thOpt;
    canbreak = canbreak && vio->canBreak && vio->canBreak(vio->user) != 0;
  }
  if (canbreak_out) *canbreak_out = canbreak;
  const int minOpt = (b->vio_opt && b->vio_opt->minOptIterations >= 0) ? b->vio_opt->minOptIterations : H.S.minOptIterations;
  if (canbreak && iteration >= minOpt) last = true;
  double newL = 0;
  if (int r = calcLEnergyR(b, &newL)) return r;
  const double newM = calcMEnergy(b, true);
  if (dynDuring) b->dynW = vioDynamicWeight(b, lastE[0], b->resInA_solve);   // before deciding whether to accept the step (FullSystemOptimize.cpp:534-538)
  // linearise the stepped state; the kernel's last workgroup sums the energy, sets the newest keyframe's threshold and decides
  fillWindow(b);
  dynFromHost(H, b->dyn_cur);
  if (int r = uploadThresholds(b)) return r;
  unsigned int D_ticket = 0;
  {
    BA_PH(2);
    BADecide D = makeDecide(b, 1, true, true);
    D_ticket = D.ticket;
    // newEnergy[0] + newEnergy[1] + newEnergyL + newEnergyM / dynamicGTSAMWeight (FullSystemOptimize.cpp:553-554): the quotients are formed here (x / 1.0 == x without hooks)
    D.lastE0 = lastE[0]; D.lastL = lastE[1]; D.lastM = lastE[2] / b->dynW; D.newL = newL; D.newM = newM / b->dynW;
    D.lastE0_from_ctl = b->pending_reject ? 1 : 0;   // the host has not seen the restored state's energy yet: the device has
    BA_LAUNCH_LINEARIZE(b, b->W, b->keep_fullJ ? b->d_fullJ : (float*)nullptr, (const unsigned char*)nullptr, shard ? packOnly(b, D) : D, BA_GATE_ALWAYS, 0, b->dyn_cur, 1, X, 1);
    HIPCHK(hipGetLastError());
    if (shard) { if (int r = decideGlobal(b, D)) return r; }
    BA_PH(3);
    if (int r = waitTicket(b, D.ticket)) return r;   // energy and the accept / reject decision are in host memory (the threshold follows: resolveTh)
    if (int r = settleReject(b, lastE, false)) return r;   // ... and so are those of an earlier relinearisation on the same stream
    BA_PH(4);
  }
  accepted = b->h_res->accept != 0;
  if (accepted) {
    // applyRes + per-point sums + point backup -> accumulation -> stitching: the system of the new state, for the next iteration's solve.
    // (Enqueuing this branch speculatively behind the linearisation, gated on the device-side decision, was measured: no gain — the four
    // launches, not the host round trip, set its length.)
    if (!last) { if (int r = accumulate(b, true, true, false, true, BA_GATE_ALWAYS)) return r; }
    else { if (int r = applyRes(b)) return r; }
    lastE[0] = b->h_res->E[0]; lastE[1] = newL; lastE[2] = newM;
    b->th_pending = true; b->th_pending_ticket = D_ticket;   // the threshold follows its decision by a few microseconds: picked up by resolveTh before anything reads it
    lambda = std::max(lambda * 0.25, 1e-5);
    if (vio && vio->acceptBAUpdate) vio->acceptBAUpdate(vio->user, lastE[0]);   // FullSystemOptimize.cpp:569-572
  } else {
    // loadSateBackup + linearizeAll (FullSystemOptimize.cpp:575-581): the points are restored by the relinearisation kernel itself; the
    // system in host memory is still the one of the restored state
    const BADecide D2 = makeDecide(b, 2, true, true);
    BA_LAUNCH_LINEARIZE(b, W_backup, b->keep_fullJ ? b->d_fullJ : (float*)nullptr, (const unsigned char*)nullptr, shard ? packOnly(b, D2) : D2, BA_GATE_ALWAYS, 1, dyn_backup, 1, b->x_none, 0);
    HIPCHK(hipGetLastError());
    if (shard) { if (int r = decideGlobal(b, D2)) return r; }
    H.restoreFrames();
    H.setPrecalcValues();
    fillWindow(b);
    b->dyn_cur = dyn_backup;
    double oldL = 0;
    if (int r = calcLEnergyR(b, &oldL)) return r;
    const double oldM = calcMEnergy(b, false);
    lastE[1] = oldL; lastE[2] = oldM;
    b->pending_reject = true; b->pending_ticket = D2.ticket; b->pending_trace = trace_slot;
    if (!defer) { if (int r = settleReject(b, lastE, true)) return r; }
    lambda *= 1e2;
  }
  BA_PH(5);
#undef BA_PH
  b->sys_ready = !(accepted && last);   // accepted: the chain left the system of the new state behind; rejected: the one at hand still is the restored state's
  b->tm.n++;
  return 0;
}

// Measurement: the kernel chain of an ACCEPTED Gauss-Newton iteration on the current state — linearise (+ energy / threshold decision pass) -> applyRes + per-point sums
// -> accumulate -> adjoint stitch -> gather — `reps` times, with HIP events recorded on the handle's stream between the launches; us5 = mean microseconds of the five
// kernels (each figure includes the gap to the next launch: what the chain costs on the stream).  The window state is left as a linearise + applyRes + accumulate leaves it.
int dmvio_hip_ba_profile_chain(dmvio_hip_ba* b, int reps, float us5[5]) {
  BA_READY(b);
  if (!us5 || reps < 1) return failmsg("ba_profile_chain: bad argument");
  if (sharded(b)) return failmsg("ba_profile_chain: single-device windows only");
  struct Events {   // destroyed on every way out
    hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ~Events() { for (int k = 0; k < 6; k++) if (e[k]) hipEventDestroy(e[k]); }
  } evs;
  hipEvent_t* ev = evs.e;
  for (int k = 0; k < 6; k++) HIPCHK(hipEventCreate(&ev[k]));
  struct ProfScope { dmvio_hip_ba* b; ~ProfScope() { b->prof = nullptr; } } profScope{b};
  double acc[5] = {0, 0, 0, 0, 0};
  int rc = 0;
  for (int r = 0; r < reps && rc == 0; r++) {
    double e = 0;
    b->prof = ev;
    rc = linearizeAll(b, false, &e, 0, false, true);   // enqueue only
    if (rc == 0) rc = accumulate(b, true, true, false, true);
    b->prof = nullptr;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int k = 0; k < 5; k++) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += 1e3 * ms; }
  }
  for (int k = 0; k < 5; k++) us5[k] = (float)(acc[k] / reps);
  return rc;
}

// diagnostics: in-kernel timeline of the last decision pass, 100 MHz ticks since its workgroup started (begin, energy, keys, selected)
int dmvio_hip_ba_last_decide_ticks(dmvio_hip_ba* b, int ticks4[4]) {
  if (!b || !b->h_res || !ticks4) return failmsg("null argument");
  BA_LOCK(b);
  for (int i = 0; i < 4; i++) ticks4[i] = b->h_res->ticks[i];
  return 0;
}

int dmvio_hip_ba_gn_iteration(dmvio_hip_ba* b, int iteration, double* lambda_io, double lastE[3], int* accepted) {
  if (!b) return failmsg("ba: null handle");
  BA_LOCK(b);
  const bool sums_fresh = b->sums_fresh, sys_ready = b->sys_ready;
  BA_READY_LOCKED(b);
  b->sums_fresh = sums_fresh; b->sys_ready = sys_ready;
  bool acc = false;
  double lam = *lambda_io;
  if (int r = gnIteration(b, iteration, lam, lastE, acc)) return r;
  *lambda_io = lam;
  if (accepted) *accepted = acc ? 1 : 0;
  return 0;
}

// ---- building blocks of a SHARDED Gauss-Newton iteration (one keyframe's points per GPU): the caller sums the packed
// local systems / energies of all ranks (RCCL all-reduce) between these calls; every rank then solves the identical reduced system.
int dmvio_hip_ba_backup(dmvio_hip_ba* b) {
  BA_READY(b);
  b->H.backupFrames();
  float d0, d1;
  return pointStep(b, 0, 0.f, &d0, &d1);
}
int dmvio_hip_ba_solve_system(dmvio_hip_ba* b, int iteration, double lambda, const double* HA, const double* bA, const double* Hsc, const double* bsc, double* x_out) {
  BA_READY(b);
  if (!HA || !bA || !Hsc || !bsc) return failmsg("ba_solve_system: null argument");
  std::vector<double> x;
  b->H.solveSystem(iteration, lambda, HA, bA, Hsc, bsc, x);
  if (x_out) memcpy(x_out, x.data(), sizeof(double) * b->H.n());
  return resubstitute(b, x);
}
int dmvio_hip_ba_step(dmvio_hip_ba* b, float stepfac, float sums6[6]) {
  BA_READY(b);
  float fs[4], sumID = 0, sumNID = 0;
  b->H.stepFrames(stepfac, fs);
  if (int r = pointStep(b, 1, stepfac, &sumID, &sumNID)) return r;
  b->H.setPrecalcValues();
  if (sums6) { for (int i = 0; i < 4; i++) sums6[i] = fs[i]; sums6[4] = sumID * b->H.N; sums6[5] = sumNID * b->H.N; }  // point sums un-normalised (local shard)
  return 0;
}
int dmvio_hip_ba_restore(dmvio_hip_ba* b) {
  BA_READY(b);
  b->H.restoreFrames();
  float d0, d1;
  if (int r = pointStep(b, 2, 0.f, &d0, &d1)) return r;
  b->H.setPrecalcValues();
  return 0;
}
// linearizeAll WITHOUT the setNewFrameEnergyTH tail: returns the local energy and the state_NewEnergyWithOutlier values of the local
// residuals that target the newest keyframe (the inputs of setNewFrameEnergyTH, FullSystemOptimize.cpp:96-149) so the caller can
// gather them over all shards.
int dmvio_hip_ba_linearize_local(dmvio_hip_ba* b, int fix, double* energy, float* new_frame_energies, int* n_new_frame_energies) {
  BA_READY(b);
  double e = 0;
  if (int r = linearizeAll(b, fix != 0, &e, 0, true)) return r;   // the threshold is set by the caller from the energies gathered over all shards
  if (energy) *energy = e;
  std::vector<float> wo(b->H.R);
  if (b->H.R) HIPCHK(b->bounce.d2h(wo.data(), b->d_newEnergyWO, sizeof(float) * b->H.R, b->stream));
  HIPCHK(b->bounce.finish(b->stream));
  int n = 0;
  for (int ri : b->h_newest)
    if (wo[ri] >= 0) { if (new_frame_energies) new_frame_energies[n] = wo[ri]; n++; }
  if (n_new_frame_energies) *n_new_frame_energies = n;
  return 0;
}
int dmvio_hip_ba_set_new_frame_energy_th(dmvio_hip_ba* b, float th) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  b->sums_fresh = false; b->sys_ready = false;
  b->H.fr[b->H.F - 1].frameEnergyTH = th;
  b->th_dirty = true;
  return 0;
}
int dmvio_hip_ba_energy_terms(dmvio_hip_ba* b, double* EL, double* EM) {
  if (!b) return failmsg("null ba");
  BA_LOCK(b);
  if (EL) { if (int r = calcLEnergyR(b, EL)) return r; }
  if (EM) *EM = b->H.calcMEnergy();
  return 0;
}

// FullSystem::optimize (FullSystemOptimize.cpp:417-647).  vio == NULL: the library's own damped LDLT (the reference's solver with setting_useGTSAMIntegration off);
// otherwise every solve, the M-energy term, the dynamic weight and the early exit go through the caller's hooks, in the order the reference calls BAGTSAMIntegration.
static int optimizeImpl(dmvio_hip_ba* b, int mnumOptIts, const dmvio_hip_ba_callbacks* vio, const dmvio_hip_ba_vio_options* opt, float* rmse, double* finalEnergy, int* iterations,
                        double* trace /* 64x4 or NULL */) {
  BAHost& H = b->H;
  if (H.F < 2) { if (rmse) *rmse = 0; return 0; }
  if (H.F < 3) mnumOptIts = 20;
  if (H.F < 4) mnumOptIts = 15;
  struct VioScope {   // the hooks are those of this call only
    dmvio_hip_ba* b;
    ~VioScope() { b->vio = nullptr; b->vio_opt = nullptr; b->H.gtsam = false; b->dynW = 1.0; }
  } scope{b};
  b->vio = vio; b->vio_opt = vio ? opt : nullptr; b->dynW = 1.0;
  const int n = H.n();
  if (vio) {
    H.gtsam = true;
    if (opt && opt->HMForGTSAM && opt->bMForGTSAM) { H.HMG.assign(opt->HMForGTSAM, opt->HMForGTSAM + (size_t)n * n); H.bMG.assign(opt->bMForGTSAM, opt->bMForGTSAM + n); }
    else { H.HMG.assign((size_t)n * n, 0.0); H.bMG.assign(n, 0.0); }
  }
  if (int r = dmvio_hip_ba_activate_all(b)) return r;
  double lastE[3];
  // initial linearisation, applyRes and the first system (per-point sums → accumulation → stitching) go out as ONE chain: nothing in it waits for the host, which
  // picks the energy and the threshold up behind the chain's last ticket
  if (int r = linearizeAll(b, false, &lastE[0], 0, false, true)) return r;
  if (int r = applyRes(b)) return r;
  if (int r = accumulate(b, true, true, false)) return r;   // backupState of the points rides in the per-point sums; the frames are backed up by the first iteration
  linearizePickUp(b, &lastE[0], false);
  b->sys_ready = true;
  if (int r = calcLEnergyR(b, &lastE[1])) return r;
  lastE[2] = calcMEnergy(b, false);
  double lambda = 1e-5;
  int done = 0;
  b->trace[0][0] = lastE[0]; b->trace[0][1] = lastE[1]; b->trace[0][2] = lastE[2]; b->trace[0][3] = 1;
  for (int iteration = 0; iteration < mnumOptIts; iteration++) {
    bool acc = false, canbreak = false;
    if (int r = gnIteration(b, iteration, lambda, lastE, acc, true, done + 1, iteration == mnumOptIts - 1, &canbreak)) return r;
    done++;
    if (done < 64) { b->trace[done][0] = lastE[0]; b->trace[done][1] = lastE[1]; b->trace[done][2] = lastE[2]; b->trace[done][3] = acc ? 1 : 0; }
    // canbreak && iteration >= setting_minOptIterations (FullSystemOptimize.cpp:586): baIntegration->canBreak() stays false without the GTSAM hooks
    const int minOpt = (opt && vio && opt->minOptIterations >= 0) ? opt->minOptIterations : H.S.minOptIterations;
    if (canbreak && iteration >= minOpt) break;
  }
  if (int r = settleReject(b, lastE, true)) return r;
  if (vio) b->dynW = vioDynamicWeight(b, lastE[0], b->resInA_solve);   // "Update again!" (FullSystemOptimize.cpp:594)
  // fix the newest frame's linearisation point, re-linearise with applyRes (FullSystemOptimize.cpp:596-609)
  BAFrameHost& last = H.fr[H.F - 1];
  double newStateZero[10] = {0, 0, 0, 0, 0, 0, last.state[6], last.state[7], 0, 0};
  last.evalPT = last.w2c;
  BAHost::frameSetState(last, newStateZero);
  BAHost::frameSetStateZero(last, newStateZero);
  H.setAdjointsF();
  if (int r = uploadAdjoints(b)) return r;
  H.setPrecalcValues();
  double fe = 0;
  if (int r = linearizeAll(b, true, &fe)) return r;
  b->final_energy = fe; b->iterations_done = done;
  if (rmse) *rmse = sqrtf((float)(fe / (8 * H.resInA)));
  if (finalEnergy) *finalEnergy = fe;
  if (iterations) *iterations = done;
  if (trace) memcpy(trace, b->trace, sizeof(b->trace));
  if (vio && vio->postOptimization) { fillFrameViews(b); vio->postOptimization(vio->user, H.F, b->vio_frames.data(), H.c_value); }   // FullSystemOptimize.cpp:641
  return 0;
}
int dmvio_hip_ba_optimize_batch(struct dmvio_hip_ba_batch* B, int W, dmvio_hip_ba* const* windows, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations, double* trace);
struct dmvio_hip_ba_batch* dmvio_hip_ba_batch_create(dmvio_hip_ctx* ctx, int max_windows);
void dmvio_hip_ba_batch_destroy(struct dmvio_hip_ba_batch* B);
int dmvio_hip_ba_optimize(dmvio_hip_ba* b, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations, double* trace /* 64x4 or NULL */) {
  BA_READY(b);
  if (b->device_loop && !sharded(b)) {
    if (!b->own_batch) b->own_batch = dmvio_hip_ba_batch_create(b->ctx, 1);
    if (!b->own_batch) return -1;
    dmvio_hip_ba* one = b;
    return dmvio_hip_ba_optimize_batch(b->own_batch, 1, &one, mnumOptIts, rmse, finalEnergy, iterations, trace);
  }
  return optimizeImpl(b, mnumOptIts, nullptr, nullptr, rmse, finalEnergy, iterations, trace);
}
// FullSystem::optimize with the reference's DEFAULT solver branch (settings.cpp:37 setting_useGTSAMIntegration = true): the same device-resident loop, the solve and
// the GTSAM energy term handed to the caller (include/dmvio_hip.h).
int dmvio_hip_ba_optimize_vio(dmvio_hip_ba* b, int mnumOptIts, const dmvio_hip_ba_callbacks* cb, const dmvio_hip_ba_vio_options* opt, float* rmse, double* finalEnergy,
                              int* iterations, double* trace) {
  BA_READY(b);
  if (!cb || !cb->computeBAUpdate) return failmsg("ba_optimize_vio: the computeBAUpdate hook is required (dmvio_hip_ba_optimize runs the library's own solver)");
  if (sharded(b)) return failmsg("ba_optimize_vio: not available for a window sharded over ranks (every rank would have to run identical hooks)");
  return optimizeImpl(b, mnumOptIts, cb, opt, rmse, finalEnergy, iterations, trace);
}
// dmvio_hip_ba_solve_ldlt with the signature of the computeBAUpdate hook: a ready-made hook for callers that fall back to the visual-only solve (user is ignored)
int dmvio_hip_ba_hook_ldlt(void* user, int n, const double* HPassed, const double* b_in, double lambda, const double* HNoLambda, int F, const dmvio_hip_ba_frame_view* frames,
                           const double calib_value[4], double* x_out) {
  (void)user; (void)lambda; (void)HNoLambda; (void)F; (void)frames; (void)calib_value;
  return dmvio_hip_ba_solve_ldlt(n, HPassed, b_in, x_out);
}
// The library's own solver as a computeBAUpdate hook (EnergyFunctional.cpp:971-973: diagonal pre-scaling (H_ii + 10)^-1/2, pivoted LDL^T): for hooks that fall back to the
// visual-only step, and the check that the hook path reproduces dmvio_hip_ba_optimize bit for bit.  HPassed is n x n row-major; host-only.
int dmvio_hip_ba_solve_ldlt(int n, const double* HPassed, const double* b_in, double* x_out) {
  if (!HPassed || !b_in || !x_out || n < 1 || n > 4 + 8 * BA_MAXF_CAP) return failmsg("ba_solve_ldlt: bad argument");
  std::vector<double> Ht((size_t)n * n, 0.0);
  double sv[4 + 8 * BA_MAXF_CAP], bs[4 + 8 * BA_MAXF_CAP];
  for (int i = 0; i < n; i++) sv[i] = 1.0 / std::sqrt(HPassed[(size_t)i * n + i] + 10);
  for (int i = 0; i < n; i++) { for (int j = 0; j <= i; j++) Ht[(size_t)j * n + i] = sv[i] * HPassed[(size_t)i * n + j] * sv[j]; bs[i] = sv[i] * b_in[i]; }
  BAHost::ldltSolveTransposed(Ht.data(), n, bs, n);
  for (int i = 0; i < n; i++) x_out[i] = sv[i] * bs[i];
  return 0;
}

}  // extern "C"

// ================================================================================================= device-resident Gauss-Newton loop, W windows per launch (round 5)
// FullSystem::optimize (FullSystemOptimize.cpp:417-647) for W windows at once, the reference's non-GTSAM solver branch: per Gauss-Newton iteration the host enqueues ONE
// fixed sequence of kernels for all windows (ba_batch_kernels.hpp) and never waits — the 68x68 solve, the frame step, the pair tables, the energies and the accept test run
// on the device (k_ba_solve + the decision pass of the linearisation), every kernel of the chain takes its window from blockIdx.y and is gated on that window's own decision.
// Two waits per call: behind the loop (the frame states come back, the host re-anchors the newest keyframe, FullSystemOptimize.cpp:596-603) and behind the final
// fix-linearisation.  Windows of one call must hold the same number of keyframes (the adjoint stitch's workgroup shape); the caller groups them.
// the per-window host work of a batch call (tables, nullspace bases, staging copies before the launches; state write-back, adjoints and pair tables behind the loop:
// 30-40 us per window each) is dealt out over a few persistent worker threads — at 64 windows it was 4.3 ms of a 12 ms call
#include <thread>
#include <condition_variable>
struct BAWorkers {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::function<int(int)> fn;
  int next = 0, count = 0, pending = 0, rc = 0, device = 0;
  unsigned long long gen = 0;
  bool quit = false;
  std::string err;
  void start(int n, int dev) {
    device = dev;
    for (int i = 0; i < n; i++) th.emplace_back([this] { run(); });
  }
  void run() {
    hipSetDevice(device);
    unsigned long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return quit || (gen != seen && next < count); });
      if (quit) return;
      while (next < count) {
        const int i = next++;
        lk.unlock();
        const int r = fn(i);
        std::string e = r ? dmv_err() : std::string();
        lk.lock();
        if (r && !rc) { rc = r; err = e; }
        if (--pending == 0) cv_done.notify_all();
      }
      seen = gen;
    }
  }
  // fn(i) for i in [0, n): on the workers and on the calling thread; returns the first non-zero result (its message becomes this thread's last error)
  int parallelFor(int n, std::function<int(int)> f, const int serial_below = 8) {
    if (th.empty() || n < serial_below) { for (int i = 0; i < n; i++) if (int r = f(i)) return r; return 0; }
    {
      std::lock_guard<std::mutex> lk(mu);
      fn = std::move(f); next = 0; count = n; pending = n; rc = 0; gen++;
    }
    cv.notify_all();
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      if (next >= count) { cv_done.wait(lk, [&] { return pending == 0; }); break; }
      const int i = next++;
      lk.unlock();
      const int r = fn(i);
      std::string e = r ? dmv_err() : std::string();
      lk.lock();
      if (r && !rc) { rc = r; err = e; }
      if (--pending == 0) cv_done.notify_all();
    }
    if (rc) { dmv_err() = err; dmv_err_epoch()++; }
    return rc;
  }
  // fn(i) for i in [0, n), handed out in index order, on the workers ALONE: the caller goes on (it enqueues one group's launches while the next group's tables are prepared)
  // and collects the result with waitAsync().  Needs workers (th.empty(): use parallelFor).
  void startAsync(int n, std::function<int(int)> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      fn = std::move(f); next = 0; count = n; pending = n; rc = 0; gen++;
    }
    cv.notify_all();
  }
  int waitAsync() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
    if (rc) { dmv_err() = err; dmv_err_epoch()++; }
    return rc;
  }
  ~BAWorkers() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};
struct dmvio_hip_ba_batch {
  BAWorkers workers;
  dmvio_hip_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  int cap = 0;
  std::mutex mu;
  BAWinDev* d_wins = nullptr;
  BAWinDev* h_wins = nullptr;      // pinned
  char* d_tab = nullptr;           // per window: [HM | bM | basis | adHostF | adTargetF] (uploaded) — stride tab_stride
  char* h_tab = nullptr;           // pinned
  char* d_out = nullptr;           // per window: [sys | trace (64 x 4) | x_last] (device-only / downloaded) — stride out_stride
  double* h_trace = nullptr;       // pinned: cap x (256 + NMAX) doubles
  size_t tab_stride = 0, out_stride = 0;
  int exact_backsub = 0;
  double host_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // host clock of the last call's phases (dmvio_hip_ba_batch_last_host_us)
  float last_ms[3] = {0, 0, 0};    // HIP-event times of the last call: the loop (init chain + iterations), the final fix-linearisation, [profile] one stepped linearisation
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // a batch of >= 4 windows is cut into groups (three by default, at most BA_BATCH_STREAMS), one stream each, their launches interleaved stage by stage (optimizeBatchGroup): while one group's
  // k_ba_solve runs (one workgroup per window) the other groups' linearisations / accumulations fill the device
  enum { BA_BATCH_STREAMS = 8 };
  hipStream_t gstream[BA_BATCH_STREAMS] = {};   // [0] = stream
  hipEvent_t gev[BA_BATCH_STREAMS][3] = {};                                        // per group: [0] its initial linearisation is enqueued (the next group's start), [1] its loop is done and its states are on the host, [2] its last kernel
  int lin_lanes = 1;               // dmvio_hip_ba_batch_set_linearize_lanes: 1 = k_ba_linearize_b1 (one lane per residual) from 4 windows on, 8 = always the eight-lane kernel
  int streams = 0;                 // dmvio_hip_ba_batch_set_streams: 0 = automatic, k >= 1 = at most k groups (1: the whole batch on one stream)
  int profile = 0;                 // dmvio_hip_ba_batch_set_profile: events around the stepped linearisation of iteration 1 (k_ba_linearize_b of all windows)
};
static constexpr int BA_BATCH_NMAX = 4 + 8 * BA_MAXF_CAP;
static size_t batchTabBytes() {
  const size_t n = BA_BATCH_NMAX, F2 = (size_t)BA_MAXF_CAP * BA_MAXF_CAP;
  return ((n * n + n + 7 * n) * sizeof(double) + 2 * F2 * 64 * sizeof(float) + F2 * sizeof(BAPrecalc) + 255) & ~(size_t)255;
}
static size_t batchOutBytes() {   // [sys | trace | x_last | H_L, b_L of the residuals kept linearised]
  const size_t n = BA_BATCH_NMAX;
  return ((2 * (n * n + n) + 1 + 256 + n + (n * n + n)) * sizeof(double) + 255) & ~(size_t)255;
}
extern "C" {
dmvio_hip_ba_batch* dmvio_hip_ba_batch_create(dmvio_hip_ctx* ctx, int max_windows) {
  if (!ctx || max_windows < 1 || max_windows > 4096) { failmsg("ba_batch_create: bad argument"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { failmsg("ba_batch_create: hipSetDevice failed"); return nullptr; }
  dmvio_hip_ba_batch* B = new dmvio_hip_ba_batch();
  B->ctx = ctx; B->cap = max_windows;
  B->tab_stride = batchTabBytes(); B->out_stride = batchOutBytes();
  bool ok = hipStreamCreateWithFlags(&B->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->d_wins, sizeof(BAWinDev) * max_windows) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&B->h_wins, sizeof(BAWinDev) * max_windows, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->d_tab, B->tab_stride * max_windows) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&B->h_tab, B->tab_stride * max_windows, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->d_out, B->out_stride * max_windows) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&B->h_trace, sizeof(double) * (257 + BA_BATCH_NMAX) * max_windows, hipHostMallocDefault) == hipSuccess;
  for (int k = 0; k < 8 && ok; k++) ok = hipEventCreate(&B->ev[k]) == hipSuccess;
  B->gstream[0] = B->stream;
  for (int g = 1; g < dmvio_hip_ba_batch::BA_BATCH_STREAMS && ok; g++) ok = hipStreamCreateWithFlags(&B->gstream[g], hipStreamNonBlocking) == hipSuccess;
  for (int g = 0; g < dmvio_hip_ba_batch::BA_BATCH_STREAMS && ok; g++) for (int k = 0; k < 3 && ok; k++) ok = hipEventCreateWithFlags(&B->gev[g][k], hipEventDisableTiming) == hipSuccess;
  if (ok) ok = hipMemset(B->d_out, 0, B->out_stride * max_windows) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
  if (!ok) { failmsg("ba_batch_create: device / pinned allocation failed"); dmvio_hip_ba_batch_destroy(B); return nullptr; }
  if (max_windows >= 8) {
    const unsigned int hw = std::thread::hardware_concurrency();
    B->workers.start((int)std::min<unsigned int>(7, hw > 2 ? hw - 2 : 0), ctx->device);
  }
  return B;
}
void dmvio_hip_ba_batch_destroy(dmvio_hip_ba_batch* B) {
  if (!B) return;
  hipSetDevice(B->ctx->device);
  if (B->stream) { hipStreamSynchronize(B->stream); hipStreamDestroy(B->stream); }
  if (B->d_wins) hipFree(B->d_wins);
  if (B->h_wins) hipHostFree(B->h_wins);
  if (B->d_tab) hipFree(B->d_tab);
  if (B->h_tab) hipHostFree(B->h_tab);
  if (B->d_out) hipFree(B->d_out);
  if (B->h_trace) hipHostFree(B->h_trace);
  for (int k = 0; k < 8; k++) if (B->ev[k]) hipEventDestroy(B->ev[k]);
  for (int g = 1; g < dmvio_hip_ba_batch::BA_BATCH_STREAMS; g++) if (B->gstream[g]) { hipStreamSynchronize(B->gstream[g]); hipStreamDestroy(B->gstream[g]); }
  for (int g = 0; g < dmvio_hip_ba_batch::BA_BATCH_STREAMS; g++) for (int k = 0; k < 3; k++) if (B->gev[g][k]) hipEventDestroy(B->gev[g][k]);
  delete B;
}
// 1: the back substitution of the 68x68 solve in the host's order (one dependent chain of n^2 / 2 subtractions: x bit-identical to BAHost::ldltSolveTransposed, ~10 us more per
// iteration); 0 (default): column-oriented — the same terms in another association (measured |dx| <= 1e-12 relative, tests/test_ba_batch_gpu.py)
int dmvio_hip_ba_batch_set_exact_backsub(dmvio_hip_ba_batch* B, int on) {
  if (!B) return failmsg("ba_batch: null handle");
  std::lock_guard<std::mutex> lk(B->mu);
  B->exact_backsub = on ? 1 : 0;
  return 0;
}
int dmvio_hip_ba_batch_last_ms(dmvio_hip_ba_batch* B, float ms3[3]) {
  if (!B || !ms3) return failmsg("ba_batch: null argument");
  std::lock_guard<std::mutex> lk(B->mu);
  ms3[0] = B->last_ms[0]; ms3[1] = B->last_ms[1]; ms3[2] = B->last_ms[2];
  return 0;
}
// diagnostics: in-kernel timeline of window 0's last k_ba_solve of the last call, 100 MHz ticks since the kernel started: staged + settled, delta + bM_top + diagonal, system
// assembled, pivot order, permuted, factorised, back-substituted, x, resubstitution inputs + stepped states, exponentials, pair tables, energies
int dmvio_hip_ba_batch_last_solve_ticks(dmvio_hip_ba_batch* B, int ticks12[12]) {
  if (!B || !ticks12) return failmsg("ba_batch: null argument");
  std::lock_guard<std::mutex> lk(B->mu);
  for (int i = 0; i < 12; i++) ticks12[i] = B->h_wins[0].S.ticks[i];
  return 0;
}
// diagnostics: how window w's last k_ba_solve of the last call found its pivot order — 0 = ranks of the scaled diagonal (all |values| distinct), 1 = ties replayed
// (selection with swaps on one wavefront), 2 = a NaN on the diagonal (the literal loop)
int dmvio_hip_ba_batch_last_pivot_branch(dmvio_hip_ba_batch* B, int w, int* branch) {
  if (!B || !branch || w < 0 || w >= B->cap) return failmsg("ba_batch_last_pivot_branch: bad argument");
  std::lock_guard<std::mutex> lk(B->mu);
  *branch = B->h_wins[w].S.pivot_branch;
  return 0;
}
// Tests / diagnostics: the solve of EnergyFunctional.cpp:971-973 for a GIVEN system on the device, exactly as k_ba_solve runs it (Jacobi scaling (H_ii + 10)^-1/2, Eigen's
// pivot order, LDL^T, forward / back substitution on one 512-thread workgroup) — the device counterpart of dmvio_hip_ba_solve_ldlt.  HPassed: n x n row-major (the lower
// triangle is read), n = 4 + 8 F <= 100.  x_out[n]; perm_out[n] (may be NULL): the index the transpositions bring to position k; branch_out (may be NULL): 0 ranks /
// 1 ties / 2 NaN; zero_out (may be NULL): the matrix's first pivot was zero (x = 0).  exact_backsub as dmvio_hip_ba_batch_set_exact_backsub.
int dmvio_hip_ba_debug_solve(dmvio_hip_ctx* ctx, int n, const double* HPassed, const double* b_in, int exact_backsub, double* x_out, int* perm_out, int* branch_out, int* zero_out) {
  if (!ctx || !HPassed || !b_in || !x_out || n < 2 || n > 4 + 8 * BA_MAXF_CAP) return failmsg("ba_debug_solve: bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  double* d = nullptr;
  const size_t nin = (size_t)n * n + n, nout = 2 * (size_t)n + 2;
  HIPCHK(hipMalloc((void**)&d, sizeof(double) * (nin + nout)));
  std::vector<double> h(nin + nout, 0.0);
  memcpy(h.data(), HPassed, sizeof(double) * n * n); memcpy(h.data() + (size_t)n * n, b_in, sizeof(double) * n);
  hipError_t e = hipMemcpy(d, h.data(), sizeof(double) * nin, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const bool small = n <= 4 + 8 * BA_MAXF;
    const size_t lds = sizeof(double) * baSolveCoreLdsDoubles(n);
    if (small) hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, d, d + (size_t)n * n, exact_backsub, d + nin);
    else hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF_CAP>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, d, d + (size_t)n * n, exact_backsub, d + nin);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h.data() + nin, d + nin, sizeof(double) * nout, hipMemcpyDeviceToHost);
  }
  hipFree(d);
  if (e != hipSuccess) return failmsg((std::string("ba_debug_solve: ") + hipGetErrorString(e)).c_str());
  const double* o = h.data() + nin;
  memcpy(x_out, o, sizeof(double) * n);
  if (perm_out) for (int i = 0; i < n; i++) perm_out[i] = (int)o[n + i];
  if (branch_out) *branch_out = (int)o[2 * n];
  if (zero_out) *zero_out = (int)o[2 * n + 1];
  return 0;
}
// 0 (default): a batch of >= 4 windows is cut into up to three groups on three streams (at least two windows each), their launches interleaved; k >= 1: at most k groups
// (1 = the whole batch on one stream).  The grouping changes no result: no arithmetic crosses windows.
int dmvio_hip_ba_batch_set_streams(dmvio_hip_ba_batch* B, int streams) {
  if (!B || streams < 0) return failmsg("ba_batch_set_streams: bad argument");
  std::lock_guard<std::mutex> lk(B->mu);
  B->streams = std::min<int>(streams, dmvio_hip_ba_batch::BA_BATCH_STREAMS);
  return 0;
}
// which linearisation kernel a batch of >= 4 windows runs: 1 (default) = k_ba_linearize_b1, one lane per residual; 8 = k_ba_linearize_b, eight lanes per residual (what a
// single window runs).  Same values either way.
int dmvio_hip_ba_batch_set_linearize_lanes(dmvio_hip_ba_batch* B, int lanes) {
  if (!B || (lanes != 1 && lanes != 8)) return failmsg("ba_batch_set_linearize_lanes: 1 or 8");
  std::lock_guard<std::mutex> lk(B->mu);
  B->lin_lanes = lanes;
  return 0;
}
// measurement: HIP events around the stepped linearisation of the second iteration (k_ba_linearize_b over all windows of the call) -> dmvio_hip_ba_batch_last_ms()[2]
int dmvio_hip_ba_batch_set_profile(dmvio_hip_ba_batch* B, int on) {
  if (!B) return failmsg("ba_batch: null handle");
  std::lock_guard<std::mutex> lk(B->mu);
  B->profile = on ? 1 : 0;
  return 0;
}
}  // extern "C"

static int optimizeBatchGroup(dmvio_hip_ba_batch* B, const int Wn, dmvio_hip_ba* const* hs, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations, double* trace,
                              double* x_last) {
  const int F = hs[0]->H.F, n = hs[0]->H.n(), F2 = F * F, tot = 2 * (n * n + n);
  if (F < 2) { for (int w = 0; w < Wn; w++) { if (rmse) rmse[w] = 0; if (iterations) iterations[w] = 0; if (finalEnergy) finalEnergy[w] = 0; } return 0; }
  if (F < 3) mnumOptIts = 20;
  if (F < 4) mnumOptIts = 15;
  if (mnumOptIts < 1) return failmsg("ba_optimize_batch: mnumOptIts < 1 (the device-resident loop writes the trace's first row in its first solve)");
  hipStream_t s = B->stream;
  const auto t_call = std::chrono::steady_clock::now();
  auto stamp = [&](const int k) { B->host_us[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count(); };
  struct StreamSwap {   // the handles' own entry points (table uploads, the reset kernel) enqueue on the batch's stream for the duration of the call
    std::vector<std::pair<dmvio_hip_ba*, hipStream_t>> saved;
    ~StreamSwap() { for (auto& kv : saved) kv.first->stream = kv.second; }
  } swap;
  int gx_lin = 0, gx_pt8 = 0, gx_acc = 0, gx_res = 0, gx_pts = 0;
  const int n_gather = (tot + 256) / 256, n_stitch = F + F2;
  for (int w = 0; w < Wn; w++) {
    dmvio_hip_ba* b = hs[w];
    if (b->stream != s) { HIPCHK(hipStreamSynchronize(b->stream)); swap.saved.emplace_back(b, b->stream); b->stream = s; }
    const int nacc = b->nsC + F2 * b->nsTop + (F2 * F * b->nsD + 3) / 4;
    gx_lin = std::max(gx_lin, b->n_lin_blocks); gx_pt8 = std::max(gx_pt8, b->n_pt8_blocks); gx_acc = std::max(gx_acc, nacc); gx_res = std::max(gx_res, (b->H.R + 255) / 256); gx_pts = std::max(gx_pts, b->H.N);
  }
  auto prepare = [&](const int w) -> int {
    dmvio_hip_ba* b = hs[w];
    BAHost& H = b->H;
    b->vio = nullptr; b->vio_opt = nullptr; b->dynW = 1.0; H.gtsam = false;
    b->pending_reject = false; b->pending_trace = -1; b->sums_fresh = false; b->sys_ready = false;
    b->fullJ_applied = false;   // the batched linearisations relinearise and apply every residual without writing d_fullJ: what the buffer holds is no longer the applied state's
    if (int r = resolveTh(b)) return r;
    if (b->adj_dirty) { if (int r = uploadAdjoints(b)) return r; }
    // (the precalc table, the thresholds and the activation of all residuals travel with the batch: one upload, one launch for all windows)
    H.getNullspaces();
    H.prepareOrthogonalize();
    // ---- the window's record
    BAWinDev& V = B->h_wins[w];
    memset(&V, 0, sizeof(V));
    fillWindow(b);
    V.W = b->W; V.Wb = b->W;
    V.P = b->P; V.Rs = b->Rs;
    V.D = makeDecide(b, 0, true, false);
    for (int f = 0; f < BA_MAXF_CAP; f++) V.frameTH[f] = f < F ? H.fr[f].frameEnergyTH : 0.0f;
    V.D.frameTH = B->d_wins[w].frameTH;   // (an address: the record's own copy on the device)
    b->th_dirty = true;                    // the handle's own device copy is stale from here on; the host's values are brought up to date below
    dynFromHost(H, V.T); V.Tb = V.T;
    b->dyn_cur = V.T;
    {
      AccumArgs& A = V.A;
      A.F = F; A.N = H.N; A.nsTop = b->nsTop; A.nsD = b->nsD; A.nsC = b->nsC;
      A.top_begin = b->d_top_begin; A.top_members = b->d_top_members; A.scd_begin = b->d_scd_begin; A.scd_members = b->d_scd_members;
      A.accTop = b->d_accTop; A.accD = b->d_accD; A.accE = b->d_accE; A.accC = b->d_accC; A.numTop = b->d_numTop; A.numD = b->d_numD;
      A.ticks = nullptr;
    }
    V.SB = b->SB; V.adHost = b->d_adHost; V.adTarget = b->d_adTarget;
    char* out = B->d_out + B->out_stride * (size_t)w;
    V.sys = reinterpret_cast<double*>(out);
    V.ctl = b->d_ctl;
    V.n_lin_blocks = b->n_lin_blocks; V.n_pt8_blocks = b->n_pt8_blocks; V.n_acc_blocks = b->nsC + F2 * b->nsTop + (F2 * F * b->nsD + 3) / 4;
    V.n_res_blocks = (H.R + 255) / 256; V.n_gather_blocks = n_gather; V.n_stitch_blocks = n_stitch; V.n_lin1_blocks = (H.R + LIN_THREADS - 1) / LIN_THREADS;
    BASolveDev& S = V.S;
    S.F = F; S.n = n; S.stepped = 0; S.iterations_done = 0; S.n_accepted = 0; S.exact_backsub = B->exact_backsub;
    S.lambda = 1e-5;
    S.lastL = calcLEnergy(b); S.lastM = H.calcMEnergy(); S.newL = S.lastL; S.newM = S.lastM;
    // residuals kept linearised (dmvio_hip_ba_fix_linearization): what the three-pass accumulation and the linearised energy read, at the deltas of the state the window enters with
    V.n_lin = b->n_lin; V.n_lin_runs = (H.N + 49) / 50; V.lin_cnt = 0;
    if (b->n_lin > 0) {
      V.fullJ = b->d_fullJ; V.lin = b->d_lin; V.rtz = b->d_rtz; V.linRec = b->d_linRec; V.linActive = b->d_linActive; V.topActive = b->d_topActive; V.linE = b->d_linE;
      std::vector<float> adHT;
      H.adHTdeltaF(adHT);
      for (int k = 0; k < 2; k++) { memcpy(V.adHTdelta[k], adHT.data(), sizeof(float) * adHT.size()); for (int i = 0; i < 4; i++) V.cDeltaF[k][i] = H.cDeltaF[i]; }
    }
    for (int i = 0; i < 4; i++) { S.c_value[i] = H.c_value[i]; S.c_value_zero[i] = H.c_value_zero[i]; S.c_value_backup[i] = H.c_value[i]; S.cPrior[i] = H.cPrior[i]; S.cPriorF[i] = H.cPriorF[i]; }
    for (int f = 0; f < F; f++) {
      BAFrameDev& q = S.fr[f]; const BAFrameHost& h = H.fr[f];
      q.evalPT = h.evalPT; q.ab_exposure = h.ab_exposure;
      for (int i = 0; i < 10; i++) { q.state[i] = h.state[i]; q.state_zero[i] = h.state_zero[i]; q.state_backup[i] = h.state[i]; }
      for (int i = 0; i < 8; i++) q.prior[i] = h.prior[i];
    }
    // ---- the uploaded tables: [HM | bM | basis | adHostF | adTargetF]
    char* tab = B->h_tab + B->tab_stride * (size_t)w;
    char* dtab = B->d_tab + B->tab_stride * (size_t)w;
    double* tHM = reinterpret_cast<double*>(tab);
    const bool haveM = H.HM.size() == (size_t)n * n;
    S.haveM = haveM ? 1 : 0;
    if (haveM) { memcpy(tHM, H.HM.data(), sizeof(double) * n * n); memcpy(tHM + (size_t)n * n, H.bM.data(), sizeof(double) * n); }
    double* tBasis = tHM + (size_t)n * n + n;
    S.nBasis = (int)H.orthoBasis.size();
    for (int k = 0; k < S.nBasis; k++) memcpy(tBasis + (size_t)k * n, H.orthoBasis[k].data(), sizeof(double) * n);
    float* tAd = reinterpret_cast<float*>(tBasis + 7 * (size_t)n);
    memcpy(tAd, H.adHostF.data(), sizeof(float) * F2 * 64); memcpy(tAd + (size_t)F2 * 64, H.adTargetF.data(), sizeof(float) * F2 * 64);
    S.HM = reinterpret_cast<const double*>(dtab); S.bM = S.HM + (size_t)n * n; S.basis = S.bM + n;
    S.adHostF = reinterpret_cast<const float*>(S.basis + 7 * (size_t)n); S.adTargetF = S.adHostF + (size_t)F2 * 64;
    BAPrecalc* tPre = reinterpret_cast<BAPrecalc*>(tAd + 2 * (size_t)F2 * 64);
    memcpy(tPre, H.pre.data(), sizeof(BAPrecalc) * F2);
    V.pre = reinterpret_cast<const BAPrecalc*>(S.adTargetF + (size_t)F2 * 64);
    b->pre_static_valid = false;           // the handle's own table was not refreshed
    S.trace = V.sys + tot + 1; S.x_last = S.trace + 256;
    V.sysL = S.x_last + BA_BATCH_NMAX;
    return 0;
  };
  stamp(0);
  const size_t used_tab = ((size_t)n * n + n + 7 * (size_t)n) * sizeof(double) + 2 * (size_t)F2 * 64 * sizeof(float) + (size_t)F2 * sizeof(BAPrecalc);
  if (used_tab > B->tab_stride) return failmsg("ba_optimize_batch: table slab too small");
  const FrameStore fs = B->ctx->fs;
  const size_t solveLds = sizeof(double) * baSolveLdsDoubles(n, F, F <= BA_MAXF ? BASolveDims<BA_MAXF>::ALIAS_HM : BASolveDims<BA_MAXF_CAP>::ALIAS_HM);
  // Up to three groups of windows by default (at most BA_BATCH_STREAMS on request), one stream each, from 4 windows on: k_ba_solve is one workgroup per window (a 50 us latency chain on a handful of CUs), so while one
  // group solves, the other groups' linearisations / accumulations fill the device.  The groups share nothing.  Their launches are enqueued STAGE BY STAGE (initial chain of
  // every group, iteration 0 of every group, ...): a stream whose commands the host has not submitted yet cannot overlap with anything (measured: with the groups enqueued one
  // after the other the second one started three iterations late).  Group g starts behind group g-1's initial linearisation, which keeps the groups out of step.  A profiled
  // call (dmvio_hip_ba_batch_set_profile) runs as ONE group: its timed linearisation then covers all windows of the call, alone on the device.
  // The host's per-window work is pipelined along the groups too: group g's tables are prepared, uploaded and its first chain enqueued while the device already works on the
  // groups before it; behind the loop group g's states are written back (and its final linearisation enqueued) while the later groups still run.
  const int maxG = B->streams > 0 ? B->streams : 3;   // measured (tools/ba_batch_streams.py): three groups are best at W = 16 and 64; a fourth stream shares a hardware queue
                                                        // with another one (GPU_MAX_HW_QUEUES = 4, one of them busy with the handles' own streams) and loses
  const int G = (Wn >= 4 && !B->profile) ? std::max(1, std::min(maxG, Wn / 2)) : 1;
  struct Grp { hipStream_t st; int w0, cnt; };
  Grp grp[dmvio_hip_ba_batch::BA_BATCH_STREAMS];
  for (int g = 0; g < G; g++) { grp[g].st = B->gstream[g]; grp[g].w0 = (int)(((long long)Wn * g) / G); grp[g].cnt = (int)(((long long)Wn * (g + 1)) / G) - grp[g].w0; }
  // the eight-lane kernel hides latency (few windows); the one-lane kernel does an eighth of the lane work (a grid that fills the device)
  const bool lin1 = B->lin_lanes == 1 && Wn >= 4;
  const int gx_lin1 = (gx_res * 256 + LIN_THREADS - 1) / LIN_THREADS;
  const size_t patchLds = sizeof(float) * LIN_THREADS * BA_PATCH_STRIDE;   // the one-lane linearisation's per-lane 8x8 image windows
  auto linearize = [&](hipStream_t st, const BAWinDev* dwq, const int cnt, const int kind) {
    if (lin1) hipLaunchKernelGGL(k_ba_linearize_b1, dim3(gx_lin1, cnt), dim3(LIN_THREADS), patchLds, st, dwq, fs, kind);
    else hipLaunchKernelGGL(k_ba_linearize_b, dim3(gx_lin, cnt), dim3(LIN_THREADS), 0, st, dwq, fs, kind);
  };
  auto solve = [&](const Grp& q, const int it, const int finish) {
    if (finish) {
      if (F <= BA_MAXF) hipLaunchKernelGGL((k_ba_solve<BA_MAXF, true>), dim3(q.cnt), dim3(BA_SOLVE_THREADS), solveLds, q.st, B->d_wins + q.w0, it);
      else hipLaunchKernelGGL((k_ba_solve<BA_MAXF_CAP, true>), dim3(q.cnt), dim3(BA_SOLVE_THREADS), solveLds, q.st, B->d_wins + q.w0, it);
    } else {
      if (F <= BA_MAXF) hipLaunchKernelGGL((k_ba_solve<BA_MAXF, false>), dim3(q.cnt), dim3(BA_SOLVE_THREADS), solveLds, q.st, B->d_wins + q.w0, it);
      else hipLaunchKernelGGL((k_ba_solve<BA_MAXF_CAP, false>), dim3(q.cnt), dim3(BA_SOLVE_THREADS), solveLds, q.st, B->d_wins + q.w0, it);
    }
  };
  // a group that holds a window with residuals kept linearised runs EnergyFunctional's three accumulations (L / A / Schur pass: accumulateLin above) for those windows; its
  // other windows take their one ordinary accumulation in the A pass
  bool grpLin[dmvio_hip_ba_batch::BA_BATCH_STREAMS];
  for (int g = 0; g < G; g++) { grpLin[g] = false; for (int w = grp[g].w0; w < grp[g].w0 + grp[g].cnt; w++) grpLin[g] = grpLin[g] || hs[w]->n_lin > 0; }
  auto linRecords = [&](const Grp& q, const int gate, const bool sums) {   // the addPoint<1> records (and the A / L activity views); sums: + the linearised residuals' per-point sums
    const BAWinDev* dwq = B->d_wins + q.w0;
    hipLaunchKernelGGL(k_ba_lin_records_b, dim3(gx_res, q.cnt), dim3(256), 0, q.st, dwq, gate);
    if (sums) hipLaunchKernelGGL(k_ba_lin_point_sums_b, dim3((gx_pts + 255) / 256, q.cnt), dim3(256), 0, q.st, dwq, gate);
  };
  auto chain = [&](const Grp& q, const int g, const int backup, const int apply, const int gate, const bool sums_done) {   // applyRes + per-point sums -> accumulate -> stitch -> gather: the system of the (new) state
    const BAWinDev* dwq = B->d_wins + q.w0;
    if (!sums_done) hipLaunchKernelGGL(k_ba_point_sums_b, dim3(gx_pt8, q.cnt), dim3(256), 0, q.st, dwq, backup, apply, gate);
    for (int pass = grpLin[g] ? (int)BA_PASS_L : (int)BA_PASS_ALL; pass <= (grpLin[g] ? (int)BA_PASS_S : (int)BA_PASS_ALL); pass++) {
      hipLaunchKernelGGL(k_ba_accumulate_b, dim3(gx_acc, q.cnt), dim3(256), 0, q.st, dwq, gate, pass);
      hipLaunchKernelGGL(k_ba_stitch_b, dim3(n_stitch, q.cnt), dim3(64 * F), sizeof(StitchWave) * F, q.st, dwq, gate, pass);
      if (F <= BA_MAXF) hipLaunchKernelGGL((k_ba_stitch_gather_b<BA_MAXF>), dim3(n_gather, q.cnt), dim3(256), 0, q.st, dwq, gate, pass);
      else hipLaunchKernelGGL((k_ba_stitch_gather_b<BA_MAXF_CAP>), dim3(n_gather, q.cnt), dim3(256), 0, q.st, dwq, gate, pass);
    }
  };
  // ---- per group: its windows' tables (host), their upload, then every residual still in the graph active again (FullSystemOptimize.cpp:431-448), initial linearisation,
  // applyRes and the first system (:450-470).  With workers the windows are prepared in index order behind the caller's back: group g + 1's while group g is enqueued.
  std::atomic<int> prepared[dmvio_hip_ba_batch::BA_BATCH_STREAMS];
  for (int g = 0; g < G; g++) prepared[g].store(0, std::memory_order_relaxed);
  const bool async_prepare = !B->workers.th.empty() && Wn >= 8 && G > 1;
  auto groupOf = [&](const int w) { int g = 0; while (g + 1 < G && w >= grp[g + 1].w0) g++; return g; };
  if (async_prepare) B->workers.startAsync(Wn, [&](const int w) -> int { const int r = prepare(w); prepared[groupOf(w)].fetch_add(1, std::memory_order_release); return r; });
  else if (int r = B->workers.parallelFor(Wn, prepare)) return r;
  int rc_launch = 0;
  for (int g = 0; g < G; g++) {
    const Grp& q = grp[g];
    const BAWinDev* dwq = B->d_wins + q.w0;
    if (async_prepare) while (prepared[g].load(std::memory_order_acquire) < q.cnt) __builtin_ia32_pause();
    if (g == 0) stamp(1);
    if (async_prepare && B->workers.rc) { rc_launch = 1; break; }   // a window's preparation failed: nothing of it (or of the groups behind it) is launched
    if (g == 0) { HIPCHK(hipEventRecord(B->ev[0], s)); for (int k = 1; k < G; k++) HIPCHK(hipStreamWaitEvent(grp[k].st, B->ev[0], 0)); }   // (the other streams: behind whatever the batch's stream still holds)
    HIPCHK(hipMemcpyAsync(B->d_wins + q.w0, B->h_wins + q.w0, sizeof(BAWinDev) * q.cnt, hipMemcpyHostToDevice, q.st));
    HIPCHK(hipMemcpyAsync(B->d_tab + B->tab_stride * (size_t)q.w0, B->h_tab + B->tab_stride * (size_t)q.w0, B->tab_stride * (size_t)(q.cnt - 1) + used_tab, hipMemcpyHostToDevice, q.st));
    if (g > 0) HIPCHK(hipStreamWaitEvent(q.st, B->gev[g - 1][0], 0));   // the stagger
    hipLaunchKernelGGL(k_ba_reset_oob_b, dim3(gx_res, q.cnt), dim3(256), 0, q.st, dwq);
    linearize(q.st, dwq, q.cnt, BA_LINB_INITIAL);
    if (g + 1 < G) HIPCHK(hipEventRecord(B->gev[g][0], q.st));
    hipLaunchKernelGGL(k_ba_apply_b, dim3(gx_res, q.cnt), dim3(256), 0, q.st, dwq, 0, (int)BA_GATE_ALWAYS);
    if (grpLin[g]) linRecords(q, BA_GATE_ALWAYS, true);
    chain(q, g, 1, 0, BA_GATE_ALWAYS, false);
  }
  if (async_prepare) {
    const int r = B->workers.waitAsync();
    if (r || rc_launch) { for (int g = 0; g < G; g++) hipStreamSynchronize(grp[g].st); return r ? r : -1; }
  }
  // ---- the loop (:485-586): nothing in it waits for the host
  for (int it = 0; it < mnumOptIts; it++)
    for (int g = 0; g < G; g++) {
      const Grp& q = grp[g];
      const BAWinDev* dwq = B->d_wins + q.w0;
      solve(q, it, 0);
      if (grpLin[g]) hipLaunchKernelGGL(k_ba_lin_energy_b, dim3(gx_res, q.cnt), dim3(256), 0, q.st, B->d_wins + q.w0);   // E_L's linearised term of the stepped state, for the accept test
      const bool prof = B->profile && g == 0 && it == std::min(1, mnumOptIts - 1);
      if (prof) HIPCHK(hipEventRecord(B->ev[4], q.st));
      if (lin1) { hipLaunchKernelGGL(k_ba_resubstitute_b, dim3(gx_pt8, q.cnt), dim3(256), 0, q.st, dwq); linearize(q.st, dwq, q.cnt, BA_LINB_STEPPED_DONE); }
      else linearize(q.st, dwq, q.cnt, BA_LINB_STEPPED);
      if (prof) HIPCHK(hipEventRecord(B->ev[5], q.st));
      // rejected: restore + relinearise | accepted: applyRes + per-point sums (the last iteration's accepted step is only applied: nobody solves its system) — one launch
      const int what = it < mnumOptIts - 1 ? 0 : 1;
      if (grpLin[g] && what == 0) linRecords(q, BA_GATE_ACCEPTED, true);   // (the per-point sums below add the linearised residuals' Hdd / bd / Hcd)
      const int gx_post = std::max(lin1 ? gx_lin1 : gx_lin, what == 0 ? gx_pt8 : gx_res);
      if (lin1) hipLaunchKernelGGL((k_ba_post_decide_b<true>), dim3(gx_post, q.cnt), dim3(LIN_THREADS), patchLds, q.st, dwq, fs, what);
      else hipLaunchKernelGGL((k_ba_post_decide_b<false>), dim3(gx_post, q.cnt), dim3(LIN_THREADS), 0, q.st, dwq, fs, what);
      if (grpLin[g] && what == 0) linRecords(q, BA_GATE_ACCEPTED, false);  // (again behind applyRes: the A pass's activity view follows the applied states)
      if (what == 0) chain(q, g, 1, 1, BA_GATE_ACCEPTED, true);
    }
  // ---- settle the last decision; every group's states and traces come back on its own stream: [resInA | trace (64 x 4) | x_last] of a window is the tail of its system
  // slab, one strided copy per group
  for (int g = 0; g < G; g++) {
    const Grp& q = grp[g];
    solve(q, mnumOptIts, 1);
    HIPCHK(hipMemcpyAsync(B->h_wins + q.w0, B->d_wins + q.w0, sizeof(BAWinDev) * q.cnt, hipMemcpyDeviceToHost, q.st));
    HIPCHK(hipMemcpy2DAsync(B->h_trace + (size_t)(257 + BA_BATCH_NMAX) * q.w0, sizeof(double) * (257 + BA_BATCH_NMAX),
                            reinterpret_cast<const double*>(B->d_out + B->out_stride * (size_t)q.w0) + tot, B->out_stride, sizeof(double) * (257 + n), q.cnt, hipMemcpyDeviceToHost, q.st));
    HIPCHK(hipEventRecord(B->gev[g][1], q.st));
  }
  HIPCHK(hipGetLastError());
  stamp(2);
  // ---- back on the host: the optimised states, then the newest keyframe's new evaluation point (:596-603) and the final fix-linearisation (:604-609)
  const size_t tab_pre_off = ((size_t)n * n + n + 7 * (size_t)n) * sizeof(double) + 2 * (size_t)F2 * 64 * sizeof(float);
  auto writeBack = [&](const int w) -> int {
    dmvio_hip_ba* b = hs[w];
    BAHost& H = b->H;
    const BAWinDev& V = B->h_wins[w];
    const BASolveDev& S = V.S;
    H.calibSetValue(S.c_value);
    for (int i = 0; i < 4; i++) H.c_value_backup[i] = S.c_value_backup[i];
    for (int f = 0; f < F; f++) {
      BAHost::frameSetState(H.fr[f], S.fr[f].state);
      for (int i = 0; i < 10; i++) H.fr[f].state_backup[i] = S.fr[f].state_backup[i];
    }
    const double* tr = B->h_trace + (size_t)(257 + BA_BATCH_NMAX) * w + 1;
    H.resInA = (int)tr[-1];   // the count the last accumulation left behind (ef->resInA after the loop)
    const int done = S.iterations_done;
    b->iterations_done = done;
    for (int k = 0; k <= done && k < 64; k++) for (int c = 0; c < 4; c++) b->trace[k][c] = tr[4 * k + c];   // row 0: the initial state (written by the first solve)
    b->H.lastX.assign(tr + 256, tr + 256 + n);
    if (x_last) memcpy(x_last + (size_t)BA_BATCH_NMAX * w, tr + 256, sizeof(double) * n);
    BAFrameHost& last = H.fr[F - 1];
    double newStateZero[10] = {0, 0, 0, 0, 0, 0, last.state[6], last.state[7], 0, 0};
    last.evalPT = last.w2c;
    BAHost::frameSetState(last, newStateZero);
    BAHost::frameSetStateZero(last, newStateZero);
    H.setAdjointsF();
    b->adj_dirty = true;                   // uploaded by the next consumer (nothing in this call stitches again)
    H.setPrecalcValues();
    memcpy(reinterpret_cast<BAPrecalc*>(B->h_tab + B->tab_stride * (size_t)w + tab_pre_off), H.pre.data(), sizeof(BAPrecalc) * F2);
    fillWindow(b);
    BAWinDev& V2 = B->h_wins[w];
    V2.W = b->W;
    dynFromHost(H, V2.T); b->dyn_cur = V2.T;
    b->th_pending = false;                 // the thresholds live in the window's record; the newest one is read back behind the final linearisation
    return 0;
  };
  // group by group, in the order they finish (the stagger): wait for the group's states, write them back (the workers share a group's windows), upload the re-anchored
  // records / pair tables and enqueue the group's final linearisation on its stream — while the groups behind it still run their last iterations
  for (int g = 0; g < G; g++) {
    const Grp& q = grp[g];
    HIPCHK(hipEventSynchronize(B->gev[g][1]));
    if (g == 0) stamp(3);
    if (int r = B->workers.parallelFor(q.cnt, [&](const int i) { return writeBack(q.w0 + i); }, 4)) { for (int k = 0; k < G; k++) hipStreamSynchronize(grp[k].st); return r; }
    if (g == G - 1) stamp(4);
    HIPCHK(hipMemcpyAsync(B->d_wins + q.w0, B->h_wins + q.w0, sizeof(BAWinDev) * q.cnt, hipMemcpyHostToDevice, q.st));
    HIPCHK(hipMemcpy2DAsync(B->d_tab + B->tab_stride * (size_t)q.w0 + tab_pre_off, B->tab_stride, B->h_tab + B->tab_stride * (size_t)q.w0 + tab_pre_off, B->tab_stride, sizeof(BAPrecalc) * F2, q.cnt,
                            hipMemcpyHostToDevice, q.st));   // the re-anchored pair tables
    if (g == 0) HIPCHK(hipEventRecord(B->ev[2], q.st));
    linearize(q.st, B->d_wins + q.w0, q.cnt, BA_LINB_FINAL);
    hipLaunchKernelGGL(k_ba_apply_b, dim3(gx_res, q.cnt), dim3(256), 0, q.st, B->d_wins + q.w0, 1, (int)BA_GATE_ALWAYS);   // applyRes + linearizeAll(true)'s removal of inactive residuals
    HIPCHK(hipGetLastError());
    if (g > 0) { HIPCHK(hipEventRecord(B->gev[g][2], q.st)); HIPCHK(hipStreamWaitEvent(s, B->gev[g][2], 0)); }
  }
  HIPCHK(hipEventRecord(B->ev[3], s));
  stamp(5);
  HIPCHK(hipStreamSynchronize(s));
  stamp(6);
  // HIP-event times: [0] the whole call on the device (first upload .. last kernel), [1] from the first group's final linearisation to the last kernel
  HIPCHK(hipEventElapsedTime(&B->last_ms[0], B->ev[0], B->ev[3]));
  HIPCHK(hipEventElapsedTime(&B->last_ms[1], B->ev[2], B->ev[3]));
  B->last_ms[0] -= B->last_ms[1];   // (callers add the two)
  B->last_ms[2] = 0;
  if (B->profile) HIPCHK(hipEventElapsedTime(&B->last_ms[2], B->ev[4], B->ev[5]));
  for (int w = 0; w < Wn; w++) {
    dmvio_hip_ba* b = hs[w];
    BAHost& H = b->H;
    if (b->n_lin > 0) {   // accumulateLF_MT's system of the last accumulation, for dmvio_hip_ba_get_lf_system and the host-side solve entry points
      std::vector<double> lf((size_t)n * n + n);
      HIPCHK(hipMemcpy(lf.data(), B->h_wins[w].sysL, sizeof(double) * lf.size(), hipMemcpyDeviceToHost));
      H.HLraw.assign(lf.begin(), lf.begin() + (size_t)n * n); H.bLraw.assign(lf.begin() + (size_t)n * n, lf.end());
    }
    const double fe = b->h_res->E[0];
    H.fr[F - 1].frameEnergyTH = b->h_res->th[0];
    b->final_energy = fe;
    if (rmse) rmse[w] = sqrtf((float)(fe / (8 * H.resInA)));
    if (finalEnergy) finalEnergy[w] = fe;
    if (iterations) iterations[w] = b->iterations_done;
    if (trace) memcpy(trace + (size_t)256 * w, b->trace, sizeof(b->trace));
    b->sums_fresh = false; b->sys_ready = false;
    if (b->bounce.used || !b->bounce.outs.empty()) HIPCHK(b->bounce.finish(s));   // the staging area of this call's uploads is free again
  }
  stamp(7);
  return 0;
}

extern "C" {
// windows[W]: handles of the batch's context, each with its window set (set_window + set_graph), all distinct.  rmse / finalEnergy / iterations: W entries each (may be
// NULL); trace: W x 64 x 4 doubles or NULL ([E_A, E_L, E_M, accepted] per iteration, row 0 = the initial state).  Windows with different keyframe counts run as separate
// groups, one after the other.  Every window's result is what a batch of that window alone gives, bit for bit (no arithmetic crosses windows).
// Diagnostics: host clock (us since the call began) at the phase boundaries of the last dmvio_hip_ba_optimize_batch group: [0] stream hand-over done, [1] per-window tables
// prepared, [2] whole loop enqueued, [3] loop finished (first wait), [4] states written back, [5] final linearisation enqueued, [6] finished (second wait), [7] results out
int dmvio_hip_ba_batch_last_host_us(dmvio_hip_ba_batch* B, double us8[8]) {
  if (!B || !us8) return failmsg("ba_batch_last_host_us: null argument");
  std::lock_guard<std::mutex> lkB(B->mu);
  for (int k = 0; k < 8; k++) us8[k] = B->host_us[k];
  return 0;
}
int dmvio_hip_ba_optimize_batch(dmvio_hip_ba_batch* B, int W, dmvio_hip_ba* const* windows, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations, double* trace) {
  if (!B || !windows || W < 1) return failmsg("ba_optimize_batch: bad argument");
  if (W > B->cap) return failmsg("ba_optimize_batch: more windows than the batch was created for");
  std::lock_guard<std::mutex> lkB(B->mu);
  HIPCHK(hipSetDevice(B->ctx->device));
  // the handles' locks, in address order (two batches sharing handles cannot deadlock)
  std::vector<dmvio_hip_ba*> order(windows, windows + W);
  std::sort(order.begin(), order.end());
  for (int i = 0; i < W; i++) {
    if (!order[i]) return failmsg("ba_optimize_batch: null window");
    if (i > 0 && order[i] == order[i - 1]) return failmsg("ba_optimize_batch: a window appears twice");
    if (order[i]->ctx != B->ctx) return failmsg("ba_optimize_batch: a window belongs to another context");
  }
  std::vector<std::unique_lock<std::recursive_mutex>> locks;
  for (dmvio_hip_ba* b : order) locks.emplace_back(b->mu);
  for (int i = 0; i < W; i++) {
    dmvio_hip_ba* b = windows[i];
    if (!b->graph_ready) return failmsg("ba_optimize_batch: set_window + set_graph first");
    if (sharded(b)) return failmsg("ba_optimize_batch: a window sharded over ranks cannot join a batch");
  }
  // groups of equal keyframe count, in the caller's order
  std::vector<char> doneW(W, 0);
  for (int i = 0; i < W; i++) {
    if (doneW[i]) continue;
    std::vector<int> idx;
    for (int j = i; j < W; j++) if (!doneW[j] && windows[j]->H.F == windows[i]->H.F) { idx.push_back(j); doneW[j] = 1; }
    const int Wn = (int)idx.size();
    std::vector<dmvio_hip_ba*> hs(Wn);
    std::vector<float> r(Wn); std::vector<double> fe(Wn), tr((size_t)256 * Wn); std::vector<int> its(Wn);
    for (int k = 0; k < Wn; k++) hs[k] = windows[idx[k]];
    if (int rc = optimizeBatchGroup(B, Wn, hs.data(), mnumOptIts, r.data(), fe.data(), its.data(), tr.data(), nullptr)) return rc;
    for (int k = 0; k < Wn; k++) {
      if (rmse) rmse[idx[k]] = r[k];
      if (finalEnergy) finalEnergy[idx[k]] = fe[k];
      if (iterations) iterations[idx[k]] = its[k];
      if (trace) memcpy(trace + (size_t)256 * idx[k], tr.data() + (size_t)256 * k, sizeof(double) * 256);
    }
  }
  return 0;
}
// dmvio_hip_ba_optimize of this handle through the device-resident loop (a batch of one window, created on first use): no PCIe poll per iteration.  The host-driven loop
// (default) stays the reference for the hook branch (dmvio_hip_ba_optimize_vio), which needs the host in every iteration anyway.
int dmvio_hip_ba_set_device_loop(dmvio_hip_ba* b, int on) {
  if (!b) return failmsg("ba: null handle");
  BA_LOCK(b);
  b->device_loop = on != 0;
  return 0;
}
// the last solve's x (n = 4 + 8F doubles, MINUS the step: EnergyFunctional::lastX) of the last optimize / gn_iteration / solve of this window
int dmvio_hip_ba_get_last_x(dmvio_hip_ba* b, double* x_out) {
  if (!b || !x_out) return failmsg("ba_get_last_x: null argument");
  BA_LOCK(b);
  if ((int)b->H.lastX.size() != b->H.n()) return failmsg("ba_get_last_x: no solve yet");
  memcpy(x_out, b->H.lastX.data(), sizeof(double) * b->H.n());
  return 0;
}
}  // extern "C"
