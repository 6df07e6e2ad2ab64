// C ABI of the immature-point path (include/dmvio_hip.h): ImmaturePoint construction and FullSystem::traceNewCoarse.
#include <vector>
#include <cmath>
#include <cstring>
#include "../../include/dmvio_hip.h"
#include "internal.h"
#include "lie_dev.h"
#include "immature_kernels.hpp"

using namespace dmv;

struct dmvio_hip_immature {
  dmvio_hip_ctx* ctx = nullptr;
  int capacity = 0, n = 0, max_tag = -1;   // max_tag: largest host_tag among the points (validated against the tables of a call)
  ImmaturePts P{};
  ImmatureSettings S;
  float* d_tables = nullptr;   // [KRKi 9H | Kt 3H | aff 2H], H <= 64
  float* h_tables = nullptr;   // pinned
  int* h_counts = nullptr;     // pinned: status histogram of the last traceNewCoarse, written by k_status_hist
  int* d_uv_stage = nullptr;   // 2 x capacity ints
  float* d_opt_tables = nullptr;   // [R 9 F*F | t 3 F*F | aff 2 F*F], F <= 8
  float* h_opt_tables = nullptr;
  int *d_result = nullptr, *d_res_state = nullptr;
  float* d_idepth = nullptr;
  unsigned char* d_select = nullptr;
  DmvBounce bounce;            // caller-owned arrays cross PCIe through the library's pinned memory (internal.h)
  std::vector<void*> allocs;
};

#define IMM_READY(m) do { if (!(m)) return failmsg("null immature handle"); HIPCHK(hipSetDevice((m)->ctx->device)); } while (0)
enum { IMM_MAX_HOSTS = 64 };

template <class T>
static int ialloc(dmvio_hip_immature* m, T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, sizeof(T) * std::max<size_t>(n, 1)));
  HIPCHK(hipMemset(*p, 0, sizeof(T) * std::max<size_t>(n, 1)));
  // hipMemset clears on the NULL stream without blocking the host, and the handle's stream is non-blocking: without this wait an upload enqueued next could
  // land before the clear does (seen with two processes sharing a GPU)
  HIPCHK(hipStreamSynchronize(nullptr));
  m->allocs.push_back(*p);
  return 0;
}

extern "C" {

dmvio_hip_immature* dmvio_hip_immature_create(dmvio_hip_ctx* ctx, int capacity) {
  if (!ctx || capacity < 1) { failmsg("immature_create: bad argument"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { failmsg("immature_create: hipSetDevice failed"); return nullptr; }
  dmvio_hip_immature* m = new dmvio_hip_immature();
  m->ctx = ctx; m->capacity = capacity;
  ImmaturePts& P = m->P;
  const size_t c = capacity;
  if (ialloc(m, &P.u, c) || ialloc(m, &P.v, c) || ialloc(m, &P.host, c) || ialloc(m, &P.color, 8 * c) || ialloc(m, &P.weights, 8 * c) || ialloc(m, &P.gradH, 4 * c) ||
      ialloc(m, &P.energyTH, c) || ialloc(m, &P.idepth_min, c) || ialloc(m, &P.idepth_max, c) || ialloc(m, &P.quality, c) || ialloc(m, &P.lastTraceUV, 2 * c) ||
      ialloc(m, &P.lastTracePixelInterval, c) || ialloc(m, &P.lastTraceStatus, c) || ialloc(m, &m->d_tables, 14 * IMM_MAX_HOSTS) || ialloc(m, &m->d_uv_stage, 2 * c) || ialloc(m, &m->d_opt_tables, 14 * 64) || ialloc(m, &m->d_result, c) || ialloc(m, &m->d_res_state, 8 * c) ||
      ialloc(m, &m->d_idepth, c) || ialloc(m, &m->d_select, c) || hipHostMalloc((void**)&m->h_opt_tables, sizeof(float) * 14 * 64, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&m->h_tables, sizeof(float) * 14 * IMM_MAX_HOSTS, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&m->h_counts, sizeof(int) * 8, hipHostMallocDefault) != hipSuccess) {
    for (void* p : m->allocs) hipFree(p);
    delete m;
    return nullptr;
  }
  return m;
}
void dmvio_hip_immature_destroy(dmvio_hip_immature* m) {
  if (!m) return;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) hipFree(p);
  if (m->h_tables) hipHostFree(m->h_tables);
  if (m->h_counts) hipHostFree(m->h_counts);
  if (m->h_opt_tables) hipHostFree(m->h_opt_tables);
  m->bounce.release();
  delete m;
}
int dmvio_hip_immature_clear(dmvio_hip_immature* m) { IMM_READY(m); m->n = 0; m->max_tag = -1; return 0; }
int dmvio_hip_immature_count(dmvio_hip_immature* m) { return m ? m->n : -1; }

int dmvio_hip_immature_add_points(dmvio_hip_immature* m, int host_tag, int host_slot, int n, const int* u, const int* v) {
  IMM_READY(m);
  dmvio_hip_ctx* c = m->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  if (n < 0 || !u || !v) return failmsg("immature_add_points: bad argument");
  if (m->n + n > m->capacity) return failmsg("immature_add_points: capacity exceeded");
  if (host_slot < 0 || host_slot >= c->n_slots || host_tag < 0 || host_tag >= IMM_MAX_HOSTS) return failmsg("immature_add_points: slot / tag out of range");
  if (int r = dmv_ensure_row_major_locked(c, host_slot)) return r;
  // the constructor reads the 2x2 cell of every pattern pixel: u +- 2 .. +1 must be inside the image (pixel selector margin, PixelSelector2.cpp)
  for (int i = 0; i < n; i++)
    if (u[i] < 2 || v[i] < 2 || u[i] + 3 >= c->w || v[i] + 3 >= c->h) return failmsg("immature_add_points: point closer than 3 px to the border");
  if (n == 0) return m->n;
  const int first = m->n;
  {
    // the integer pixel positions become the float arrays the kernels read, written straight into the pinned staging memory (no wait: the copies are ordered before the
    // constructor kernel on the stream, and the staging area is not reused before the next synchronisation)
    size_t off;   // ONE reservation for both arrays: a second one could drain and rewind the staging area under the first
    HIPCHK(m->bounce.reserve(sizeof(float) * 2 * (size_t)n, c->stream, &off));
    float* uf = reinterpret_cast<float*>(m->bounce.h + off); float* vf = uf + n;
    for (int i = 0; i < n; i++) { uf[i] = (float)u[i]; vf[i] = (float)v[i]; }
    HIPCHK(hipMemcpyAsync(m->P.u + first, uf, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(m->P.v + first, vf, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  }
  m->P.n = first + n;
  hipLaunchKernelGGL(k_immature_init, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->levelPtr(host_slot, 0), c->w, first, n, m->P, host_tag, m->S);
  HIPCHK(hipGetLastError());
  m->n = first + n;
  m->max_tag = std::max(m->max_tag, host_tag);
  return first;
}

int dmvio_hip_immature_get_static(dmvio_hip_immature* m, float* u, float* v, int* host_tag, float* color8, float* weights8, float* gradH4, float* energyTH) {
  IMM_READY(m);
  hipStream_t s = m->ctx->stream;
  const size_t n = m->n;
  if (u) HIPCHK(m->bounce.d2h(u, m->P.u, sizeof(float) * n, s));
  if (v) HIPCHK(m->bounce.d2h(v, m->P.v, sizeof(float) * n, s));
  if (host_tag) HIPCHK(m->bounce.d2h(host_tag, m->P.host, sizeof(int) * n, s));
  if (color8) HIPCHK(m->bounce.d2h(color8, m->P.color, sizeof(float) * 8 * n, s));
  if (weights8) HIPCHK(m->bounce.d2h(weights8, m->P.weights, sizeof(float) * 8 * n, s));
  if (gradH4) HIPCHK(m->bounce.d2h(gradH4, m->P.gradH, sizeof(float) * 4 * n, s));
  if (energyTH) HIPCHK(m->bounce.d2h(energyTH, m->P.energyTH, sizeof(float) * n, s));
  HIPCHK(m->bounce.finish(s));
  return 0;
}
int dmvio_hip_immature_get_state(dmvio_hip_immature* m, float* idepth_min, float* idepth_max, float* quality, float* lastTraceUV2, float* lastTracePixelInterval,
                                 int* lastTraceStatus) {
  IMM_READY(m);
  hipStream_t s = m->ctx->stream;
  const size_t n = m->n;
  if (idepth_min) HIPCHK(m->bounce.d2h(idepth_min, m->P.idepth_min, sizeof(float) * n, s));
  if (idepth_max) HIPCHK(m->bounce.d2h(idepth_max, m->P.idepth_max, sizeof(float) * n, s));
  if (quality) HIPCHK(m->bounce.d2h(quality, m->P.quality, sizeof(float) * n, s));
  if (lastTraceUV2) HIPCHK(m->bounce.d2h(lastTraceUV2, m->P.lastTraceUV, sizeof(float) * 2 * n, s));
  if (lastTracePixelInterval) HIPCHK(m->bounce.d2h(lastTracePixelInterval, m->P.lastTracePixelInterval, sizeof(float) * n, s));
  if (lastTraceStatus) HIPCHK(m->bounce.d2h(lastTraceStatus, m->P.lastTraceStatus, sizeof(int) * n, s));
  HIPCHK(m->bounce.finish(s));
  return 0;
}
int dmvio_hip_immature_set_state(dmvio_hip_immature* m, const float* idepth_min, const float* idepth_max, const float* quality, const int* lastTraceStatus) {
  IMM_READY(m);
  hipStream_t s = m->ctx->stream;
  const size_t n = m->n;
  if (idepth_min) HIPCHK(m->bounce.h2d(m->P.idepth_min, idepth_min, sizeof(float) * n, s));
  if (idepth_max) HIPCHK(m->bounce.h2d(m->P.idepth_max, idepth_max, sizeof(float) * n, s));
  if (quality) HIPCHK(m->bounce.h2d(m->P.quality, quality, sizeof(float) * n, s));
  if (lastTraceStatus) HIPCHK(m->bounce.h2d(m->P.lastTraceStatus, lastTraceStatus, sizeof(int) * n, s));
  HIPCHK(m->bounce.finish(s));
  return 0;
}

// traceOn of every point against the frame in new_slot; per-host tables indexed by the points' host_tag
int dmvio_hip_immature_trace(dmvio_hip_immature* m, int new_slot, int n_hosts, const float* KRKi9, const float* Kt3, const float* aff2) {
  IMM_READY(m);
  dmvio_hip_ctx* c = m->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  if (!KRKi9 || !Kt3 || !aff2 || n_hosts < 1 || n_hosts > IMM_MAX_HOSTS) return failmsg("immature_trace: bad argument");
  if (new_slot < 0 || new_slot >= c->n_slots) return failmsg("immature_trace: frame slot out of range");
  if (int r = dmv_ensure_row_major_locked(c, new_slot)) return r;
  if (m->n == 0) return 0;
  if (m->max_tag >= n_hosts) return failmsg("immature_trace: a point's host_tag has no table row (host_tag >= n_hosts)");
  m->P.n = m->n;
  if (n_hosts <= IMM_ARG_HOSTS) {
    TraceTablesArg T;
    memset(&T, 0, sizeof(T));
    memcpy(T.KRKi, KRKi9, sizeof(float) * 9 * n_hosts);
    memcpy(T.Kt, Kt3, sizeof(float) * 3 * n_hosts);
    memcpy(T.aff, aff2, sizeof(float) * 2 * n_hosts);
    hipLaunchKernelGGL(k_immature_trace<TraceTablesArg>, dim3((m->n + 3) / 4), dim3(256), 0, c->stream, c->levelPtr(new_slot, 0), c->w, c->h, m->P, T, m->S);
  } else {
    HIPCHK(hipStreamSynchronize(c->stream));   // the pinned tables of a previous call may still be in flight
    float* t = m->h_tables;
    memcpy(t, KRKi9, sizeof(float) * 9 * n_hosts);
    memcpy(t + 9 * IMM_MAX_HOSTS, Kt3, sizeof(float) * 3 * n_hosts);
    memcpy(t + 12 * IMM_MAX_HOSTS, aff2, sizeof(float) * 2 * n_hosts);
    HIPCHK(hipMemcpyAsync(m->d_tables, t, sizeof(float) * 14 * IMM_MAX_HOSTS, hipMemcpyHostToDevice, c->stream));
    TraceTables T;
    T.KRKi = m->d_tables; T.Kt = m->d_tables + 9 * IMM_MAX_HOSTS; T.aff = m->d_tables + 12 * IMM_MAX_HOSTS;
    hipLaunchKernelGGL(k_immature_trace<TraceTables>, dim3((m->n + 3) / 4), dim3(256), 0, c->stream, c->levelPtr(new_slot, 0), c->w, c->h, m->P, T, m->S);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// FullSystem::traceNewCoarse (FullSystem.cpp:541-584): per-host KRKi / Kt / affine tables from the poses, traceOn of every immature point,
// status histogram counts6 = {good, oob, outlier, skipped, badcondition, uninitialized}
int dmvio_hip_trace_new_coarse(dmvio_hip_immature* m, int new_slot, const double new_w2c7[7], const double new_aff[2], float new_exposure, int n_hosts,
                               const double* host_c2w7, const double* host_aff2, const float* host_exposure, const double fxfycxcy[4], int counts6[6]) {
  IMM_READY(m);
  if (!new_w2c7 || !new_aff || !host_c2w7 || !host_aff2 || !host_exposure || !fxfycxcy || n_hosts < 1 || n_hosts > IMM_MAX_HOSTS) return failmsg("trace_new_coarse: bad argument");
  std::vector<float> KRKi(9 * (size_t)n_hosts), Kt(3 * (size_t)n_hosts), aff(2 * (size_t)n_hosts);
  const float fx = (float)fxfycxcy[0], fy = (float)fxfycxcy[1], cx = (float)fxfycxcy[2], cy = (float)fxfycxcy[3];
  const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
  // K.inverse(): Eigen's 3x3 cofactor inverse
  const float a = K[0], e = K[4], cc = K[2], ff = K[5];
  const float det = a * (e * 1.0f - ff * 0.0f), invdet = 1.0f / det;
  const float Ki[9] = {(e * 1.0f - ff * 0.0f) * invdet, (cc * 0.0f - 0.0f * 1.0f) * invdet, (0.0f * ff - cc * e) * invdet,
                       (ff * 0.0f - 0.0f * 1.0f) * invdet, (a * 1.0f - cc * 0.0f) * invdet, (cc * 0.0f - a * ff) * invdet,
                       (0.0f * 0.0f - e * 0.0f) * invdet, (0.0f * 0.0f - a * 0.0f) * invdet, (a * e - 0.0f * 0.0f) * invdet};
  const Pose Tn = poseFrom7(new_w2c7);
  for (int hI = 0; hI < n_hosts; hI++) {
    const Pose T = poseMul(Tn, poseFrom7(host_c2w7 + 7 * hI));
    double Rd[9];
    quatToR(T.q, Rd);
    float R[9], t[3], KR[9];
    for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
    for (int i = 0; i < 3; i++) t[i] = (float)T.t[i];
    for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) KR[r * 3 + q] = K[r * 3 + 0] * R[q] + K[r * 3 + 1] * R[3 + q] + K[r * 3 + 2] * R[6 + q];
    for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) KRKi[9 * hI + r * 3 + q] = KR[r * 3 + 0] * Ki[q] + KR[r * 3 + 1] * Ki[3 + q] + KR[r * 3 + 2] * Ki[6 + q];
    for (int r = 0; r < 3; r++) Kt[3 * hI + r] = K[r * 3 + 0] * t[0] + K[r * 3 + 1] * t[1] + K[r * 3 + 2] * t[2];
    double ab[2];
    affFromTo(host_exposure[hI], new_exposure, host_aff2[2 * hI], host_aff2[2 * hI + 1], new_aff[0], new_aff[1], ab);
    aff[2 * hI] = (float)ab[0]; aff[2 * hI + 1] = (float)ab[1];
  }
  if (int r = dmvio_hip_immature_trace(m, new_slot, n_hosts, KRKi.data(), Kt.data(), aff.data())) return r;
  if (counts6) {
    for (int k = 0; k < 6; k++) counts6[k] = 0;
    if (m->n > 0) {
      dmvio_hip_ctx* c = m->ctx;
      std::lock_guard<std::mutex> lk(c->mu);
      hipLaunchKernelGGL(k_status_hist, dim3(1), dim3(1024), 0, c->stream, (const int*)m->P.lastTraceStatus, m->n, m->h_counts);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));
      for (int k = 0; k < 6; k++) counts6[k] = m->h_counts[k];
    }
  }
  return 0;
}

// FullSystem::optimizeImmaturePoint for the selected points (FullSystemOptPoint.cpp:51-205; caller: activatePointsMT_Reductor,
// FullSystem.cpp:587-602).  The pair tables PRE_RTll / PRE_tTll / PRE_aff_mode are FrameFramePrecalc::set (HessianBlocks.cpp:193-223).
int dmvio_hip_immature_optimize(dmvio_hip_immature* m, int F, const int* frame_slots, const double* w2c7, const double* aff2, const float* exposure,
                                const double fxfycxcy[4], const unsigned char* select, int minObs, int* result, float* idepth, int* res_state) {
  IMM_READY(m);
  dmvio_hip_ctx* c = m->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  if (F < 2 || F > 8 || !frame_slots || !w2c7 || !aff2 || !exposure || !fxfycxcy || !result || !idepth) return failmsg("immature_optimize: bad argument");
  if (m->n == 0) return 0;
  if (m->max_tag >= F) return failmsg("immature_optimize: a point's host_tag is not a keyframe index of this window (host_tag >= F)");
  HIPCHK(hipStreamSynchronize(c->stream));
  float* tb = m->h_opt_tables;
  float *R = tb, *t = tb + 9 * 64, *aff = tb + 12 * 64;
  OptTables T;
  T.F = F;
  for (int f = 0; f < 8; f++) T.slot[f] = 0;
  for (int f = 0; f < F; f++) {
    if (frame_slots[f] < 0 || frame_slots[f] >= c->n_slots) return failmsg("immature_optimize: frame slot out of range");
    if (int r = dmv_ensure_row_major_locked(c, frame_slots[f])) return r;
    T.slot[f] = frame_slots[f];
  }
  for (int hI = 0; hI < F; hI++) {
    const Pose c2w = poseInv(poseFrom7(w2c7 + 7 * hI));
    for (int tI = 0; tI < F; tI++) {
      const Pose l = poseMul(poseFrom7(w2c7 + 7 * tI), c2w);
      double Rd[9];
      quatToR(l.q, Rd);
      const int o = hI * F + tI;
      for (int i = 0; i < 9; i++) R[9 * o + i] = (float)Rd[i];
      for (int i = 0; i < 3; i++) t[3 * o + i] = (float)l.t[i];
      double ab[2];
      affFromTo(exposure[hI], exposure[tI], aff2[2 * hI], aff2[2 * hI + 1], aff2[2 * tI], aff2[2 * tI + 1], ab);
      aff[2 * o] = (float)ab[0]; aff[2 * o + 1] = (float)ab[1];
    }
  }
  HIPCHK(hipMemcpyAsync(m->d_opt_tables, tb, sizeof(float) * 14 * 64, hipMemcpyHostToDevice, c->stream));
  T.R = m->d_opt_tables; T.t = m->d_opt_tables + 9 * 64; T.aff = m->d_opt_tables + 12 * 64;
  T.fxl = (float)fxfycxcy[0]; T.fyl = (float)fxfycxcy[1]; T.cxl = (float)fxfycxcy[2]; T.cyl = (float)fxfycxcy[3];
  T.fxli = 1.0f / T.fxl; T.fyli = 1.0f / T.fyl;   // CalibHessian::setValueScaled (HessianBlocks.h:373-387)
  if (select) HIPCHK(m->bounce.h2d(m->d_select, select, m->n, c->stream));
  m->P.n = m->n;
  hipLaunchKernelGGL(k_immature_optimize, dim3((m->n + 3) / 4), dim3(256), 0, c->stream, c->fs, c->w, c->h, m->P, T, select ? m->d_select : nullptr, minObs,
                     100.0f /* setting_minIdepthH_act */, 3 /* setting_GNItsOnPointActivation */, m->S.huberTH, m->d_result, m->d_idepth, m->d_res_state);
  HIPCHK(hipGetLastError());
  HIPCHK(m->bounce.d2h(result, m->d_result, sizeof(int) * m->n, c->stream));
  HIPCHK(m->bounce.d2h(idepth, m->d_idepth, sizeof(float) * m->n, c->stream));
  if (res_state) HIPCHK(m->bounce.d2h(res_state, m->d_res_state, sizeof(int) * (size_t)m->n * F, c->stream));
  HIPCHK(m->bounce.finish(c->stream));
  return 0;
}

}  // extern "C"
