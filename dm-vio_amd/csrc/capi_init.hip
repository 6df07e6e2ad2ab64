// C ABI of the initializer hot function (include/dmvio_hip.h): CoarseInitializer::calcResAndGS.
#include <vector>
#include <cmath>
#include <cstring>
#include "../../include/dmvio_hip.h"
#include "internal.h"
#include "lie_dev.h"
#include "init_kernels.hpp"

using namespace dmv;

struct dmvio_hip_initializer {
  dmvio_hip_ctx* ctx = nullptr;
  int capacity = 0, n = 0;
  // ONE device slab for the inputs and one for the per-point results, laid out per call for the n points of the level ([u | v | iR | outlierTH | energy(2) | idepth_new | isGood]
  // and [energy_new(2) | maxstep | lastHessian_new | JbBuffer_new(10) | isGood_new]); each crosses PCIe as ONE copy from / into pinned staging memory — a calcResAndGS call of
  // the initializer's LM loop is ~15 us of kernels, thirteen separate pageable copies cost ten times that
  float *d_in = nullptr, *d_res = nullptr, *h_in = nullptr, *h_res = nullptr;
  float *d_partials = nullptr, *d_out = nullptr, *h_out = nullptr;
  bool upload_pending = false;
  std::vector<void*> allocs;
};
static inline size_t padn(int n) { return ((size_t)n + 63) & ~(size_t)63; }   // every array of a slab starts on a 256-byte boundary
enum { IN_FLOATS = 7, RES_FLOATS = 14 };   // per point, + one byte each (isGood / isGood_new) at the end of the slab
enum { INIT_MAX_BLOCKS = 256 };

template <class T>
static int nalloc(dmvio_hip_initializer* m, T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, sizeof(T) * std::max<size_t>(n, 1)));
  HIPCHK(hipMemset(*p, 0, sizeof(T) * std::max<size_t>(n, 1)));
  // hipMemset clears on the NULL stream without blocking the host, and the handle's stream is non-blocking: without this wait an upload enqueued next could
  // land before the clear does (seen with two processes sharing a GPU)
  HIPCHK(hipStreamSynchronize(nullptr));
  m->allocs.push_back(*p);
  return 0;
}
#define INIT_READY(m) do { if (!(m)) return failmsg("null initializer handle"); HIPCHK(hipSetDevice((m)->ctx->device)); } while (0)

extern "C" {

dmvio_hip_initializer* dmvio_hip_initializer_create(dmvio_hip_ctx* ctx, int capacity) {
  if (!ctx || capacity < 1) { failmsg("initializer_create: bad argument"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { failmsg("initializer_create: hipSetDevice failed"); return nullptr; }
  dmvio_hip_initializer* m = new dmvio_hip_initializer();
  m->ctx = ctx; m->capacity = capacity;
  const size_t c = padn(capacity);
  if (nalloc(m, &m->d_in, (IN_FLOATS + 1) * c) || nalloc(m, &m->d_res, (RES_FLOATS + 1) * c) || nalloc(m, &m->d_partials, (size_t)INIT_MAX_BLOCKS * IN_PART) ||
      nalloc(m, &m->d_out, IN_PART) || hipHostMalloc((void**)&m->h_out, sizeof(float) * IN_PART, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&m->h_in, sizeof(float) * (IN_FLOATS + 1) * c, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&m->h_res, sizeof(float) * (RES_FLOATS + 1) * c, hipHostMallocDefault) != hipSuccess) {
    for (void* p : m->allocs) hipFree(p);
    if (m->h_out) hipHostFree(m->h_out);
    if (m->h_in) hipHostFree(m->h_in);
    if (m->h_res) hipHostFree(m->h_res);
    delete m;
    return nullptr;
  }
  return m;
}
void dmvio_hip_initializer_destroy(dmvio_hip_initializer* m) {
  if (!m) return;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) hipFree(p);
  if (m->h_out) hipHostFree(m->h_out);
  if (m->h_in) hipHostFree(m->h_in);
  if (m->h_res) hipHostFree(m->h_res);
  delete m;
}

// the per-level point set (struct Pnt, CoarseInitializer.h:44-83): the fields calcResAndGS reads
int dmvio_hip_initializer_set_points(dmvio_hip_initializer* m, int n, const float* u, const float* v, const float* iR, const unsigned char* isGood,
                                     const float* energy2, const float* outlierTH) {
  INIT_READY(m);
  if (n < 0 || n > m->capacity || !u || !v || !iR || !isGood || !energy2 || !outlierTH) return failmsg("initializer_set_points: bad argument");
  hipStream_t s = m->ctx->stream;
  if (m->upload_pending) { HIPCHK(hipStreamSynchronize(s)); m->upload_pending = false; }   // the staging buffer is still being read by the previous upload
  const size_t P = padn(n);
  memcpy(m->h_in, u, sizeof(float) * n); memcpy(m->h_in + P, v, sizeof(float) * n); memcpy(m->h_in + 2 * P, iR, sizeof(float) * n);
  memcpy(m->h_in + 3 * P, outlierTH, sizeof(float) * n); memcpy(m->h_in + 4 * P, energy2, sizeof(float) * 2 * n);
  memcpy(m->h_in + 7 * P, isGood, n);
  // [u | v | iR | outlierTH | energy] and, behind the gap idepth_new fills per evaluation, isGood: two copies, no wait — the evaluation's own synchronisation covers them
  HIPCHK(hipMemcpyAsync(m->d_in, m->h_in, sizeof(float) * 6 * P, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(m->d_in + 7 * P, m->h_in + 7 * P, n, hipMemcpyHostToDevice, s));
  m->upload_pending = true;
  m->n = n;
  return 0;
}

int dmvio_hip_initializer_calc_res_and_gs(dmvio_hip_initializer* m, int lvl, int first_slot, int new_slot, const double Ki9[9], const float fxfycxcy_lvl[4],
                                          const double refToNew7[7], const double aff_ab[2], const float* idepth_new, float alphaW, float alphaK,
                                          float couplingWeight, double priorY, double priorX, float* H_out64, float* b_out8, float* H_sc64, float* b_sc8,
                                          float res3[3], float* energy_new2, unsigned char* isGood_new, float* maxstep, float* lastHessian_new, float* JbBuffer_new10) {
  INIT_READY(m);
  dmvio_hip_ctx* c = m->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  if (!Ki9 || !fxfycxcy_lvl || !refToNew7 || !aff_ab || !idepth_new || !H_out64 || !b_out8 || !H_sc64 || !b_sc8 || !res3) return failmsg("initializer_calc: null argument");
  if (lvl < 0 || lvl >= c->levels || first_slot < 0 || first_slot >= c->n_slots || new_slot < 0 || new_slot >= c->n_slots) return failmsg("initializer_calc: level / slot out of range");
  if (lvl == 0) { if (int r = dmv_ensure_row_major_locked(c, first_slot)) return r; if (int r = dmv_ensure_row_major_locked(c, new_slot)) return r; }
  const int n = m->n;
  hipStream_t s = c->stream;
  const Pose T = poseFrom7(refToNew7);
  InitArgs A;
  {
    double Rd[9];
    quatToR(T.q, Rd);
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) A.RKi[r * 3 + q] = (float)(Rd[r * 3 + 0] * Ki9[q] + Rd[r * 3 + 1] * Ki9[3 + q] + Rd[r * 3 + 2] * Ki9[6 + q]);
    for (int i = 0; i < 3; i++) A.t[i] = (float)T.t[i];
  }
  A.aff0 = (float)std::exp(aff_ab[0]); A.aff1 = (float)aff_ab[1];
  A.fxl = fxfycxcy_lvl[0]; A.fyl = fxfycxcy_lvl[1]; A.cxl = fxfycxcy_lvl[2]; A.cyl = fxfycxcy_lvl[3];
  A.wl = c->wl[lvl]; A.hl = c->hl[lvl];
  A.couplingWeight = couplingWeight; A.huberTH = 9.0f;
  // alpha energy (CoarseInitializer.cpp:497-535); EAlpha.A is identically 0 in the reference
  const double tsq = T.t[0] * T.t[0] + T.t[1] * T.t[1] + T.t[2] * T.t[2];
  float alphaEnergy = alphaW * (0.0f + tsq * n);
  float alphaOpt;
  if (alphaEnergy > alphaK * n) { alphaOpt = 0; alphaEnergy = alphaK * n; }
  else alphaOpt = alphaW;
  A.alphaOpt = alphaOpt;
  const size_t PN = padn(n);
  memcpy(m->h_in + 6 * PN, idepth_new, sizeof(float) * n);
  HIPCHK(hipMemcpyAsync(m->d_in + 6 * PN, m->h_in + 6 * PN, sizeof(float) * n, hipMemcpyHostToDevice, s));
  InitPts P;
  P.n = n; P.u = m->d_in; P.v = m->d_in + PN; P.iR = m->d_in + 2 * PN; P.outlierTH = m->d_in + 3 * PN; P.energy = m->d_in + 4 * PN; P.idepth_new = m->d_in + 6 * PN;
  P.isGood = reinterpret_cast<unsigned char*>(m->d_in + 7 * PN);
  P.energy_new = m->d_res; P.maxstep = m->d_res + 2 * PN; P.lastHessian_new = m->d_res + 3 * PN; P.JbBuffer_new = m->d_res + 4 * PN;
  P.isGood_new = reinterpret_cast<unsigned char*>(m->d_res + 14 * PN);
  const int G = std::max(1, std::min((int)INIT_MAX_BLOCKS, (n + 255) / 256));
  hipLaunchKernelGGL(k_init_partial, dim3(G), dim3(256), 0, s, c->levelPtr(first_slot, lvl), c->levelPtr(new_slot, lvl), P, A, m->d_partials);
  hipLaunchKernelGGL(k_init_final, dim3(1), dim3(128), 0, s, m->d_partials, G, m->d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(m->h_out, m->d_out, sizeof(float) * IN_PART, hipMemcpyDeviceToHost, s));
  const bool want = energy_new2 || isGood_new || maxstep || lastHessian_new || JbBuffer_new10;
  if (want) HIPCHK(hipMemcpyAsync(m->h_res, m->d_res, sizeof(float) * 14 * PN + n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  m->upload_pending = false;
  if (energy_new2) memcpy(energy_new2, m->h_res, sizeof(float) * 2 * n);
  if (maxstep) memcpy(maxstep, m->h_res + 2 * PN, sizeof(float) * n);
  if (lastHessian_new) memcpy(lastHessian_new, m->h_res + 3 * PN, sizeof(float) * n);
  if (JbBuffer_new10) memcpy(JbBuffer_new10, m->h_res + 4 * PN, sizeof(float) * 10 * n);
  if (isGood_new) memcpy(isGood_new, m->h_res + 14 * PN, n);
  // unpack the two upper triangles, then the tail of calcResAndGS (:582-613)
  float M[2][9][9];
  for (int w = 0; w < 2; w++) {
    int k = 0;
    for (int r = 0; r < 9; r++) for (int q = r; q < 9; q++) { M[w][r][q] = M[w][q][r] = m->h_out[45 * w + k]; k++; }
  }
  for (int r = 0; r < 8; r++) {
    for (int q = 0; q < 8; q++) { H_out64[r * 8 + q] = M[0][r][q]; H_sc64[r * 8 + q] = M[1][r][q]; }
    b_out8[r] = M[0][r][8]; b_sc8[r] = M[1][r][8];
  }
  H_out64[0] += alphaOpt * n; H_out64[9] += alphaOpt * n; H_out64[18] += alphaOpt * n;
  double lg[6];
  poseLogHost(T, lg);
  const float tlog[3] = {(float)lg[0], (float)lg[1], (float)lg[2]};
  b_out8[0] += tlog[0] * alphaOpt * n; b_out8[1] += tlog[1] * alphaOpt * n; b_out8[2] += tlog[2] * alphaOpt * n;
  H_out64[9] += priorY; b_out8[1] += priorY * T.t[1];
  H_out64[0] += priorX; b_out8[0] += priorX * T.t[0];
  res3[0] = m->h_out[90]; res3[1] = alphaEnergy; res3[2] = (float)(2 * (size_t)n);   // accE[0].num: every point once per loop (:351-470, :503-516)
  return 0;
}

}  // extern "C"
