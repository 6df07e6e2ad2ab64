// libdmvio_hip.so — C ABI implementation (include/dmvio_hip.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include <atomic>

#include "../../include/dmvio_hip.h"
#include "common.h"
#include "lie_dev.h"
#include "image_kernels.hpp"
#include "ref_kernels.hpp"
#include "tracker_kernels.hpp"

using namespace dmv;

#include "internal.h"

struct dmvio_hip_tracker {
  dmvio_hip_ctx* ctx = nullptr;
  TrackerDev dev{};
  bool haveK = false, haveRef = false;
  RefLevels R{};
  int n_tiles = 0;
  float *d_idp = nullptr, *d_wsp = nullptr, *d_idp2 = nullptr, *d_wsp2 = nullptr, *d_dense = nullptr;
  int *d_tile_count = nullptr, *d_tile_base = nullptr, *d_pc_n = nullptr, *d_seg = nullptr;
  unsigned long long* d_flow_mask = nullptr;
  size_t flow_words = 0;
  float4* d_pc[DMV_MAX_LEVELS] = {};
  float4** d_pc_ptrs = nullptr;
  float* d_pts = nullptr;
  int pts_cap = 0;
  float *d_partials = nullptr, *h_tot = nullptr;
  unsigned int* d_arrive = nullptr;   // arrive counter of k_eval_fused (zero between launches)
  unsigned int* d_leave = nullptr;    // evaluation server: the launch (by its first ticket) whose workgroups have been told to leave by an idle time-out
  unsigned int eval_ticket = 0;       // ticket of the last fused evaluation; the kernel stores it behind the sums in h_tot
  // evaluation server (k_eval_server): one launch per tracked frame, requests through a mailbox in host-coherent memory
  unsigned int* h_mail = nullptr;     // EVAL_MAIL_DWORDS dwords: [0] request ticket, [1..] EvalP, [last] the ticket again (written before [0])
  float* h_rec = nullptr;             // EVAL_SERVER_MAX_BLOCKS records of EVAL_RECORD_FLOATS floats: every server workgroup stores its partial sums + the ticket into its own
  bool server_on = false;             // a server kernel was launched for server_slot and has not been told to quit
  int server_slot = -1, server_G = 0;
  // hypothesis-parallel trackNewCoarse: the tries after the first are split over `xworld` ranks, their per-try records summed over the ranks (every record is written by
  // exactly one rank, the others add zeros) by `xchg`
  std::function<int(double*, size_t)> xchg;
  int xrank = 0, xworld = 0;
  bool debug_split1 = false;          // dmvio_hip_tracker_debug_split_single_rank: a group of ONE rank still takes the split path (tests of the transport on a one-device box)
  unsigned int server_session = 0;    // identity of the current serverStart .. serverStop session (mailbox dword EVAL_MAIL_SESSION, kernel argument)
  long long server_idle_ticks = 500000;   // the server leaves after this long without a request (100 MHz ticks: 5 ms); dmvio_hip_tracker_set_server_idle_us
  int use_server = 1;                 // DMVIO_HIP_EVAL_SERVER=0: one k_eval_fused launch per evaluation instead
  int single_host_lm = 1;             // DMVIO_HIP_SINGLE_HOST_LM=0: a single alignment problem runs the device-resident LM (cluster mode) instead of the host LM + server
  int last_vio_iterations = 0;
  int eval_blocks_override = 0;
  int max_eval_blocks = 1024;
  LMProblemIn *h_in = nullptr;   // 2 x batch_cap entries of pinned host memory, read by the kernel directly (each workgroup copies its 120 B into LDS)
  std::vector<int> h_rank_keys, h_rank_cnt;    // set_ref: host-side ranking of the points that share a pixel
  std::vector<unsigned char> h_rank;
  std::vector<int> last_repeat_lvl;            // per problem of the last fetch: the level that ran twice (or -1) ...
  std::vector<double> last_first_pass_res;     // ... and its residual after the first pass
  int batch_kernel = 0;          // dmvio_hip_tracker_set_batch_kernel: 1 = full batches on k_track_lm_pp (control steps beside the evaluations)
  unsigned int* d_pp_next = nullptr;
  int debug_mode = 0, log_cap = 0, log_B = 0;   // dmvio_hip_tracker_debug_record_replay
  EvalP* d_log = nullptr; int* d_log_n = nullptr; float* d_log_sink = nullptr;
  LMProblemOut *d_out = nullptr, *h_out = nullptr;   // h_out: 2 x batch_cap entries of pinned host memory the kernel writes its results into (alternating per launch)
  int out_cur = 0, out_fetch = 0, staged_half = 0;    // half of the last launch / half a pending fetch_begin refers to / half staged for the next launch
  int batch_cap = 0, staged_B = 0, staged_coarsest = 0;
  bool staged_unlaunched = false;   // _track_batch_stage ran and _track_batch_launch has not yet: a single-frame call must not slip in front of that batch
  long long last_evals = 0, last_point_evals = 0, last_ticks_step = 0, last_ticks_eval = 0;
  int lm_threads_override = 0, lm_waves_override = 0, lm_cluster_override = 0, last_cluster = 0, last_threads = 0;
  hipEvent_t done_event[2] = {nullptr, nullptr};   // recorded behind each launch: the results of that half are in host memory once it has completed
  int fetch_pending_B = 0;
  float* d_cl_part = nullptr;          // cluster mode: B x 2 x C x ACC_PAD partial sums
  unsigned int* d_cl_cnt = nullptr;    // cluster mode: arrive counters
  size_t cl_part_cap = 0; int cl_cnt_cap = 0;
};

std::string& dmv_err() { static thread_local std::string e; return e; }
unsigned int& dmv_err_epoch() { static thread_local unsigned int n = 0; return n; }

extern "C" {

const char* dmvio_hip_last_error(void) { return dmv_err().c_str(); }

int dmvio_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int pyrLevels(int w, int h) {
  // setGlobalCalib (src/dso/util/globalCalib.cpp:47-55)
  int wl = w, hl = h, used = 1;
  while (wl % 2 == 0 && hl % 2 == 0 && wl * hl > 5000 && used < DMV_MAX_LEVELS) { wl /= 2; hl /= 2; used++; }
  return used;
}

dmvio_hip_ctx* dmvio_hip_create(int device, int w, int h, int n_frame_slots) {
  if (w <= 0 || h <= 0 || n_frame_slots <= 0) { failmsg("dmvio_hip_create: bad arguments"); return nullptr; }
  int ndev = 0;
  HIPCHKP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { failmsg("dmvio_hip_create: no such device"); return nullptr; }
  HIPCHKP(hipSetDevice(device));
  dmvio_hip_ctx* c = new dmvio_hip_ctx();
  c->device = device; c->w = w; c->h = h; c->n_slots = n_frame_slots;
  c->levels = pyrLevels(w, h);
  size_t off = 0;
  for (int l = 0; l < c->levels; l++) {
    c->wl[l] = w >> l; c->hl[l] = h >> l;
    c->fs.level_off[l] = off;
    off += (size_t)c->wl[l] * c->hl[l];
  }
  c->fs.levels = c->levels;
  c->fs.slot_stride = (off + 63) & ~(size_t)63;
  HIPCHKP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHKP(hipMalloc((void**)&c->fs.base, sizeof(float) * c->fs.slot_stride * n_frame_slots));
  HIPCHKP(hipMemsetAsync(c->fs.base, 0, sizeof(float) * c->fs.slot_stride * n_frame_slots, c->stream));
  HIPCHKP(hipMalloc((void**)&c->fs.build_gen, sizeof(unsigned int) * 2 * n_frame_slots));   // build_gen | bad_gen: equal (0) = not known to be clean
  HIPCHKP(hipMemsetAsync(c->fs.build_gen, 0, sizeof(unsigned int) * 2 * n_frame_slots, c->stream));
  c->fs.bad_gen = c->fs.build_gen + n_frame_slots;
  HIPCHKP(hipMalloc((void**)&c->fs.lvl0, sizeof(const float*) * n_frame_slots));
  c->h_lvl0.resize(n_frame_slots);
  for (int s = 0; s < n_frame_slots; s++) c->h_lvl0[s] = c->fs.own_level(s, 0);
  HIPCHKP(hipMemcpyAsync(c->fs.lvl0, c->h_lvl0.data(), sizeof(const float*) * n_frame_slots, hipMemcpyHostToDevice, c->stream));
  HIPCHKP(hipMalloc((void**)&c->fs.tiled0, n_frame_slots));
  HIPCHKP(hipMemsetAsync(c->fs.tiled0, 0, n_frame_slots, c->stream));
  c->h_tiled.assign(n_frame_slots, 0);
  HIPCHKP(hipMalloc((void**)&c->d_upload, sizeof(float) * w * h));
  c->pg.levels = c->levels;
  for (int l = 0; l < c->levels; l++) { c->pg.w[l] = c->wl[l]; c->pg.h[l] = c->hl[l]; }
  {
    // tile shape of the pyramid build: as wide as the image (contiguous level-0 memory per workgroup), at least 2^(levels-1) rows for the 2x2 reductions
    int twl = 9;
    while (twl > 7 && ((1 << (twl - 1)) >= w || (PYR_TILE_PX >> twl) < (1 << (c->levels - 1)))) twl--;
    c->pg.tw_log2 = twl;
    const int TW = 1 << twl, TH = PYR_TILE_PX >> twl;
    c->pg.tiles_x = (w + TW - 1) / TW; c->pg.tiles_y = (h + TH - 1) / TH;
  }
  HIPCHKP(hipMalloc((void**)&c->d_f3, sizeof(float) * 3 * w * h));
  HIPCHKP(hipStreamSynchronize(c->stream));
  return c;
}

// Tile shape of the LDS-tile pyramid build (k_build_pyramids / k_build_pyramids_raw): 2^tw_log2 pixels wide, 4096 / 2^tw_log2 high; 7 .. 9, and the tile must keep
// 2^(levels-1) rows.  A measurement knob (profiles/r03_stream_ceilings.md); the default is as wide as the image allows.
int dmvio_hip_set_pyramid_tile_log2(dmvio_hip_ctx* c, int tw_log2) {
  if (!c) return failmsg("null context");
  if (tw_log2 < 7 || tw_log2 > 9 || (PYR_TILE_PX >> tw_log2) < (1 << (c->levels - 1))) return failmsg("set_pyramid_tile_log2: 7 .. 9, with at least 2^(levels-1) rows per tile");
  std::lock_guard<std::mutex> lk(c->mu);
  c->pg.tw_log2 = tw_log2;
  const int TW = 1 << tw_log2, TH = PYR_TILE_PX >> tw_log2;
  c->pg.tiles_x = (c->w + TW - 1) / TW; c->pg.tiles_y = (c->h + TH - 1) / TH;
  return 0;
}

void dmvio_hip_destroy(dmvio_hip_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  hipFree(c->fs.base);
  hipFree(c->fs.build_gen);
  hipFree(c->fs.lvl0);
  hipFree(c->fs.tiled0);
  hipFree(c->d_upload);
  c->bounce.release();
  hipFree(c->d_f3);
  for (auto& L : c->slot_lists) { if (L.d) hipFree(L.d); if (L.h) hipHostFree(L.h); }
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int dmvio_hip_pyr_levels(const dmvio_hip_ctx* c) { return c ? c->levels : 0; }

int dmvio_hip_set_stream(dmvio_hip_ctx* c, void* s) {
  if (!c) return failmsg("null ctx");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (s) {
    if (c->own_stream) { HIPCHK(hipStreamDestroy(c->stream)); }
    c->stream = (hipStream_t)s; c->own_stream = false;
  } else if (!c->own_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true;
  }
  return 0;
}

static hipStream_t buildStream(const dmvio_hip_ctx* c) { return c->build_stream ? c->build_stream : c->stream; }
// The stream the BATCHED pyramid builds (dmvio_hip_frames_from_device_batch / _attach_device_batch / _from_raw_device_batch) are enqueued on; NULL (default): the context's
// stream.  With a stream of its own the build of batch k+1 overlaps the tracking of batch k — the build is bound by HBM, k_track_lm by its L1 miss path and VALU issue.  The
// CALLER orders the two streams (events): a build must not start before the consumers of the slots it rewrites have finished, a consumer not before the build of its slots.
int dmvio_hip_set_build_stream(dmvio_hip_ctx* c, void* s) {
  if (!c) return failmsg("null ctx");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(buildStream(c)));
  c->build_stream = (hipStream_t)s;
  return 0;
}

int dmvio_hip_synchronize(dmvio_hip_ctx* c) {
  if (!c) return failmsg("null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->build_stream) HIPCHK(hipStreamSynchronize(c->build_stream));
  return 0;
}

// ------------------------------------------------------------------ frames
// the wave-autonomous builds (a 4 x 8 pixel block per thread, levels in registers: image_kernels.hpp) take pyramids of at most four levels on images whose sides are
// multiples of 8; everything else (and dmvio_hip_set_raw_batch_kernel(ctx, 0)) goes to the LDS-tile builds
static bool regBuild(const dmvio_hip_ctx* c) { return c->raw_batch_kernel && c->levels <= 4 && (c->w % 8) == 0 && (c->h % 8) == 0; }
static dim3 regGrid(const dmvio_hip_ctx* c, int B) { return dim3(((c->w / 4) * (c->h / 8) + 255) / 256, B); }
static int buildPyramid(dmvio_hip_ctx* c, int slot, const float* d_color) {
  hipLaunchKernelGGL(k_build_pyramids, dim3(c->pg.tiles_x * c->pg.tiles_y, 1), dim3(256), 0, c->stream, d_color, (size_t)0, c->pg, c->fs,
                     (const int*)nullptr, slot, ++c->build_gen, 0);
  c->h_lvl0[slot] = c->fs.own_level(slot, 0); c->h_tiled[slot] = 0;
  HIPCHK(hipGetLastError());
  return 0;
}

// see internal.h
int dmv_ensure_row_major_locked(dmvio_hip_ctx* c, int slot) {
  if (slot < 0 || slot >= c->n_slots || !c->h_tiled[slot]) return 0;
  HIPCHK(hipSetDevice(c->device));
  const int n = c->w * c->h;
  float* own = c->fs.own_level(slot, 0);
  hipLaunchKernelGGL(k_untile_level0, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float*)own, c->w, c->h, c->d_upload);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(own, c->d_upload, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->fs.tiled0 + slot, 0, 1, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));   // consumers on other streams (a window optimiser's) read the plane next
  c->h_tiled[slot] = 0;
  return 0;
}
int dmv_ensure_row_major(dmvio_hip_ctx* c, int slot) {
  if (!c) return failmsg("null ctx");
  std::lock_guard<std::mutex> lk(c->mu);
  return dmv_ensure_row_major_locked(c, slot);
}
// What dmvio_hip_frames_from_raw_device_batch writes as level 0: 0 = row-major (default), 1 = 8x4 tiles (the coarse tracker's batch kernel reads them natively; other consumers
// convert the slot back on first use).  Tiles need w % 8 == 0 and h % 4 == 0; other sizes are always row-major.  Measured (profiles/r04_*): the tiled plane saves lines per tap
// (2.4 instead of 4.3) but costs twelve dword loads with their own tile addresses instead of four vector loads — k_track_lm 3.78 vs 3.50 ms per 4096 frames — hence not the default.
int dmvio_hip_set_raw_batch_layout(dmvio_hip_ctx* c, int tiled) {
  if (!c) return failmsg("null ctx");
  std::lock_guard<std::mutex> lk(c->mu);
  c->raw_batch_tiled = tiled ? 1 : 0;
  return 0;
}
// Which kernels build the pyramids of the raw-image batch and of frames attached in place: 1 (default) = the wave-autonomous register builds where the geometry allows it
// (at most four pyramid levels, both sides multiples of 8), 0 = always the LDS-tile builds.  Both write the same bits; the switch exists for A/B measurements and the parity test of the two.
int dmvio_hip_set_raw_batch_kernel(dmvio_hip_ctx* c, int variant) {
  if (!c) return failmsg("null ctx");
  if (variant != 0 && variant != 1) return failmsg("set_raw_batch_kernel: variant must be 0 or 1");
  std::lock_guard<std::mutex> lk(c->mu);
  c->raw_batch_kernel = variant;
  return 0;
}
int dmvio_hip_frame_level0_is_tiled(dmvio_hip_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= c->n_slots) return failmsg("frame_level0_is_tiled: bad argument");
  std::lock_guard<std::mutex> lk(c->mu);
  return c->h_tiled[slot] ? 1 : 0;
}

int dmvio_hip_frame_upload(dmvio_hip_ctx* c, int slot, const float* host) {
  if (!c || !host) return failmsg("frame_upload: null argument");
  if (slot < 0 || slot >= c->n_slots) return failmsg("frame_upload: slot out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(c->bounce.h2d(c->d_upload, host, sizeof(float) * c->w * c->h, c->stream));   // a pageable 1 MB source costs ~0.27 ms per copy, a memcpy + a pinned copy ~0.07 ms
  if (int r = buildPyramid(c, slot, c->d_upload)) return r;
  HIPCHK(c->bounce.finish(c->stream));
  return 0;
}

// ---- raw camera image -> photometric + geometric undistortion -> pyramids (Undistort::undistort + FrameHessian::makeImages)
struct dmvio_hip_undistorter {
  dmvio_hip_ctx* ctx = nullptr;
  UndistortDev U{};
  int bytes_per_px = 1;
  float *d_G = nullptr, *d_vig = nullptr, *d_rx = nullptr, *d_ry = nullptr;
  void *d_raw = nullptr, *h_raw = nullptr;
};
dmvio_hip_undistorter* dmvio_hip_undistorter_create(dmvio_hip_ctx* c, int wOrg, int hOrg, int bits, const float* G, const float* vignetteMapInv, const float* remapX,
                                                    const float* remapY) {
  if (!c || wOrg < 1 || hOrg < 1 || (bits != 8 && bits != 16)) { failmsg("undistorter_create: bad argument"); return nullptr; }
  if ((remapX == nullptr) != (remapY == nullptr)) { failmsg("undistorter_create: remapX / remapY must both be given or both be NULL"); return nullptr; }
  if (!remapX && (wOrg != c->w || hOrg != c->h)) { failmsg("undistorter_create: passthrough needs wOrg x hOrg == w x h"); return nullptr; }
  if (remapX) {
    // the bilinear taps of a remapped pixel are (xi, yi), (xi+1, yi), (xi, yi+1), (xi+1, yi+1) of the raw image with xi = (int)x, yi = (int)y: they are in bounds
    // exactly when 0 <= x < wOrg-1 and 0 <= y < hOrg-1.  The reference's remap generation keeps 0 < x < wOrg-1, 0 < y < hOrg-1 and marks everything else with -1
    // (Undistort.cpp:920-942, "make rounding resistant"), so its tables always pass; a table that does not would read out of bounds.
    const size_t nOutChk = (size_t)c->w * c->h;
    for (size_t i = 0; i < nOutChk; i++) {
      const float x = remapX[i], y = remapY[i];
      if (x < 0) continue;
      if (!(x < (float)(wOrg - 1)) || !(y >= 0) || !(y < (float)(hOrg - 1)) || (int)x + 1 > wOrg - 1 || (int)y + 1 > hOrg - 1) {
        failmsg("undistorter_create: remap entry outside the raw image (mark invalid pixels with remapX = -1)");
        return nullptr;
      }
    }
  }
  if (hipSetDevice(c->device) != hipSuccess) { failmsg("undistorter_create: hipSetDevice failed"); return nullptr; }
  dmvio_hip_undistorter* u = new dmvio_hip_undistorter();
  u->ctx = c; u->bytes_per_px = bits / 8;
  const size_t nOrg = (size_t)wOrg * hOrg, nOut = (size_t)c->w * c->h, nG = bits == 8 ? 256 : 65536;
  bool ok = hipMalloc(&u->d_raw, nOrg * u->bytes_per_px) == hipSuccess && hipHostMalloc(&u->h_raw, nOrg * u->bytes_per_px, hipHostMallocDefault) == hipSuccess;
  auto up = [&](float** d, const float* h, size_t n) {
    if (!h) return true;
    return hipMalloc((void**)d, sizeof(float) * n) == hipSuccess && hipMemcpy(*d, h, sizeof(float) * n, hipMemcpyHostToDevice) == hipSuccess;
  };
  ok = ok && up(&u->d_G, G, nG) && up(&u->d_vig, G ? vignetteMapInv : nullptr, nOrg) && up(&u->d_rx, remapX, nOut) && up(&u->d_ry, remapY, nOut);
  if (!ok) { failmsg("undistorter_create: device allocation failed"); dmvio_hip_undistorter_destroy(u); return nullptr; }
  u->U.wOrg = wOrg; u->U.hOrg = hOrg; u->U.w = c->w; u->U.h = c->h;
  u->U.G = u->d_G; u->U.vignetteMapInv = u->d_vig; u->U.remapX = u->d_rx; u->U.remapY = u->d_ry; u->U.factor = 1.0f;
  return u;
}
void dmvio_hip_undistorter_destroy(dmvio_hip_undistorter* u) {
  if (!u) return;
  hipSetDevice(u->ctx->device);
  hipStreamSynchronize(u->ctx->stream);
  if (u->d_raw) hipFree(u->d_raw);
  if (u->h_raw) hipHostFree(u->h_raw);
  if (u->d_G) hipFree(u->d_G);
  if (u->d_vig) hipFree(u->d_vig);
  if (u->d_rx) hipFree(u->d_rx);
  if (u->d_ry) hipFree(u->d_ry);
  delete u;
}
int dmvio_hip_frame_upload_raw(dmvio_hip_ctx* c, dmvio_hip_undistorter* u, int slot, const void* raw, float factor, float* undistorted_out) {
  if (!c || !u || !raw || u->ctx != c) return failmsg("frame_upload_raw: bad argument");
  if (slot < 0 || slot >= c->n_slots) return failmsg("frame_upload_raw: slot out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  const size_t nOrg = (size_t)u->U.wOrg * u->U.hOrg, nOut = (size_t)c->w * c->h;
  HIPCHK(hipStreamSynchronize(c->stream));   // the pinned staging buffer of a previous upload
  memcpy(u->h_raw, raw, nOrg * u->bytes_per_px);
  HIPCHK(hipMemcpyAsync(u->d_raw, u->h_raw, nOrg * u->bytes_per_px, hipMemcpyHostToDevice, c->stream));
  UndistortDev U = u->U;
  U.factor = factor;
  if (u->bytes_per_px == 1) hipLaunchKernelGGL((k_undistort<unsigned char>), dim3((nOut + 255) / 256), dim3(256), 0, c->stream, (const unsigned char*)u->d_raw, U, c->d_upload);
  else hipLaunchKernelGGL((k_undistort<unsigned short>), dim3((nOut + 255) / 256), dim3(256), 0, c->stream, (const unsigned short*)u->d_raw, U, c->d_upload);
  HIPCHK(hipGetLastError());
  if (int r = buildPyramid(c, slot, c->d_upload)) return r;
  if (undistorted_out) HIPCHK(c->bounce.d2h(undistorted_out, c->d_upload, sizeof(float) * nOut, c->stream));
  HIPCHK(c->bounce.finish(c->stream));
  return 0;
}

int dmvio_hip_frame_from_device(dmvio_hip_ctx* c, int slot, const float* dev) {
  if (!c || !dev) return failmsg("frame_from_device: null argument");
  if (slot < 0 || slot >= c->n_slots) return failmsg("frame_from_device: slot out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  return buildPyramid(c, slot, dev);  // asynchronous on the ctx stream
}

// slot list of a batched build -> device (kept while the same list comes again); caller holds c->mu
static int stageSlots(dmvio_hip_ctx* c, int B, const int* slots) {
  for (int i = 0; i < B; i++) if (slots[i] < 0 || slots[i] >= c->n_slots) return failmsg("frame batch: slot out of range");
  dmvio_hip_ctx::SlotList* hit = nullptr;
  dmvio_hip_ctx::SlotList* lru = &c->slot_lists[0];
  for (auto& L : c->slot_lists) {
    if (L.n == B && L.h && !memcmp(L.h, slots, sizeof(int) * B)) { hit = &L; break; }
    if (L.used < lru->used) lru = &L;
  }
  if (!hit) {
    hit = lru;
    // the pinned copy may still be the source of an in-flight upload, the device copy the argument of a running build: drain before rewriting
    HIPCHK(hipStreamSynchronize(buildStream(c)));
    if (c->build_stream) HIPCHK(hipStreamSynchronize(c->stream));
    if (B > hit->cap) {
      if (hit->d) { HIPCHK(hipFree(hit->d)); HIPCHK(hipHostFree(hit->h)); hit->d = nullptr; hit->h = nullptr; }
      hit->cap = std::max(B, 64);
      HIPCHK(hipMalloc((void**)&hit->d, sizeof(int) * hit->cap));
      HIPCHK(hipHostMalloc((void**)&hit->h, sizeof(int) * hit->cap, hipHostMallocDefault));
    }
    memcpy(hit->h, slots, sizeof(int) * B);
    hit->n = B;
    HIPCHK(hipMemcpyAsync(hit->d, hit->h, sizeof(int) * B, hipMemcpyHostToDevice, buildStream(c)));
  }
  hit->used = ++c->slot_clock;
  c->d_slots = hit->d;
  return 0;
}

static int framesFromDeviceBatch(dmvio_hip_ctx* c, int B, const int* slots, const float* dev_base, size_t stride_bytes, const bool attach) {
  if (!c || !slots || !dev_base) return failmsg("frames_from_device_batch: null argument");
  if (B <= 0 || stride_bytes % sizeof(float)) return failmsg("frames_from_device_batch: bad B / stride");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = stageSlots(c, B, slots)) return r;
  // attached in place: the register build (read-dominated: 5 % ahead); level 0 copied: the LDS-tile build (4 % ahead there) — both measured, tools/time_pyramids.py
  if (regBuild(c) && attach)
    hipLaunchKernelGGL((k_build_pyramids_reg<true>), regGrid(c, B), dim3(256), 0, buildStream(c), dev_base, stride_bytes / sizeof(float), c->pg, c->fs, (const int*)c->d_slots, 0, ++c->build_gen);
  else
    hipLaunchKernelGGL(k_build_pyramids, dim3(c->pg.tiles_x * c->pg.tiles_y, B), dim3(256), 0, buildStream(c), dev_base, stride_bytes / sizeof(float),
                       c->pg, c->fs, (const int*)c->d_slots, 0, ++c->build_gen, attach ? 1 : 0);
  for (int i = 0; i < B; i++) { c->h_lvl0[slots[i]] = attach ? dev_base + (size_t)i * (stride_bytes / sizeof(float)) : c->fs.own_level(slots[i], 0); c->h_tiled[slots[i]] = 0; }
  HIPCHK(hipGetLastError());
  return 0;
}

// B raw camera images already on the device (the caller's own pinned buffers / copy stream brought them there, 1 or 2 bytes per pixel) -> undistorted level 0
// + pyramids of B slots in one launch; asynchronous on the ctx stream like the fp32 variants.  Undistort::undistort + FrameHessian::makeImages per frame.
int dmvio_hip_frames_from_raw_device_batch(dmvio_hip_ctx* c, dmvio_hip_undistorter* u, int B, const int* slots, const void* raw_dev_base, size_t stride_bytes, float factor) {
  if (!c || !u || !slots || !raw_dev_base || u->ctx != c) return failmsg("frames_from_raw_device_batch: bad argument");
  const size_t nOrg = (size_t)u->U.wOrg * u->U.hOrg;
  if (B <= 0 || stride_bytes % u->bytes_per_px || stride_bytes < nOrg * u->bytes_per_px || (uintptr_t)raw_dev_base % u->bytes_per_px)
    return failmsg("frames_from_raw_device_batch: bad B / stride / alignment");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = stageSlots(c, B, slots)) return r;
  UndistortDev U = u->U;
  U.factor = factor;
  // level 0 in 8x4 tiles on request (dmvio_hip_set_raw_batch_layout)
  const bool tiled = c->raw_batch_tiled && (c->w % 8) == 0 && (c->h % 4) == 0;
  // the wave-autonomous build (a 4 x 8 pixel block per thread, levels in registers) where the pyramid allows it, the LDS-tile build otherwise
  const bool reg = regBuild(c);
  const dim3 grid = reg ? regGrid(c, B) : dim3(c->pg.tiles_x * c->pg.tiles_y, B);
#define DMV_LAUNCH_RAW(K, T, TILED, stride) hipLaunchKernelGGL((K<T, TILED>), grid, dim3(256), 0, buildStream(c), (const T*)raw_dev_base, stride, U, c->pg, c->fs, (const int*)c->d_slots, ++c->build_gen)
  if (u->bytes_per_px == 1) {
    if (reg) { if (tiled) DMV_LAUNCH_RAW(k_build_pyramids_raw_reg, unsigned char, true, stride_bytes); else DMV_LAUNCH_RAW(k_build_pyramids_raw_reg, unsigned char, false, stride_bytes); }
    else { if (tiled) DMV_LAUNCH_RAW(k_build_pyramids_raw, unsigned char, true, stride_bytes); else DMV_LAUNCH_RAW(k_build_pyramids_raw, unsigned char, false, stride_bytes); }
  } else {
    if (reg) { if (tiled) DMV_LAUNCH_RAW(k_build_pyramids_raw_reg, unsigned short, true, stride_bytes / 2); else DMV_LAUNCH_RAW(k_build_pyramids_raw_reg, unsigned short, false, stride_bytes / 2); }
    else { if (tiled) DMV_LAUNCH_RAW(k_build_pyramids_raw, unsigned short, true, stride_bytes / 2); else DMV_LAUNCH_RAW(k_build_pyramids_raw, unsigned short, false, stride_bytes / 2); }
  }
#undef DMV_LAUNCH_RAW
  for (int i = 0; i < B; i++) { c->h_lvl0[slots[i]] = c->fs.own_level(slots[i], 0); c->h_tiled[slots[i]] = tiled ? 1 : 0; }
  HIPCHK(hipGetLastError());
  return 0;
}

int dmvio_hip_frames_from_device_batch(dmvio_hip_ctx* c, int B, const int* slots, const float* dev_base, size_t stride_bytes) {
  return framesFromDeviceBatch(c, B, slots, dev_base, stride_bytes, false);
}
// Zero-copy variant: the intensity plane of level 0 IS the caller's image (this library stores no gradient channels), so only the coarser
// levels are built; the slots reference dev_base until they are rebuilt, and the caller keeps those images valid and unchanged meanwhile.
int dmvio_hip_frames_attach_device_batch(dmvio_hip_ctx* c, int B, const int* slots, const float* dev_base, size_t stride_bytes) {
  return framesFromDeviceBatch(c, B, slots, dev_base, stride_bytes, true);
}

// Diagnostics: withdraw the "every pixel finite" stamp of a slot, so that its consumers take the guarded code path (tests compare the two)
int dmvio_hip_frame_mark_unclean(dmvio_hip_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= c->n_slots) return failmsg("frame_mark_unclean: bad argument");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->fs.bad_gen + slot, c->fs.build_gen + slot, sizeof(unsigned int), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// Diagnostics: 1 when the slot's last build stamped it clean (every pixel finite, |I| <= 1e30), 0 when not, < 0 on error.  Waits for the context's stream.
int dmvio_hip_frame_is_clean(dmvio_hip_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= c->n_slots) return failmsg("frame_is_clean: bad argument");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  unsigned int g[2] = {0, 0};
  HIPCHK(c->bounce.d2h(&g[0], c->fs.build_gen + slot, sizeof(unsigned int), c->stream));
  HIPCHK(c->bounce.d2h(&g[1], c->fs.bad_gen + slot, sizeof(unsigned int), c->stream));
  HIPCHK(c->bounce.finish(c->stream));
  return g[0] != g[1] ? 1 : 0;
}

int dmvio_hip_frame_download(dmvio_hip_ctx* c, int slot, int lvl, float* out) {
  if (!c || !out) return failmsg("frame_download: null argument");
  if (slot < 0 || slot >= c->n_slots || lvl < 0 || lvl >= c->levels) return failmsg("frame_download: out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  const int n = c->wl[lvl] * c->hl[lvl];
  if (lvl == 0) { if (int r = dmv_ensure_row_major_locked(c, slot)) return r; }
  hipLaunchKernelGGL(k_level_to_f3, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->levelPtr(slot, lvl), c->wl[lvl], c->hl[lvl], c->d_f3);
  HIPCHK(hipGetLastError());
  HIPCHK(c->bounce.d2h(out, c->d_f3, sizeof(float) * 3 * n, c->stream));
  HIPCHK(c->bounce.finish(c->stream));
  return 0;
}

// FrameHessian::makeImages' absSquaredGrad planes (the input of PixelSelector::makeMaps / makeHists) of a resident frame
int dmvio_hip_frame_abs_squared_grad(dmvio_hip_ctx* c, int slot, int n_levels, const float* B_lut256, float* const* out_host) {
  if (!c || !out_host) return failmsg("frame_abs_squared_grad: null argument");
  if (slot < 0 || slot >= c->n_slots || n_levels < 1 || n_levels > c->levels) return failmsg("frame_abs_squared_grad: out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = dmv_ensure_row_major_locked(c, slot)) return r;
  // scratch layout inside d_f3 (3*w*h floats): [256-entry response table | level 0 | level 1 | ...]  (sum of the levels < 1.34 w*h)
  float* d_lut = nullptr;
  float* d_out = c->d_f3 + 256;
  if ((size_t)256 + (size_t)c->wl[0] * c->hl[0] * 4 / 3 + 16 > (size_t)3 * c->wl[0] * c->hl[0]) return failmsg("frame_abs_squared_grad: frame too small");
  if (B_lut256) {
    d_lut = c->d_f3;
    HIPCHK(c->bounce.h2d(d_lut, B_lut256, sizeof(float) * 256, c->stream));
  }
  size_t off = 0;
  for (int l = 0; l < n_levels; l++) {
    const int n = c->wl[l] * c->hl[l];
    hipLaunchKernelGGL(k_abs_squared_grad, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->levelPtr(slot, l), c->wl[l], c->hl[l], (const float*)d_lut, d_out + off);
    if (out_host[l]) HIPCHK(c->bounce.d2h(out_host[l], d_out + off, sizeof(float) * n, c->stream));
    off += n;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(c->bounce.finish(c->stream));
  return 0;
}

// ------------------------------------------------------------------ tracker
dmvio_hip_tracker* dmvio_hip_tracker_create(dmvio_hip_ctx* c) {
  if (!c) { failmsg("tracker_create: null ctx"); return nullptr; }
  HIPCHKP(hipSetDevice(c->device));
  dmvio_hip_tracker* t = new dmvio_hip_tracker();
  t->ctx = c;
  t->dev.levels = c->levels;
  t->dev.huberTH = 9.0f; t->dev.coarseCutoffTH = 20.0f; t->dev.modeA = 1e12f; t->dev.modeB = 1e8f;
  t->dev.ref_exposure = 1.0f; t->dev.ref_aff_a = 0; t->dev.ref_aff_b = 0;
  RefLevels& R = t->R;
  R.levels = c->levels;
  size_t off = 0; int tiles = 0;
  for (int l = 0; l < c->levels; l++) {
    R.w[l] = c->wl[l]; R.h[l] = c->hl[l]; R.off[l] = off; R.tile_off[l] = tiles;
    off += (size_t)R.w[l] * R.h[l];
    R.blocks_x[l] = (R.w[l] + 15) / 16;
    tiles += R.blocks_x[l] * ((R.h[l] + 15) / 16);
    t->dev.g[l].w = R.w[l]; t->dev.g[l].h = R.h[l];
  }
  R.tile_off[c->levels] = tiles; R.total = off; t->n_tiles = tiles;
  R.seg_x0 = (R.w[0] + 7) / 8;
  R.order = 0;
  {
    int so = 0;
    for (int l = 0; l < c->levels; l++) { R.seg_x[l] = (R.w[l] + 7) / 8; R.seg_off[l] = so; so += R.seg_x[l] * R.h[l]; }
    R.seg_off[c->levels] = so;
  }
  t->flow_words = ((size_t)R.w[0] * R.h[0] + 63) / 64;
  HIPCHKP(hipMalloc((void**)&t->d_seg, sizeof(int) * R.seg_off[c->levels]));
  HIPCHKP(hipMalloc((void**)&t->d_flow_mask, sizeof(unsigned long long) * t->flow_words));
  HIPCHKP(hipMemset(t->d_flow_mask, 0, sizeof(unsigned long long) * t->flow_words));
  t->dev.flow_mask = t->d_flow_mask;
  HIPCHKP(hipMalloc((void**)&t->d_idp, sizeof(float) * off));
  HIPCHKP(hipMalloc((void**)&t->d_wsp, sizeof(float) * off));
  HIPCHKP(hipMalloc((void**)&t->d_idp2, sizeof(float) * off));
  HIPCHKP(hipMalloc((void**)&t->d_wsp2, sizeof(float) * off));
  HIPCHKP(hipMalloc((void**)&t->d_dense, sizeof(float) * off));
  HIPCHKP(hipMalloc((void**)&t->d_tile_count, sizeof(int) * tiles));
  HIPCHKP(hipMalloc((void**)&t->d_tile_base, sizeof(int) * tiles));
  HIPCHKP(hipMalloc((void**)&t->d_pc_n, sizeof(int) * DMV_MAX_LEVELS));
  for (int l = 0; l < c->levels; l++) HIPCHKP(hipMalloc((void**)&t->d_pc[l], sizeof(float4) * R.w[l] * R.h[l]));
  HIPCHKP(hipMalloc((void**)&t->d_pc_ptrs, sizeof(float4*) * DMV_MAX_LEVELS));
  HIPCHKP(hipMemcpy(t->d_pc_ptrs, t->d_pc, sizeof(float4*) * DMV_MAX_LEVELS, hipMemcpyHostToDevice));
  HIPCHKP(hipMalloc((void**)&t->d_partials, sizeof(float) * ACC_PAD * t->max_eval_blocks));
  // host-coherent (fine-grained) pinned memory: the fused evaluation stores its sums and then a ticket there, the host spins on the ticket
  HIPCHKP(hipHostMalloc((void**)&t->h_tot, sizeof(float) * (ACC_PAD + 16), hipHostMallocCoherent | hipHostMallocMapped));
  memset(t->h_tot, 0, sizeof(float) * (ACC_PAD + 16));
  HIPCHKP(hipHostMalloc((void**)&t->h_mail, sizeof(unsigned int) * EVAL_MAIL_DWORDS, hipHostMallocCoherent | hipHostMallocMapped));
  memset(t->h_mail, 0, sizeof(unsigned int) * EVAL_MAIL_DWORDS);
  HIPCHKP(hipHostMalloc((void**)&t->h_rec, sizeof(float) * EVAL_RECORD_FLOATS * EVAL_SERVER_MAX_BLOCKS, hipHostMallocCoherent | hipHostMallocMapped));
  memset(t->h_rec, 0, sizeof(float) * EVAL_RECORD_FLOATS * EVAL_SERVER_MAX_BLOCKS);
  HIPCHKP(hipMalloc((void**)&t->d_arrive, sizeof(unsigned int)));
  HIPCHKP(hipMemset(t->d_arrive, 0, sizeof(unsigned int)));
  HIPCHKP(hipMalloc((void**)&t->d_leave, sizeof(unsigned int)));
  HIPCHKP(hipMemset(t->d_leave, 0xFF, sizeof(unsigned int)));
  HIPCHKP(hipStreamSynchronize(nullptr));   // the clears above run on the NULL stream; the context's stream does not wait for it
  return t;
}

void dmvio_hip_tracker_destroy(dmvio_hip_tracker* t) {
  if (!t) return;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  t->xchg = nullptr;   // the exchange owns a device buffer (RCCL transport): released now, while the context it was allocated under is still alive
  hipFree(t->d_idp); hipFree(t->d_wsp); hipFree(t->d_idp2); hipFree(t->d_wsp2); hipFree(t->d_dense);
  hipFree(t->d_tile_count); hipFree(t->d_tile_base); hipFree(t->d_pc_n); hipFree(t->d_seg); hipFree(t->d_flow_mask);
  if (t->d_pp_next) hipFree(t->d_pp_next);
  if (t->d_log) { hipFree(t->d_log); hipFree(t->d_log_n); hipFree(t->d_log_sink); }
  for (int l = 0; l < t->ctx->levels; l++) hipFree(t->d_pc[l]);
  hipFree(t->d_pc_ptrs); hipFree(t->d_pts); hipFree(t->d_partials); if (t->h_mail) hipHostFree(t->h_mail); if (t->h_rec) hipHostFree(t->h_rec);
  hipHostFree(t->h_tot); hipFree(t->d_arrive); hipFree(t->d_leave);
  hipFree(t->d_out);
  for (hipEvent_t e : t->done_event) if (e) hipEventDestroy(e);
  if (t->d_cl_part) hipFree(t->d_cl_part);
  if (t->d_cl_cnt) hipFree(t->d_cl_cnt);
  if (t->h_in) hipHostFree(t->h_in);
  if (t->h_out) hipHostFree(t->h_out);
  delete t;
}

int dmvio_hip_tracker_set_settings(dmvio_hip_tracker* t, const dmvio_hip_tracker_settings* s) {
  if (!t || !s) return failmsg("tracker_set_settings: null argument");
  t->dev.huberTH = s->huberTH; t->dev.coarseCutoffTH = s->coarseCutoffTH; t->dev.modeA = s->affineOptModeA; t->dev.modeB = s->affineOptModeB;
  return 0;
}

int dmvio_hip_tracker_make_k(dmvio_hip_tracker* t, const float k[4]) {
  if (!t || !k) return failmsg("tracker_make_k: null argument");
  const int L = t->ctx->levels;
  LevelGeom* g = t->dev.g;
  g[0].fx = k[0]; g[0].fy = k[1]; g[0].cx = k[2]; g[0].cy = k[3];
  for (int l = 1; l < L; l++) {
    g[l].fx = g[l - 1].fx * 0.5;
    g[l].fy = g[l - 1].fy * 0.5;
    g[l].cx = (g[0].cx + 0.5) / ((int)1 << l) - 0.5;
    g[l].cy = (g[0].cy + 0.5) / ((int)1 << l) - 0.5;
  }
  for (int l = 0; l < L; l++) {
    // K^-1 by cofactors / determinant in float (what Eigen's Matrix3f::inverse() evaluates for
    // K = [fx 0 cx; 0 fy cy; 0 0 1]; CoarseTracker.cpp:127-128)
    const float a = g[l].fx, e = g[l].fy, c = g[l].cx, f = g[l].cy;
    const float det = a * (e * 1.0f - f * 0.0f);
    const float invdet = 1.0f / det;
    float* Ki = g[l].Ki;
    Ki[0] = (e * 1.0f - f * 0.0f) * invdet; Ki[1] = (c * 0.0f - 0.0f * 1.0f) * invdet; Ki[2] = (0.0f * f - c * e) * invdet;
    Ki[3] = (f * 0.0f - 0.0f * 1.0f) * invdet; Ki[4] = (a * 1.0f - c * 0.0f) * invdet; Ki[5] = (c * 0.0f - a * f) * invdet;
    Ki[6] = (0.0f * 0.0f - e * 0.0f) * invdet; Ki[7] = (0.0f * 0.0f - a * 0.0f) * invdet; Ki[8] = (a * e - 0.0f * 0.0f) * invdet;
  }
  t->haveK = true;
  return 0;
}

int dmvio_hip_tracker_set_ref(dmvio_hip_tracker* t, int ref_slot, float ref_exposure, double aff_a, double aff_b,
                              int n, const float* u, const float* v, const float* idepth, const float* hdiF) {
  if (!t) return failmsg("tracker_set_ref: null tracker");
  dmvio_hip_ctx* c = t->ctx;
  if (ref_slot < 0 || ref_slot >= c->n_slots) return failmsg("tracker_set_ref: slot out of range");
  if (n < 0 || (n > 0 && (!u || !v || !idepth || !hdiF))) return failmsg("tracker_set_ref: bad point arrays");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = dmv_ensure_row_major_locked(c, ref_slot)) return r;
  hipStream_t s = c->stream;
  if (n > t->pts_cap) {
    if (t->d_pts) HIPCHK(hipFree(t->d_pts));
    t->pts_cap = std::max(n, 4096);
    HIPCHK(hipMalloc((void**)&t->d_pts, sizeof(float) * 5 * t->pts_cap));   // u, v, idepth, hdiF, [per-pixel rank bytes]
  }
  const RefLevels& R = t->R;
  HIPCHK(hipMemsetAsync(t->d_idp, 0, sizeof(float) * R.w[0] * R.h[0], s));
  HIPCHK(hipMemsetAsync(t->d_wsp, 0, sizeof(float) * R.w[0] * R.h[0], s));
  if (n > 0) {
    HIPCHK(c->bounce.h2d(t->d_pts + 0 * (size_t)t->pts_cap, u, sizeof(float) * n, s));      // through the library's pinned memory (internal.h: DmvBounce)
    HIPCHK(c->bounce.h2d(t->d_pts + 1 * (size_t)t->pts_cap, v, sizeof(float) * n, s));
    HIPCHK(c->bounce.h2d(t->d_pts + 2 * (size_t)t->pts_cap, idepth, sizeof(float) * n, s));
    HIPCHK(c->bounce.h2d(t->d_pts + 3 * (size_t)t->pts_cap, hdiF, sizeof(float) * n, s));
    // rank of every point among the points of its pixel (index order), by open addressing on the host: pixels with more than two points are scattered
    // rank by rank (k_ref_scatter)
    int maxRank = 0;
    {
      size_t cap = 64;
      while (cap < 2 * (size_t)n + 16) cap <<= 1;
      t->h_rank_keys.assign(cap, -1); t->h_rank_cnt.assign(cap, 0); t->h_rank.resize(n);
      for (int i = 0; i < n; i++) {
        const int ui = (int)(u[i] + 0.5f), vi = (int)(v[i] + 0.5f);
        if (ui < 0 || vi < 0 || ui >= R.w[0] || vi >= R.h[0]) { t->h_rank[i] = 0; continue; }
        const int key = ui + R.w[0] * vi;
        size_t hpos = ((unsigned)key * 2654435761u) & (cap - 1);
        while (t->h_rank_keys[hpos] != -1 && t->h_rank_keys[hpos] != key) hpos = (hpos + 1) & (cap - 1);
        t->h_rank_keys[hpos] = key;
        const int r = t->h_rank_cnt[hpos]++;
        t->h_rank[i] = (unsigned char)std::min(r, 255);
        maxRank = std::max(maxRank, std::min(r, 255));
      }
    }
    const unsigned char* d_rank = nullptr;
    if (maxRank >= 2) {
      HIPCHK(c->bounce.h2d(t->d_pts + 4 * (size_t)t->pts_cap, t->h_rank.data(), (size_t)n, s));
      d_rank = (const unsigned char*)(t->d_pts + 4 * (size_t)t->pts_cap);
    }
    for (int r = 1; r <= std::max(maxRank, 1); r++)
      hipLaunchKernelGGL(k_ref_scatter, dim3((n + 255) / 256), dim3(256), 0, s, n, t->d_pts, t->d_pts + t->pts_cap, t->d_pts + 2 * (size_t)t->pts_cap,
                         t->d_pts + 3 * (size_t)t->pts_cap, t->d_idp, t->d_wsp, R.w[0], R.h[0], d_rank, r == 1 ? 0 : r, r);
  }
  if (R.levels > 1) {
    const size_t npool = R.total - R.off[1];
    hipLaunchKernelGGL(k_ref_pool, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, s, R, t->d_idp, t->d_wsp);
  }
  hipLaunchKernelGGL(k_ref_dilate, dim3((unsigned)((R.total + 255) / 256)), dim3(256), 0, s, R, t->d_idp, t->d_wsp, t->d_idp2, t->d_wsp2);
  HIPCHK(hipMemsetAsync(t->d_flow_mask, 0, sizeof(unsigned long long) * t->flow_words, s));
  hipLaunchKernelGGL(k_ref_count, dim3(t->n_tiles), dim3(256), 0, s, R, t->d_idp2, t->d_wsp2, c->fs, ref_slot, t->d_tile_count, t->d_seg);
  hipLaunchKernelGGL(k_ref_scan, dim3(2 * R.levels), dim3(1024), 0, s, R, t->d_tile_count, t->d_tile_base, t->d_pc_n, t->d_seg);
  hipLaunchKernelGGL(k_ref_write, dim3(t->n_tiles), dim3(256), 0, s, R, t->d_idp2, t->d_wsp2, c->fs, ref_slot, t->d_tile_base, t->d_seg, t->d_pc_ptrs,
                     t->d_dense, t->d_flow_mask);
  HIPCHK(hipGetLastError());
  int pcn[DMV_MAX_LEVELS] = {};
  HIPCHK(c->bounce.d2h(pcn, t->d_pc_n, sizeof(int) * R.levels, s));
  HIPCHK(c->bounce.finish(s));
  for (int l = 0; l < R.levels; l++) { t->dev.pc_n[l] = pcn[l]; t->dev.pc[l] = t->d_pc[l]; }
  t->dev.ref_exposure = ref_exposure; t->dev.ref_aff_a = aff_a; t->dev.ref_aff_b = aff_b;
  t->haveRef = true;
  return 0;
}

int dmvio_hip_tracker_pc_n(dmvio_hip_tracker* t, int lvl) {
  if (!t || lvl < 0 || lvl >= t->ctx->levels) return failmsg("tracker_pc_n: bad argument");
  return t->dev.pc_n[lvl];
}

int dmvio_hip_tracker_get_pc(dmvio_hip_tracker* t, int lvl, float* u, float* v, float* idepth, float* color) {
  if (!t || lvl < 0 || lvl >= t->ctx->levels) return failmsg("tracker_get_pc: bad argument");
  dmvio_hip_ctx* c = t->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  const int n = t->dev.pc_n[lvl];
  std::vector<float4> tmp(n);
  HIPCHK(c->bounce.d2h(tmp.data(), t->d_pc[lvl], sizeof(float4) * n, c->stream));
  HIPCHK(c->bounce.finish(c->stream));
  // the device keeps the template in tile order; hand it out in the reference's row-major order (y, then x)
  std::sort(tmp.begin(), tmp.end(), [](const float4& a, const float4& b) { return a.y < b.y || (a.y == b.y && a.x < b.x); });
  for (int i = 0; i < n; i++) { u[i] = tmp[i].x; v[i] = tmp[i].y; idepth[i] = tmp[i].z; color[i] = tmp[i].w; }
  return 0;
}

// The dense maps CoarseTracker keeps next to the template: idepth[lvl] and weightSums[lvl] as makeCoarseDepthL0 leaves them (CoarseTracker.cpp:249-293; read by
// debugPlotIDepthMap / debugPlotIDepthMapFloat, :772-880, when output wrappers exist).  Debug path: the dilated planes come back from the device and the normalisation loop is
// replayed on the host; whether a pixel with weight became a template point (finite reference colour, idepth > 0) is taken from the template itself.
int dmvio_hip_tracker_get_idepth_map(dmvio_hip_tracker* t, int lvl, float* idepth_out, float* weightSums_out) {
  if (!t || lvl < 0 || lvl >= t->ctx->levels || !idepth_out) return failmsg("tracker_get_idepth_map: bad argument");
  if (!t->haveRef) return failmsg("tracker_get_idepth_map: setCoarseTrackingRef not called");
  dmvio_hip_ctx* c = t->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  const RefLevels& R = t->R;
  const int wl = R.w[lvl], hl = R.h[lvl], n = t->dev.pc_n[lvl];
  const size_t npx = (size_t)wl * hl;
  std::vector<float> ws(npx);
  std::vector<float4> pc(n);
  HIPCHK(c->bounce.d2h(idepth_out, t->d_idp2 + R.off[lvl], sizeof(float) * npx, c->stream));
  HIPCHK(c->bounce.d2h(ws.data(), t->d_wsp2 + R.off[lvl], sizeof(float) * npx, c->stream));
  if (n) HIPCHK(c->bounce.d2h(pc.data(), t->d_pc[lvl], sizeof(float4) * n, c->stream));
  HIPCHK(c->bounce.finish(c->stream));
  std::vector<unsigned char> kept(npx, 0);
  for (int i = 0; i < n; i++) kept[(size_t)pc[i].x + (size_t)pc[i].y * wl] = 1;
  for (int y = 2; y < hl - 2; y++)
    for (int x = 2; x < wl - 2; x++) {
      const size_t i = (size_t)x + (size_t)y * wl;
      if (ws[i] > 0) {
        idepth_out[i] /= ws[i];
        if (!kept[i]) { idepth_out[i] = -1; continue; }   // the reference's "just skip if something is wrong": weightSums keeps its value
      } else
        idepth_out[i] = -1;
      ws[i] = 1;
    }
  if (weightSums_out) memcpy(weightSums_out, ws.data(), sizeof(float) * npx);
  return 0;
}

// One fused calcRes + calcGSSSE evaluation, result in t->h_tot when the call returns: ONE launch, no stream synchronisation — the
// last workgroup stores the sums and then the launch's ticket into host-coherent memory and the host spins on the ticket.
// G workgroups of 256 threads split the template (point index first = rank * 256 + thread, stride G * 256) and the partial sums are
// added in rank order: with G = the cluster size of a one-problem k_track_lm launch the sums are bit-identical to the device-resident LM's.
static int evalFused(dmvio_hip_tracker* t, int lvl, int new_slot, const EvalP& e, int G) {
  dmvio_hip_ctx* c = t->ctx;
  constexpr int T = 256;
  G = std::max(1, std::min(G, t->max_eval_blocks));
  const unsigned int ticket = ++t->eval_ticket;
  hipLaunchKernelGGL(k_eval_fused<T>, dim3(G), dim3(T), 0, c->stream, t->dev, c->fs, new_slot, e, t->d_partials, t->d_arrive, t->h_tot, ticket);
  HIPCHK(hipGetLastError());
  volatile unsigned int* flag = reinterpret_cast<volatile unsigned int*>(t->h_tot) + ACC_PAD;
  unsigned long long spins = 0;
  while (*flag != ticket) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {   // every ~million polls (a few ms): has the launch failed instead of finishing?
      const hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail("k_eval_fused", __FILE__, __LINE__, q);
      if (q == hipSuccess && *flag != ticket) return failmsg("tracker evaluation finished without publishing its result");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// ---- evaluation server session: serverStart launches k_eval_server for one frame slot, serverEval posts a request and polls the sums, serverStop tells the kernel to leave.
// The kernel also leaves by itself after 5 ms without a request (a callback that takes longer — a factor-graph solve — simply finds it gone: serverEval relaunches it).
// ticket into the mailbox: the copy in the last dword first, then the first dword (the device accepts a read only when both show the same new value)
static void mailTicket(dmvio_hip_tracker* t, unsigned int v) {
  __atomic_store_n(&t->h_mail[EVAL_MAIL_DWORDS - 1], v, __ATOMIC_RELEASE);
  __atomic_store_n(&t->h_mail[0], v, __ATOMIC_RELEASE);
}
static int serverLaunch(dmvio_hip_tracker* t) {
  dmvio_hip_ctx* c = t->ctx;
  hipLaunchKernelGGL(k_eval_server<256>, dim3(t->server_G), dim3(256), 0, c->stream, t->dev, c->fs, t->server_slot, (const unsigned int*)t->h_mail, t->d_leave, t->eval_ticket,
                     t->server_idle_ticks, t->h_rec, t->server_session);
  HIPCHK(hipGetLastError());
  return 0;
}
static int serverStart(dmvio_hip_tracker* t, int new_slot, int G) {
  t->server_G = std::max(1, std::min(G, (int)EVAL_SERVER_MAX_BLOCKS));
  t->server_slot = new_slot;
  t->eval_ticket = (t->eval_ticket + 1) & ~EVAL_QUIT_BIT;   // a number of its own for the launch
  // a session number of its own too: a workgroup of the PREVIOUS launch that has not polled its quit ticket yet (serverStop does not wait) must not take this
  // session's tickets for requests — it compares the mailbox's session dword with its own and leaves
  t->server_session++;
  __atomic_store_n(&t->h_mail[EVAL_MAIL_SESSION], t->server_session, __ATOMIC_RELEASE);
  mailTicket(t, t->eval_ticket);   // "nothing new" for the kernel about to start
  if (int r = serverLaunch(t)) return r;
  t->server_on = true;
  return 0;
}
static void serverStop(dmvio_hip_tracker* t) {
  if (!t->server_on) return;
  mailTicket(t, t->eval_ticket | EVAL_QUIT_BIT);
  t->server_on = false;
}
static int serverEval(dmvio_hip_tracker* t, const EvalP& e) {
  dmvio_hip_ctx* c = t->ctx;
  static_assert(sizeof(EvalP) / 4 + 3 <= EVAL_MAIL_DWORDS, "EvalP must fit the mailbox between the ticket and the session / ticket copy");
  const int G = t->server_G;
  auto nextTicket = [&]() { unsigned int v = (t->eval_ticket + 1) & ~EVAL_QUIT_BIT; if (v == 0) v = 1; t->eval_ticket = v; return v; };
  memcpy((void*)(t->h_mail + 1), &e, sizeof(EvalP));
  unsigned int ticket = nextTicket();
  mailTicket(t, ticket);
  // wait for the record of every workgroup (ticket behind its sums), last rank first: the others are usually there by then
  auto pending = [&]() -> bool {
    for (int g = G - 1; g >= 0; g--)
      if (reinterpret_cast<volatile unsigned int*>(t->h_rec + (size_t)g * EVAL_RECORD_FLOATS)[ACC_PAD] != ticket) return true;
    return false;
  };
  unsigned long long spins = 0;
  int restarts = 0;
  const auto t_begin = std::chrono::steady_clock::now();
  while (pending()) {
    __builtin_ia32_pause();
    if ((++spins & 0x7FFF) == 0) {   // a request takes ~12 us; every ~30 thousand polls (about a hundred microseconds) ask whether the server is still there
      if (std::chrono::steady_clock::now() - t_begin > std::chrono::seconds(5)) {   // never spin forever: report, and let the kernel leave
        mailTicket(t, t->eval_ticket | EVAL_QUIT_BIT);
        t->server_on = false;
        return failmsg("evaluation server did not answer within 5 s");
      }
      const hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail("k_eval_server", __FILE__, __LINE__, q);
      if (q == hipSuccess && pending()) {
        // the kernel left (idle time-out) without serving this request: start it again (it takes the mailbox's current ticket as seen) and post the request again under
        // the next number
        if (++restarts > 8) return failmsg("evaluation server keeps leaving before it serves the request");
        if (int r = serverLaunch(t)) return r;             // first_seen = eval_ticket = the unanswered ticket: nothing new for the kernel about to start ...
        ticket = nextTicket();                             // ... and the request itself under a fresh number
        mailTicket(t, ticket);
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  // the sums of the evaluation: the partial sums added in rank order — ((0 + p0) + p1) + ... per slot, exactly what the device-side reductions compute
  for (int k = 0; k < ACC_PAD; k++) {
    float sacc = 0.0f;
    for (int g = 0; g < G; g++) sacc += t->h_rec[(size_t)g * EVAL_RECORD_FLOATS + k];
    t->h_tot[k] = sacc;
  }
  return 0;
}

int dmvio_hip_tracker_eval(dmvio_hip_tracker* t, int lvl, int new_slot, float new_exposure, const double pose7[7], const double aff[2],
                           float cutoffTH, double res6[6], double H[64], double b[8]) {
  if (!t || !pose7 || !aff) return failmsg("tracker_eval: null argument");
  dmvio_hip_ctx* c = t->ctx;
  if (!t->haveK || !t->haveRef) return failmsg("tracker_eval: makeK / setCoarseTrackingRef not called");
  if (lvl < 0 || lvl >= c->levels || new_slot < 0 || new_slot >= c->n_slots) return failmsg("tracker_eval: out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = dmv_ensure_row_major_locked(c, new_slot)) return r;
  EvalP e;
  makeEvalP(t->dev, lvl, poseFrom7(pose7), aff[0], aff[1], new_exposure, cutoffTH, e);
  const int n = t->dev.pc_n[lvl];
  if (int r = evalFused(t, lvl, new_slot, e, (n + 255) / 256)) return r;
  if (res6) res6FromSums(t->h_tot, res6);
  if (H && b) systemFromSums(t->h_tot, H, b);
  return 0;
}

static int ensureBatch(dmvio_hip_tracker* t, int B) {
  if (B <= t->batch_cap) return 0;
  if (t->fetch_pending_B > 0) return failmsg("track_batch_stage: a larger batch cannot be staged while the results of the previous one are still to be fetched");
  if (t->h_in) { HIPCHK(hipStreamSynchronize(t->ctx->stream)); HIPCHK(hipFree(t->d_out)); HIPCHK(hipHostFree(t->h_in)); HIPCHK(hipHostFree(t->h_out)); }
  t->batch_cap = std::max(B, 64);
  HIPCHK(hipMalloc((void**)&t->d_out, sizeof(LMProblemOut)));   // the discard entry of cluster mode (non-leading workgroups)
  HIPCHK(hipHostMalloc((void**)&t->h_in, sizeof(LMProblemIn) * 2 * t->batch_cap, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&t->h_out, sizeof(LMProblemOut) * 2 * t->batch_cap, hipHostMallocDefault));
  return 0;
}

int dmvio_hip_tracker_track_batch_stage(dmvio_hip_tracker* t, int B, const int* new_slots, const float* new_exposures,
                                        const double* pose7_in, const double* aff_in, int coarsestLvl, const double* minRes) {
  if (!t || !new_slots || !pose7_in || !aff_in) return failmsg("track_batch_stage: null argument");
  dmvio_hip_ctx* c = t->ctx;
  if (!t->haveK || !t->haveRef) return failmsg("track: makeK / setCoarseTrackingRef not called");
  if (B <= 0) return failmsg("track: B must be positive");
  if (coarsestLvl < 0 || coarsestLvl >= c->levels || coarsestLvl >= 5) return failmsg("track: coarsestLvl out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = ensureBatch(t, B)) return r;
  // problems and results live in two halves of pinned host memory used alternately: the launch before the last one (same half) must have
  // finished before its inputs are overwritten; the last launch may still be running
  const int half = t->out_cur ^ 1;
  // ... and its results must have been picked up: with a fetch_begin outstanding on that half, a second batch staged behind it would have the kernel
  // overwrite results nobody has read yet
  if (t->fetch_pending_B > 0 && half == t->out_fetch)
    return failmsg("track_batch_stage: the results marked by track_batch_fetch_begin have not been fetched yet (one batch may be staged behind them, not two)");
  if (t->done_event[half]) HIPCHK(hipEventSynchronize(t->done_event[half]));
  for (int i = 0; i < B; i++) {
    if (new_slots[i] < 0 || new_slots[i] >= c->n_slots) return failmsg("track: frame slot out of range");
    LMProblemIn& p = t->h_in[(size_t)half * t->batch_cap + i];
    memcpy(p.pose7, pose7_in + 7 * i, sizeof(double) * 7);
    p.aff[0] = aff_in[2 * i]; p.aff[1] = aff_in[2 * i + 1];
    for (int k = 0; k < 5; k++) p.minRes[k] = minRes ? minRes[5 * i + k] : NAN;
    p.new_slot = new_slots[i];
    p.new_exposure = new_exposures ? new_exposures[i] : 1.0f;
  }
  t->staged_B = B; t->staged_coarsest = coarsestLvl; t->staged_half = half; t->staged_unlaunched = true;
  return 0;
}

// Workgroups per alignment problem.  The evaluation time falls with 1/C while every evaluation pays one device-scope barrier whose cost
// grows with the number of workgroups in flight, so the best C grows with the template (about sqrt(points)) and shrinks with the batch;
// the table is the argmin of a measured sweep (tools/sweep_tracking.py --cluster, profiles/r01_tracking_sweep.md) over
// template size (level-0 points) x batch size.
static int clusterSize(const int B, const int pc0) {
  if (B > 128) return 1;
  const int sz = pc0 < 20000 ? 0 : (pc0 < 70000 ? 1 : (pc0 < 180000 ? 2 : 3));
  const int bb = B <= 4 ? 0 : (B <= 8 ? 1 : (B <= 16 ? 2 : (B <= 32 ? 3 : (B <= 64 ? 4 : 5))));
  static const int tab[4][6] = {{8, 8, 4, 4, 4, 1}, {8, 8, 8, 4, 4, 4}, {32, 16, 8, 8, 4, 4}, {32, 16, 16, 8, 8, 4}};   // re-measured in round 2 (profiles/r02_tracking_cluster_sweep.md)
  return tab[sz][bb];
}

int dmvio_hip_tracker_track_batch_launch(dmvio_hip_tracker* t) {
  if (!t || t->staged_B <= 0) return failmsg("track_batch_launch: nothing staged");
  dmvio_hip_ctx* c = t->ctx;
  HIPCHK(hipSetDevice(c->device));
  const int B = t->staged_B;
  t->staged_unlaunched = false;
  // Up to 128 problems -> cluster mode: C workgroups of 256 threads per problem (latency; measured on MI355X: B=1 240 us with C=8 vs
  // 367 us for one 1024-thread workgroup); more -> one workgroup per problem (512 threads up to 512 problems, then 256 threads,
  // four resident per CU: throughput).
  // Overridable for experiments: dmvio_hip_tracker_set_launch_shape (lm_threads / lm_waves / lm_cluster).
  int C = 1;
  if (t->lm_cluster_override > 0) C = t->lm_cluster_override;
  else if (!t->lm_threads_override) C = clusterSize(B, t->dev.pc_n[0]);
  // cluster mode synchronises the workgroups of a problem with a device-scope barrier: all of them must be resident at once (256 CUs x 4); one
  // workgroup per problem has no such limit — batches beyond the resident slots simply queue
  if (C > 1 && (long)B * C > 1024) return failmsg("track_batch_launch: cluster size too large for the batch (B*C must be <= 1024 resident workgroups)");
  // threads per workgroup: 256 in cluster mode and for full batches (four workgroups per CU); batches that cannot fill the CUs that way
  // (129..512 problems) take 512 threads per problem (measured: B=256 0.47 -> 0.40 ms, B=512 0.62 -> 0.58 ms)
  const int T = t->lm_threads_override ? t->lm_threads_override : (C > 1 ? 256 : (B <= 128 ? 1024 : (B <= 512 ? 512 : 256)));
  const int W = t->lm_waves_override ? t->lm_waves_override : 4;
  ClusterArgs cl; cl.C = C; cl.part = nullptr; cl.cnt = nullptr; cl.discard = t->d_out; cl.log = nullptr; cl.log_n = nullptr;
  if (t->debug_mode) {
    // diagnostics: 1 = record the evaluations of this launch, 2 = run the recorded evaluations again without the control steps (dmvio_hip_tracker_debug_record_replay)
    if (C != 1 || T != 256) return failmsg("tracker record / replay: full batches only (one 256-thread workgroup per problem)");
    if (B > t->log_cap) {
      if (t->d_log) { HIPCHK(hipFree(t->d_log)); HIPCHK(hipFree(t->d_log_n)); HIPCHK(hipFree(t->d_log_sink)); }
      HIPCHK(hipMalloc((void**)&t->d_log, sizeof(EvalP) * (size_t)B * LM_LOG_EVALS)); HIPCHK(hipMalloc((void**)&t->d_log_n, sizeof(int) * B));
      HIPCHK(hipMalloc((void**)&t->d_log_sink, sizeof(float) * (size_t)B * ACC_PAD));
      t->log_cap = B; t->log_B = 0;
    }
    if (t->debug_mode == 1) { cl.log = t->d_log; cl.log_n = t->d_log_n; t->log_B = B; }
    else {
      if (t->log_B != B) return failmsg("tracker replay: record a launch of the same batch first");
      hipLaunchKernelGGL((k_track_replay<256, 4>), dim3(B), dim3(256), 0, c->stream, t->dev, c->fs, t->h_in + (size_t)t->staged_half * t->batch_cap, (const EvalP*)t->d_log,
                         (const int*)t->d_log_n, t->d_log_sink);
      HIPCHK(hipGetLastError());
      return 0;   // nothing to fetch: the caller times the launch and synchronises the stream itself
    }
  }
  t->last_cluster = C; t->last_threads = T;
  t->out_cur = t->staged_half;   // the host may still be unpacking the previous launch's half (fetch_begin pipeline)
  if (C > 1) {
    const size_t need = (size_t)B * 2 * C * ACC_PAD;
    if (need > t->cl_part_cap) { if (t->d_cl_part) HIPCHK(hipFree(t->d_cl_part)); HIPCHK(hipMalloc((void**)&t->d_cl_part, sizeof(float) * need)); t->cl_part_cap = need; }
    if (B > t->cl_cnt_cap) { if (t->d_cl_cnt) HIPCHK(hipFree(t->d_cl_cnt)); HIPCHK(hipMalloc((void**)&t->d_cl_cnt, sizeof(unsigned int) * B)); t->cl_cnt_cap = B; }
    HIPCHK(hipMemsetAsync(t->d_cl_cnt, 0, sizeof(unsigned int) * B, c->stream));
    cl.part = t->d_cl_part; cl.cnt = t->d_cl_cnt;
  }
  // frames whose level 0 the batched raw-image build stored in 8x4 tiles: the (256, 4) and (512, 4) configurations have an instantiation that gathers from them directly
  // (decided per problem inside the kernel); any other configuration has those slots converted back first
  bool any_tiled = false;
  // the context's lock from the scan of the layout flags to the launch: a batched raw-image build on another thread (dmvio_hip_set_build_stream invites one) must not flip a
  // slot's layout between the choice of the instantiation and the launch that relies on it
  std::lock_guard<std::mutex> lk_layout(c->mu);
  {
    const LMProblemIn* pin = t->h_in + (size_t)t->out_cur * t->batch_cap;
    for (int i = 0; i < B && !any_tiled; i++) any_tiled = c->h_tiled[pin[i].new_slot] != 0;
    const bool has_variant = (T == 256 && W < 6) || (T == 512 && W < 6 && C == 1);
    if (any_tiled && !has_variant) {
      for (int i = 0; i < B; i++) if (int r = dmv_ensure_row_major_locked(c, pin[i].new_slot)) return r;
      any_tiled = false;
    }
  }
  if (t->batch_kernel == 1 && C == 1 && T == 256 && !any_tiled && !t->debug_mode && B >= 512) {
    // full batches: five-wave workgroups holding two problems each, the LM control step of one beside the evaluation of the other; persistent grid, problems dealt out by a
    // device-wide counter
    if (!t->d_pp_next) HIPCHK(hipMalloc((void**)&t->d_pp_next, sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(t->d_pp_next, 0, sizeof(unsigned int), c->stream));
    const int grid = std::min((B + 1) / 2, 1024);   // 256 CUs x 4 resident workgroups at 128 registers
    hipLaunchKernelGGL(k_track_lm_pp, dim3(grid), dim3(256), 0, c->stream, t->dev, c->fs, t->h_in + (size_t)t->out_cur * t->batch_cap, t->h_out + (size_t)t->out_cur * t->batch_cap,
                       t->staged_coarsest, B, t->d_pp_next);
    HIPCHK(hipGetLastError());
    t->last_threads = 256;
    if (!t->done_event[t->out_cur]) HIPCHK(hipEventCreateWithFlags(&t->done_event[t->out_cur], hipEventDisableTiming));
    HIPCHK(hipEventRecord(t->done_event[t->out_cur], c->stream));
    return 0;
  }
#define DMV_LAUNCH_LM(TT, WW) hipLaunchKernelGGL((k_track_lm<TT, WW>), dim3(B * C), dim3(TT), 0, c->stream, t->dev, c->fs, t->h_in + (size_t)t->out_cur * t->batch_cap, t->h_out + (size_t)t->out_cur * t->batch_cap, t->staged_coarsest, cl)
#define DMV_LAUNCH_LM_TILED(TT, WW) hipLaunchKernelGGL((k_track_lm<TT, WW, true>), dim3(B * C), dim3(TT), 0, c->stream, t->dev, c->fs, t->h_in + (size_t)t->out_cur * t->batch_cap, t->h_out + (size_t)t->out_cur * t->batch_cap, t->staged_coarsest, cl)
  if (any_tiled && T == 256) DMV_LAUNCH_LM_TILED(256, 4);
  else if (any_tiled && T == 512) DMV_LAUNCH_LM_TILED(512, 4);
  else if (T == 1024 && C == 1) DMV_LAUNCH_LM(1024, 4);
  else if (T == 512 && W >= 6 && C == 1) DMV_LAUNCH_LM(512, 6);
  else if (T == 512 && C == 1) DMV_LAUNCH_LM(512, 4);
  else if (T == 256 && W >= 6) DMV_LAUNCH_LM(256, 6);
  else if (T == 256) DMV_LAUNCH_LM(256, 4);
  else if (T == 128 && C == 1) DMV_LAUNCH_LM(128, 4);
  else return failmsg("track_batch_launch: lm_threads (dmvio_hip_tracker_set_launch_shape) must be 128/256/512/1024 (cluster mode: 256)");
#undef DMV_LAUNCH_LM
#undef DMV_LAUNCH_LM_TILED
  HIPCHK(hipGetLastError());
  if (!t->done_event[t->out_cur]) HIPCHK(hipEventCreateWithFlags(&t->done_event[t->out_cur], hipEventDisableTiming));
  HIPCHK(hipEventRecord(t->done_event[t->out_cur], c->stream));
  return 0;
}

// Marks the results of the last launch as "to be fetched later": the caller may queue the NEXT batch (pyramids, stage, launch) before
// blocking in _fetch, so the device never waits for the host to unpack results.
int dmvio_hip_tracker_track_batch_fetch_begin(dmvio_hip_tracker* t) {
  if (!t || t->staged_B <= 0) return failmsg("track_batch_fetch_begin: nothing staged");
  dmvio_hip_ctx* c = t->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  // the kernel writes its results into pinned host memory itself and an event follows every launch: nothing to enqueue, the next
  // launch goes into the other half and _fetch waits for this launch's event only
  t->fetch_pending_B = t->staged_B; t->out_fetch = t->out_cur;
  return 0;
}

int dmvio_hip_tracker_track_batch_fetch(dmvio_hip_tracker* t, double* pose7_out, double* aff_out, double* lastResiduals, double* lastFlow,
                                        double* H, double* b, int* good, int* iterations) {
  if (!t || t->staged_B <= 0) return failmsg("track_batch_fetch: nothing staged");
  dmvio_hip_ctx* c = t->ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  int B = t->staged_B, half = t->out_cur;
  if (t->fetch_pending_B > 0) {   // results marked by _fetch_begin: they belong to the launch before the last one
    B = t->fetch_pending_B; t->fetch_pending_B = 0; half = t->out_fetch;
  }
  if (!t->done_event[half]) return failmsg("track_batch_fetch: nothing launched");
  HIPCHK(hipEventSynchronize(t->done_event[half]));
  long long ev = 0, pe = 0;
  t->last_ticks_step = t->last_ticks_eval = 0;
  for (int i = 0; i < B; i++) {
    const LMProblemOut& o = t->h_out[(size_t)half * t->batch_cap + i];
    if (pose7_out) memcpy(pose7_out + 7 * i, o.pose7, sizeof(double) * 7);
    if (aff_out) { aff_out[2 * i] = o.aff[0]; aff_out[2 * i + 1] = o.aff[1]; }
    if (lastResiduals) memcpy(lastResiduals + 5 * i, o.lastRes, sizeof(double) * 5);
    if (lastFlow) memcpy(lastFlow + 3 * i, o.flow, sizeof(double) * 3);
    if (H) memcpy(H + 64 * i, o.H, sizeof(double) * 64);
    if (b) memcpy(b + 8 * i, o.b, sizeof(double) * 8);
    if (good) good[i] = o.good;
    if (iterations) iterations[i] = o.iterations;
    if ((int)t->last_repeat_lvl.size() < B) { t->last_repeat_lvl.resize(B); t->last_first_pass_res.resize(B); }
    t->last_repeat_lvl[i] = o.repeated_lvl; t->last_first_pass_res[i] = o.first_pass_res;
    ev += o.n_evals; pe += o.n_point_evals; t->last_ticks_step += o.ticks_step; t->last_ticks_eval += o.ticks_eval;
  }
  t->last_evals = ev; t->last_point_evals = pe;
  return 0;
}

int dmvio_hip_tracker_track_batch(dmvio_hip_tracker* t, int B, const int* new_slots, const float* new_exposures, double* pose7_io, double* aff_io,
                                  int coarsestLvl, const double* minRes, double* lastResiduals, double* lastFlow, double* H, double* b,
                                  int* good, int* iterations) {
  // ONE alignment problem: the LM control step (8x8 pivoted LDL^T, SE3 exp: one dependent chain) takes 6.5 us per iteration on a wavefront and well under a microsecond on
  // the host, so the loop runs on the host against the evaluation server (one launch per frame, requests through host-coherent memory).  Same split of the template and
  // same order of the partial sums as the device-resident LM's cluster mode, same arithmetic in the step: identical sums, residuals, H, b and iteration counts, the pose
  // to the last bit or two of its fp64 components (tests/test_vio_gpu.py, tests/test_edge_gpu.py); 0.165 instead of 0.25 ms per frame.
  // (not while a staged batch waits for its launch: the single-frame path would run in front of it and the caller's later _launch would find nothing)
  if (B == 1 && t && t->single_host_lm && t->use_server && !t->lm_threads_override && !t->lm_cluster_override && t->fetch_pending_B == 0 && !t->staged_unlaunched && new_slots && pose7_io &&
      aff_io) {
    int g = 0, ne = 0;
    const float ex = new_exposures ? new_exposures[0] : 1.0f;
    if (int r = dmvio_hip_tracker_track_vio(t, new_slots[0], ex, pose7_io, aff_io, coarsestLvl, minRes, nullptr, lastResiduals, lastFlow, H, b, &g, &ne)) return r;
    if (good) good[0] = g;
    if (iterations) iterations[0] = t->last_vio_iterations;
    return 0;
  }
  if (int r = dmvio_hip_tracker_track_batch_stage(t, B, new_slots, new_exposures, pose7_io, aff_io, coarsestLvl, minRes)) return r;
  if (int r = dmvio_hip_tracker_track_batch_launch(t)) return r;
  return dmvio_hip_tracker_track_batch_fetch(t, pose7_io, aff_io, lastResiduals, lastFlow, H, b, good, iterations);
}

int dmvio_hip_tracker_track(dmvio_hip_tracker* t, int new_slot, float new_exposure, double pose7_io[7], double aff_io[2], int coarsestLvl,
                            const double minRes[5], double lastResiduals[5], double lastFlow[3], double H[64], double b[8], int* good) {
  return dmvio_hip_tracker_track_batch(t, 1, &new_slot, &new_exposure, pose7_io, aff_io, coarsestLvl, minRes, lastResiduals, lastFlow, H, b, good, nullptr);
}

// The visual-only LM step of CoarseTracker.cpp:639-682 on the host: damped H, the 6 / 7 / 8-dof LDL^T variants selected by
// affineOptModeA / B, extrapolation, SCALE_* scaling, SE3::exp(inc) * refToNew_current.  This is what the reference itself executes
// while the IMU is not initialised (CoarseTracker.cpp:612: !imuIntegration.isCoarseInitialized()), and the default update of
// dmvio_hip_tracker_track_vio.  incA / incB are returned UNSCALED like computeCoarseUpdate's (the caller applies SCALE_A / SCALE_B, :633-637).
int dmvio_hip_coarse_update_visual(const dmvio_hip_tracker_settings* st, const double H[64], const double b[8], float extrapFac, float lambda,
                                   const double pose7_cur[7], double pose7_new[7], double* incA, double* incB, double* incNorm) {
  if (!H || !b || !pose7_cur || !pose7_new) return failmsg("coarse_update_visual: null argument");
  const float modeA = st ? st->affineOptModeA : 1e12f, modeB = st ? st->affineOptModeB : 1e8f;
  double Hl[64];
  memcpy(Hl, H, sizeof(Hl));
  for (int i = 0; i < 8; i++) Hl[i * 9] *= (1 + lambda);
  double inc[8];
  const bool fixA = modeA < 0, fixB = modeB < 0;
  if (!fixA && !fixB) {
    double m[64]; memcpy(m, Hl, sizeof(m));
    for (int i = 0; i < 8; i++) inc[i] = -b[i];
    ldltSolveInPlace<8>(m, 8, inc, 8);
  } else if (fixA && fixB) {
    double m[64]; memcpy(m, Hl, sizeof(m));
    for (int i = 0; i < 6; i++) inc[i] = -b[i];
    ldltSolveInPlace<8>(m, 8, inc, 6);
    inc[6] = inc[7] = 0;
  } else if (!fixA && fixB) {
    double m[64]; memcpy(m, Hl, sizeof(m));
    for (int i = 0; i < 7; i++) inc[i] = -b[i];
    ldltSolveInPlace<8>(m, 8, inc, 7);
    inc[7] = 0;
  } else {   // fix a: b's row / column take the place of a's (CoarseTracker.cpp:653-664)
    double m[64]; memcpy(m, Hl, sizeof(m));
    double bs[8]; memcpy(bs, b, sizeof(bs));
    for (int r = 0; r < 8; r++) m[r * 8 + 6] = m[r * 8 + 7];
    for (int cc = 0; cc < 8; cc++) m[6 * 8 + cc] = m[7 * 8 + cc];
    bs[6] = bs[7];
    double x[8];
    for (int i = 0; i < 7; i++) x[i] = -bs[i];
    ldltSolveInPlace<8>(m, 8, x, 7);
    for (int i = 0; i < 6; i++) inc[i] = x[i];
    inc[6] = 0; inc[7] = x[6];
  }
  for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
  double incScaled[8];
  for (int i = 0; i < 6; i++) incScaled[i] = inc[i] * 1.0f;   // SCALE_XI_ROT / SCALE_XI_TRANS
  incScaled[6] = inc[6] * 10.0f; incScaled[7] = inc[7] * 1000.0f;
  double ssum = 0;
  for (int i = 0; i < 8; i++) ssum += incScaled[i];
  if (!std::isfinite(ssum)) for (int i = 0; i < 8; i++) incScaled[i] = 0;
  const Pose nxt = poseMul(poseExp(incScaled), poseFrom7(pose7_cur));
  poseTo7(nxt, pose7_new);
  // the reference's two branches differ here: the LDLT branch adds incScaled[6..7] to the affine parameters, the IMU branch receives
  // the unscaled increments and scales them itself — both give the same numbers as long as incScaled was not zeroed for being non-finite
  if (incA) *incA = std::isfinite(ssum) ? inc[6] : 0.0;
  if (incB) *incB = std::isfinite(ssum) ? inc[7] : 0.0;
  double nn = 0;
  for (int i = 0; i < 8; i++) nn += inc[i] * inc[i];
  if (incNorm) *incNorm = sqrt(nn);
  return 0;
}

// CoarseTracker::trackNewestCoarse with the LM step handed to the caller (CoarseTracker.cpp:539-770, the setting_useIMU branch
// :612-637): the host loop of the reference over fused device evaluations.  Every iteration costs one kernel launch whose result the
// host picks up by polling host-coherent memory (evalFused), plus the callback.
int dmvio_hip_tracker_set_single_frame_mode(dmvio_hip_tracker* t, int host_lm) {
  if (!t) return failmsg("null tracker");
  t->single_host_lm = host_lm ? 1 : 0;
  return 0;
}
// Measurement knobs (tools/sweep_tracking.py, profiles/): explicit calls instead of environment variables read behind the caller's back.  0 = the library's own choice.
//   eval_blocks: workgroups per fused evaluation / evaluation server (changes how the fp32 partial sums are grouped)
//   lm_threads / lm_waves: workgroup shape of k_track_lm (256 or 512 threads; 1, 2 or 4 tap sets in flight)
//   lm_cluster: workgroups that share one alignment problem in cluster mode (2 .. 32; changes the grouping of the sums)
int dmvio_hip_tracker_set_launch_shape(dmvio_hip_tracker* t, int eval_blocks, int lm_threads, int lm_waves, int lm_cluster) {
  if (!t) return failmsg("null tracker");
  if (eval_blocks < 0 || eval_blocks > t->max_eval_blocks) return failmsg("tracker_set_launch_shape: eval_blocks out of range");
  if (lm_threads != 0 && lm_threads != 256 && lm_threads != 512) return failmsg("tracker_set_launch_shape: lm_threads is 0, 256 or 512");
  if (lm_waves != 0 && lm_waves != 1 && lm_waves != 2 && lm_waves != 4) return failmsg("tracker_set_launch_shape: lm_waves is 0, 1, 2 or 4");
  if (lm_cluster < 0 || lm_cluster > 32) return failmsg("tracker_set_launch_shape: lm_cluster out of range");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  t->eval_blocks_override = eval_blocks; t->lm_threads_override = lm_threads; t->lm_waves_override = lm_waves; t->lm_cluster_override = lm_cluster;
  return 0;
}
// 1 (default): the evaluations of a host-driven LM go to the resident evaluation server (one launch per tracked frame); 0: one fused launch per evaluation
// Diagnostics for profiles/: mode 1 = the next dmvio_hip_tracker_track_batch_launch calls record the parameters of every evaluation they run (full batches: one 256-thread
// workgroup per problem), mode 2 = they run the recorded evaluations again WITHOUT the LM control steps between them (k_track_replay: same points, same taps, same fused
// sums; nothing to fetch — time the launch on the context's stream), 0 = normal operation.
int dmvio_hip_tracker_debug_record_replay(dmvio_hip_tracker* t, int mode) {
  if (!t || mode < 0 || mode > 2) return failmsg("tracker_debug_record_replay: bad argument");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  t->debug_mode = mode;
  return 0;
}
// Order in which setCoarseTrackingRef stores the template points of every level: 0 (default) = 8x8-pixel tiles, Z-ordered inside 16x16 blocks; 1 = the reference's
// row-major order (CoarseTracker.cpp:249-293).  Takes effect with the next dmvio_hip_tracker_set_ref.  The sums of an evaluation are formed per 64-point group and then in
// group order, so the two orders group the fp32 additions differently (results agree to rounding, like cluster sizes do); profiles/r05_tracker_floor.md has the measurement.
int dmvio_hip_tracker_set_template_order(dmvio_hip_tracker* t, int row_major) {
  if (!t) return failmsg("null tracker");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  t->R.order = row_major ? 1 : 0;
  return 0;
}
// Kernel of FULL batches (>= 512 problems, one 256-thread evaluation group per problem): 0 = k_track_lm (four wavefronts per problem: they evaluate, then three wait
// while the first solves), 1 = k_track_lm_pp (a four-wavefront workgroup holds two problems: wavefront 0 runs the control step of one, then joins the evaluation of the other; per problem the same
// arithmetic in the same order — identical results).
int dmvio_hip_tracker_set_batch_kernel(dmvio_hip_tracker* t, int mode) {
  if (!t || mode < 0 || mode > 1) return failmsg("tracker_set_batch_kernel: 0 or 1");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  t->batch_kernel = mode;
  return 0;
}
int dmvio_hip_tracker_set_eval_server(dmvio_hip_tracker* t, int on) {
  if (!t) return failmsg("null tracker");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  t->use_server = on ? 1 : 0;
  return 0;
}

// How long the evaluation server of a single-frame track / track_vio call stays resident without a request (default 5000 us).  A computeCoarseUpdate hook that regularly
// takes longer (a factor-graph solve) would otherwise pay a relaunch of the kernel per LM iteration.
int dmvio_hip_tracker_set_server_idle_us(dmvio_hip_tracker* t, int microseconds) {
  if (!t || microseconds < 100 || microseconds > 2000000) return failmsg("tracker_set_server_idle_us: 100 us .. 2 s");
  t->server_idle_ticks = (long long)microseconds * 100;   // wall_clock64 runs at 100 MHz
  return 0;
}

int dmvio_hip_tracker_track_vio(dmvio_hip_tracker* t, int new_slot, float new_exposure, double pose7_io[7], double aff_io[2], int coarsestLvl,
                                const double minResForAbort[5], const dmvio_hip_coarse_callbacks* cb, double lastResiduals[5], double lastFlow[3],
                                double H_out[64], double b_out[8], int* good, int* n_evals) {
  if (!t || !pose7_io || !aff_io) return failmsg("track_vio: null argument");
  dmvio_hip_ctx* c = t->ctx;
  if (!t->haveK || !t->haveRef) return failmsg("track_vio: makeK / setCoarseTrackingRef not called");
  if (coarsestLvl < 0 || coarsestLvl >= c->levels || coarsestLvl >= 5) return failmsg("track_vio: coarsestLvl out of range");
  if (new_slot < 0 || new_slot >= c->n_slots) return failmsg("track_vio: frame slot out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  if (int r = dmv_ensure_row_major_locked(c, new_slot)) return r;   // the evaluation server reads row-major planes
  const TrackerDev& trk = t->dev;
  const int G = t->eval_blocks_override > 0 ? t->eval_blocks_override : clusterSize(1, trk.pc_n[0]);
  dmvio_hip_tracker_settings st; st.huberTH = trk.huberTH; st.coarseCutoffTH = trk.coarseCutoffTH; st.affineOptModeA = trk.modeA; st.affineOptModeB = trk.modeB;
  double lastRes[5], flow[3] = {1000, 1000, 1000};
  for (int i = 0; i < 5; i++) lastRes[i] = NAN;
  const int maxIterations[5] = {10, 20, 50, 50, 50};
  const float lambdaExtrapolationLimit = 0.001f;
  Pose cur = poseFrom7(pose7_io);
  double affA = aff_io[0], affB = aff_io[1];
  bool haveRepeated = false, failed = false;
  double H[64] = {0}, b[8] = {0};
  int lastLvl = -1, evals = 0, totalIts = 0, repeatedLvl = -1;
  double firstPassRes = NAN;
  EvalP e;
  // one server launch for the whole call (requests through the mailbox) or one fused launch per evaluation
  struct Session {
    dmvio_hip_tracker* t; bool on;
    ~Session() { if (on) serverStop(t); }
  } session{t, false};
  if (t->use_server) { if (int r = serverStart(t, new_slot, G)) return r; session.on = true; }
  long long point_evals = 0;   // sum over evaluations of pc_n[lvl]: the unit count behind the roofline's algorithmic bytes (dmvio_hip_tracker_last_work)
  auto evaluate = [&](const EvalP& ep, int lv) -> int { point_evals += trk.pc_n[lv]; return session.on ? serverEval(t, ep) : evalFused(t, lv, new_slot, ep, G); };
  for (int lvl = coarsestLvl; lvl >= 0 && !failed; lvl--) {
    float levelCutoffRepeat = 1;
    double resOld[6];
    makeEvalP(trk, lvl, cur, affA, affB, new_exposure, trk.coarseCutoffTH * levelCutoffRepeat, e);
    if (int r = evaluate(e, lvl)) return r;
    evals++;
    res6FromSums(t->h_tot, resOld);
    while (resOld[5] > 0.6 && (levelCutoffRepeat < 50 || resOld[5] > 0.99)) {
      levelCutoffRepeat *= 2;
      makeEvalP(trk, lvl, cur, affA, affB, new_exposure, trk.coarseCutoffTH * levelCutoffRepeat, e);
      if (int r = evaluate(e, lvl)) return r;
      evals++;
      res6FromSums(t->h_tot, resOld);
    }
    systemFromSums(t->h_tot, H, b);
    float lambda = 0.01f;
    for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
      double cur7[7], nxt7[7], incA = 0, incB = 0, incNorm = 0;
      poseTo7(cur, cur7);
      int rc;
      if (cb && cb->update) {
        const double affc[2] = {affA, affB};
        rc = cb->update(cb->user, H, b, extrapFac, lambda, cur7, affc, nxt7, &incA, &incB, &incNorm);
        if (rc) return failmsg("track_vio: the coarse-update callback reported an error");
      } else if ((rc = dmvio_hip_coarse_update_visual(&st, H, b, extrapFac, lambda, cur7, nxt7, &incA, &incB, &incNorm))) return rc;
      const Pose nxt = poseFrom7(nxt7);
      const double affA_n = affA + incA * 10.0f, affB_n = affB + incB * 1000.0f;   // SCALE_A, SCALE_B (CoarseTracker.cpp:633-637)
      makeEvalP(trk, lvl, nxt, affA_n, affB_n, new_exposure, trk.coarseCutoffTH * levelCutoffRepeat, e);
      if (int r = evaluate(e, lvl)) return r;
      evals++;
      double resNew[6];
      res6FromSums(t->h_tot, resNew);
      const bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        systemFromSums(t->h_tot, H, b);
        for (int i = 0; i < 6; i++) resOld[i] = resNew[i];
        affA = affA_n; affB = affB_n; cur = nxt;
        if (cb && cb->accept) cb->accept(cb->user);
        lambda *= 0.5f;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      lastLvl = lvl;
      totalIts++;
      if (!(incNorm > 1e-3)) break;
    }
    lastRes[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    flow[0] = resOld[2]; flow[1] = resOld[3]; flow[2] = resOld[4];
    if (std::isnan(lastRes[lvl]) || (minResForAbort && lastRes[lvl] > 1.5 * minResForAbort[lvl])) { failed = true; break; }
    if (levelCutoffRepeat > 1 && !haveRepeated) { repeatedLvl = lvl; firstPassRes = lastRes[lvl]; lvl++; haveRepeated = true; }
  }
  t->last_vio_iterations = totalIts;
  // bookkeeping of the "last launch" queries: this call's own numbers, not those of an earlier batch
  t->last_evals = evals; t->last_point_evals = point_evals; t->last_ticks_step = t->last_ticks_eval = 0;
  t->last_cluster = session.on ? t->server_G : G; t->last_threads = 256;
  if (t->last_repeat_lvl.empty()) { t->last_repeat_lvl.resize(1); t->last_first_pass_res.resize(1); }
  t->last_repeat_lvl[0] = repeatedLvl; t->last_first_pass_res[0] = firstPassRes;
  if (lastResiduals) memcpy(lastResiduals, lastRes, sizeof(lastRes));
  if (lastFlow) memcpy(lastFlow, flow, sizeof(flow));
  if (n_evals) *n_evals = evals;
  if (H_out) memcpy(H_out, H, sizeof(H));
  if (b_out) memcpy(b_out, b, sizeof(b));
  if (failed) { if (good) *good = 0; return 0; }   // the reference returns false without touching lastToNew_out / aff_g2l_out (:731-732)
  double aff[2] = {affA, affB};
  bool trackingGood = true;
  if ((trk.modeA != 0 && (fabsf((float)aff[0]) > 1.2f)) || (trk.modeB != 0 && (fabsf((float)aff[1]) > 200.0f))) trackingGood = false;
  double rel[2];
  affFromTo(trk.ref_exposure, new_exposure, trk.ref_aff_a, trk.ref_aff_b, aff[0], aff[1], rel);
  if ((trk.modeA == 0 && (fabsf(logf((float)rel[0])) > 1.5f)) || (trk.modeB == 0 && (fabsf((float)rel[1]) > 200.0f))) trackingGood = false;
  if (trk.modeA < 0) aff[0] = 0;
  if (trk.modeB < 0) aff[1] = 0;
  poseTo7(cur, pose7_io);
  aff_io[0] = aff[0]; aff_io[1] = aff[1];
  if (good) *good = trackingGood ? 1 : 0;
  if (lastLvl == 0 && cb && cb->visual) cb->visual(cb->user, H, b, trackingGood ? 1 : 0);   // addVisualToCoarseGraph (:763-767)
  return 0;
}

// lastF_2_fh_tries of FullSystem::trackNewCoarse (FullSystem.cpp:364-402) from three camToWorld poses
int dmvio_hip_make_track_hypotheses(const double slast_c2w[7], const double sprelast_c2w[7], const double lastF_c2w[7], double* out, int max_out) {
  if (!slast_c2w || !sprelast_c2w || !lastF_c2w || !out) return failmsg("make_track_hypotheses: null argument");
  const Pose slast = poseFrom7(slast_c2w), sprelast = poseFrom7(sprelast_c2w), lastF = poseFrom7(lastF_c2w);
  const Pose fh_2_slast = poseMul(poseInv(sprelast), slast);          // slast_2_sprelast, assumed equal to fh_2_slast
  const Pose lastF_2_slast = poseMul(poseInv(slast), lastF);
  const Pose fhInv = poseInv(fh_2_slast);
  std::vector<Pose> tries;
  tries.push_back(poseMul(fhInv, lastF_2_slast));                       // constant motion
  tries.push_back(poseMul(poseMul(fhInv, fhInv), lastF_2_slast));       // double motion (frame skipped)
  { double lg[6]; poseLogHost(fh_2_slast, lg); for (int i = 0; i < 6; i++) lg[i] *= 0.5; tries.push_back(poseMul(poseInv(poseExp(lg)), lastF_2_slast)); }  // half motion
  tries.push_back(lastF_2_slast);                                       // zero motion
  Pose ident; ident.q.w = 1; ident.q.x = ident.q.y = ident.q.z = 0; ident.t[0] = ident.t[1] = ident.t[2] = 0;
  tries.push_back(ident);                                               // zero motion from the keyframe
  const double d = (double)0.02f;  /* float rotDelta widened by Sophus::Quaterniond */                                              // the reference's rotDelta loop runs exactly once (:376)
  static const int sg[26][3] = {{1,0,0},{0,1,0},{0,0,1},{-1,0,0},{0,-1,0},{0,0,-1},{1,1,0},{0,1,1},{1,0,1},{-1,1,0},{0,-1,1},{-1,0,1},{1,-1,0},{0,1,-1},{1,0,-1},
                                {-1,-1,0},{0,-1,-1},{-1,0,-1},{-1,-1,-1},{-1,-1,1},{-1,1,-1},{-1,1,1},{1,-1,-1},{1,-1,1},{1,1,-1},{1,1,1}};
  const Pose base = poseMul(fhInv, lastF_2_slast);
  for (int k = 0; k < 26; k++) {
    Pose R = ident;
    Quatd q = {1, sg[k][0] * d, sg[k][1] * d, sg[k][2] * d};
    R.q = qnormalize(q);
    tries.push_back(poseMul(base, R));
  }
  const int n = std::min((int)tries.size(), max_out);
  for (int i = 0; i < n; i++) poseTo7(tries[i], out + 7 * i);
  return n;
}

// The try loop of FullSystem::trackNewCoarse (FullSystem.cpp:419-489).  The reference runs the hypotheses one after the other, feeding
// the best residuals so far (achievedRes) to the next try as abort thresholds.  Here try 0 runs alone (the usual winner); only if it
// misses the re-track threshold the remaining tries run as ONE batch without thresholds, and the sequential rule is replayed on
// their per-level residuals: a level residual above 1.5x the running threshold marks the try as aborted at that level, exactly as
// CoarseTracker.cpp:731-732 would have (a level repeated after levelCutoffRepeat is tested with the residual of each of its two passes) — per-level
// results do not depend on the thresholds, so the outcome is identical.
int dmvio_hip_tracker_track_new_coarse(dmvio_hip_tracker* t, int new_slot, float new_exposure, int n_tries, const double* tries7, const double aff_last[2],
                                       double lastCoarseRMSE_io[5], double reTrackThreshold, double pose7_out[7], double aff_out[2], double flow_out[3],
                                       int* winner, int* tries_used, int* tracking_good) {
  if (!t || !tries7 || !aff_last || !lastCoarseRMSE_io || !pose7_out || !aff_out) return failmsg("track_new_coarse: null argument");
  if (n_tries < 1) return failmsg("track_new_coarse: no hypotheses");
  const int L = t->ctx->levels;
  std::vector<double> poses(tries7, tries7 + 7 * (size_t)n_tries), affs(2 * (size_t)n_tries), lr(5 * (size_t)n_tries), fl(3 * (size_t)n_tries);
  std::vector<int> good(n_tries), slots(n_tries, new_slot), rep_lvl(n_tries, -1);
  std::vector<double> rep_first(n_tries, NAN);
  std::vector<float> exps(n_tries, new_exposure);
  for (int i = 0; i < n_tries; i++) { affs[2 * i] = aff_last[0]; affs[2 * i + 1] = aff_last[1]; }
  double achieved[5];
  for (int k = 0; k < 5; k++) achieved[k] = NAN;
  bool haveOneGood = false, trackingGoodRet = false;
  int win = -1, used = 0;
  double flow[3] = {100, 100, 100}, bestPose[7], bestAff[2] = {0, 0};
  memcpy(bestPose, tries7, sizeof(bestPose));
  int computed = 0;
  const bool split = t->xworld >= 1 && (bool)t->xchg;   // xworld == 1 only through the test hook of dmv_tracker_set_exchange: the whole exchange path on one device
  for (int i = 0; i < n_tries; i++) {
    if (i >= computed) {
      const int first = computed, cnt = (first == 0) ? 1 : n_tries - first;
      if (first == 0 || !split) {
        if (int r = dmvio_hip_tracker_track_batch(t, cnt, slots.data() + first, exps.data() + first, poses.data() + 7 * first, affs.data() + 2 * first, L - 1, nullptr,
                                                  lr.data() + 5 * first, fl.data() + 3 * first, nullptr, nullptr, good.data() + first, nullptr)) return r;
        for (int k = 0; k < cnt; k++) { rep_lvl[first + k] = t->last_repeat_lvl[k]; rep_first[first + k] = t->last_first_pass_res[k]; }
      } else {
        // the remaining hypotheses over the ranks (try 0 ran on every rank: same inputs, same bits, no exchange when it already ends the loop): rank r takes the tries
        // first + r, first + r + world, ...; one record of 20 doubles per try [pose7 | aff2 | lastResiduals5 | flow3 | good | repeated level | its first-pass residual],
        // written by its owner, zero elsewhere; the fp64 sum over the ranks (x + 0 = x, NaN stays NaN) hands every rank the complete list, and the sequential
        // abort / winner rule below is replayed identically everywhere
        enum { REC = 20 };
        std::vector<int> mine;
        for (int j = first + t->xrank; j < n_tries; j += t->xworld) mine.push_back(j);
        const int m = (int)mine.size();
        // one extra slot behind the records counts the ranks whose share FAILED locally: a rank must never leave before the exchange (its peers would wait in the
        // all-reduce for ever) — it enters with its error flagged, and every rank fails together afterwards
        std::vector<double> rec((size_t)REC * cnt + 1, 0.0);
        int local_rc = 0;
        std::string local_err;
        if (m > 0) {
          std::vector<double> mp(7 * (size_t)m), ma(2 * (size_t)m), mlr(5 * (size_t)m), mfl(3 * (size_t)m);
          std::vector<int> mg(m), ms(m, new_slot);
          std::vector<float> me(m, new_exposure);
          for (int k = 0; k < m; k++) { memcpy(&mp[7 * (size_t)k], &poses[7 * (size_t)mine[k]], sizeof(double) * 7); ma[2 * k] = aff_last[0]; ma[2 * k + 1] = aff_last[1]; }
          // a share of ONE try runs the device-resident LM like the batch of the unsplit call does (never the host LM of single-frame tracking)
          const int keep_mode = t->single_host_lm;
          t->single_host_lm = 0;
          local_rc = dmvio_hip_tracker_track_batch(t, m, ms.data(), me.data(), mp.data(), ma.data(), L - 1, nullptr, mlr.data(), mfl.data(), nullptr, nullptr, mg.data(), nullptr);
          t->single_host_lm = keep_mode;
          if (local_rc) { local_err = dmv_err(); rec.back() = 1.0; }
          else
            for (int k = 0; k < m; k++) {
              double* q = &rec[(size_t)REC * (mine[k] - first)];
              memcpy(q, &mp[7 * (size_t)k], sizeof(double) * 7); q[7] = ma[2 * k]; q[8] = ma[2 * k + 1];
              memcpy(q + 9, &mlr[5 * (size_t)k], sizeof(double) * 5); memcpy(q + 14, &mfl[3 * (size_t)k], sizeof(double) * 3);
              q[17] = mg[k]; q[18] = t->last_repeat_lvl[k]; q[19] = t->last_first_pass_res[k];
            }
        }
        if (int r = t->xchg(rec.data(), rec.size())) return r;
        if (local_rc) return failmsg("track_new_coarse: this rank's share of the hypotheses failed (" + local_err + ")");
        if (rec.back() != 0.0) return failmsg("track_new_coarse: the share of the hypotheses of " + std::to_string((int)rec.back()) + " other rank(s) failed");
        for (int k = 0; k < cnt; k++) {
          const double* q = &rec[(size_t)REC * k];
          const int j = first + k;
          memcpy(&poses[7 * (size_t)j], q, sizeof(double) * 7); affs[2 * j] = q[7]; affs[2 * j + 1] = q[8];
          memcpy(&lr[5 * (size_t)j], q + 9, sizeof(double) * 5); memcpy(&fl[3 * (size_t)j], q + 14, sizeof(double) * 3);
          good[j] = q[17] != 0; rep_lvl[j] = (int)q[18]; rep_first[j] = q[19];
        }
      }
      computed = first + cnt;
    }
    used++;
    // replay the abort thresholds of the sequential loop on this try's per-level residuals
    double res[5];
    for (int k = 0; k < 5; k++) res[k] = NAN;
    bool ok = good[i] != 0, aborted = false;
    for (int lvl = L - 1; lvl >= 0; lvl--) {
      // a level that ran twice (levelCutoffRepeat) met the abort test after each pass, the first time with the residual the repeat later overwrote
      if (lvl == rep_lvl[i]) {
        const double r1 = rep_first[i];
        res[lvl] = r1;
        if (std::isnan(r1) || r1 > 1.5 * achieved[lvl]) { aborted = true; break; }
      }
      const double rv = lr[5 * i + lvl];
      res[lvl] = rv;
      if (std::isnan(rv) || rv > 1.5 * achieved[lvl]) { aborted = true; break; }
    }
    if (aborted) ok = false;
    if (ok) trackingGoodRet = true;
    if (ok && std::isfinite((float)res[0]) && !(res[0] >= achieved[0])) {
      for (int k = 0; k < 3; k++) flow[k] = fl[3 * i + k];
      bestAff[0] = affs[2 * i]; bestAff[1] = affs[2 * i + 1];
      memcpy(bestPose, poses.data() + 7 * i, sizeof(bestPose));
      haveOneGood = true; win = i;
    }
    if (haveOneGood)
      for (int k = 0; k < 5; k++)
        if (!std::isfinite((float)achieved[k]) || achieved[k] > res[k]) achieved[k] = res[k];
    if (haveOneGood && achieved[0] < lastCoarseRMSE_io[0] * reTrackThreshold) break;
  }
  if (!haveOneGood) { flow[0] = flow[1] = flow[2] = 0; bestAff[0] = aff_last[0]; bestAff[1] = aff_last[1]; memcpy(bestPose, tries7, sizeof(bestPose)); }
  for (int k = 0; k < 5; k++) lastCoarseRMSE_io[k] = achieved[k];
  memcpy(pose7_out, bestPose, sizeof(bestPose));
  aff_out[0] = bestAff[0]; aff_out[1] = bestAff[1];
  if (flow_out) for (int k = 0; k < 3; k++) flow_out[k] = flow[k];
  if (winner) *winner = win;
  if (tries_used) *tries_used = used;
  if (tracking_good) *tracking_good = trackingGoodRet ? 1 : 0;
  return 0;
}

}  // extern "C"
int dmv_tracker_set_exchange(dmvio_hip_tracker* t, std::function<int(double*, size_t)> allreduce_sum, int rank, int world) {
  if (!t) return failmsg("null tracker");
  // dmvio_hip_tracker_debug_split_single_rank(t, 1) (tests): a group of ONE rank still takes the split path — every try is "mine", the all-reduce is the identity — so
  // that the exchange (RCCL on the context's stream included) runs on a one-device box.  An explicit call on THIS tracker, never the environment.
  const bool force1 = world == 1 && allreduce_sum && t->debug_split1;
  if ((world <= 1 && !force1) || !allreduce_sum) { t->xchg = nullptr; t->xrank = 0; t->xworld = 0; return 0; }
  if (rank < 0 || rank >= world) return failmsg("tracker_set_comm: 0 <= rank < world");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  t->xchg = std::move(allreduce_sum); t->xrank = rank; t->xworld = world;
  return 0;
}
dmvio_hip_ctx* dmv_tracker_ctx(dmvio_hip_tracker* t) { return t ? t->ctx : nullptr; }
bool dmv_tracker_debug_split1(dmvio_hip_tracker* t) { return t && t->debug_split1; }
extern "C" int dmvio_hip_tracker_debug_split_single_rank(dmvio_hip_tracker* t, int on) {
  if (!t) return failmsg("null tracker");
  t->debug_split1 = on != 0;
  return 0;
}
extern "C" {
#ifdef DMV_LM_TICKS
extern "C" int dmvio_hip_debug_lm_ticks(double out8[8], int reset) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lm_ticks), sizeof(double) * 8));
  if (reset) { double z[8] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_lm_ticks), z, sizeof(z))); }
  return 0;
}
#endif
int dmvio_hip_tracker_last_ticks(dmvio_hip_tracker* t, long long* ticks_step, long long* ticks_eval) {
  if (!t) return failmsg("null tracker");
  if (ticks_step) *ticks_step = t->last_ticks_step;
  if (ticks_eval) *ticks_eval = t->last_ticks_eval;
  return 0;
}

// Diagnostics: the evaluation loop's division (refined reciprocal shared by the quotients of one denominator) next to the compiler's
// IEEE division, for n operand pairs — lets a test pin the claim that both give the same bits in the operand range of the path.
int dmvio_hip_selftest_divide(dmvio_hip_ctx* c, int n, const float* a, const float* b, float* q_shared, float* q_ieee) {
  if (!c || n <= 0 || !a || !b || !q_shared || !q_ieee) return failmsg("selftest_divide: bad argument");
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(hipSetDevice(c->device));
  float* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, sizeof(float) * 4 * (size_t)n));
  HIPCHK(c->bounce.h2d(d, a, sizeof(float) * n, c->stream));
  HIPCHK(c->bounce.h2d(d + n, b, sizeof(float) * n, c->stream));
  hipLaunchKernelGGL(k_selftest_divide, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n);
  HIPCHK(hipGetLastError());
  HIPCHK(c->bounce.d2h(q_shared, d + 2 * (size_t)n, sizeof(float) * n, c->stream));
  HIPCHK(c->bounce.d2h(q_ieee, d + 3 * (size_t)n, sizeof(float) * n, c->stream));
  HIPCHK(c->bounce.finish(c->stream));
  HIPCHK(hipFree(d));
  return 0;
}

int dmvio_hip_tracker_last_launch(dmvio_hip_tracker* t, int* workgroups_per_problem, int* threads_per_workgroup) {
  if (!t) return failmsg("null tracker");
  if (workgroups_per_problem) *workgroups_per_problem = t->last_cluster;
  if (threads_per_workgroup) *threads_per_workgroup = t->last_threads;
  return 0;
}

int dmvio_hip_tracker_last_work(dmvio_hip_tracker* t, long long* n_evals, long long* n_point_evals) {
  if (!t) return failmsg("null tracker");
  if (n_evals) *n_evals = t->last_evals;
  if (n_point_evals) *n_point_evals = t->last_point_evals;
  return 0;
}

// FullSystem::printResult (FullSystem.cpp:256-298): one line "timestamp tx ty tz qx qy qz qw" per frame with a valid pose, 15
// significant digits, poses relative to the first frame (camToFirst = firstPose^-1 * camToWorld); frames that are not keyframes are
// re-based on their tracking reference's CURRENT pose when tracking_ref / camToTrackingRef7 are given (useCamToTrackingRef).
// Host-only (no device work): the on-disk edge of the path, so that trajectories of both pipelines compare file against file.
int dmvio_hip_write_result_txt(const char* path, int n, const double* timestamps, const double* camToWorld7, const unsigned char* pose_valid,
                               const int* tracking_ref, const double* camToTrackingRef7, const double firstPose7[7]) {
  if (!path || n < 0 || (n > 0 && (!timestamps || !camToWorld7)) || !firstPose7) return failmsg("write_result_txt: bad argument");
  if (tracking_ref && !camToTrackingRef7) return failmsg("write_result_txt: tracking_ref without camToTrackingRef7");
  FILE* f = fopen(path, "w");
  if (!f) return failmsg("write_result_txt: cannot open the file");
  const Pose firstInv = poseInv(poseFrom7(firstPose7));
  for (int i = 0; i < n; i++) {
    if (pose_valid && !pose_valid[i]) continue;
    Pose c2w = poseFrom7(camToWorld7 + 7 * i);
    if (tracking_ref && tracking_ref[i] >= 0) {
      if (tracking_ref[i] >= n) { fclose(f); return failmsg("write_result_txt: tracking_ref out of range"); }
      c2w = poseMul(poseFrom7(camToWorld7 + 7 * tracking_ref[i]), poseFrom7(camToTrackingRef7 + 7 * i));
    }
    double p[7];
    poseTo7(poseMul(firstInv, c2w), p);
    fprintf(f, "%.15g %.15g %.15g %.15g %.15g %.15g %.15g %.15g\n", timestamps[i], p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
  }
  if (fclose(f) != 0) return failmsg("write_result_txt: write failed");
  return 0;
}

}  // extern "C"
