// internals shared by the translation units of libdmvio_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include <string>
#include <functional>
#include "common.h"

std::string& dmv_err();
// counts the errors recorded on this thread: a download registered with DmvBounce::d2h is only delivered by a finish() of the SAME error epoch — an entry point that
// failed between its d2h() and its finish() returns early and leaves registrations that point at dead stack buffers or at caller arrays that may be gone by then
unsigned int& dmv_err_epoch();
static inline int fail(const char* what, const char* file, int line, hipError_t e) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
  dmv_err() = buf;
  dmv_err_epoch()++;
  return -1;
}
static inline int failmsg(const std::string& m) { dmv_err() = m; dmv_err_epoch()++; return -2; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(#x, __FILE__, __LINE__, _e); } while (0)
#define HIPCHKP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fail(#x, __FILE__, __LINE__, _e); return nullptr; } } while (0)

// Host <-> device copies of caller-owned arrays go through pinned memory the LIBRARY owns.  Handing a pageable pointer to hipMemcpyAsync makes the runtime pin the caller's pages
// for larger transfers and keep that registration cached; when the caller later frees the array (the reference deletes a FrameHessian, a std::vector goes out of scope) and the
// C library returns the pages to the kernel, the MMU notifier evicts the process's GPU queues and the NEXT submission waits 8-20 ms for their restore (measured inside the
// reference's FullSystem: every frame marginalisation was followed by one such stall in dmvio_hip_frame_upload; GPU_PINNED_MIN_XFER_SIZE=<huge> removes it, as does this).
// One Bounce per handle (context, tracker, immature set, initializer, window optimiser): h2d() stages the source and enqueues the copy, d2h() enqueues the copy into the
// bounce and remembers where the caller wants it, finish() waits for the stream and delivers.  Staged sources stay intact until the next finish().
struct DmvBounce {
  char* h = nullptr;
  size_t cap = 0, used = 0;
  struct Out { void* dst; size_t off, bytes; unsigned int epoch; };
  std::vector<Out> outs;
  hipError_t reserve(size_t bytes, hipStream_t s, size_t* off) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (used + need > cap) {
      if (used || !outs.empty()) { hipError_t e = finish(s); if (e != hipSuccess) return e; }
      if (need > cap) {
        if (h) { hipError_t e = hipHostFree(h); h = nullptr; cap = 0; if (e != hipSuccess) return e; }
        const size_t want = need > ((size_t)4 << 20) ? need + need / 2 : ((size_t)4 << 20);
        hipError_t e = hipHostMalloc((void**)&h, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        cap = want;
      }
    }
    *off = used; used += need;
    return hipSuccess;
  }
  hipError_t h2d(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    if (grouping && used + ((bytes + 255) & ~(size_t)255) > cap) { hipError_t e = flush_group(s); if (e != hipSuccess) return e; }   // reserve() is about to recycle the staging area
    size_t off; hipError_t e = reserve(bytes, s, &off); if (e != hipSuccess) return e;
    memcpy(h + off, h_src, bytes);
    if (grouping) { group.push_back({(char*)d_dst, off, bytes}); return hipSuccess; }
    return hipMemcpyAsync(d_dst, h + off, bytes, hipMemcpyHostToDevice, s);
  }
  // Uploads of one call whose destinations lie back to back on the device (arena allocations: 256-byte granules, like the staging area's) leave as ONE copy: between
  // begin_group() and end_group() h2d() only stages; end_group() merges pieces that follow each other both in the staging area and on the device at the same padded
  // distance (the padding it copies along is the unused tail of the previous allocation) and enqueues what is left.  Nothing that consumes the data may be enqueued
  // before end_group().
  struct Piece { char* dst; size_t off, bytes; };
  std::vector<Piece> group;
  bool grouping = false;
  void begin_group() { grouping = true; group.clear(); }
  hipError_t flush_group(hipStream_t s) {
    size_t i = 0;
    while (i < group.size()) {
      size_t j = i;
      while (j + 1 < group.size()) {
        const size_t step = (group[j].bytes + 255) & ~(size_t)255;
        // merged only between arena granules (256-byte aligned destinations): the padding copied along is then the unused tail of the previous allocation,
        // never live data of an array that holds both pieces
        if (group[j + 1].off != group[j].off + step || group[j + 1].dst != group[j].dst + step || ((size_t)group[j].dst & 255) || ((size_t)group[j + 1].dst & 255)) break;
        j++;
      }
      hipError_t e = hipMemcpyAsync(group[i].dst, h + group[i].off, group[j].off + group[j].bytes - group[i].off, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) { group.clear(); return e; }
      i = j + 1;
    }
    group.clear();
    return hipSuccess;
  }
  hipError_t end_group(hipStream_t s) { grouping = false; return flush_group(s); }
  hipError_t d2h(void* h_dst, const void* d_src, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    size_t off; hipError_t e = reserve(bytes, s, &off); if (e != hipSuccess) return e;
    e = hipMemcpyAsync(h + off, d_src, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) outs.push_back({h_dst, off, bytes, dmv_err_epoch()});   // registered only once the copy is really on its way
    return e;
  }
  hipError_t finish(hipStream_t s) {
    hipError_t e = hipStreamSynchronize(s);
    const unsigned int now = dmv_err_epoch();
    if (e == hipSuccess) for (const Out& o : outs) if (o.epoch == now) memcpy(o.dst, h + o.off, o.bytes);   // registrations of a call that failed since are dropped
    outs.clear(); used = 0;
    return e;
  }
  void release() { if (h) hipHostFree(h); h = nullptr; cap = used = 0; outs.clear(); group.clear(); grouping = false; }
};

struct dmvio_hip_ctx {
  int device = 0, w = 0, h = 0, levels = 0, n_slots = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  dmv::FrameStore fs{};
  float* d_upload = nullptr;  // staging for host uploads (w*h)
  float* d_f3 = nullptr;      // download scratch (w*h*3)
  dmv::PyrGeom pg{};
  int wl[DMV_MAX_LEVELS] = {}, hl[DMV_MAX_LEVELS] = {};
  // slot lists of the batched builds, staged on the device: the last few distinct lists stay (a double-buffered pipeline alternates between two)
  struct SlotList { int *d = nullptr, *h = nullptr; int cap = 0, n = 0; unsigned long long used = 0; };
  SlotList slot_lists[4];
  unsigned long long slot_clock = 0;
  int* d_slots = nullptr;               // the list the current batched build reads (one of slot_lists[].d)
  hipStream_t build_stream = nullptr;   // dmvio_hip_set_build_stream: where the batched builds are enqueued (NULL: the context's stream)
  std::vector<const float*> h_lvl0;   // host mirror of FrameStore::lvl0 (kept by the build entry points)
  const float* levelPtr(int slot, int lvl) const { return lvl == 0 ? h_lvl0[slot] : fs.own_level(slot, lvl); }
  unsigned int build_gen = 0;   // generation counter of pyramid builds (FrameStore::build_gen / bad_gen stamps)
  std::vector<unsigned char> h_tiled;   // host mirror of FrameStore::tiled0: level 0 of the slot is stored in 8x4 tiles (written so by the batched raw-image build)
  int raw_batch_kernel = 1;             // dmvio_hip_set_raw_batch_kernel: 1 = the wave-autonomous register build where the geometry allows (<= 4 levels, sides % 8 == 0), 0 = the LDS-tile build
  int raw_batch_tiled = 0;              // dmvio_hip_set_raw_batch_layout: what dmvio_hip_frames_from_raw_device_batch writes (1 = tiles where the image size allows)
  DmvBounce bounce;             // caller-owned arrays cross PCIe through here (used under `mu`)
  std::mutex mu;
};


// Level 0 of `slot` back into the row-major layout if the batched raw-image build stored it in 8x4 tiles (FrameStore::tiled0): called by every consumer of a level-0 plane
// other than the coarse tracker's batch kernel (reference template, single-frame tracking, window optimiser, immature points, initializer, downloads) before it reads the
// slot.  No-op (one host-side flag test) for every other slot.  _locked: the caller holds c->mu; the conversion runs on the context's stream and is waited for.
extern "C" int dmv_ensure_row_major_locked(dmvio_hip_ctx* c, int slot);
extern "C" int dmv_ensure_row_major(dmvio_hip_ctx* c, int slot);

// Host-side mirror of the window's point / residual graph, mutated the way EnergyFunctional mutates its own (capi_graph.hip; include/dmvio_hip.h "window graph").
#define DMV_GRAPH_MAX_FRAMES 16   // keyframes of a window, and residuals of a point (one per other keyframe)
struct DmvGraphPoint {
  float u, v, idepth;
  float color[8], weights[8];
  short target[DMV_GRAPH_MAX_FRAMES];   // EFPoint::residualsAll order: frame index of each residual's target; -1 = its frame was removed, the residual not yet dropped
  int lin[DMV_GRAPH_MAX_FRAMES];        // EFResidual::isLinearized: index of the residual's frozen linearisation in dmvio_hip_graph::linPool, -1 = not linearised
  int nres;
  unsigned char prior;
};
struct DmvGraphLin { float J[74]; float res_toZeroF[8]; };   // EFResidual::J (RawResidualJacobian, dmvio_hip_ba_get_full_jacobians' layout) and ::res_toZeroF
struct dmvio_hip_graph {
  std::mutex mu;
  std::vector<std::vector<DmvGraphPoint>> frames;   // EnergyFunctional::frames -> EFFrame::points
  std::vector<DmvGraphLin> linPool;                  // linearisations of the residuals that carry one; linFree: slots given back by dropped residuals / removed points
  std::vector<int> linFree;
  int nLin = 0;
  int nPoints = 0, nRes = 0, nDangling = 0;
  unsigned long long version = 0;                    // counts structural changes (not value updates)
  unsigned long long flat_version = ~0ull;           // `version` when the graph was last flattened (dmvio_hip_graph_export / dmvio_hip_ba_set_graph_from): values that come back in
                                                     // flat order (dmvio_hip_graph_set_idepths) are only accepted while the two agree
};

// hypothesis-parallel trackNewCoarse: the element-wise fp64 sum over all ranks of a small HOST buffer, in place (set by dmvio_hip_tracker_set_comm /
// _set_comm_callbacks in capi_ba.hip, which owns the RCCL calls; used by dmvio_hip_tracker_track_new_coarse in capi.hip)
struct dmvio_hip_tracker;
int dmv_tracker_set_exchange(dmvio_hip_tracker* t, std::function<int(double*, size_t)> allreduce_sum, int rank, int world);
dmvio_hip_ctx* dmv_tracker_ctx(dmvio_hip_tracker* t);
bool dmv_tracker_debug_split1(dmvio_hip_tracker* t);
