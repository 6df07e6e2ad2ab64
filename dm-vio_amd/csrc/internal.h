// internals shared by the translation units of libdmvio_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <mutex>
#include <vector>
#include <string>
#include <functional>
#include "common.h"

std::string& dmv_err();
static inline int fail(const char* what, const char* file, int line, hipError_t e) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
  dmv_err() = buf;
  return -1;
}
static inline int failmsg(const std::string& m) { dmv_err() = m; return -2; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(#x, __FILE__, __LINE__, _e); } while (0)
#define HIPCHKP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fail(#x, __FILE__, __LINE__, _e); return nullptr; } } while (0)

struct dmvio_hip_ctx {
  int device = 0, w = 0, h = 0, levels = 0, n_slots = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  dmv::FrameStore fs{};
  float* d_upload = nullptr;  // staging for host uploads (w*h)
  float* d_f3 = nullptr;      // download scratch (w*h*3)
  dmv::PyrGeom pg{};
  int wl[DMV_MAX_LEVELS] = {}, hl[DMV_MAX_LEVELS] = {};
  int *d_slots = nullptr, *h_slots = nullptr;
  int slots_cap = 0, slots_valid = 0;
  std::vector<const float*> h_lvl0;   // host mirror of FrameStore::lvl0 (kept by the build entry points)
  const float* levelPtr(int slot, int lvl) const { return lvl == 0 ? h_lvl0[slot] : fs.own_level(slot, lvl); }
  unsigned int build_gen = 0;   // generation counter of pyramid builds (FrameStore::build_gen / bad_gen stamps)
  std::mutex mu;
};


// hypothesis-parallel trackNewCoarse: the element-wise fp64 sum over all ranks of a small HOST buffer, in place (set by dmvio_hip_tracker_set_comm /
// _set_comm_callbacks in capi_ba.hip, which owns the RCCL calls; used by dmvio_hip_tracker_track_new_coarse in capi.hip)
struct dmvio_hip_tracker;
int dmv_tracker_set_exchange(dmvio_hip_tracker* t, std::function<int(double*, size_t)> allreduce_sum, int rank, int world);
dmvio_hip_ctx* dmv_tracker_ctx(dmvio_hip_tracker* t);
