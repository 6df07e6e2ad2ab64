// Shared POD types of libdmvio_hip (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DMV_MAX_LEVELS 6

namespace dmv {

// Per pyramid level geometry + intrinsics (CoarseTracker::makeK, CoarseTracker.cpp:105-134).
struct LevelGeom {
  int w, h;
  float fx, fy, cx, cy;
  float Ki[9];
};

// Resident image pyramids: every frame slot holds the INTENSITY plane of all levels back to back (float, 4 B/px).
// The reference materialises (I, dx, dy) as Eigen::Vector3f per pixel (HessianBlocks.cpp:128-191); here the
// central-difference gradients are recomputed from the 4x4 intensity neighbourhood at every tap — bit-identical
// values (same fp32 operations), 4x less pyramid traffic and footprint, whole pyramids stay L2-resident.
struct FrameStore {
  float* base;           // n_slots * slot_stride floats
  size_t slot_stride;    // floats per slot
  size_t level_off[DMV_MAX_LEVELS];
  int levels;
  // per slot: generation stamp of the last pyramid build, and of the last build that met a pixel that is not finite or beyond 1e30
  // (k_build_pyramids).  bad_gen[slot] != build_gen[slot]  <=>  every pixel of every level is finite and so is every central difference:
  // consumers may then skip the reference's isfinite guards (HessianBlocks.cpp:172-181, CoarseTracker.cpp:455) — they cannot fire.
  unsigned int *build_gen, *bad_gen;
  // level 0 of a slot is reached through a pointer table: it is either the slot's own plane or — frames attached in place
  // (dmvio_hip_frames_attach_device_batch) — the caller's resident image itself: the intensity plane IS the input image, so nothing is copied.
  // The table lives in device memory (written by k_build_pyramids); host code uses dmvio_hip_ctx::levelPtr.
  const float** lvl0;
  // tiled0[slot] != 0: the slot's OWN level-0 plane is stored in 8x4-pixel tiles (32 floats = one 128-byte line per tile, tiles row-major, w/8 per tile row) instead of
  // row-major — written so by the batched raw-image build (k_build_pyramids_raw<T, true>), which produces level 0 anyway, for the coarse tracker's level-0 gather:
  // the 4x4 footprint of a bilinear tap then touches 2.4 lines on average instead of 4.3.  Only k_track_lm's tiled instantiation reads such a plane; every other
  // consumer has the slot converted back first (dmv_ensure_row_major, host side).  Every build stamps the flag of its slot.
  unsigned char* tiled0;
  // (the table entry is read through an lvalue whose pointee carries the GLOBAL address space: a pointer loaded from memory is otherwise a generic one to the compiler and
  // every image tap behind it a flat_load, which counts on both wait counters and ties the taps to the LDS traffic of the kernels that gather through it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wincompatible-pointer-types-discards-qualifiers"
  __device__ const float* level(int slot, int lvl) const {
    const float* l0 = (const float*)(*reinterpret_cast<const __attribute__((address_space(1))) float* const*>(&lvl0[slot]));
    return lvl == 0 ? l0 : base + (size_t)slot * slot_stride + level_off[lvl];
  }
#pragma clang diagnostic pop
  __host__ __device__ float* own_level(int slot, int lvl) const { return base + (size_t)slot * slot_stride + level_off[lvl]; }
};

// a level's base address made wave-uniform (scalar registers) and typed as a GLOBAL pointer: an integer cast straight to `const float*` is a generic pointer, and every tap
// through it a flat_load
__device__ __forceinline__ const float* dmvUniformGlobal(const float* p) {
  const unsigned long long ia = (unsigned long long)p;
  const unsigned long long ua = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(ia >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)ia);
  return (const float*)(const __attribute__((address_space(1))) float*)ua;
}
// element (x, y) of a level-0 plane stored in 8x4 tiles (tpr = w / 8 tiles per tile row): float offset from the plane's base
__host__ __device__ __forceinline__ unsigned int tiled84Offset(const int x, const int y, const int tpr) {
  return ((((unsigned int)(y >> 2) * (unsigned int)tpr) + (unsigned int)(x >> 3)) << 5) + (unsigned int)(((y & 3) << 3) + (x & 7));
}

struct PyrGeom {
  int levels;
  int w[DMV_MAX_LEVELS], h[DMV_MAX_LEVELS];
  int tiles_x, tiles_y;  // level-0 tiles of k_build_pyramids (2^tw_log2 x 4096 / 2^tw_log2 pixels)
  int tw_log2;           // log2 of the tile width: 7..9
};

// Reference template of the coarse tracker on the device (pc_* of CoarseTracker.h:113-118 as one
// float4 {u, v, idepth, color} record per template point: one coalesced 16-byte load per point).
struct TrackerDev {
  int levels;
  LevelGeom g[DMV_MAX_LEVELS];
  const float4* pc[DMV_MAX_LEVELS];
  int pc_n[DMV_MAX_LEVELS];
  // level 0 only: bit j of word k set <=> template entry 64k+j is one of the reference's "every 32nd point in
  // row-major order" flow-indicator samples (CoarseTracker.cpp:416); entries are stored in tile order.
  const unsigned long long* flow_mask;
  float ref_exposure;
  double ref_aff_a, ref_aff_b;
  float huberTH, coarseCutoffTH, modeA, modeB;
};

// Uniform inputs of one calcRes+calcGS evaluation.
struct EvalP {
  float RKi[9];
  float t[3];
  float aff0, aff1;   // affLL (CoarseTracker.cpp:379)
  float b0;           // lastRef_aff_g2l.b (CoarseTracker.cpp:305)
  float cutoff, maxEnergy;
  int lvl;
};

// accumulator slots of one evaluation
enum {
  ACC_H = 0,        // 45 upper-triangular sums of w * [J0..J7, r] [J0..J7, r]^T  (Accumulator9 order)
  ACC_E = 45,
  ACC_NE = 46,      // numTermsInE
  ACC_NSAT = 47,    // numSaturated
  ACC_NW = 48,      // numTermsInWarped (before padding to x4)
  ACC_FT = 49,      // sumSquaredShiftT
  ACC_FRT = 50,     // sumSquaredShiftRT
  ACC_FN = 51,      // sumSquaredShiftNum
  ACC_N = 52,
  ACC_PAD = 64
};

struct LMProblemIn {
  double pose7[7];
  double aff[2];
  double minRes[5];
  int new_slot;
  float new_exposure;
};
struct LMProblemOut {
  double pose7[7];
  double aff[2];
  double lastRes[5];
  double flow[3];
  double H[64];
  double b[8];
  int good;
  int iterations;
  int n_evals;
  int repeated_lvl;        // the level that ran twice (levelCutoffRepeat, CoarseTracker.cpp:735-739) or -1
  long long n_point_evals;
  long long ticks_step, ticks_eval;  // wall_clock64 (100 MHz) spent in LM control steps / evaluations
  double first_pass_res;   // lastResiduals[repeated_lvl] after its FIRST pass (the abort rule :731 saw that value before the repeat overwrote it)
};

}  // namespace dmv
