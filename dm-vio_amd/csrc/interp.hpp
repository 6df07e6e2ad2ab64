// Bilinear (I, dx, dy) tap on the intensity-only pyramid planes (shared by the coarse tracker and the BA linearisation).
#pragma once
#include "common.h"

namespace dmv {

// getInterpolatedElement33 (src/dso/util/globalFuncs.h:103-118) on an intensity-only plane: the four taps' gradient
// channels are the central differences the reference stored in dIp[.][1..2] (HessianBlocks.cpp:172-181), rebuilt from the
// 4x4 intensity neighbourhood: rows iy-1 (2 px), iy (4 px), iy+1 (4 px), iy+2 (2 px) = 48 B in four unaligned vector loads.
// Callers guarantee 1 <= ix, ix+2 <= w-1, 1 <= iy, iy+2 <= h-1 (true for every in-bounds tap of tracker and BA).
__device__ __forceinline__ float fin0(const float v) { return isfinite(v) ? v : 0.0f; }
template <bool GUARD>
__device__ __forceinline__ float fin0g(const float v) { return GUARD ? fin0(v) : v; }
__device__ __forceinline__ float3 interp33(const float* __restrict__ img, const float x, const float y, const int width) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy;
  const float dxdy = dx * dy;
  const float* bp = img + ix + iy * width;
  float2 A, D;
  float4 B, C;
  __builtin_memcpy(&A, bp - width, 8);
  __builtin_memcpy(&B, bp - 1, 16);
  __builtin_memcpy(&C, bp + width - 1, 16);
  __builtin_memcpy(&D, bp + 2 * width, 8);
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  // taps: p00 = (ix,iy) = B.y, p10 = B.z, p01 = C.y, p11 = C.z
  const float gx00 = fin0(0.5f * (B.z - B.x)), gx10 = fin0(0.5f * (B.w - B.y));
  const float gx01 = fin0(0.5f * (C.z - C.x)), gx11 = fin0(0.5f * (C.w - C.y));
  const float gy00 = fin0(0.5f * (C.y - A.x)), gy10 = fin0(0.5f * (C.z - A.y));
  const float gy01 = fin0(0.5f * (D.x - B.y)), gy11 = fin0(0.5f * (D.y - B.z));
  float3 r;
  r.x = w11 * C.z + w01 * C.y + w10 * B.z + w00 * B.y;
  r.y = w11 * gx11 + w01 * gx01 + w10 * gx10 + w00 * gx00;
  r.z = w11 * gy11 + w01 * gy01 + w10 * gy10 + w00 * gy00;
  return r;
}

// The same tap split in two: the four unaligned vector loads (issued early, software pipelining) and the arithmetic on them.
struct Taps33 { float2 A, D; float4 B, C; };
__device__ __forceinline__ void interp33Load(const float* __restrict__ img, const float x, const float y, const int width, Taps33& t) {
  const int ix = (int)x, iy = (int)y;
  const float* bp = img + ix + iy * width;
  __builtin_memcpy(&t.A, bp - width, 8);
  __builtin_memcpy(&t.B, bp - 1, 16);
  __builtin_memcpy(&t.C, bp + width - 1, 16);
  __builtin_memcpy(&t.D, bp + 2 * width, 8);
}
// The same twelve values out of a plane stored in 8x4 tiles (FrameStore::tiled0): one dword load per element, each with its own tile address — a 4-pixel row segment
// crosses a tile column for 3/8 of the taps, a row pair a tile row for 3/4, so no vector load is safe; what the layout buys is lines per tap (2.4 instead of 4.3), and
// the level-0 gather is bound by missed lines, not by load instructions (profiles/r02_gather_bounds.md: 12 scalar loads cost 4 % more than 4 vector loads).
__device__ __forceinline__ void interp33LoadTiled(const float* __restrict__ img, const float x, const float y, const int tpr, Taps33& t) {
  const int ix = (int)x, iy = (int)y;
  const unsigned int strip = (unsigned int)tpr << 5;   // floats per strip of four image rows
  unsigned int ro[4], co[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = iy - 1 + k, c = ix - 1 + k;
    ro[k] = (unsigned int)(r >> 2) * strip + (unsigned int)((r & 3) << 3);
    co[k] = ((unsigned int)(c >> 3) << 5) + (unsigned int)(c & 7);
  }
  t.A.x = img[ro[0] + co[1]]; t.A.y = img[ro[0] + co[2]];
  t.B.x = img[ro[1] + co[0]]; t.B.y = img[ro[1] + co[1]]; t.B.z = img[ro[1] + co[2]]; t.B.w = img[ro[1] + co[3]];
  t.C.x = img[ro[2] + co[0]]; t.C.y = img[ro[2] + co[1]]; t.C.z = img[ro[2] + co[2]]; t.C.w = img[ro[2] + co[3]];
  t.D.x = img[ro[3] + co[1]]; t.D.y = img[ro[3] + co[2]];
}
// GUARD = false: for planes stamped clean by k_build_pyramids (FrameStore::bad_gen) — the guards cannot fire, the values are the same
template <bool GUARD = true>
__device__ __forceinline__ float3 interp33Finish(const Taps33& t, const float x, const float y) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy;
  const float dxdy = dx * dy;
  const float2 A = t.A, D = t.D;
  const float4 B = t.B, C = t.C;
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  const float gx00 = fin0g<GUARD>(0.5f * (B.z - B.x)), gx10 = fin0g<GUARD>(0.5f * (B.w - B.y));
  const float gx01 = fin0g<GUARD>(0.5f * (C.z - C.x)), gx11 = fin0g<GUARD>(0.5f * (C.w - C.y));
  const float gy00 = fin0g<GUARD>(0.5f * (C.y - A.x)), gy10 = fin0g<GUARD>(0.5f * (C.z - A.y));
  const float gy01 = fin0g<GUARD>(0.5f * (D.x - B.y)), gy11 = fin0g<GUARD>(0.5f * (D.y - B.z));
  float3 r;
  r.x = w11 * C.z + w01 * C.y + w10 * B.z + w00 * B.y;
  r.y = w11 * gx11 + w01 * gx01 + w10 * gx10 + w00 * gx00;
  r.z = w11 * gy11 + w01 * gy01 + w10 * gy10 + w00 * gy00;
  return r;
}

// dIp[lvl][idx][1], [2] of the reference at pixel (x, y) of a level plane: central differences with the reference's
// flat-index range (rows 1..h-2) and isfinite guard (HessianBlocks.cpp:172-181).
__device__ __forceinline__ float2 gradAt(const float* __restrict__ I, const int w, const int h, const int x, const int y) {
  const int idx = x + y * w;
  float dx = 0.f, dy = 0.f;
  if (idx >= w && idx < w * (h - 1)) {
    dx = 0.5f * (I[idx + 1] - I[idx - 1]);
    dy = 0.5f * (I[idx + w] - I[idx - w]);
    if (!isfinite(dx)) dx = 0.f;
    if (!isfinite(dy)) dy = 0.f;
  }
  return make_float2(dx, dy);
}

}  // namespace dmv
