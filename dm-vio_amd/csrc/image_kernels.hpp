// Image pyramid on the device.
// Replaces FrameHessian::makeImages (src/dso/FullSystem/HessianBlocks.cpp:128-191).  Only the intensity plane of
// every level is stored (common.h FrameStore); level l+1 = 0.25f * (((a+b)+c)+d) of the 2x2 block of level l, in the
// reference's operand order, so every level is bit-identical to the reference's dIp[l][.][0].  The gradient channels
// dIp[l][.][1..2] are central differences of that plane over the flat index range [w, w*(h-1)) and are recomputed
// where they are consumed (gradAt below = the reference's formula incl. its isfinite guard); absSquaredGrad (pixel
// selector only) is not produced.
//
// One launch builds ALL levels of B frames: a workgroup owns a level-0 tile of 4096 pixels (512x8 for a 512-wide, 4-level pyramid: see PYR_TILE_PX),
// keeps the successive 2x2 reductions in LDS and streams each level out.  HBM traffic per frame:
// read 4 B/px + write 4 B/px * (1 + 1/4 + 1/16 + ...).
#pragma once
#include "common.h"
#include "interp.hpp"

namespace dmv {

typedef float pyr_f4 __attribute__((ext_vector_type(4)));
// level-0 tile of a workgroup: PYR_TILE_PX pixels, 2^tw_log2 wide (PyrGeom::tw_log2, chosen per context: as wide as the image allows — a 512-pixel-wide tile of a
// 512-pixel-wide image is 8 rows x 2 KB = 16 KB of CONTIGUOUS level-0 memory per workgroup, where a 128 x 32 tile wrote 32 separate 512-byte pieces — and as high as the
// 2x2 reductions of the coarser levels need: 2^(levels-1) rows)
#define PYR_TILE_PX 4096
// levels 1.. of a workgroup's tile: successive 2x2 means of the level-0 tile in s_a (the caller's barrier has made it visible), streamed out level by level
// pitch0: row pitch (floats) of the level-0 tile in s_a (the tiled build pads it, see k_build_pyramids_raw)
__device__ __forceinline__ void pyrReduceLevels(float* s_a, float* s_b, const PyrGeom& G, const FrameStore& fs, const int slot, const int x0, const int y0, const int pitch0) {
  float* cur = s_a;
  float* nxt = s_b;
  int sw = 1 << G.tw_log2, sh = PYR_TILE_PX >> G.tw_log2;
  for (int l = 1; l < G.levels; l++) {
    const int nw = sw >> 1, nh = sh >> 1;
    const int pitch = l == 1 ? pitch0 : sw;
    for (int o = threadIdx.x; o < nw * nh; o += 256) {
      const int lx = o % nw, ly = o / nw;
      const int b = 2 * lx + 2 * ly * pitch;
      const float val = 0.25f * (cur[b] + cur[b + 1] + cur[b + pitch] + cur[b + pitch + 1]);
      nxt[ly * nw + lx] = val;
      const int x = (x0 >> l) + lx, y = (y0 >> l) + ly;
      if (x < G.w[l] && y < G.h[l]) fs.own_level(slot, l)[(size_t)y * G.w[l] + x] = val;
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
    sw = nw; sh = nh;
  }
}

__global__ void __launch_bounds__(256) k_build_pyramids(const float* __restrict__ in_base, const size_t in_stride, const PyrGeom G,
                                                         const FrameStore fs, const int* __restrict__ slots, const int single_slot, const unsigned int gen, const int attach) {
  __shared__ float s_a[PYR_TILE_PX];
  __shared__ float s_b[PYR_TILE_PX / 4];
  const int f = blockIdx.y;
  const int slot = slots ? slots[f] : single_slot;
  const float* __restrict__ src = in_base + (size_t)f * in_stride;
  const int tx = blockIdx.x % G.tiles_x, ty = blockIdx.x / G.tiles_x;
  const int w0 = G.w[0], h0 = G.h[0];
  const int TW = 1 << G.tw_log2, rowsPerPass = 1024 >> G.tw_log2;   // 256 threads x 4 pixels = 1024 pixels per pass, four passes per tile
  const int x0 = tx * TW, y0 = ty * (PYR_TILE_PX >> G.tw_log2);
  bool bad = false;
  // level 0: 256 threads x 4 passes x 4 consecutive pixels; all four 16-byte loads of a thread are issued before the first store.
  // The raw image is read once and the level-0 plane is far larger than the caches it would pollute: non-temporal loads / stores
  // (measured: 4.4 -> 5.5 TB/s)
  {
    const int lx = (threadIdx.x & ((TW >> 2) - 1)) * 4, lyb = threadIdx.x >> (G.tw_log2 - 2);   // TW / 4 threads x 4 pixels per row, rowsPerPass rows per pass
    float* __restrict__ dst = fs.own_level(slot, 0);
    float4 v[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int x = x0 + lx, y = y0 + lyb + rowsPerPass * p;
      v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < h0) {
        if (x + 3 < w0 && ((uintptr_t)(src + (size_t)y * w0 + x) & 15) == 0) { const pyr_f4 t = __builtin_nontemporal_load(reinterpret_cast<const pyr_f4*>(src + (size_t)y * w0 + x)); v[p] = make_float4(t[0], t[1], t[2], t[3]); }
        else if (x + 3 < w0) __builtin_memcpy(&v[p], src + (size_t)y * w0 + x, 16);
        else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int k = 0; k < 4; k++) if (x + k < w0) t[k] = src[(size_t)y * w0 + x + k];
          v[p] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int x = x0 + lx, y = y0 + lyb + rowsPerPass * p, ly = lyb + rowsPerPass * p;
      if (y < h0 && !attach) {   // attached in place: level 0 is the caller's image itself
        if (x + 3 < w0 && ((uintptr_t)(dst + (size_t)y * w0 + x) & 15) == 0) { const pyr_f4 t = {v[p].x, v[p].y, v[p].z, v[p].w}; __builtin_nontemporal_store(t, reinterpret_cast<pyr_f4*>(dst + (size_t)y * w0 + x)); }
        else if (x + 3 < w0) __builtin_memcpy(dst + (size_t)y * w0 + x, &v[p], 16);
        else {
          const float t[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
          for (int k = 0; k < 4; k++) if (x + k < w0) dst[(size_t)y * w0 + x + k] = t[k];
        }
      }
      *reinterpret_cast<float4*>(&s_a[ly * TW + lx]) = v[p];
      // NaN fails the comparison too; out-of-image lanes hold zeros
      bad |= !(fabsf(v[p].x) <= 1e30f) || !(fabsf(v[p].y) <= 1e30f) || !(fabsf(v[p].z) <= 1e30f) || !(fabsf(v[p].w) <= 1e30f);
    }
  }
  // stamps of this build (see FrameStore): the barrier the level reduction needs anyway carries the tile's verdict
  if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) fs.bad_gen[slot] = gen;
  if (blockIdx.x == 0 && threadIdx.x == 0) { fs.build_gen[slot] = gen; fs.lvl0[slot] = attach ? src : fs.own_level(slot, 0); fs.tiled0[slot] = 0; }
  pyrReduceLevels(s_a, s_b, G, fs, slot, x0, y0, TW);
}

// Photometric + geometric undistortion of a raw camera image on the upload path:
//   PhotometricUndistorter::processFrame (src/dso/util/Undistort.cpp:214-250): data = G[raw] (* vignetteMapInv), or factor * raw without calibration,
//   Undistort::undistort            (src/dso/util/Undistort.cpp:386-481): bilinear remap of that image through remapX / remapY (xx < 0 -> 0).
// Fused: each output pixel evaluates the photometric value of its four source taps — the same per-pixel arithmetic, bit-identical output.
// T = unsigned char / unsigned short.  remapX == NULL: passthrough (no geometric step, wOrg x hOrg == w x h).
struct UndistortDev {
  int wOrg, hOrg, w, h;
  const float* G;               // 256 (u8) or 65536 (u16) entries, NULL = no photometric calibration
  const float* vignetteMapInv;  // wOrg*hOrg or NULL
  const float *remapX, *remapY; // w*h or NULL
  float factor;
};
template <typename T>
__device__ __forceinline__ float photoAt(const T* __restrict__ raw, const UndistortDev& U, const int i) {
  if (!U.G) return U.factor * raw[i];
  float v = U.G[raw[i]];
  if (U.vignetteMapInv) v *= U.vignetteMapInv[i];
  return v;
}
template <typename T>
__device__ __forceinline__ float undistortPixel(const T* __restrict__ raw, const UndistortDev& U, const int idx) {
  if (!U.remapX) return photoAt(raw, U, idx);
  float xx = U.remapX[idx], yy = U.remapY[idx];
  if (xx < 0) return 0.f;
  const int xxi = (int)xx, yyi = (int)yy;
  xx -= xxi; yy -= yyi;
  const float xxyy = xx * yy;
  const int base = xxi + yyi * U.wOrg;
  return xxyy * photoAt(raw, U, base + 1 + U.wOrg) + (yy - xxyy) * photoAt(raw, U, base + U.wOrg) + (xx - xxyy) * photoAt(raw, U, base + 1) +
         (1 - xx - yy + xxyy) * photoAt(raw, U, base);
}
template <typename T>
__global__ void __launch_bounds__(256) k_undistort(const T* __restrict__ raw, const UndistortDev U, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= U.w * U.h) return;
  out[idx] = undistortPixel(raw, U, idx);
}

// Raw camera images of B frames -> undistorted level 0 + all coarser levels in ONE launch (k_undistort fused into the level-0 pass of k_build_pyramids: the
// fp32 image is never staged, a frame enters as 1 or 2 bytes per pixel).  Same per-pixel arithmetic as the two kernels in sequence, so the same bits.
// TILED: level 0 is written in 8x4-pixel tiles (FrameStore::tiled0; needs w % 8 == 0, h % 4 == 0) for the coarse tracker's gather.  The level-0 values pass through
// LDS for the coarser levels anyway; the tiled store is issued from there, behind the barrier, chunk c of 16 bytes -> lane c: a wavefront writes 1 KB of CONTIGUOUS
// tiled memory per instruction (storing from the computing thread would write 32-byte pieces of 32 different lines).  The LDS rows are padded by 8 floats so that the
// eight 16-byte reads of a tile (4 rows x 2 halves) fall into eight different bank groups.
#define PYR_TILED_PAD 8
template <typename T, bool TILED = false>
__global__ void __launch_bounds__(256) k_build_pyramids_raw(const T* __restrict__ raw_base, const size_t raw_stride, const UndistortDev U, const PyrGeom G,
                                                             const FrameStore fs, const int* __restrict__ slots, const unsigned int gen) {
  __shared__ float s_a[PYR_TILE_PX + (TILED ? PYR_TILED_PAD * 32 : 0)];   // at most 32 tile rows (128-pixel-wide tiles)
  __shared__ float s_b[PYR_TILE_PX / 4];
  const int f = blockIdx.y;
  const int slot = slots[f];
  const T* __restrict__ raw = raw_base + (size_t)f * raw_stride;
  const int tx = blockIdx.x % G.tiles_x, ty = blockIdx.x / G.tiles_x;
  const int w0 = G.w[0], h0 = G.h[0];
  const int TW = 1 << G.tw_log2, rowsPerPass = 1024 >> G.tw_log2;   // 256 threads x 4 pixels = 1024 pixels per pass, four passes per tile
  const int pitch = TILED ? TW + PYR_TILED_PAD : TW;
  const int x0 = tx * TW, y0 = ty * (PYR_TILE_PX >> G.tw_log2);
  bool bad = false;
  float* __restrict__ dst = fs.own_level(slot, 0);
  {
    const int lx = (threadIdx.x & ((TW >> 2) - 1)) * 4, lyb = threadIdx.x >> (G.tw_log2 - 2);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int x = x0 + lx, y = y0 + lyb + rowsPerPass * p, ly = lyb + rowsPerPass * p;
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      if (y < h0) {
        const int i0 = y * w0 + x;
        if (!U.remapX && x + 3 < w0 && ((uintptr_t)(raw + i0) & (4 * sizeof(T) - 1)) == 0) {
          // passthrough geometry: the four raw values of this thread are one aligned 4- / 8-byte load
          T r4[4];
          __builtin_memcpy(r4, __builtin_assume_aligned(raw + i0, 4 * sizeof(T)), 4 * sizeof(T));
          if (!U.G) {
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = U.factor * r4[k];
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = U.G[r4[k]];
            if (U.vignetteMapInv) {
#pragma unroll
              for (int k = 0; k < 4; k++) t[k] *= U.vignetteMapInv[i0 + k];
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) if (x + k < w0) t[k] = undistortPixel(raw, U, i0 + k);
        }
        if (!TILED) {
          if (x + 3 < w0 && ((uintptr_t)(dst + (size_t)y * w0 + x) & 15) == 0) { const pyr_f4 v = {t[0], t[1], t[2], t[3]}; __builtin_nontemporal_store(v, reinterpret_cast<pyr_f4*>(dst + (size_t)y * w0 + x)); }
          else for (int k = 0; k < 4; k++) if (x + k < w0) dst[(size_t)y * w0 + x + k] = t[k];
        }
      }
      *reinterpret_cast<float4*>(&s_a[ly * pitch + lx]) = make_float4(t[0], t[1], t[2], t[3]);
      bad |= !(fabsf(t[0]) <= 1e30f) || !(fabsf(t[1]) <= 1e30f) || !(fabsf(t[2]) <= 1e30f) || !(fabsf(t[3]) <= 1e30f);
    }
  }
  if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) fs.bad_gen[slot] = gen;
  if (blockIdx.x == 0 && threadIdx.x == 0) { fs.build_gen[slot] = gen; fs.lvl0[slot] = fs.own_level(slot, 0); fs.tiled0[slot] = TILED ? 1 : 0; }
  if (TILED) {
    // the workgroup's 2^tw_log2 x (4096 >> tw_log2) pixels = strips of four rows, TW / 8 tiles each; chunk c = (strip, tile, row in tile, half): consecutive chunks are
    // consecutive 16-byte pieces of the tiled plane within a strip
    const int tpr = w0 >> 3;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int c = p * 256 + threadIdx.x;
      const int strip = c >> G.tw_log2, cs = c & (TW - 1);
      const int lx = ((cs >> 3) << 3) + ((cs & 1) << 2), ly = (strip << 2) + ((cs & 7) >> 1);
      const int x = x0 + lx, y = y0 + ly;
      if (x < w0 && y < h0) {   // w0 % 8 == 0, h0 % 4 == 0: a chunk is inside or outside as a whole
        const float4 v = *reinterpret_cast<const float4*>(&s_a[ly * pitch + lx]);
        const pyr_f4 q = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(q, reinterpret_cast<pyr_f4*>(dst + tiled84Offset(x, y, tpr)));
      }
    }
  }
  pyrReduceLevels(s_a, s_b, G, fs, slot, x0, y0, pitch);
}

// Levels 1..3 of a thread's 4 x 8 pixel block (k_build_pyramids_raw_reg, k_build_pyramids_reg): 2 x 4 values of level 1 and two of level 2 in registers, level 3 by the even
// lane of a lane pair.  0.25f * (((a + b) + c) + d) per level, the reference's operand order (HessianBlocks.cpp:150-166).
__device__ __forceinline__ void pyrRegCoarse(const float (&v)[8][4], const PyrGeom& G, const FrameStore& fs, const int slot, const int tx, const int ty, const bool live, const int lane) {
  if (G.levels < 2) return;
  float l1[4][2];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 2; c++) l1[r][c] = 0.25f * (v[2 * r][2 * c] + v[2 * r][2 * c + 1] + v[2 * r + 1][2 * c] + v[2 * r + 1][2 * c + 1]);
  if (live) {
    float* __restrict__ d1 = fs.own_level(slot, 1);
    const int w1 = G.w[1];
#pragma unroll
    for (int r = 0; r < 4; r++) { const float2 q = make_float2(l1[r][0], l1[r][1]); __builtin_memcpy(d1 + (size_t)(4 * ty + r) * w1 + 2 * tx, &q, 8); }
  }
  if (G.levels < 3) return;
  float l2[2];
#pragma unroll
  for (int r = 0; r < 2; r++) l2[r] = 0.25f * (l1[2 * r][0] + l1[2 * r][1] + l1[2 * r + 1][0] + l1[2 * r + 1][1]);
  if (live) {
    float* __restrict__ d2 = fs.own_level(slot, 2);
    const int w2 = G.w[2];
    d2[(size_t)(2 * ty) * w2 + tx] = l2[0];
    d2[(size_t)(2 * ty + 1) * w2 + tx] = l2[1];
  }
  if (G.levels < 4) return;
  // level 3: the 2 x 2 block of level 2 = this lane's two values and those of the lane to its right (w0 % 8 == 0: a lane pair never straddles a block row)
  const float nb0 = __shfl_down(l2[0], 1, 64), nb1 = __shfl_down(l2[1], 1, 64);
  if (live && !(lane & 1)) fs.own_level(slot, 3)[(size_t)ty * G.w[3] + (tx >> 1)] = 0.25f * (l2[0] + nb0 + l2[1] + nb1);
}

// The same build without workgroup coupling — the default of dmvio_hip_frames_from_raw_device_batch for pyramids of at most four levels on images whose sides are
// multiples of 8 (512x512, 640x480, 800x400 ...).  A thread owns a 4 x 8 pixel block: its eight level-0 rows, the 2 x 4 values of level 1 and the two of level 2 they
// reduce to never leave registers, level 3 is formed by the even lane of a lane pair (one cross-lane read).  No barrier and — row-major level 0 — no LDS: a wavefront
// streams 256 x 8 pixels in and four levels out on its own (the LDS-tile kernel above held every store of a tile behind three workgroup barriers and kept 7 workgroups of
// 20 KB per CU; profiles/r04_pyramid_build.md).  Per wave instruction the lanes write consecutive pieces: 16 B (level 0), 8 B (level 1), 4 B (levels 2, 3).
// Same per-pixel arithmetic and the same 0.25f * (((a + b) + c) + d) per level: bit-identical planes.
// TILED: level 0 in 8x4 tiles; the four rows of a strip pass through a wave-private LDS transpose (row stride 68 float4: the sixteen 16-byte reads of a quarter wave fall
// into sixteen different bank groups) so that instruction k of a wave still writes 1 KB of contiguous tiled memory.
#define PYR_REG_ROWF4 68
template <typename T, bool TILED>
__global__ void __launch_bounds__(256) k_build_pyramids_raw_reg(const T* __restrict__ raw_base, const size_t raw_stride, const UndistortDev U, const PyrGeom G,
                                                                 const FrameStore fs, const int* __restrict__ slots, const unsigned int gen) {
  __shared__ float4 s_rows[TILED ? 4 * 4 * PYR_REG_ROWF4 : 1];
  __shared__ unsigned int s_off[TILED ? 256 : 1];
  const int f = blockIdx.y;
  const int slot = slots[f];
  const T* __restrict__ raw = raw_base + (size_t)f * raw_stride;
  const int w0 = G.w[0], h0 = G.h[0];
  const int tpr4 = w0 >> 2, nthr = tpr4 * (h0 >> 3);
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < nthr;
  const int tx = live ? t % tpr4 : 0, ty = live ? t / tpr4 : 0;
  const int x = 4 * tx, y = 8 * ty;
  const int lane = threadIdx.x & 63;
  float v[8][4];
  {
    const int i0 = y * w0 + x;
    if (!U.remapX && ((uintptr_t)(raw + i0) & (4 * sizeof(T) - 1)) == 0) {   // i0 % 4 == 0 and w0 % 8 == 0: the alignment of the frame's base decides, for all rows
      // passthrough geometry: the four raw values of a row are one aligned 4- / 8-byte load; all eight rows are requested before the first conversion
      T r4[8][4];
#pragma unroll
      for (int r = 0; r < 8; r++) __builtin_memcpy(r4[r], __builtin_assume_aligned(raw + i0 + r * w0, 4 * sizeof(T)), 4 * sizeof(T));
      if (!U.G) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
          for (int k = 0; k < 4; k++) v[r][k] = U.factor * r4[r][k];
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
          for (int k = 0; k < 4; k++) v[r][k] = U.G[r4[r][k]];
        if (U.vignetteMapInv) {
#pragma unroll
          for (int r = 0; r < 8; r++) {
            float4 g;
            __builtin_memcpy(&g, U.vignetteMapInv + i0 + r * w0, 16);
            v[r][0] *= g.x; v[r][1] *= g.y; v[r][2] *= g.z; v[r][3] *= g.w;
          }
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) v[r][k] = undistortPixel(raw, U, i0 + r * w0 + k);
    }
  }
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) bad |= !(fabsf(v[r][k]) <= 1e30f);   // NaN fails the comparison too
  if (__any(bad && live) && lane == 0) fs.bad_gen[slot] = gen;
  if (t == 0) { fs.build_gen[slot] = gen; fs.lvl0[slot] = fs.own_level(slot, 0); fs.tiled0[slot] = TILED ? 1 : 0; }
  float* __restrict__ dst0 = fs.own_level(slot, 0);
  if (!TILED) {
    if (live) {
      if (((uintptr_t)dst0 & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) { const pyr_f4 q = {v[r][0], v[r][1], v[r][2], v[r][3]}; __builtin_nontemporal_store(q, reinterpret_cast<pyr_f4*>(dst0 + (size_t)(y + r) * w0 + x)); }
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
          for (int k = 0; k < 4; k++) dst0[(size_t)(y + r) * w0 + x + k] = v[r][k];
      }
    }
  } else {
    float4* __restrict__ wrow = s_rows + (threadIdx.x >> 6) * (4 * PYR_REG_ROWF4);
    unsigned int* __restrict__ woff = s_off + (threadIdx.x & ~63);
    const int tpr = w0 >> 3;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      // offset of this lane's half row 0 inside its tile of strip s (rows add 8 floats each); 0xffffffff: nothing to write
      woff[lane] = live ? tiled84Offset(x, y + 4 * s, tpr) : 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 4; r++) wrow[r * PYR_REG_ROWF4 + lane] = make_float4(v[4 * s + r][0], v[4 * s + r][1], v[4 * s + r][2], v[4 * s + r][3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // chunk c of the wave's 32 tiles: tile c / 8, row (c % 8) / 2, half c % 2 — consecutive chunks are consecutive 16-byte pieces of the tiled plane
        const int c = 64 * k + lane;
        const int r = (c & 7) >> 1, src = ((c >> 3) << 1) + (c & 1);
        const unsigned int o = woff[src];
        const float4 q4 = wrow[r * PYR_REG_ROWF4 + src];
        if (o != 0xffffffffu) { const pyr_f4 q = {q4.x, q4.y, q4.z, q4.w}; __builtin_nontemporal_store(q, reinterpret_cast<pyr_f4*>(dst0 + o + 8 * r)); }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  pyrRegCoarse(v, G, fs, slot, tx, ty, live, lane);
}

// fp32 images resident on the device: the same wave-autonomous build.  ATTACH: level 0 is the caller's image itself (nothing stored for it) — the case the library launches
// this kernel for (dmvio_hip_frames_attach_device_batch: 1.075 -> 1.023 ms per 4096 frames of 512x512); with level 0 copied the LDS-tile build stays 4 % ahead
// (1.74 vs 1.81 ms) and keeps that path.  Rows are read with non-temporal 16-byte loads where the frame's base allows.
template <bool ATTACH>
__global__ void __launch_bounds__(256) k_build_pyramids_reg(const float* __restrict__ in_base, const size_t in_stride, const PyrGeom G, const FrameStore fs,
                                                             const int* __restrict__ slots, const int single_slot, const unsigned int gen) {
  const int f = blockIdx.y;
  const int slot = slots ? slots[f] : single_slot;
  const float* __restrict__ src = in_base + (size_t)f * in_stride;
  const int w0 = G.w[0], h0 = G.h[0];
  const int tpr4 = w0 >> 2, nthr = tpr4 * (h0 >> 3);
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < nthr;
  const int tx = live ? t % tpr4 : 0, ty = live ? t / tpr4 : 0;
  const int x = 4 * tx, y = 8 * ty;
  const int lane = threadIdx.x & 63;
  float v[8][4];
  const float* __restrict__ p0 = src + (size_t)y * w0 + x;
  if (((uintptr_t)src & 15) == 0) {   // x % 4 == 0, w0 % 8 == 0: the frame's base decides for every row
#pragma unroll
    for (int r = 0; r < 8; r++) { const pyr_f4 q = __builtin_nontemporal_load(reinterpret_cast<const pyr_f4*>(p0 + (size_t)r * w0)); v[r][0] = q[0]; v[r][1] = q[1]; v[r][2] = q[2]; v[r][3] = q[3]; }
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_memcpy(v[r], p0 + (size_t)r * w0, 16);
  }
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) bad |= !(fabsf(v[r][k]) <= 1e30f);
  if (__any(bad && live) && lane == 0) fs.bad_gen[slot] = gen;
  if (t == 0) { fs.build_gen[slot] = gen; fs.lvl0[slot] = ATTACH ? src : fs.own_level(slot, 0); fs.tiled0[slot] = 0; }
  if (!ATTACH && live) {
    float* __restrict__ dst0 = fs.own_level(slot, 0);
    if (((uintptr_t)dst0 & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 8; r++) { const pyr_f4 q = {v[r][0], v[r][1], v[r][2], v[r][3]}; __builtin_nontemporal_store(q, reinterpret_cast<pyr_f4*>(dst0 + (size_t)(y + r) * w0 + x)); }
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++) __builtin_memcpy(dst0 + (size_t)(y + r) * w0 + x, v[r], 16);
    }
  }
  pyrRegCoarse(v, G, fs, slot, tx, ty, live, lane);
}

// level 0 of a slot out of the 8x4-tile layout into a row-major plane (consumers other than the coarse tracker's batch kernel: dmv_ensure_row_major)
__global__ void __launch_bounds__(256) k_untile_level0(const float* __restrict__ tiled, const int w, const int h, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < w * h) out[idx] = tiled[tiled84Offset(idx % w, idx / w, w >> 3)];
}

// level plane -> the reference's Eigen::Vector3f AoS (I, dx, dy)   (parity tests / debug download)
__global__ void __launch_bounds__(256) k_level_to_f3(const float* __restrict__ I, const int w, const int h, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < w * h) {
    const float2 g = gradAt(I, w, h, idx % w, idx / w);
    out[3 * idx + 0] = I[idx]; out[3 * idx + 1] = g.x; out[3 * idx + 2] = g.y;
  }
}

// absSquaredGrad[lvl] of FrameHessian::makeImages (HessianBlocks.cpp:169-189): dx*dx + dy*dy, times (B[c+1] - B[c])^2 of CalibHessian::getBGradOnly
// (HessianBlocks.h:394-400) when a response table is given (setting_gammaWeightsPixelSelect == 1).  Rows 0 and h-1, which the reference never writes, are zero.
__global__ void __launch_bounds__(256) k_abs_squared_grad(const float* __restrict__ I, const int w, const int h, const float* __restrict__ B, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= w * h) return;
  float v = 0.f;
  if (idx >= w && idx < w * (h - 1)) {
    const float2 g = gradAt(I, w, h, idx % w, idx / w);
    v = g.x * g.x + g.y * g.y;
    if (B) {
      int c = (int)(I[idx] + 0.5f);
      if (c < 5) c = 5;
      if (c > 250) c = 250;
      const float gw = B[c + 1] - B[c];
      v *= gw * gw;
    }
  }
  out[idx] = v;
}

}  // namespace dmv
