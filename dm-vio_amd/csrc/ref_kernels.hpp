// Reference-template construction of the coarse tracker on the device.
// Replaces CoarseTracker::makeCoarseDepthL0 (src/dso/FullSystem/CoarseTracker.cpp:138-295):
//   scatter of weighted inverse depths (:144-161), 2x2 sum-pool down the pyramid (:164-189),
//   1-pixel dilation — diagonal neighbours on levels 0/1 (:193-220), 4-neighbourhood above (:224-245) —
//   normalisation and row-major compaction into pc_u/pc_v/pc_idepth/pc_color (:249-293).
// The compaction is an ORDERED (deterministic) stream compaction: counts -> exclusive scan -> ranked write.
// Template points are emitted in TILE order (8x8-pixel tiles, Z-ordered inside 16x16 blocks, blocks row-major) rather
// than the reference's row-major order: neighbouring lanes then tap neighbouring pixels, so the dilated 3x3 clusters
// around each active point share cache lines.  Sums do not depend on the order; the one order-dependent rule of the
// reference — flow indicators are sampled at every 32nd point of the ROW-MAJOR list (CoarseTracker.cpp:416) — is kept
// exactly: each point's row-major rank is reconstructed from per-(row, tile) counts and stored as a bit mask.
#pragma once
#include "common.h"

namespace dmv {

struct RefLevels {
  int levels;
  int w[DMV_MAX_LEVELS], h[DMV_MAX_LEVELS];
  size_t off[DMV_MAX_LEVELS];      // offset of the level inside the concatenated per-pixel float planes
  int tile_off[DMV_MAX_LEVELS + 1];  // first 16x16 block of each level
  int blocks_x[DMV_MAX_LEVELS];      // 16x16 blocks per row of the level
  int seg_x0;                        // level 0: 8-pixel segments per row (row-major rank reconstruction)
  int seg_x[DMV_MAX_LEVELS];         // 8-pixel segments per row of every level, and where the level's (row, segment) counts start in seg_count
  int seg_off[DMV_MAX_LEVELS + 1];
  int order;                         // 0: the template is stored in tile order; 1: in the reference's row-major order (dmvio_hip_tracker_set_template_order)
  size_t total;                    // total pixels over all levels
};

// The reference adds the points of a pixel one after the other (CoarseTracker.cpp:151-165).  Two float atomics on one pixel commute (0 + a + b); a third
// would make the sum depend on the arrival order, so the host ranks the points of every pixel by index (`rank`, NULL when no pixel holds more than two) and
// launches ranks [0,1] together and every further rank as its own launch behind them: sequential order, no data race.
__global__ void __launch_bounds__(256) k_ref_scatter(const int n, const float* __restrict__ u, const float* __restrict__ v,
                                                      const float* __restrict__ idepth, const float* __restrict__ hdiF,
                                                      float* __restrict__ id0, float* __restrict__ ws0, const int w0, const int h0,
                                                      const unsigned char* __restrict__ rank, const int rank_lo, const int rank_hi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (rank) { const int r = rank[i]; if (r < rank_lo || r > rank_hi) return; }
  const int ui = (int)(u[i] + 0.5f);
  const int vi = (int)(v[i] + 0.5f);
  if (ui < 0 || vi < 0 || ui >= w0 || vi >= h0) return;  // the reference would write out of bounds here
  const float weight = sqrtf((float)(1e-3 / ((double)hdiF[i] + 1e-12)));
  atomicAdd(&id0[ui + w0 * vi], idepth[i] * weight);
  atomicAdd(&ws0[ui + w0 * vi], weight);
}

// nested 2x2 sums in the reference's operand order: ((a + b) + c) + d, level by level
template <int L>
__device__ __forceinline__ float pooledSum(const float* __restrict__ p0, const int w0, const int x, const int y) {
  if constexpr (L == 0) {
    return p0[x + y * w0];
  } else if constexpr (L <= 2) {
    return pooledSum<L - 1>(p0, w0, 2 * x, 2 * y) + pooledSum<L - 1>(p0, w0, 2 * x + 1, 2 * y) +
           pooledSum<L - 1>(p0, w0, 2 * x, 2 * y + 1) + pooledSum<L - 1>(p0, w0, 2 * x + 1, 2 * y + 1);
  } else {
    // from level 3 on the four children are visited by a ROLLED loop (same operand order): fully unrolled, level 5 issued its 1024 loads at once and took 256 VGPRs +
    // 62 AGPRs for a kernel that runs once per keyframe
    float t = 0.f;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
      const float v = pooledSum<L - 1>(p0, w0, 2 * x + (q & 1), 2 * y + (q >> 1));
      t = q == 0 ? v : t + v;
    }
    return t;
  }
}

// all levels >= 1 in one launch; idx runs over the concatenated pixels of levels 1..L-1
__global__ void __launch_bounds__(256) k_ref_pool(const RefLevels R, float* __restrict__ idp, float* __restrict__ wsp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x + R.off[1];
  if (idx >= R.total) return;
  int lvl = 1;
  while (lvl + 1 < R.levels && idx >= R.off[lvl + 1]) lvl++;
  const int li = (int)(idx - R.off[lvl]);
  const int x = li % R.w[lvl], y = li / R.w[lvl];
  float a, b;
  switch (lvl) {
    case 1: a = pooledSum<1>(idp, R.w[0], x, y); b = pooledSum<1>(wsp, R.w[0], x, y); break;
    case 2: a = pooledSum<2>(idp, R.w[0], x, y); b = pooledSum<2>(wsp, R.w[0], x, y); break;
    case 3: a = pooledSum<3>(idp, R.w[0], x, y); b = pooledSum<3>(wsp, R.w[0], x, y); break;
    case 4: a = pooledSum<4>(idp, R.w[0], x, y); b = pooledSum<4>(wsp, R.w[0], x, y); break;
    default: a = pooledSum<5>(idp, R.w[0], x, y); b = pooledSum<5>(wsp, R.w[0], x, y); break;
  }
  idp[idx] = a;
  wsp[idx] = b;
}

// dilation: reads the un-dilated planes (the reference's weightSums_bak + never-overwritten idepth
// entries), writes new planes — race free by construction.
__global__ void __launch_bounds__(256) k_ref_dilate(const RefLevels R, const float* __restrict__ idp, const float* __restrict__ wsp,
                                                     float* __restrict__ idp2, float* __restrict__ wsp2) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R.total) return;
  int lvl = 0;
  while (lvl + 1 < R.levels && idx >= R.off[lvl + 1]) lvl++;
  const int wl = R.w[lvl], hl = R.h[lvl];
  const int i = (int)(idx - R.off[lvl]);
  const float* id = idp + R.off[lvl];
  const float* ws = wsp + R.off[lvl];
  float oid = id[i], ows = ws[i];
  const int wh = wl * hl - wl;
  if (i >= wl + 1 && i < wh - 1 && ows <= 0) {
    float sum = 0, num = 0, numn = 0;
    int n0, n1, n2, n3;
    if (lvl < 2) { n0 = i + 1 + wl; n1 = i - 1 - wl; n2 = i + wl - 1; n3 = i - wl + 1; }
    else         { n0 = i + 1;      n1 = i - 1;      n2 = i + wl;     n3 = i - wl; }
    if (ws[n0] > 0) { sum += id[n0]; num += ws[n0]; numn++; }
    if (ws[n1] > 0) { sum += id[n1]; num += ws[n1]; numn++; }
    if (ws[n2] > 0) { sum += id[n2]; num += ws[n2]; numn++; }
    if (ws[n3] > 0) { sum += id[n3]; num += ws[n3]; numn++; }
    if (numn > 0) { oid = sum / numn; ows = num / numn; }
  }
  idp2[idx] = oid;
  wsp2[idx] = ows;
}

// normalise; returns whether pixel (x,y) of level lvl becomes a template point, and its record
__device__ __forceinline__ bool refPixel(const RefLevels& R, const int lvl, const int x, const int y, const float* __restrict__ idp2,
                                         const float* __restrict__ wsp2, const float* __restrict__ refI, float4& rec, float& idOut) {
  const int wl = R.w[lvl], hl = R.h[lvl];
  const int li = x + y * wl;
  const float idv = idp2[R.off[lvl] + li];
  idOut = idv;
  if (!(x >= 2 && x < wl - 2 && y >= 2 && y < hl - 2)) return false;
  const float wsv = wsp2[R.off[lvl] + li];
  if (wsv > 0) {
    const float idn = idv / wsv;
    const float color = refI[li];
    if (!isfinite(color) || !(idn > 0)) { idOut = -1; return false; }
    idOut = idn;
    rec = make_float4((float)x, (float)y, idn, color);
    return true;
  }
  idOut = -1;
  return false;
}

// workgroup = one 16x16 pixel block; wave k = 8x8 tile (k&1, k>>1) of the block; lane = (lx = l&7, ly = l>>3)
__device__ __forceinline__ void blockToPixel(const RefLevels& R, const int blk, int& lvl, int& x, int& y) {
  lvl = 0;
  while (lvl + 1 < R.levels && blk >= R.tile_off[lvl + 1]) lvl++;
  const int b = blk - R.tile_off[lvl];
  const int bx = b % R.blocks_x[lvl], by = b / R.blocks_x[lvl];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  x = bx * 16 + (wave & 1) * 8 + (lane & 7);
  y = by * 16 + (wave >> 1) * 8 + (lane >> 3);
}

// pass 1: per-block counts (+ per (row, 8-px segment) counts on level 0)
__global__ void __launch_bounds__(256) k_ref_count(const RefLevels R, const float* __restrict__ idp2, const float* __restrict__ wsp2,
                                                    const FrameStore fs, const int ref_slot, int* __restrict__ blk_count, int* __restrict__ seg_count) {
  int lvl, x, y;
  blockToPixel(R, blockIdx.x, lvl, x, y);
  bool flag = false;
  const bool inside = x < R.w[lvl] && y < R.h[lvl];
  if (inside) {
    float4 rec; float idn;
    flag = refPixel(R, lvl, x, y, idp2, wsp2, fs.level(ref_slot, lvl), rec, idn);
  }
  __shared__ int s_cnt[4];
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63;
  if (lane == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  if (inside && (lane & 7) == 0) seg_count[R.seg_off[lvl] + y * R.seg_x[lvl] + (x >> 3)] = __popcll((m >> (lane & 56)) & 0xffull);
  __syncthreads();
  if (threadIdx.x == 0) blk_count[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// block-wide exclusive scan helper over 1024 threads (value per thread) ; returns exclusive prefix, total via s_total
__device__ __forceinline__ int scan1024(const int c, int* s_wave, int& total) {
  int v = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v += o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = v;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int k = 0; k < 16; k++) { const int sv = s_wave[k]; if (k < (int)(threadIdx.x >> 6)) woff += sv; tot += sv; }
  total = tot;
  return woff + v - c;
}

// pass 2: blockIdx.x < levels: exclusive scan of the block counts of that level -> blk_base, pc_n[lvl].
//         blockIdx.x >= levels: row-major rank tables of level blockIdx.x - levels: seg_count -> exclusive prefix in row-major order (in place).
__global__ void __launch_bounds__(1024) k_ref_scan(const RefLevels R, const int* __restrict__ blk_count, int* __restrict__ blk_base,
                                                    int* __restrict__ pc_n, int* __restrict__ seg_count) {
  __shared__ int s_wave[16];
  int carry = 0;
  if ((int)blockIdx.x < R.levels) {
    const int lvl = blockIdx.x;
    const int t0 = R.tile_off[lvl], t1 = R.tile_off[lvl + 1];
    for (int base = t0; base < t1; base += 1024) {
      const int t = base + threadIdx.x;
      const int c = (t < t1) ? blk_count[t] : 0;
      int total;
      const int ex = scan1024(c, s_wave, total);
      if (t < t1) blk_base[t] = carry + ex;
      carry += total;
    }
    if (threadIdx.x == 0) pc_n[lvl] = carry;
  } else {
    const int lvl = blockIdx.x - R.levels;
    const int n = R.h[lvl] * R.seg_x[lvl];  // row-major (y, segment) order == raster order of the segments
    int* sc = seg_count + R.seg_off[lvl];
    for (int base = 0; base < n; base += 1024) {
      const int t = base + threadIdx.x;
      const int c = (t < n) ? sc[t] : 0;
      int total;
      const int ex = scan1024(c, s_wave, total);
      if (t < n) sc[t] = carry + ex;
      carry += total;
    }
  }
}

// pass 3: ranked write of the template records in tile order, the dense normalised idepth map (debugPlotIDepthMap)
// and, on level 0, the flow-sample bit of every entry whose ROW-MAJOR rank is a multiple of 32.
__global__ void __launch_bounds__(256) k_ref_write(const RefLevels R, const float* __restrict__ idp2, const float* __restrict__ wsp2,
                                                    const FrameStore fs, const int ref_slot, const int* __restrict__ blk_base,
                                                    const int* __restrict__ seg_prefix, float4* const* __restrict__ pc,
                                                    float* __restrict__ idepth_dense, unsigned long long* __restrict__ flow_mask) {
  int lvl, x, y;
  blockToPixel(R, blockIdx.x, lvl, x, y);
  bool flag = false;
  float4 rec = make_float4(0, 0, 0, 0);
  float idn = -1;
  const bool inside = x < R.w[lvl] && y < R.h[lvl];
  if (inside) flag = refPixel(R, lvl, x, y, idp2, wsp2, fs.level(ref_slot, lvl), rec, idn);
  __shared__ int s_cnt[4];
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = __popcll(m);
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < wave; k++) woff += s_cnt[k];
  if (flag) {
    const unsigned long long rowbits = (m >> (lane & 56)) & 0xffull;
    const int raster = seg_prefix[R.seg_off[lvl] + y * R.seg_x[lvl] + (x >> 3)] + __popcll(rowbits & ((1ull << (lane & 7)) - 1ull));   // rank in the reference's row-major list
    const int pos = R.order ? raster : blk_base[blockIdx.x] + woff + __popcll(m & ((1ull << lane) - 1ull));
    pc[lvl][pos] = rec;
    if (lvl == 0 && (raster & 31) == 0) atomicOr(&flow_mask[pos >> 6], 1ull << (pos & 63));
  }
  if (inside) idepth_dense[R.off[lvl] + x + y * R.w[lvl]] = idn;
}

}  // namespace dmv
