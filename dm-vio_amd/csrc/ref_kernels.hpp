// Reference-template construction of the coarse tracker on the device.
// Replaces CoarseTracker::makeCoarseDepthL0 (src/dso/FullSystem/CoarseTracker.cpp:138-295):
//   scatter of weighted inverse depths (:144-161), 2x2 sum-pool down the pyramid (:164-189),
//   1-pixel dilation — diagonal neighbours on levels 0/1 (:193-220), 4-neighbourhood above (:224-245) —
//   normalisation and row-major compaction into pc_u/pc_v/pc_idepth/pc_color (:249-293).
// The compaction is an ORDERED stream compaction (tile counts -> exclusive scan -> ranked write), so the
// template point order — which decides the every-32nd-point flow-indicator sample of calcRes — is the
// reference's row-major order.
#pragma once
#include "common.h"

namespace dmv {

struct RefLevels {
  int levels;
  int w[DMV_MAX_LEVELS], h[DMV_MAX_LEVELS];
  size_t off[DMV_MAX_LEVELS];      // offset of the level inside the concatenated per-pixel float planes
  int tile_off[DMV_MAX_LEVELS + 1];  // first tile of each level (tiles of 256 consecutive pixels)
  size_t total;                    // total pixels over all levels
};

__global__ void __launch_bounds__(256) k_ref_scatter(const int n, const float* __restrict__ u, const float* __restrict__ v,
                                                      const float* __restrict__ idepth, const float* __restrict__ hdiF,
                                                      float* __restrict__ id0, float* __restrict__ ws0, const int w0, const int h0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
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
  } else {
    return pooledSum<L - 1>(p0, w0, 2 * x, 2 * y) + pooledSum<L - 1>(p0, w0, 2 * x + 1, 2 * y) +
           pooledSum<L - 1>(p0, w0, 2 * x, 2 * y + 1) + pooledSum<L - 1>(p0, w0, 2 * x + 1, 2 * y + 1);
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

// normalise; returns whether pixel li of level lvl becomes a template point, and its record
__device__ __forceinline__ bool refPixel(const RefLevels& R, const int lvl, const int li, const float* __restrict__ idp2,
                                         const float* __restrict__ wsp2, const float4* __restrict__ refImg, float4& rec, float& idOut) {
  const int wl = R.w[lvl], hl = R.h[lvl];
  const int x = li % wl, y = li / wl;
  const float idv = idp2[R.off[lvl] + li];
  idOut = idv;
  if (!(x >= 2 && x < wl - 2 && y >= 2 && y < hl - 2)) return false;
  const float wsv = wsp2[R.off[lvl] + li];
  if (wsv > 0) {
    const float idn = idv / wsv;
    const float color = refImg[li].x;
    if (!isfinite(color) || !(idn > 0)) { idOut = -1; return false; }
    idOut = idn;
    rec = make_float4((float)x, (float)y, idn, color);
    return true;
  }
  idOut = -1;
  return false;
}

__device__ __forceinline__ void tileToLevel(const RefLevels& R, const int tile, int& lvl, int& li0) {
  lvl = 0;
  while (lvl + 1 < R.levels && tile >= R.tile_off[lvl + 1]) lvl++;
  li0 = (tile - R.tile_off[lvl]) * 256;
}

// pass 1: per-tile counts.  One workgroup (256 threads) per tile of 256 consecutive pixels.
__global__ void __launch_bounds__(256) k_ref_count(const RefLevels R, const float* __restrict__ idp2, const float* __restrict__ wsp2,
                                                    const FrameStore fs, const int ref_slot, int* __restrict__ tile_count) {
  int lvl, li0;
  tileToLevel(R, blockIdx.x, lvl, li0);
  const int li = li0 + threadIdx.x;
  bool flag = false;
  if (li < R.w[lvl] * R.h[lvl]) {
    float4 rec; float idn;
    flag = refPixel(R, lvl, li, idp2, wsp2, fs.level(ref_slot, lvl), rec, idn);
  }
  __shared__ int s_cnt[4];
  const unsigned long long m = __ballot(flag);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// pass 2: exclusive scan of the tile counts of each level (one workgroup per level), pc_n[lvl]
__global__ void __launch_bounds__(1024) k_ref_scan(const RefLevels R, const int* __restrict__ tile_count, int* __restrict__ tile_base, int* __restrict__ pc_n) {
  const int lvl = blockIdx.x;
  const int t0 = R.tile_off[lvl], t1 = R.tile_off[lvl + 1];
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = t0; base < t1; base += 1024) {
    const int t = base + threadIdx.x;
    const int c = (t < t1) ? tile_count[t] : 0;
    // inclusive scan inside the wave
    int v = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d, 64);
      if ((threadIdx.x & 63) >= d) v += o;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) woff += s_wave[k];
    const int carry = s_carry;
    if (t < t1) tile_base[t] = carry + woff + v - c;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + woff + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) pc_n[lvl] = s_carry;
}

// pass 3: ranked write of the template records + the dense normalised idepth map (debugPlotIDepthMap)
__global__ void __launch_bounds__(256) k_ref_write(const RefLevels R, const float* __restrict__ idp2, const float* __restrict__ wsp2,
                                                    const FrameStore fs, const int ref_slot, const int* __restrict__ tile_base,
                                                    float4* const* __restrict__ pc, float* __restrict__ idepth_dense) {
  int lvl, li0;
  tileToLevel(R, blockIdx.x, lvl, li0);
  const int li = li0 + threadIdx.x;
  bool flag = false;
  float4 rec = make_float4(0, 0, 0, 0);
  float idn = -1;
  const bool inside = li < R.w[lvl] * R.h[lvl];
  if (inside) flag = refPixel(R, lvl, li, idp2, wsp2, fs.level(ref_slot, lvl), rec, idn);
  __shared__ int s_cnt[4];
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = __popcll(m);
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < wave; k++) woff += s_cnt[k];
  const int rank = __popcll(m & ((1ull << lane) - 1ull));
  if (flag) pc[lvl][tile_base[blockIdx.x] + woff + rank] = rec;
  if (inside) idepth_dense[R.off[lvl] + li] = idn;
}

}  // namespace dmv
