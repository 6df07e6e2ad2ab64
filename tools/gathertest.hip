// Diagnostic: what bounds the bilinear 4x4-footprint gather of the coarse tracker — L2->L1 line fills, tag look-ups or load
// instruction count?  Emulates k_track_lm's access pattern (one workgroup per frame, template points in 8x8-tile order, own
// image per frame) for several image layouts / load shapes and prints ns per point-evaluation per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

enum { ROWMAJOR_VEC = 0, ROWMAJOR_SCALAR = 1, TILED84_SCALAR = 2, ROWPAIR_VEC = 3, TILED84_ROWVEC = 4, ROWMAJOR_VEC_X2 = 5, ROWMAJOR_VEC_NT = 6, ROWMAJOR_VEC_X4 = 7 };

__device__ __forceinline__ int tiled84(int x, int y, int tilesPerRow) { return (((y >> 2) * tilesPerRow + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7); }

template <int MODE>
__global__ void __launch_bounds__(256, 4) k_gather(const float* __restrict__ imgs, size_t img_stride, const float2* __restrict__ pts, int n, int w, int h,
                                                   int passes, float* __restrict__ out) {
  const float* __restrict__ img = imgs + (size_t)blockIdx.x * img_stride;
  float acc = 0.f;
  for (int p = 0; p < passes; p++) {
    const float sx = 0.37f * p, sy = 0.21f * p;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float2 P = pts[i];
      const float x = fminf(fmaxf(P.x + sx, 2.5f), w - 3.5f), y = fminf(fmaxf(P.y + sy, 2.5f), h - 3.5f);
      const int ix = (int)x, iy = (int)y;
      if (MODE == ROWMAJOR_VEC_X2 || MODE == ROWMAJOR_VEC_X4) {
        // U points of one lane in flight at once (memory-level parallelism test): handled in the strided loop below
        continue;
      }
      if (MODE == ROWMAJOR_VEC_NT) {
        const float* bp = img + ix + iy * w;
        typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
        typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
        const f2 A = __builtin_nontemporal_load((const f2*)(bp - w)); const f4 Bv = __builtin_nontemporal_load((const f4*)(bp - 1));
        const f4 C = __builtin_nontemporal_load((const f4*)(bp + w - 1)); const f2 D = __builtin_nontemporal_load((const f2*)(bp + 2 * w));
        acc += A.x + A.y + Bv.x + Bv.y + Bv.z + Bv.w + C.x + C.y + C.z + C.w + D.x + D.y;
        continue;
      }
      float s = 0.f;
      if (MODE == ROWMAJOR_VEC) {
        const float* bp = img + ix + iy * w;
        float2 A, D; float4 B, C;
        __builtin_memcpy(&A, bp - w, 8); __builtin_memcpy(&B, bp - 1, 16); __builtin_memcpy(&C, bp + w - 1, 16); __builtin_memcpy(&D, bp + 2 * w, 8);
        s = A.x + A.y + B.x + B.y + B.z + B.w + C.x + C.y + C.z + C.w + D.x + D.y;
      } else if (MODE == ROWMAJOR_SCALAR) {
        const float* bp = img + ix + iy * w;
        s = bp[-w] + bp[-w + 1] + bp[-1] + bp[0] + bp[1] + bp[2] + bp[w - 1] + bp[w] + bp[w + 1] + bp[w + 2] + bp[2 * w] + bp[2 * w + 1];
      } else if (MODE == TILED84_SCALAR) {
        const int tpr = w >> 3;
        int co[4], ro[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int c = ix - 1 + k, r = iy - 1 + k; co[k] = ((c >> 3) << 5) + (c & 7); ro[k] = ((r >> 2) * tpr << 5) + ((r & 3) << 3); }
        s = img[ro[0] + co[1]] + img[ro[0] + co[2]] + img[ro[1] + co[0]] + img[ro[1] + co[1]] + img[ro[1] + co[2]] + img[ro[1] + co[3]] +
            img[ro[2] + co[0]] + img[ro[2] + co[1]] + img[ro[2] + co[2]] + img[ro[2] + co[3]] + img[ro[3] + co[1]] + img[ro[3] + co[2]];
      } else if (MODE == ROWPAIR_VEC) {
        // 16 px x 2 rows per 128-B tile; vector loads (chunk-crossing ignored: timing only)
        const int tpr = w >> 4;
        auto idx = [&](int c, int r) { return (((r >> 1) * tpr + (c >> 4)) << 5) + ((r & 1) << 4) + (c & 15); };
        float2 A, D; float4 B, C;
        __builtin_memcpy(&A, img + idx(ix, iy - 1), 8); __builtin_memcpy(&B, img + idx(ix - 1, iy), 16);
        __builtin_memcpy(&C, img + idx(ix - 1, iy + 1), 16); __builtin_memcpy(&D, img + idx(ix, iy + 2), 8);
        s = A.x + A.y + B.x + B.y + B.z + B.w + C.x + C.y + C.z + C.w + D.x + D.y;
      } else {
        // 8x4 tiles, one vector load per row (tile-column crossing ignored: timing only) — lower bound for a tiled layout
        const int tpr = w >> 3;
        float2 A, D; float4 B, C;
        __builtin_memcpy(&A, img + tiled84(ix, iy - 1, tpr), 8); __builtin_memcpy(&B, img + tiled84(ix - 1, iy, tpr), 16);
        __builtin_memcpy(&C, img + tiled84(ix - 1, iy + 1, tpr), 16); __builtin_memcpy(&D, img + tiled84(ix, iy + 2, tpr), 8);
        s = A.x + A.y + B.x + B.y + B.z + B.w + C.x + C.y + C.z + C.w + D.x + D.y;
      }
      acc += s;
    }
    if (MODE == ROWMAJOR_VEC_X2 || MODE == ROWMAJOR_VEC_X4) {
      constexpr int U = MODE == ROWMAJOR_VEC_X2 ? 2 : 4;
      for (int i = threadIdx.x; i < n; i += 256 * U) {
        float2 A[U], D[U]; float4 Bq[U], C[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int j = min(i + 256 * u, n - 1);
          const float2 P = pts[j];
          const float x = fminf(fmaxf(P.x + sx, 2.5f), w - 3.5f), y = fminf(fmaxf(P.y + sy, 2.5f), h - 3.5f);
          const float* bp = img + (int)x + (int)y * w;
          __builtin_memcpy(&A[u], bp - w, 8); __builtin_memcpy(&Bq[u], bp - 1, 16); __builtin_memcpy(&C[u], bp + w - 1, 16); __builtin_memcpy(&D[u], bp + 2 * w, 8);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += A[u].x + A[u].y + Bq[u].x + Bq[u].y + Bq[u].z + Bq[u].w + C[u].x + C[u].y + C[u].z + C[u].w + D[u].x + D[u].y;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static float run(const float* imgs, size_t stride, const float2* pts, int n, int w, int h, int passes, int B, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_gather<MODE><<<B, 256>>>(imgs, stride, pts, n, w, h, passes, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) k_gather<MODE><<<B, 256>>>(imgs, stride, pts, n, w, h, passes, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int B = 1024, passes = 4;
  std::mt19937 rng(7);
  for (int lvl = 0; lvl < 4; lvl++) {
    const int w = 512 >> lvl, h = 512 >> lvl;
    // 2000 seeds (at level-0 resolution) + diagonal dilation at levels 0/1, 4-neighbour at 2/3, deduplicated, 8x8-tile order
    std::vector<std::pair<int, int>> px;
    std::vector<char> occ((size_t)w * h, 0);
    for (int k = 0; k < 2000; k++) {
      const int x = (4 + rng() % 504) >> lvl, y = (4 + rng() % 504) >> lvl;
      const int dxs[5] = {0, 1, -1, 1, -1}, dys[5] = {0, 1, -1, -1, 1}, dx4[5] = {0, 1, -1, 0, 0}, dy4[5] = {0, 0, 0, 1, -1};
      for (int q = 0; q < 5; q++) {
        const int xx = x + (lvl < 2 ? dxs[q] : dx4[q]), yy = y + (lvl < 2 ? dys[q] : dy4[q]);
        if (xx < 3 || yy < 3 || xx >= w - 3 || yy >= h - 3 || occ[yy * w + xx]) continue;
        occ[yy * w + xx] = 1; px.push_back({xx, yy});
      }
    }
    std::sort(px.begin(), px.end(), [&](auto a, auto b) {
      auto key = [&](std::pair<int, int> p) { return ((long)(p.second >> 4) << 40) | ((long)(p.first >> 4) << 20) | ((long)((p.second >> 3) & 1) << 12) | ((long)((p.first >> 3) & 1) << 11) | ((p.second & 7) << 3) | (p.first & 7); };
      return key(a) < key(b); });
    const int n = (int)px.size();
    std::vector<float2> hp(n);
    for (int i = 0; i < n; i++) hp[i] = make_float2(px[i].first + 0.3f, px[i].second + 0.6f);
    float2* dp; hipMalloc(&dp, sizeof(float2) * n); hipMemcpy(dp, hp.data(), sizeof(float2) * n, hipMemcpyHostToDevice);
    const size_t stride = (size_t)w * h + 4096;
    float* imgs; hipMalloc(&imgs, sizeof(float) * stride * B); hipMemset(imgs, 0, sizeof(float) * stride * B);
    float* out; hipMalloc(&out, sizeof(float) * 256 * B);
    const double evals = (double)B * n * passes;
    const float t0 = run<ROWMAJOR_VEC>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t1 = run<ROWMAJOR_SCALAR>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t2 = run<TILED84_SCALAR>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t3 = run<ROWPAIR_VEC>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t4 = run<TILED84_ROWVEC>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t5 = run<ROWMAJOR_VEC_X2>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t6 = run<ROWMAJOR_VEC_NT>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t7 = run<ROWMAJOR_VEC_X4>(imgs, stride, dp, n, w, h, passes, B, out);
    const float t8 = run<ROWMAJOR_VEC>(imgs, stride, dp, n, w, h, passes, B / 2, out) * 2;   // 2 waves per SIMD
    const float t9 = run<ROWMAJOR_VEC_X4>(imgs, stride, dp, n, w, h, passes, B / 2, out) * 2;
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 * 256.0 / evals; };   // CU-cycles per point-evaluation at 2.4 GHz, 256 CUs
    printf("lvl %d (%dx%d, n=%d): CU-cycles/point  rowmajor-vec %.2f | rowmajor-scalar %.2f | tiled8x4-scalar %.2f | rowpair16x2-vec %.2f | tiled8x4-rowvec %.2f\n",
           lvl, w, h, n, cyc(t0), cyc(t1), cyc(t2), cyc(t3), cyc(t4));
    printf("        two points in flight %.2f | four %.2f | non-temporal %.2f | 2 waves/SIMD: %.2f, with four in flight %.2f\n", cyc(t5), cyc(t7), cyc(t6), cyc(t8), cyc(t9));
    hipFree(dp); hipFree(imgs); hipFree(out);
  }
  return 0;
}
