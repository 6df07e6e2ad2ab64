// TEST INFRASTRUCTURE ONLY.  Accumulator9 / Accumulator11 of the reference (src/dso/OptimizationBackend/MatrixAccumulators.h:975-1345, 92-141):
// 4-lane SSE partial sums with the 1k / 1M shift-up hierarchy.
#pragma once
#include <emmintrin.h>
#include <cstring>
#include <cstddef>

struct Acc9 {
  alignas(16) float SSEData[4 * 45];
  alignas(16) float SSEData1k[4 * 45];
  alignas(16) float SSEData1m[4 * 45];
  float numIn1, numIn1k, numIn1m;
  size_t num;
  float H[9][9];
  void initialize() {
    memset(SSEData, 0, sizeof(SSEData)); memset(SSEData1k, 0, sizeof(SSEData1k)); memset(SSEData1m, 0, sizeof(SSEData1m));
    num = 0; numIn1 = numIn1k = numIn1m = 0; memset(H, 0, sizeof(H));
  }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < 45; i++)
        _mm_store_ps(SSEData1k + 4 * i, _mm_add_ps(_mm_load_ps(SSEData + 4 * i), _mm_load_ps(SSEData1k + 4 * i)));
      numIn1k += numIn1; numIn1 = 0; memset(SSEData, 0, sizeof(SSEData));
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < 45; i++)
        _mm_store_ps(SSEData1m + 4 * i, _mm_add_ps(_mm_load_ps(SSEData1k + 4 * i), _mm_load_ps(SSEData1m + 4 * i)));
      numIn1m += numIn1k; numIn1k = 0; memset(SSEData1k, 0, sizeof(SSEData1k));
    }
  }
  inline void updateSSE_eighted(const __m128 J[9], const __m128 w) {
    float* pt = SSEData;
    for (int r = 0; r < 9; r++) {
      __m128 Jrw = _mm_mul_ps(J[r], w);
      for (int c = r; c < 9; c++) {
        _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(Jrw, J[c])));
        pt += 4;
      }
    }
    num += 4; numIn1++;
    shiftUp(false);
  }
  // Accumulator9::updateSSE (MatrixAccumulators.h:1027-1089): unweighted 4-lane update
  inline void updateSSE(const __m128 J[9]) {
    float* pt = SSEData;
    for (int r = 0; r < 9; r++)
      for (int c = r; c < 9; c++) {
        _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(J[r], J[c])));
        pt += 4;
      }
    num += 4; numIn1++;
    shiftUp(false);
  }
  // Accumulator9::updateSingleWeighted (MatrixAccumulators.h:1242-1318): lane 0 only; the diagonal is J*J*w, the row then continues with J*w
  inline void updateSingleWeighted(float J[9], const float w) {
    float* pt = SSEData;
    for (int r = 0; r < 9; r++) {
      *pt += J[r] * J[r] * w; pt += 4; J[r] *= w;
      for (int c = r + 1; c < 9; c++) { *pt += J[c] * J[r]; pt += 4; }
    }
    num++; numIn1++;
    shiftUp(false);
  }
  void finish() {
    memset(H, 0, sizeof(H));
    shiftUp(true);
    int idx = 0;
    for (int r = 0; r < 9; r++)
      for (int c = r; c < 9; c++) {
        float d = SSEData1m[idx + 0] + SSEData1m[idx + 1] + SSEData1m[idx + 2] + SSEData1m[idx + 3];
        H[r][c] = H[c][r] = d;
        idx += 4;
      }
  }
};


struct Acc11 {
  alignas(16) float SSEData[4], SSEData1k[4], SSEData1m[4];
  float A;
  size_t num;
  float numIn1, numIn1k, numIn1m;
  void initialize() { A = 0; memset(SSEData, 0, 16); memset(SSEData1k, 0, 16); memset(SSEData1m, 0, 16); num = 0; numIn1 = numIn1k = numIn1m = 0; }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) {
      _mm_store_ps(SSEData1k, _mm_add_ps(_mm_load_ps(SSEData), _mm_load_ps(SSEData1k)));
      numIn1k += numIn1; numIn1 = 0; memset(SSEData, 0, 16);
    }
    if (numIn1k > 1000 || force) {
      _mm_store_ps(SSEData1m, _mm_add_ps(_mm_load_ps(SSEData1k), _mm_load_ps(SSEData1m)));
      numIn1m += numIn1k; numIn1k = 0; memset(SSEData1k, 0, 16);
    }
  }
  void updateSingle(const float val) { SSEData[0] += val; num++; numIn1++; shiftUp(false); }
  void finish() { shiftUp(true); A = SSEData1m[0 + 0] + SSEData1m[0 + 1] + SSEData1m[0 + 2] + SSEData1m[0 + 3]; }
};
