// CoarseInitializer::calcResAndGS (src/dso/FullSystem/CoarseInitializer.cpp:331-624) on the device.
// One thread per initializer point evaluates its 8 pattern residuals exactly as the reference does (per-point outputs — energy_new,
// isGood_new, maxstep, JbBuffer_new, lastHessian_new — are bit-identical); the two 9x9 systems (sum over all accepted residual rows,
// and the Schur rows JbBuffer * 1/(1+Hdd)) are accumulated on the matrix cores like the tracker's (tracker_kernels.hpp), fixed order.
// alphaOpt is known before the launch: the reference's EAlpha accumulator is never fed (its A stays 0, CoarseInitializer.cpp:497-520),
// so alphaEnergy = alphaW * |t|^2 * npts depends on the pose only.
#pragma once
#include "common.h"
#include "interp.hpp"

namespace dmv {

typedef float init_f32x4 __attribute__((ext_vector_type(4)));
enum { IN_STRIDE = 66, IN_ROWS = 16, IN_WAVE_FLOATS = IN_ROWS * IN_STRIDE + 64, IN_PART = 96 };   // partial: 45 H + 45 SC + E + pad

struct InitPts {   // struct Pnt (CoarseInitializer.h:44-83), structure of arrays
  int n;
  const float *u, *v, *idepth_new, *iR, *energy /* 2n */, *outlierTH;
  const unsigned char* isGood;
  float *energy_new /* 2n */, *maxstep, *lastHessian_new, *JbBuffer_new /* 10n */;
  unsigned char* isGood_new;
};
struct InitArgs {
  float RKi[9], t[3], aff0, aff1;   // (R * Ki).cast<float>(), translation, (exp(a), b)
  float fxl, fyl, cxl, cyl;
  int wl, hl;
  float alphaOpt, couplingWeight, huberTH;
};

__constant__ int c_initPattern8[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

// getInterpolatedElement31 on the intensity plane
__device__ __forceinline__ float initInterp31(const float* __restrict__ I, const float x, const float y, const int w) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float* bp = I + ix + iy * w;
  return dxdy * bp[1 + w] + (dy - dxdy) * bp[w] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
}
// getInterpolatedElement33 with the reference's flat-index gradient rule at any in-image position
__device__ __forceinline__ float3 initInterp33(const float* __restrict__ I, const float x, const float y, const int w, const int h) {
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  const float* bp = I + ix + iy * w;
  const float2 g00 = gradAt(I, w, h, ix, iy), g10 = gradAt(I, w, h, ix + 1, iy), g01 = gradAt(I, w, h, ix, iy + 1), g11 = gradAt(I, w, h, ix + 1, iy + 1);
  float3 r;
  r.x = w11 * bp[1 + w] + w01 * bp[w] + w10 * bp[1] + w00 * bp[0];
  r.y = w11 * g11.x + w01 * g01.x + w10 * g10.x + w00 * g00.x;
  r.z = w11 * g11.y + w01 * g01.y + w10 * g10.y + w00 * g00.y;
  return r;
}

// 64 rows (one per lane) J[0..8], weight w  ->  acc += sum_lane w * J J^T  (16 v_mfma_f32_16x16x4_f32; rows/cols 9..15 stay zero)
__device__ __forceinline__ void waveOuter9(const float (&J)[9], const float w, float* __restrict__ wJ, float* __restrict__ wW, const int lane,
                                           init_f32x4& a0, init_f32x4& a1, init_f32x4& a2, init_f32x4& a3) {
  const int mi = lane & 15, mk = lane >> 4;
#pragma unroll
  for (int k = 0; k < 9; k++) wJ[k * IN_STRIDE + lane] = J[k];
  wW[lane] = w;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int m = 0; m < 16; m += 4) {
    const float x0 = wJ[mi * IN_STRIDE + 4 * m + mk], x1 = wJ[mi * IN_STRIDE + 4 * m + 4 + mk];
    const float x2 = wJ[mi * IN_STRIDE + 4 * m + 8 + mk], x3 = wJ[mi * IN_STRIDE + 4 * m + 12 + mk];
    const float y0 = x0 * wW[4 * m + mk], y1 = x1 * wW[4 * m + 4 + mk], y2 = x2 * wW[4 * m + 8 + mk], y3 = x3 * wW[4 * m + 12 + mk];
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, y0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, y1, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, y2, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x3, y3, a3, 0, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(256) k_init_partial(const float* __restrict__ Iref, const float* __restrict__ Inew, const InitPts P, const InitArgs A,
                                                       float* __restrict__ partials /* gridDim.x x IN_PART */) {
  __shared__ float s_stage[4 * IN_WAVE_FLOATS];
  __shared__ float s_partH[4 * 256], s_partS[4 * 256], s_partE[4];
  for (int k = threadIdx.x; k < 4 * IN_WAVE_FLOATS; k += 256) s_stage[k] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* __restrict__ wJ = s_stage + wave * IN_WAVE_FLOATS;
  float* __restrict__ wW = wJ + IN_ROWS * IN_STRIDE;
  init_f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0, h2 = h0, h3 = h0, c0 = h0, c1 = h0, c2 = h0, c3 = h0;
  float Esum = 0.f;
  const int stride = gridDim.x * 256, base0 = blockIdx.x * 256 + wave * 64;
  for (int base = base0; base < P.n; base += stride) {   // wave-uniform trip count
    const int i = base + lane;
    float rows[8][9];
    float Jb[10];
    bool accepted = false;
#pragma unroll
    for (int k = 0; k < 10; k++) Jb[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
      for (int k = 0; k < 9; k++) rows[q][k] = 0.f;
    if (i < P.n) {
      float maxstep = 1e10f;
      const float e0 = P.energy[2 * i], e1 = P.energy[2 * i + 1];
      if (!P.isGood[i]) {
        Esum += e0;
        P.energy_new[2 * i] = e0; P.energy_new[2 * i + 1] = e1;
        P.isGood_new[i] = 0;
      } else {
        const float pu = P.u[i], pv = P.v[i], id = P.idepth_new[i];
        bool good = true;
        float en = 0.f;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
          if (good) {
            const int dx = c_initPattern8[idx][0], dy = c_initPattern8[idx][1];
            const float X = pu + dx, Y = pv + dy;
            const float pt0 = A.RKi[0] * X + A.RKi[1] * Y + A.RKi[2] * 1.0f + A.t[0] * id;
            const float pt1 = A.RKi[3] * X + A.RKi[4] * Y + A.RKi[5] * 1.0f + A.t[1] * id;
            const float pt2 = A.RKi[6] * X + A.RKi[7] * Y + A.RKi[8] * 1.0f + A.t[2] * id;
            const float u = pt0 / pt2, v = pt1 / pt2;
            const float Ku = A.fxl * u + A.cxl, Kv = A.fyl * v + A.cyl;
            const float new_idepth = id / pt2;
            if (!(Ku > 1 && Kv > 1 && Ku < A.wl - 2 && Kv < A.hl - 2 && new_idepth > 0)) { good = false; }
            else {
              const float3 hit = initInterp33(Inew, Ku, Kv, A.wl, A.hl);
              const float rlR = initInterp31(Iref, X, Y, A.wl);
              if (!isfinite(rlR) || !isfinite(hit.x)) { good = false; }
              else {
                const float residual = hit.x - A.aff0 * rlR - A.aff1;
                float hw = fabsf(residual) < A.huberTH ? 1 : A.huberTH / fabsf(residual);
                en += hw * residual * residual * (2 - hw);
                const float dxdd = (A.t[0] - A.t[2] * u) / pt2, dydd = (A.t[1] - A.t[2] * v) / pt2;
                if (hw < 1) hw = sqrtf(hw);
                const float dxInterp = hw * hit.y * A.fxl, dyInterp = hw * hit.z * A.fyl;
                float* r = rows[idx];
                r[0] = new_idepth * dxInterp;
                r[1] = new_idepth * dyInterp;
                r[2] = -new_idepth * (u * dxInterp + v * dyInterp);
                r[3] = -u * v * dxInterp - (1 + v * v) * dyInterp;
                r[4] = (1 + u * u) * dxInterp + u * v * dyInterp;
                r[5] = -v * dxInterp + u * dyInterp;
                r[6] = -hw * A.aff0 * rlR;
                r[7] = -hw * 1;
                const float dd = dxInterp * dxdd + dyInterp * dydd;
                r[8] = hw * residual;
                const float a = dxdd * A.fxl, b = dydd * A.fyl;
                const float ms = 1.0f / sqrtf(a * a + b * b);
                if (ms < maxstep) maxstep = ms;
#pragma unroll
                for (int k = 0; k < 8; k++) Jb[k] += r[k] * dd;
                Jb[8] += r[8] * dd;
                Jb[9] += dd * dd;
              }
            }
          }
        }
        if (!good || en > P.outlierTH[i] * 20) {
          Esum += e0;
          P.isGood_new[i] = 0;
          P.energy_new[2 * i] = e0; P.energy_new[2 * i + 1] = e1;
        } else {
          accepted = true;
          Esum += en;
          P.isGood_new[i] = 1;
          P.energy_new[2 * i] = en;
          P.energy_new[2 * i + 1] = (id - 1) * (id - 1);
          P.lastHessian_new[i] = Jb[9];
          Jb[8] += A.alphaOpt * (id - 1);
          Jb[9] += A.alphaOpt;
          if (A.alphaOpt == 0) { Jb[8] += A.couplingWeight * (id - P.iR[i]); Jb[9] += A.couplingWeight; }
          Jb[9] = 1 / (1 + Jb[9]);
        }
      }
      P.maxstep[i] = maxstep;
#pragma unroll
      for (int k = 0; k < 10; k++) P.JbBuffer_new[10 * i + k] = Jb[k];
    }
    // residual rows of accepted points -> H; Schur row -> Hsc
#pragma unroll
    for (int q = 0; q < 8; q++) {
      float J[9];
#pragma unroll
      for (int k = 0; k < 9; k++) J[k] = accepted ? rows[q][k] : 0.f;
      waveOuter9(J, accepted ? 1.0f : 0.0f, wJ, wW, lane, h0, h1, h2, h3);
    }
    {
      float J[9];
#pragma unroll
      for (int k = 0; k < 9; k++) J[k] = accepted ? Jb[k] : 0.f;
      waveOuter9(J, accepted ? Jb[9] : 0.0f, wJ, wW, lane, c0, c1, c2, c3);
    }
  }
  const int mi = lane & 15, mk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    s_partH[wave * 256 + (mk * 4 + r) * 16 + mi] = (h0[r] + h1[r]) + (h2[r] + h3[r]);
    s_partS[wave * 256 + (mk * 4 + r) * 16 + mi] = (c0[r] + c1[r]) + (c2[r] + c3[r]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) Esum += __shfl_xor(Esum, off, 64);
  if (lane == 0) s_partE[wave] = Esum;
  __syncthreads();
  if (threadIdx.x < IN_PART) {
    const int k = threadIdx.x;
    float s = 0.f;
    if (k < 90) {
      const int kk = k % 45;
      int r = 0, off = 0;
      while (kk >= off + (9 - r)) { off += 9 - r; r++; }
      const int c = r + (kk - off);
      const float* src = k < 45 ? s_partH : s_partS;
      for (int wv = 0; wv < 4; wv++) s += src[wv * 256 + r * 16 + c];
    } else if (k == 90) {
      for (int wv = 0; wv < 4; wv++) s += s_partE[wv];
    }
    partials[blockIdx.x * IN_PART + k] = s;
  }
}

__global__ void __launch_bounds__(128) k_init_final(const float* __restrict__ partials, const int G, float* __restrict__ out /* IN_PART */) {
  if (threadIdx.x >= IN_PART) return;
  float s = 0.f;
  for (int g = 0; g < G; g++) s += partials[g * IN_PART + threadIdx.x];
  out[threadIdx.x] = s;
}

}  // namespace dmv
