// Image pyramid + gradients on the device.
// Replaces FrameHessian::makeImages (src/dso/FullSystem/HessianBlocks.cpp:128-191): 2x2 box pyramid
// (0.25f * ((a+b)+c)+d, the reference's operand order) and central-difference gradients over the flat
// index range [w, w*(h-1)) — including the reference's row wrap-around at x=0 / x=w-1.
// Output layout: float4 (I, dx, dy, 0) per pixel per level (common.h FrameStore).  Rows 0 and h-1
// (never written by the reference) carry dx=dy=0.  absSquaredGrad (pixel selector only) is not produced.
#pragma once
#include "common.h"

namespace dmv {

// One launch per level: reads planar intensity of level l (Il), writes the float4 level image and the
// planar intensity of level l+1 (Inext, may be null on the coarsest level).  Pure streaming kernel.
__global__ void __launch_bounds__(256) k_make_level(const float* __restrict__ Il, const int w, const int h,
                                                     float4* __restrict__ out, float* __restrict__ Inext) {
  const int n = w * h;
  const int stride = gridDim.x * blockDim.x;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const float I = Il[idx];
    float dx = 0.0f, dy = 0.0f;
    if (idx >= w && idx < w * (h - 1)) {
      dx = 0.5f * (Il[idx + 1] - Il[idx - 1]);
      dy = 0.5f * (Il[idx + w] - Il[idx - w]);
      if (!isfinite(dx)) dx = 0.0f;
      if (!isfinite(dy)) dy = 0.0f;
    }
    out[idx] = make_float4(I, dx, dy, 0.0f);
  }
  if (Inext) {
    const int w2 = w >> 1, h2 = h >> 1, n2 = w2 * h2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n2; idx += stride) {
      const int x = idx % w2, y = idx / w2;
      const int b = 2 * x + 2 * y * w;
      Inext[idx] = 0.25f * (Il[b] + Il[b + 1] + Il[b + w] + Il[b + w + 1]);
    }
  }
}

// Batched variant: blockIdx.y = frame of the batch.  in_base/in_stride address the planar intensity of level l of
// frame f (raw images for l = 0, per-frame scratch above); slots[f] selects the destination pyramid.
__global__ void __launch_bounds__(256) k_make_level_batch(const float* __restrict__ in_base, const size_t in_stride, const int w, const int h,
                                                           const FrameStore fs, const int* __restrict__ slots, const int lvl,
                                                           float* __restrict__ next_base, const size_t next_stride) {
  const int f = blockIdx.y;
  const float* __restrict__ Il = in_base + (size_t)f * in_stride;
  float4* __restrict__ out = fs.level_mut(slots[f], lvl);
  const int n = w * h;
  const int stride = gridDim.x * blockDim.x;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const float I = Il[idx];
    float dx = 0.0f, dy = 0.0f;
    if (idx >= w && idx < w * (h - 1)) {
      dx = 0.5f * (Il[idx + 1] - Il[idx - 1]);
      dy = 0.5f * (Il[idx + w] - Il[idx - w]);
      if (!isfinite(dx)) dx = 0.0f;
      if (!isfinite(dy)) dy = 0.0f;
    }
    out[idx] = make_float4(I, dx, dy, 0.0f);
  }
  if (next_base) {
    float* __restrict__ Inext = next_base + (size_t)f * next_stride;
    const int w2 = w >> 1, h2 = h >> 1, n2 = w2 * h2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n2; idx += stride) {
      const int x = idx % w2, y = idx / w2;
      const int b = 2 * x + 2 * y * w;
      Inext[idx] = 0.25f * (Il[b] + Il[b + 1] + Il[b + w] + Il[b + w + 1]);
    }
  }
}

// float4 level image -> the reference's float3 AoS (parity tests / debug download)
__global__ void __launch_bounds__(256) k_level_to_f3(const float4* __restrict__ in, const int n, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) {
    const float4 p = in[idx];
    out[3 * idx + 0] = p.x; out[3 * idx + 1] = p.y; out[3 * idx + 2] = p.z;
  }
}

}  // namespace dmv
