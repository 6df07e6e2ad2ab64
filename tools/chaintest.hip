// Diagnostic: latency of a dependent fp32 add chain / a dependent LDS read chain / a dependent global load chain for ONE wavefront,
// with the GPU kept busy (back-to-back launches) vs launched every ~400 us (the BA loop's duty cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void chain(int n, const int* __restrict__ idx, long long* out, float* sink) {
  __shared__ int s_next[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_next[i] = (i * 17 + 1) & 1023;
  __syncthreads();
  float x = threadIdx.x;
  long long w0 = wall_clock64();
  for (int i = 0; i < n; i += 64) {
#pragma unroll
    for (int k = 0; k < 64; k++) x += 1.0f;
  }
  long long w1 = wall_clock64();
  int p = threadIdx.x;
  for (int i = 0; i < n / 8; i += 16) {
#pragma unroll
    for (int k = 0; k < 16; k++) p = s_next[p];
  }
  long long w2 = wall_clock64();
  int g = threadIdx.x;
  for (int i = 0; i < n / 64; i++) g = idx[g];
  long long w3 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = w2 - w1; out[2] = w3 - w2; }
  sink[threadIdx.x] = x + p + g;
}
int main() {
  const int M = 1 << 24;
  int* hidx = new int[M];
  for (int i = 0; i < M; i++) hidx[i] = (int)(((long long)i * 1000003 + 12345) % M);
  int* didx; hipMalloc(&didx, sizeof(int) * M); hipMemcpy(didx, hidx, sizeof(int) * M, hipMemcpyHostToDevice);
  long long* d; hipMalloc(&d, 64); float* sink; hipMalloc(&sink, 4096); long long h[3];
  const int n = 65536;
  for (int mode = 0; mode < 3; mode++) {
    double a = 0, l = 0, g = 0; const int reps = 50;
    for (int r = 0; r < reps; r++) {
      chain<<<1, 64>>>(n, didx, d, sink);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      if (mode == 1) usleep(400);
      if (mode == 2) usleep(5000);
      if (r >= 10) { a += h[0]; l += h[1]; g += h[2]; }
    }
    const double k = 10.0 / (reps - 10);   // ns per tick
    printf("mode %d (%s): fp32 add %.2f ns, LDS dependent read %.1f ns, global dependent load %.0f ns\n", mode,
           mode == 0 ? "back-to-back" : mode == 1 ? "400us gaps" : "5ms gaps", a * k / n, l * k / (n / 8), g * k / (n / 64));
  }
  return 0;
}
