// Diagnostic: effective shader clock (s_memtime ticks per 100 MHz wall-clock tick) for short and long kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long wall_ticks, long long* out) {
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  float x = threadIdx.x;
  while (wall_clock64() - w0 < wall_ticks) { for (int i = 0; i < 64; i++) x = x * 1.0001f + 0.5f; }
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[3];
  const long long durs[] = {1000, 10000, 100000, 1000000, 10000000, 1000, 10000, 100000};
  for (long long t : durs) {
    for (int grid : {1, 256, 2048}) {
      spin<<<grid, 256>>>(t, d);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("wall %8.1f us grid %4d: shader clock %.0f MHz\n", h[1] / 100.0, grid, 100.0 * h[0] / h[1]);
    }
  }
  return 0;
}
