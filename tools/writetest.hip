// What a write-dominated streaming kernel reaches on MI355X: 16-byte stores per lane, plain vs non-temporal, per-workgroup contiguous chunks of 4 KB / 16 KB / 64 KB,
// with and without a 1-byte-per-4-written read stream beside them (the shape of k_build_pyramids_raw<u8>).   hipcc --offload-arch=gfx950 -O3 tools/writetest.hip -o writetest
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT, int RD>
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ dst, const unsigned int* __restrict__ src, const size_t n4, const int chunk4) {
  // workgroup b owns f4 elements [b*chunk4, (b+1)*chunk4)
  for (size_t base = (size_t)blockIdx.x * chunk4; base < n4; base += (size_t)gridDim.x * chunk4)
    for (int i = threadIdx.x; i < chunk4; i += 256) {
      f4 v = {1.f, 2.f, 3.f, 4.f};
      if (RD) { const unsigned int r = src[base + i]; v[0] = (float)(r & 255); v[1] = (float)((r >> 8) & 255); v[2] = (float)((r >> 16) & 255); v[3] = (float)(r >> 24); }
      if (NT) __builtin_nontemporal_store(v, dst + base + i); else dst[base + i] = v;
    }
}
template <int NT, int RD>
static void run(f4* d, unsigned int* s, size_t n4, int chunk4, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (int)((n4 + chunk4 - 1) / chunk4);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_write<NT, RD>), dim3(grid), dim3(256), 0, 0, d, s, n4, chunk4);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_write<NT, RD>), dim3(grid), dim3(256), 0, 0, d, s, n4, chunk4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-44s chunk %6d B: %.3f ms, %.2f TB/s written\n", name, chunk4 * 16, ms, n4 * 16.0 / ms / 1e9);
}
int main() {
  const size_t n4 = (size_t)1 << 28;   // 4 GiB written
  f4* d; unsigned int* s;
  hipMalloc(&d, n4 * 16); hipMalloc(&s, n4 * 4); hipMemset(s, 7, n4 * 4);
  for (int chunk4 : {256, 1024, 4096}) {
    run<0, 0>(d, s, n4, chunk4, "plain stores, no reads");
    run<1, 0>(d, s, n4, chunk4, "non-temporal stores, no reads");
    run<0, 1>(d, s, n4, chunk4, "plain stores + 1 B read per 4 B written");
    run<1, 1>(d, s, n4, chunk4, "non-temporal stores + 1 B read per 4 B written");
  }
  return 0;
}
