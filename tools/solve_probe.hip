// Measurement tool (not part of the library): k_ba_solve_debug on a random SPD system with cycle stamps inside one step of the factorisation (-DBA_SOLVE_PROBE=<step>).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DBA_SOLVE_PROBE=20 -I dm-vio_amd/csrc -I include -o tools/_solve_probe tools/solve_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "ba_batch_kernels.hpp"
using namespace dmv;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 68;
  std::vector<double> H((size_t)n * n), b(n), A((size_t)n * (n + 8));
  unsigned long long sd = 12345;
  auto rnd = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return ((double)(sd >> 11) / 9007199254740992.0) * 2.0 - 1.0; };
  for (auto& v : A) v = rnd();
  for (int i = 0; i < n; i++) { b[i] = rnd(); for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n + 8; k++) s += A[(size_t)i * (n + 8) + k] * A[(size_t)j * (n + 8) + k]; H[(size_t)i * n + j] = s; } }
  if (argc > 2 && atoi(argv[2])) { for (int i = 1; i < n; i += 2) H[(size_t)i * n + i] = 1e13; }   // tie groups
  double *dH, *db, *dout;
  hipMalloc((void**)&dH, sizeof(double) * n * n); hipMalloc((void**)&db, sizeof(double) * n); hipMalloc((void**)&dout, sizeof(double) * (2 * n + 2));
  hipMemcpy(dH, H.data(), sizeof(double) * n * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  const bool small = n <= 4 + 8 * BA_MAXF;
  const size_t lds = sizeof(double) * baSolveCoreLdsDoubles(n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int exact = 0; exact < 2; exact++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; i++) {
        if (small) hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, dH, db, exact, dout);
        else hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF_CAP>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, dH, db, exact, dout);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("n = %d, exact_backsub = %d: %.2f us per launch (20 back to back), %s\n", n, exact, 1e3 * ms / 20, hipGetErrorString(hipGetLastError()));
    }
  }
#ifdef BA_SOLVE_PROBE
  long long pr[64];
  hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_ba_probe), sizeof(pr));
  const char* names[4] = {"owner, row group 0", "owner, row group 1", "", ""};
  printf("step %d (shader-clock cycles since the owner's slot start):\n", BA_SOLVE_PROBE);
  const long long t0 = pr[0];
  printf("  owner: start 0, column formed %lld, pivot read %lld, divided %lld, published %lld, past barrier %lld\n", pr[1] - t0, pr[2] - t0, pr[3] - t0, pr[4] - t0, pr[5] - t0);
  printf("  another wave: start %lld, updated %lld, past barrier %lld;  the next owner: start %lld, updated %lld, past barrier %lld\n", pr[16] - t0, pr[17] - t0, pr[21] - t0, pr[32] - t0, pr[33] - t0, pr[37] - t0);
  (void)names;
  printf("  the solve three times inside one launch (100 MHz ticks): %lld, %lld, %lld\n", pr[48], pr[49], pr[50]);
#endif
  std::vector<double> out(2 * n + 2);
  hipMemcpy(out.data(), dout, sizeof(double) * (2 * n + 2), hipMemcpyDeviceToHost);
  double res = 0; for (int i = 0; i < n; i++) { double s = -b[i]; for (int j = 0; j < n; j++) s += H[(size_t)i * n + j] * out[j]; res = fmax(res, fabs(s)); }
  printf("residual %.3e, branch %d\n", res, (int)out[2 * n]);
  return 0;
}
