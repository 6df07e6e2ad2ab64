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
  {
    // the same launches with a DIFFERENT kernel of similar code size between them (the other instantiation on its own system): what a cold instruction cache costs a launch — in
    // the real loop the linearisation / accumulation kernels run on the CU between two solves
    const int n2 = small ? 100 : 68;
    std::vector<double> H2((size_t)n2 * n2, 0.0), b2(n2, 1.0);
    for (int i = 0; i < n2; i++) H2[(size_t)i * n2 + i] = 10.0 + i;
    double *dH2, *db2, *dout2;
    hipMalloc((void**)&dH2, sizeof(double) * n2 * n2); hipMalloc((void**)&db2, sizeof(double) * n2); hipMalloc((void**)&dout2, sizeof(double) * (2 * n2 + 2));
    hipMemcpy(dH2, H2.data(), sizeof(double) * n2 * n2, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), sizeof(double) * n2, hipMemcpyHostToDevice);
    const size_t lds2 = sizeof(double) * baSolveCoreLdsDoubles(n2);
    hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
    float t_mine = 0, t_other = 0;
    for (int rep = 0; rep < 3; rep++) {
      t_mine = 0; t_other = 0;
      for (int i = 0; i < 20; i++) {
        hipEventRecord(a0);
        if (small) hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, dH, db, 0, dout);
        else hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF_CAP>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, dH, db, 0, dout);
        hipEventRecord(a1); hipEventSynchronize(a1);
        float ms; hipEventElapsedTime(&ms, a0, a1); t_mine += ms;
        hipEventRecord(a0);
        if (small) hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF_CAP>), dim3(1), dim3(BA_SOLVE_THREADS), lds2, nullptr, n2, dH2, db2, 0, dout2);
        else hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF>), dim3(1), dim3(BA_SOLVE_THREADS), lds2, nullptr, n2, dH2, db2, 0, dout2);
        hipEventRecord(a1); hipEventSynchronize(a1);
        hipEventElapsedTime(&ms, a0, a1); t_other += ms;
      }
    }
    // and the same launch pattern (one launch per event pair) without the other kernel
    float t_alone = 0;
    for (int i = 0; i < 20; i++) {
      hipEventRecord(a0);
      if (small) hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, dH, db, 0, dout);
      else hipLaunchKernelGGL((k_ba_solve_debug<BA_MAXF_CAP>), dim3(1), dim3(BA_SOLVE_THREADS), lds, nullptr, n, dH, db, 0, dout);
      hipEventRecord(a1); hipEventSynchronize(a1);
      float ms; hipEventElapsedTime(&ms, a0, a1); t_alone += ms;
    }
    printf("one launch per event pair: %.2f us alone, %.2f us with the other instantiation launched between two launches (that one: %.2f us)\n", 1e3 * t_alone / 20, 1e3 * t_mine / 20, 1e3 * t_other / 20);
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
