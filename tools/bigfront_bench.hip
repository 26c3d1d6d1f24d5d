// Diagnostic (not part of the product): the three big-front kernels alone, on `count` identical fronts, HIP-event timed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bigfront_bench.hip -Igtsam_petercdev_amd/csrc \
//        -Lgtsam_petercdev_amd/csrc -lgsx -Wl,-rpath,'$ORIGIN/../gtsam_petercdev_amd/csrc' -o build/bigfront_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "kernels.h"
using namespace gsx;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 400, F = argc > 2 ? atoi(argv[2]) : 144, count = argc > 3 ? atoi(argv[3]) : 1;
  const int reps = 50;
  const i64 per = big_panel_offset(n) + (i64)n * F;
  std::vector<double> h((size_t)per, 0.0);
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  std::vector<double> B((size_t)n * n);
  for (auto& x : B) x = nd(rng);
  for (int c = 0; c < n; ++c)
    for (int r = c; r < n; ++r) {
      double s = (r == c) ? n : 0.0;
      for (int k = 0; k < 8; ++k) s += B[(size_t)r * n + k] * B[(size_t)c * n + k];
      h[r + (size_t)c * n] = s;
    }
  double* arena; BigDesc* d; DevStatus* st;
  hipMalloc(&arena, (size_t)per * count * sizeof(double)); hipMalloc(&d, count * sizeof(BigDesc)); hipMalloc(&st, sizeof(DevStatus));
  std::vector<BigDesc> hd(count);
  for (int k = 0; k < count; ++k) hd[k] = BigDesc{per * k, per * k + big_panel_offset(n), n, F, k, -1};
  hipMemcpy(d, hd.data(), count * sizeof(BigDesc), hipMemcpyHostToDevice);
  BigPlan plan;
  plan_big_group(hd.data(), count, plan);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tot[3] = {0, 0, 0};
  for (int rep = 0; rep < reps + 3; ++rep) {
    for (int k = 0; k < count; ++k) hipMemcpyAsync(arena + per * k, h.data(), (size_t)per * sizeof(double), hipMemcpyHostToDevice, 0);
    hipDeviceSynchronize();
    for (int r = 0; r < plan.rounds(); ++r)
      for (int ph = 0; ph < 3; ++ph) {
        hipEventRecord(e0, 0);
        if (ph == 0) launch_big_diag(d, count, plan, r, arena, st, 0);
        if (ph == 1) launch_big_rows(d, count, plan, r, arena, 0);
        if (ph == 2) launch_big_schur(d, count, plan, r, arena, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3) tot[ph] += ms;
      }
  }
  printf("n=%d F=%d count=%d rounds=%d chunk=%d: diag %.1f us  rows %.1f us  schur %.1f us  (per factorization, event-timed incl. ~launch)\n",
         n, F, count, plan.rounds(), plan.chunk, 1e3 * tot[0] / reps, 1e3 * tot[1] / reps, 1e3 * tot[2] / reps);
#ifdef GSX_STAMP
  big_stamp_dump("bench (totals over all repetitions; s_memtime ticks of 100 MHz)");
#endif
  return 0;
}
