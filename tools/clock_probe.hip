// Diagnostic (not part of the product): in-kernel clock and latency of dependent chains for a
// single-wave kernel on MI355X.  Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain_kernel(double* out, unsigned long long* stamps, int nfma, int nlds, int nrsq) {
  __shared__ double buf[64];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double x = out[threadIdx.x], y = 1.000001;
  for (int i = 0; i < nfma; ++i) x = __builtin_fma(x, y, 1e-9);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  buf[threadIdx.x] = x;
  int idx = threadIdx.x;
  for (int i = 0; i < nlds; ++i) { idx = (int)buf[idx & 63] & 63; buf[(idx + 1) & 63] = idx; }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  double z = x + 2.0;
  for (int i = 0; i < nrsq; ++i) z = rsqrt(z) + 1.5;
  unsigned long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x + idx + z;
  if (threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = t3 - t0; stamps[4] = r3 - r0; }
}
__global__ void heavy_kernel(double* a, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] * 1.0001 + 1.0;
}
int main() {
  double* out; unsigned long long* st; double* big; size_t nbig = 1ull << 28;
  hipMalloc(&out, 64 * 8); hipMalloc(&st, 64); hipMalloc(&big, nbig * 8);
  hipMemset(out, 0, 64 * 8); hipMemset(big, 0, nbig * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long h[5];
  auto run = [&](const char* name, bool heavy_first, int reps) {
    if (heavy_first) heavy_kernel<<<4096, 256>>>(big, nbig);
    float ms_tot = 0;
    for (int r = 0; r < reps; ++r) {
      hipEventRecord(e0);
      chain_kernel<<<1, 64>>>(out, st, 2000, 200, 200);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms_tot += ms;
    }
    hipMemcpy(h, st, 40, hipMemcpyDeviceToHost);
    double clk = (double)h[3] / (double)h[4] * 100.0;  // MHz (memrealtime ticks at 100 MHz)
    printf("%-28s event_us %.1f | cycles: fma %.1f/op lds %.1f/iter rsqrt %.1f/op total %llu | clock %.0f MHz\n", name,
           ms_tot / reps * 1e3, h[0] / 2000.0, h[1] / 200.0, h[2] / 200.0, h[3], clk);
  };
  run("cold", false, 1);
  run("warm x20", false, 20);
  run("after heavy", true, 1);
  run("after heavy x20", true, 20);
  // back-to-back launches of tiny kernels, one event pair
  hipEventRecord(e0);
  for (int r = 0; r < 200; ++r) chain_kernel<<<1, 64>>>(out, st, 2000, 200, 200);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, st, 40, hipMemcpyDeviceToHost);
  printf("200 back-to-back: %.1f us each, last clock %.0f MHz\n", ms / 200 * 1e3, (double)h[3] / (double)h[4] * 100.0);
  return 0;
}
