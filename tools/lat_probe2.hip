// Diagnostic (not part of the product): dependent-op latencies with the loop overhead unrolled away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, real; };
#define T0 unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define T1(slot) { unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
  if (threadIdx.x == 0 && blockIdx.x == 0) { st[slot].cyc = c1 - c0; st[slot].real = r1 - r0; } }
#define U _Pragma("unroll 32")

__global__ void k_fma(double* out, Stamp* st, int n, int slot) {
  double x = out[threadIdx.x], y = 1.000001;
  T0
  U for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 1e-9);
  T1(slot)
  out[threadIdx.x] = x;
}
__global__ void k_fma2(double* out, Stamp* st, int n) {  // two independent chains
  double x = out[threadIdx.x], y = 1.000001, z = x + 1;
  T0
  U for (int i = 0; i < n; ++i) { x = __builtin_fma(x, y, 1e-9); z = __builtin_fma(z, y, 1e-9); }
  T1(1)
  out[threadIdx.x] = x + z;
}
__global__ void k_rcp(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 2.0;
  T0
  U for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rcp(x);
  T1(2)
  out[threadIdx.x] = x;
}
__global__ void k_rsq(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 2.0;
  T0
  U for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rsq(x);
  T1(3)
  out[threadIdx.x] = x;
}
__global__ void k_sqrt(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 2.0;
  T0
  U for (int i = 0; i < n; ++i) x = __builtin_amdgcn_sqrt(x);
  T1(4)
  out[threadIdx.x] = x;
}
__global__ void k_rsqf(double* out, Stamp* st, int n) {  // f32 rsq seed (cvt, rsq, cvt)
  double x = out[threadIdx.x] + 2.0;
  T0
  U for (int i = 0; i < n; ++i) x = (double)__builtin_amdgcn_rsqf((float)x);
  T1(5)
  out[threadIdx.x] = x;
}
__global__ void k_mfma4(double* out, Stamp* st, int n) {
  double a = out[threadIdx.x] * 1e-3, b = 1e-3;
  double acc = 0;
  T0
  U for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
  T1(6)
  out[threadIdx.x] = acc;
}
__global__ void k_mfma16(double* out, Stamp* st, int n) {
  double a = out[threadIdx.x] * 1e-3, b = 1e-3;
  v4d acc = {0, 0, 0, 0};
  T0
  U for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  T1(7)
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_mfma16_fma(double* out, Stamp* st, int n) {  // mfma -> fma on its result -> mfma operand
  double a = out[threadIdx.x] * 1e-3, b = 1e-3;
  v4d acc = {0, 0, 0, 0};
  T0
  U for (int i = 0; i < n; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    a = __builtin_fma(acc[0], 1e-9, a);
  }
  T1(8)
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a;
}
__global__ void k_readlane(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 1.0;
  T0
  U for (int i = 0; i < n; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
    x = __builtin_fma(x, 1e-9, __hiloint2double(hi, lo));
  }
  T1(9)
  out[threadIdx.x] = x;
}
__global__ void k_dpp(double* out, Stamp* st, int n) {  // row_shr:1 on both halves + fma
  double x = out[threadIdx.x] + 1.0;
  T0
  U for (int i = 0; i < n; ++i) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, false);
    x = __builtin_fma(x, 1e-9, __hiloint2double(hi, lo));
  }
  T1(10)
  out[threadIdx.x] = x;
}
__global__ void k_bperm(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 1.0;
  int idx = ((threadIdx.x + 1) & 63) * 4;
  T0
  U for (int i = 0; i < n; ++i) {
    const int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(x));
    x = __builtin_fma(__hiloint2double(hi, lo), 1e-9, 1.0);
  }
  T1(11)
  out[threadIdx.x] = x;
}
__global__ void k_ldsrt(double* out, Stamp* st, int n) {  // single wave: LDS write -> read other lane's (no barrier needed)
  __shared__ double buf[64];
  double x = out[threadIdx.x];
  T0
  U for (int i = 0; i < n; ++i) {
    buf[threadIdx.x] = x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    x = buf[(threadIdx.x + 17) & 63] + 1.0;
  }
  T1(12)
  out[threadIdx.x] = x;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ void k_ldsbar(double* out, Stamp* st, int n, int slot) {
  __shared__ double buf[2][1024];
  double x = out[threadIdx.x];
  const int nb = blockDim.x;
  const int other = (threadIdx.x + 17) % nb;
  T0
  U for (int i = 0; i < n; ++i) {
    buf[i & 1][threadIdx.x] = x;
    lds_barrier();
    x = buf[i & 1][other] + 1.0;
  }
  T1(slot)
  out[threadIdx.x] = x;
}
__global__ void k_bar_only(double* out, Stamp* st, int n, int slot) {
  T0
  U for (int i = 0; i < n; ++i) __builtin_amdgcn_s_barrier();
  T1(slot)
  out[threadIdx.x] = 1;
}
// 16 waves of one workgroup (4 per SIMD), all running the dependent fma chain: per-wave latency under co-residency
__global__ void k_fma_many(double* out, Stamp* st, int n, int slot) {
  double x = out[threadIdx.x & 63], y = 1.000001;
  T0
  U for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 1e-9);
  T1(slot)
  out[threadIdx.x] = x;
}
int main() {
  double* out; Stamp* st;
  hipMalloc(&out, 2048 * 8); hipMalloc(&st, 64 * sizeof(Stamp));
  std::vector<double> h(2048, 1.0);
  hipMemcpy(out, h.data(), 2048 * 8, hipMemcpyHostToDevice);
  hipMemset(st, 0, 64 * sizeof(Stamp));
  const int n = 4096;
  for (int rep = 0; rep < 3; ++rep) {
    k_fma<<<1, 64>>>(out, st, n, 0);
    k_fma2<<<1, 64>>>(out, st, n);
    k_rcp<<<1, 64>>>(out, st, n);
    k_rsq<<<1, 64>>>(out, st, n);
    k_sqrt<<<1, 64>>>(out, st, n);
    k_rsqf<<<1, 64>>>(out, st, n);
    k_mfma4<<<1, 64>>>(out, st, n);
    k_mfma16<<<1, 64>>>(out, st, n);
    k_mfma16_fma<<<1, 64>>>(out, st, n);
    k_readlane<<<1, 64>>>(out, st, n);
    k_dpp<<<1, 64>>>(out, st, n);
    k_bperm<<<1, 64>>>(out, st, n);
    k_ldsrt<<<1, 64>>>(out, st, n);
    k_ldsbar<<<1, 256>>>(out, st, n, 13);
    k_ldsbar<<<1, 512>>>(out, st, n, 14);
    k_ldsbar<<<1, 1024>>>(out, st, n, 15);
    k_bar_only<<<1, 256>>>(out, st, n, 16);
    k_bar_only<<<1, 1024>>>(out, st, n, 17);
    k_fma_many<<<1, 256>>>(out, st, n, 18);
    k_fma_many<<<1, 1024>>>(out, st, n, 19);
    hipDeviceSynchronize();
  }
  Stamp hs[64];
  hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
  const char* names[] = {"fma f64 dep", "fma f64 2 chains (per pair)", "rcp f64 dep", "rsq f64 dep", "sqrt f64 dep", "cvt+rsq f32+cvt dep",
                         "mfma f64 4x4x4 dep", "mfma f64 16x16x4 dep", "mfma16 + fma dep", "2 readlane + fma", "2 dpp + fma",
                         "2 bpermute + fma", "lds write-wait-read 1 wave", "lds round+barrier 256", "lds round+barrier 512",
                         "lds round+barrier 1024", "s_barrier only 256", "s_barrier only 1024", "fma dep, 4 waves (1/SIMD)",
                         "fma dep, 16 waves (4/SIMD)"};
  for (int i = 0; i < 20; ++i)
    printf("%-34s cycles/op %8.1f  ns/op %8.1f\n", names[i], (double)hs[i].cyc / n, (double)hs[i].real * 10.0 / n);
  return 0;
}
