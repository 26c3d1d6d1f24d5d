// Diagnostic (not part of the product): numeric check of the in-wave 16 x 16 tile factorization loop (phase D of big_diag)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rsqrt_refined(double d) {
  const double y0 = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
template <int VAR>
__global__ void k_tile(const double* in, double* out, int carrier_in) {
  __shared__ double Pv[16][4][16];
  __shared__ int progress;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wv = threadIdx.x >> 6;
  if (wv != 0) { __builtin_amdgcn_s_barrier(); return; }
  const bool carrier = carrier_in != 0;
  v4d pt, E;
  for (int q = 0; q < 4; ++q) pt[q] = in[li + 16 * (4 * q + lk)];
  for (int q = 0; q < 4; ++q) E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
  const int jb = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = j >> 2, lkj = j & 3;
    const bool colj = lk == lkj;
    const double dj = readlane_f64(pt[q], lkj * 16 + j);
    const double sj = rsqrt_refined(dj);
    const double xj = pt[q] * sj;
    const double xm = (li >= j) ? xj : 0.0;
    Pv[j][lk][li] = colj ? xm : sj;
    asm volatile("" ::: "memory");
    if (lane == 0) __atomic_store_n(&progress, jb + j + 1, __ATOMIC_RELAXED);
    asm volatile("" ::: "memory");
    if (j < 15 || carrier) {
      const double xu = (colj && li > j) ? xj : 0.0;
      if (j < 15) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
      if (carrier) {
        const double ej = colj ? E[q] * sj : 0.0;
        E[q] = colj ? ej : E[q];
        if (j < 15) E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int q = 0; q < 4; ++q) pt[q] = Pv[4 * q + lk][lk][li];
  for (int q = 0; q < 4; ++q) out[li + 16 * (4 * q + lk)] = pt[q];
  for (int q = 0; q < 4; ++q) out[256 + li + 16 * (4 * q + lk)] = E[q];
  __builtin_amdgcn_s_barrier();
}
int main() {
  std::vector<double> h(256), L(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i + 16 * j] = (i == j) ? 20.0 + i : 1.0 / (1 + abs(i - j));
  // CPU Cholesky (lower)
  std::vector<double> a = h;
  for (int j = 0; j < 16; ++j) {
    double d = a[j + 16 * j];
    for (int k = 0; k < j; ++k) d -= L[j + 16 * k] * L[j + 16 * k];
    L[j + 16 * j] = sqrt(d);
    for (int i = j + 1; i < 16; ++i) {
      double s = a[i + 16 * j];
      for (int k = 0; k < j; ++k) s -= L[i + 16 * k] * L[j + 16 * k];
      L[i + 16 * j] = s / L[j + 16 * j];
    }
  }
  double *in, *out;
  hipMalloc(&in, 256 * 8); hipMalloc(&out, 512 * 8);
  hipMemcpy(in, h.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int threads : {64, 256, 512, 768}) for (int car : {0, 1}) {
    k_tile<0><<<1, threads>>>(in, out, car);
    hipDeviceSynchronize();
    std::vector<double> g(512);
    hipMemcpy(g.data(), out, 512 * 8, hipMemcpyDeviceToHost);
    double worst = 0; int wi = -1, wj = -1;
    for (int j = 0; j < 16; ++j) for (int i = j; i < 16; ++i) { double e = fabs(g[i + 16 * j] - L[i + 16 * j]); if (e > worst) { worst = e; wi = i; wj = j; } }
    printf("threads %4d carrier %d: max |L - Lcpu| = %.3e at (%d,%d)\n", threads, car, worst, wi, wj);
  }
  return 0;
}
