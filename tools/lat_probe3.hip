// Diagnostic (not part of the product): cost of the in-wave 16 x 16 tile factorization loop of big_diag (phase D), variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, real; };
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
__global__ void k_tile(const double* in, double* out, Stamp* st, int reps, int slot) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wv = threadIdx.x >> 6;
  if (wv != 0) { __builtin_amdgcn_s_barrier(); return; }
  v4d pt0;
  for (int q = 0; q < 4; ++q) pt0[q] = in[li + 16 * (4 * q + lk)];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  v4d pt = pt0, E;
  double accum = 0;
  for (int rep = 0; rep < reps; ++rep) {
    pt = pt0;
    pt[0] += accum * 1e-300;
#pragma unroll
    for (int q = 0; q < 4; ++q) E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = j >> 2, lkj = j & 3;
      const double dj = (VAR == 3) ? 4.0 + j : readlane_f64(pt[q], lkj * 16 + j);
      const double sj = rsqrt_refined(dj);
      const bool colj = lk == lkj;
      const double xm = (colj && li >= j) ? pt[q] * sj : 0.0;
      const double ej = colj ? E[q] * sj : 0.0;
      pt[q] = colj ? xm : pt[q];
      E[q] = colj ? ej : E[q];
      if (j < 15) {
        const double xu = (li > j) ? xm : 0.0;
        if (VAR != 2) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
        else pt[q] = fma(-xu, xu, pt[q]) + 1.0;
        if (VAR == 0 || VAR == 3) E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
      }
    }
    accum += pt[3] + E[3];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { st[slot].cyc = c1 - c0; st[slot].real = r1 - r0; }
  for (int q = 0; q < 4; ++q) out[li + 16 * (4 * q + lk)] = pt[q] + E[q] + accum;
  __builtin_amdgcn_s_barrier();
}
int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i + 16 * j] = (i == j) ? 20.0 : 1.0 / (1 + abs(i - j));
  double *in, *out; Stamp* st;
  hipMalloc(&in, 256 * 8); hipMalloc(&out, 256 * 8); hipMalloc(&st, 32 * sizeof(Stamp));
  hipMemcpy(in, h.data(), 256 * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int r = 0; r < 2; ++r) {
    k_tile<0><<<1, 64>>>(in, out, st, reps, 0);
    k_tile<1><<<1, 64>>>(in, out, st, reps, 1);
    k_tile<2><<<1, 64>>>(in, out, st, reps, 2);
    k_tile<3><<<1, 64>>>(in, out, st, reps, 3);
    k_tile<0><<<1, 768>>>(in, out, st, reps, 4);
    k_tile<0><<<256, 768>>>(in, out, st, reps, 5);
    hipDeviceSynchronize();
  }
  Stamp hs[32];
  hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
  const char* names[] = {"full (pt + E mfma), 1 wave", "no E mfma", "no mfma (valu only)", "no readlane dependency", "full, 12 waves (11 at barrier)", "full, 256 blocks x 12 waves"};
  for (int i = 0; i < 6; ++i)
    printf("%-36s per 16x16 tile: %8.0f cycles  %7.2f us   (per pivot %.0f cycles)\n", names[i], (double)hs[i].cyc / reps,
           (double)hs[i].real * 0.01 / reps, (double)hs[i].cyc / reps / 16);
  return 0;
}
