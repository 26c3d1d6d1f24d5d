// Diagnostic (not part of the product): is straight-line unrolled code instruction-fetch bound the first time it runs?
// The in-wave 16-pivot loop of big_diag (phase D), per repetition: the first pass runs cold code, later ones warm.
#include <hip/hip_runtime.h>
#include <cstdio>
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
template <int COPY>
__device__ __noinline__ void tile_factor(v4d& pt, v4d& E, int li, int lk) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = j >> 2, lkj = j & 3;
    const double dj = readlane_f64(pt[q], lkj * 16 + j);
    const double sj = rsqrt_refined(dj);
    const bool colj = lk == lkj;
    const double xm = (colj && li >= j) ? pt[q] * sj : 0.0;
    const double ej = colj ? E[q] * sj : 0.0;
    pt[q] = colj ? xm : pt[q];
    E[q] = colj ? ej : E[q];
    if (j < 15) {
      const double xu = (li > j) ? xm : 0.0;
      pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
      E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
    }
  }
}
__global__ void k_reps(const double* in, double* out, unsigned long long* t, int ncopies) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  v4d pt0;
  for (int q = 0; q < 4; ++q) pt0[q] = in[li + 16 * (4 * q + lk)];
  double accum = 0;
  for (int rep = 0; rep < 24; ++rep) {
    v4d pt = pt0, E;
    pt[0] += accum * 1e-300;
    for (int q = 0; q < 4; ++q) E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    switch (rep % ncopies) {
      case 0: tile_factor<0>(pt, E, li, lk); break;
      case 1: tile_factor<1>(pt, E, li, lk); break;
      case 2: tile_factor<2>(pt, E, li, lk); break;
      case 3: tile_factor<3>(pt, E, li, lk); break;
      case 4: tile_factor<4>(pt, E, li, lk); break;
      case 5: tile_factor<5>(pt, E, li, lk); break;
      case 6: tile_factor<6>(pt, E, li, lk); break;
      default: tile_factor<7>(pt, E, li, lk); break;
    }
    accum += pt[3] + E[3];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) t[rep] = c1 - c0;
  }
  out[lane] = accum;
}
int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i + 16 * j] = (i == j) ? 20.0 : 1.0 / (1 + abs(i - j));
  double *in, *out; unsigned long long* t;
  hipMalloc(&in, 256 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&t, 64 * 8);
  hipMemcpy(in, h.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int nc : {1, 2, 4, 8}) {
    k_reps<<<1, 64>>>(in, out, t, nc);
    hipDeviceSynchronize();
    unsigned long long ht[24];
    hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
    printf("%d code copies in rotation, cycles per 16-pivot pass:", nc);
    for (int r = 0; r < 24; ++r) printf(" %llu", ht[r]);
    printf("\n");
  }
  return 0;
}
