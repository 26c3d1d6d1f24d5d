// Diagnostic (not part of the product): the D loop of big_diag with its mailbox, beside waves that poll / trail it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
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
// mode 0: others wait at the barrier; 1: others spin on progress only; 2: others run the trailing (O) loop
__global__ void k_col(const double* in, double* out, Stamp* st, int reps, int mode, int slot) {
  __shared__ double Pv[16][4][16];
  __shared__ int progress;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) progress = 0;
  __syncthreads();
  v4d pt0;
  for (int q = 0; q < 4; ++q) pt0[q] = in[li + 16 * (4 * q + lk)];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double accum = 0;
  for (int rep = 0; rep < reps; ++rep) {
    const int jb = 16 * rep;
    v4d pt = pt0;
    pt[0] += accum * 1e-300;
    if (wv == 0) {
      __builtin_amdgcn_s_setprio(3);
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
        const double xu = (colj && li > j) ? xj : 0.0;
        if (j < 15) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      accum += pt[3];
    } else if (mode == 1) {
      int spins = 0;
      while (__atomic_load_n(&progress, __ATOMIC_RELAXED) < jb + 16 && ++spins < (1 << 22)) {}
    } else if (mode == 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, lkj = j & 3;
        const bool colj = lk == lkj;
        int spins = 0;
        while (__atomic_load_n(&progress, __ATOMIC_RELAXED) < jb + j + 1 && ++spins < (1 << 22)) {}
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        const double sj = Pv[j][(lkj + 1) & 3][0];
        const double ld = Pv[j][lkj][li];
        const double xj = colj ? pt[q] * sj : 0.0;
        pt[q] = colj ? xj : pt[q];
        const double a = (colj && li > j) ? -ld : 0.0;
        if (j < 15) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xj, pt, 0, 0, 0);
      }
      accum += pt[3];
    }
    __syncthreads();
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { st[slot].cyc = c1 - c0; st[slot].real = r1 - r0; }
  out[threadIdx.x] = accum;
}
int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i + 16 * j] = (i == j) ? 20.0 + i : 1.0 / (1 + abs(i - j));
  double *in, *out; Stamp* st;
  hipMalloc(&in, 256 * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&st, 32 * sizeof(Stamp));
  hipMemcpy(in, h.data(), 256 * 8, hipMemcpyHostToDevice);
  const int reps = 100;
  int slot = 0;
  const int threads[] = {64, 256, 768}, modes[] = {0, 1, 2};
  for (int r = 0; r < 2; ++r) { slot = 0; for (int t : threads) for (int m : modes) { k_col<<<1, t>>>(in, out, st, reps, m, slot++); } hipDeviceSynchronize(); }
  Stamp hs[32];
  hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
  slot = 0;
  for (int t : threads) for (int m : modes) {
    printf("threads %4d mode %d (0 idle, 1 spin, 2 trail): per column %8.0f cycles  %6.2f us\n", t, m, (double)hs[slot].cyc / reps, (double)hs[slot].real * 0.01 / reps);
    ++slot;
  }
  return 0;
}
