// Diagnostic (not part of the product): latency constants that price the dense-Cholesky critical path on MI355X.
// Build here:  hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o build/lat_probe   (build/ travels to the GPU box)
// Run there:   build/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };
#define T0 unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define T1(slot) { unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
  if (threadIdx.x == 0 && blockIdx.x == 0) { st[slot].cyc = c1 - c0; st[slot].real = r1 - r0; } }

__global__ void k_fma(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x], y = 1.000001;
  T0
  for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 1e-9);
  T1(0)
  out[threadIdx.x] = x;
}
__global__ void k_rcp(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 2.0;
  T0
  for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;
  T1(1)
  out[threadIdx.x] = x;
}
__global__ void k_rsq(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 2.0;
  T0
  for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
  T1(2)
  out[threadIdx.x] = x;
}
__global__ void k_rcpf(double* out, Stamp* st, int n) {  // f32 rcp seed + 2 f64 newton steps
  double x = out[threadIdx.x] + 2.0;
  T0
  for (int i = 0; i < n; ++i) {
    double y = (double)__builtin_amdgcn_rcpf((float)x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    x = y + 1.5;
  }
  T1(3)
  out[threadIdx.x] = x;
}
__global__ void k_mfma_dep(double* out, Stamp* st, int n) {
  double a = out[threadIdx.x] * 1e-3, b = 1e-3;
  v4d acc = {0, 0, 0, 0};
  T0
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  T1(4)
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_mfma_ind(double* out, Stamp* st, int n) {
  double a = out[threadIdx.x] * 1e-3, b = 1e-3;
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  T0
  for (int i = 0; i < n; i += 4) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0);
  }
  T1(5)
  out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
// mfma whose A operand depends on the previous result through one fma (the shape of a blocked panel step)
__global__ void k_mfma_fma(double* out, Stamp* st, int n) {
  double a = out[threadIdx.x] * 1e-3, b = 1e-3;
  v4d acc = {0, 0, 0, 0};
  T0
  for (int i = 0; i < n; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    a = __builtin_fma(acc[0], 1e-9, a);
  }
  T1(6)
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a;
}
__global__ void k_readlane(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 1.0;
  T0
  for (int i = 0; i < n; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 63);
    const double s = __hiloint2double(hi, lo);
    x = __builtin_fma(x, 1e-9, s);
  }
  T1(7)
  out[threadIdx.x] = x;
}
__global__ void k_bpermute(double* out, Stamp* st, int n) {
  double x = out[threadIdx.x] + 1.0;
  int idx = ((threadIdx.x + 1) & 63) * 4;
  T0
  for (int i = 0; i < n; ++i) {
    const int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(x));
    x = __builtin_fma(__hiloint2double(hi, lo), 1e-9, 1.0);
  }
  T1(8)
  out[threadIdx.x] = x;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// one LDS hand-off round: every thread writes a double, barrier, reads a neighbour's
__global__ void k_lds_round(double* out, Stamp* st, int n, int slot) {
  __shared__ double buf[2][1024];
  double x = out[threadIdx.x];
  const int nb = blockDim.x;
  T0
  for (int i = 0; i < n; ++i) {
    buf[i & 1][threadIdx.x] = x;
    lds_barrier();
    x = buf[i & 1][(threadIdx.x + 17) % nb] + 1.0;
  }
  T1(slot)
  out[threadIdx.x] = x;
}
// LDS round trip without a barrier (single wave): write, wait, read
__global__ void k_lds_nobar(double* out, Stamp* st, int n) {
  __shared__ double buf[64];
  double x = out[threadIdx.x];
  T0
  for (int i = 0; i < n; ++i) {
    buf[threadIdx.x] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    x = buf[(threadIdx.x + 17) & 63] + 1.0;
  }
  T1(12)
  out[threadIdx.x] = x;
}
__global__ void k_empty(double* out) { if (threadIdx.x == 1000) out[0] = 1; }

// cross-workgroup ping-pong through agent-scope atomics (two blocks): block 0 sets flag[0] = i, block 1 answers flag[1] = i
__global__ void k_pingpong(unsigned* flags, Stamp* st, int n) {
  if (threadIdx.x != 0) return;
  T0
  if (blockIdx.x == 0) {
    for (int i = 1; i <= n; ++i) {
      __hip_atomic_store(&flags[0], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(&flags[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i && ++spins < 1000000) {}
    }
  } else if (blockIdx.x == 1) {
    for (int i = 1; i <= n; ++i) {
      int spins = 0;
      while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i && ++spins < 1000000) {}
      __hip_atomic_store(&flags[64], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  T1(13)
}

int main() {
  double* out; Stamp* st; unsigned* flags;
  hipMalloc(&out, 1024 * 8); hipMalloc(&st, 32 * sizeof(Stamp)); hipMalloc(&flags, 4096);
  std::vector<double> h(1024, 1.0);
  hipMemcpy(out, h.data(), 1024 * 8, hipMemcpyHostToDevice);
  hipMemset(st, 0, 32 * sizeof(Stamp)); hipMemset(flags, 0, 4096);
  const int n = 4096;
  for (int rep = 0; rep < 3; ++rep) {
    k_fma<<<1, 64>>>(out, st, n);
    k_rcp<<<1, 64>>>(out, st, n);
    k_rsq<<<1, 64>>>(out, st, n);
    k_rcpf<<<1, 64>>>(out, st, n);
    k_mfma_dep<<<1, 64>>>(out, st, n);
    k_mfma_ind<<<1, 64>>>(out, st, n);
    k_mfma_fma<<<1, 64>>>(out, st, n);
    k_readlane<<<1, 64>>>(out, st, n);
    k_bpermute<<<1, 64>>>(out, st, n);
    k_lds_round<<<1, 64>>>(out, st, n, 9);
    k_lds_round<<<1, 256>>>(out, st, n, 10);
    k_lds_round<<<1, 1024>>>(out, st, n, 11);
    k_lds_nobar<<<1, 64>>>(out, st, n);
    hipMemset(flags, 0, 4096);
    k_pingpong<<<2, 64>>>(flags, st, 2000);
    hipDeviceSynchronize();
  }
  Stamp hs[32];
  hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
  const char* names[] = {"fma f64 dep", "rcp f64 + add dep", "rsq f64 + add dep", "rcp f32 seed + 2 newton + add", "mfma f64 16x16x4 dep",
                         "mfma f64 16x16x4 4 indep", "mfma + fma dep", "2 readlane + fma", "2 bpermute + fma",
                         "lds round 64 thr", "lds round 256 thr", "lds round 1024 thr", "lds round no barrier 64", "pingpong round trip"};
  for (int i = 0; i < 14; ++i) {
    const int cnt = (i == 13) ? 2000 : n;
    printf("%-34s cycles/op %8.1f  ns/op %8.1f  clock %.0f MHz\n", names[i], (double)hs[i].cyc / cnt,
           (double)hs[i].real * 10.0 / cnt, (double)hs[i].cyc / (double)hs[i].real * 100.0);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 64, 1024}) {
    hipEventRecord(e0);
    for (int r = 0; r < 1000; ++r) k_empty<<<grid, 256>>>(out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel grid %4d back-to-back: %.2f us each\n", grid, ms);
  }
  return 0;
}
