// bigfront.hip — partial Cholesky of the BIG fronts (n > kSmallMaxN) of a level, three launches per round of
// <= 256 frontal columns instead of one launch per 32-column panel step (gfx950, wave64, FP64 matrix cores).
//
// choleskyPartial (gtsam/base/cholesky.cpp:108-159) on the front [A11 A21'; A21 A22] (lower, column-major, rhs = last
// row) is three products with very different shapes, and each gets the kernel that fits it:
//
//   A. big_diag   L11 = chol(A11)                one workgroup per front.  The F x F block lives in REGISTERS for the
//                 whole factorization, as 16 x 16 tiles in the matrix-core accumulator layout spread over the
//                 workgroup's waves; only the four current columns pass through LDS (two LDS barriers per four
//                 pivots).  This is the only sequential chain of a front: F pivots.
//   B. big_rows   L21 = A21 L11^-T               one workgroup per 32 rows of A21 (and the rhs row): blocked forward
//                 substitution along the row block, every product on the matrix cores, no inter-workgroup
//                 dependency at all — the rows of L21 are independent of each other once L11 is known.
//   C. big_schur  A22 -= L21 L21'                one workgroup per lower 32 x 32 tile pair, operands streamed straight
//                 from the L panel into matrix-core registers, every tile read and written exactly once.
//
// The front's storage is unchanged (kernels.h: BigDesc): the n x n square keeps the factored 32 x 32 diagonal tiles
// (L lower, (L^-1)' strictly upper) and, after C, the Schur complement handed to the parent; the n x F L-panel area
// right after it receives the rows of L below each diagonal tile and 1 / L_cc on its diagonal.  Back-substitution,
// marginals, the parents' gather and the partial re-elimination read exactly what they read before.
//
// Measured constants behind the design (tools/lat_probe2.hip on MI355X): a dependent FP64 FMA issues every 6 cycles, rsq /
// rcp every 17, v_mfma_f64_16x16x4 every 64 (latency = issue), an LDS write -> barrier -> read round 176-295 cycles
// (256-1024 threads), a kernel boundary 2.4 us, a cross-workgroup flag 0.6 us one way.  So a pivot costs ~250 cycles
// when four of them share two barriers, and a front's chain is F x 0.1 us; the old schedule paid a launch, a tile
// load and a 32 x 32 factorization with 14 barriers per 32 columns.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>

#include "gsx_internal.h"
#include "kernels.h"

namespace gsx {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int T = kTile;          // 32: edge of the diagonal tiles of the stored layout
constexpr int kMaxChunk = 192;    // frontal columns one round factors (12 tile rows of 16: 78 tiles over 8 waves)

#ifdef GSX_STAMP
__device__ unsigned long long g_bstamp[16];
#define BST_BEGIN unsigned long long bst0__ = __builtin_amdgcn_s_memtime();
#define BST_ADD(slot)                                                          \
  {                                                                            \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();                     \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_bstamp[slot] += t__ - bst0__;   \
    bst0__ = t__;                                                              \
  }
#else
#define BST_BEGIN
#define BST_ADD(slot)
#endif

__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 1 / sqrt(d) to full precision from the hardware seed (one cubic correction step)
__device__ __forceinline__ double rsqrt_refined(double d) {
  const double y0 = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

__device__ inline void report_failure(DevStatus* status, int front) {
  atomicAdd(&status->n_fail, 1);
  atomicMin(&status->first_front, front);
}

// ---- A. L11 = chol(A11[c0 .. c0 + fw)) ------------------------------------------------------------------------------
// Lower 16 x 16 tiles (ti >= tj), column-major over the triangle, tile t -> wave t % NW, slot t / NW (so a wave's slots
// are sorted by tile column, and a wave owns at most one tile of any column).  A lane (li = lane & 15, lk = lane >> 4)
// holds of a tile the entries (row 16 ti + li, column 16 tj + 4 q + lk), q = 0..3 — the accumulator layout of
// v_mfma_f64_16x16x4 with the tile computed transposed, so that the rank-4 update of a tile by four finished columns
// is ONE instruction whose operands are the tile row's and the tile column's L values.
// Per tile column: the wave that owns a tile of it moves that tile to `pt`, the four stages (4 pivots each) run
//   1. owners publish their rows of the 4 current columns            -> LDS barrier
//   2. owners factor the 4 x 4 pivot block (each lane alike), scale their rows, publish L -> LDS barrier
//   3. every tile to the right takes its rank-4 update (one matrix-core instruction per tile)
// and the finished tile goes out to the front.  The slots still to the right are a suffix s >= s0 of the wave's slots:
// the update chain is entered through a switch, so every slot's code exists once and there is no per-slot branch.
__device__ __forceinline__ void tile_coords(int t, int nt16, int& ti, int& tj) {
  int c = 0;
  while (t >= nt16 - c) {
    t -= nt16 - c;
    ++c;
  }
  tj = c;
  ti = c + t;
}

template <int NW, int MAXS>
__global__ void __launch_bounds__(NW * 64) big_diag_kernel(const BigDesc* descs, int c0, int chunk, double* arena,
                                                           DevStatus* status) {
  __shared__ double P[kMaxChunk][4];    // the four current columns, raw (updated through the previous stage)
  __shared__ double Lc[kMaxChunk][4];   // the same columns of L
  __shared__ double dinv[kMaxChunk];    // 1 / L_cc
  __shared__ double ldiag[kMaxChunk];   // L_cc
  __shared__ int sfail;
  extern __shared__ double dyn[];       // the factored 32 x 32 diagonal tiles [kb][r][33], then the 16 x 16 inverses [b][i][17]
  BST_BEGIN
  const BigDesc d = descs[blockIdx.x];
  const int n = d.N, F = d.F;
  if (c0 >= F) return;
  const int fw = min(chunk, F - c0);
  const int nt16 = (fw + 15) >> 4, nt32 = (fw + T - 1) / T;
  const int ntile = nt16 * (nt16 + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* A = arena + d.off;
  double* X = arena + d.xoff;
  double* tiles = dyn;
  double* Xd = dyn + (size_t)nt32 * T * (T + 1);
  if (tid == 0) sfail = 0;

  int ti[MAXS], tj[MAXS];
  v4d acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int t = wv + s * NW;
    ti[s] = tj[s] = 0;  // an empty slot updates a tile of zeros with valid operands: harmless
    if (t < ntile) tile_coords(t, nt16, ti[s], tj[s]);
    const int r = 16 * ti[s] + li;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = 16 * tj[s] + 4 * q + lk;
      double v = (r == cc) ? 1.0 : 0.0;  // padding beyond fw: identity (its pivots are 1, its L columns zero)
      if (t < ntile && r < fw && cc < fw) {
        const int hi = max(r, cc), lo = min(r, cc);  // a diagonal tile is kept symmetric (both triangles updated alike)
        v = A[(c0 + hi) + (i64)(c0 + lo) * n];
      }
      acc[s][q] = v;
    }
  }

  int fail = 0;
  int s0 = 0;  // first slot whose tile column has not been finished
  BST_ADD(0)
  for (int tc = 0; tc < nt16; ++tc) {
    const int jb = 16 * tc;
    // this wave's tiles of column tc (a column has at most 2 NW tiles: at most two, its first unfinished slots)
    bool own0 = false, own1 = false;
    int r0 = 0, r1 = 0;
    v4d pt0 = {0.0, 0.0, 0.0, 0.0}, pt1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int t = wv + s0 * NW;
      if (s0 < MAXS && t < ntile) {
        int cti, ctj;
        tile_coords(t, nt16, cti, ctj);
        if (ctj == tc) {
#pragma unroll
          for (int s = 0; s < MAXS; ++s)
            if (s == s0) (o == 0 ? pt0 : pt1) = acc[s];
          (o == 0 ? own0 : own1) = true;
          (o == 0 ? r0 : r1) = 16 * cti + li;
          ++s0;
        }
      }
    }
    BST_ADD(1)
    // 2. of a stage: factor the 4 x 4 pivot block (every owner lane alike), scale the tile's rows, publish L
#define GSX_PHASE2(PT, R, M)                                                                                          \
  {                                                                                                                   \
    const double p00 = P[j][0];                                                                                       \
    const double p10 = P[j + 1][0], p11 = P[j + 1][1];                                                                \
    const double p20 = P[j + 2][0], p21 = P[j + 2][1], p22 = P[j + 2][2];                                             \
    const double p30 = P[j + 3][0], p31 = P[j + 3][1], p32 = P[j + 3][2], p33 = P[j + 3][3];                          \
    const double a0 = P[R][0], a1 = P[R][1], a2 = P[R][2], a3 = P[R][3];                                              \
    /* pivot block: l_kk = sqrt(d_k), i_k = 1 / l_kk; a non-positive pivot fails the front (Eigen::LLT NumericalIssue) */ \
    const double d0 = p00;                                                                                            \
    const double i0 = d0 > 0 ? rsqrt_refined(d0) : 1.0;                                                               \
    const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;                                                      \
    const double d1 = fma(-l10, l10, p11);                                                                            \
    const double i1 = d1 > 0 ? rsqrt_refined(d1) : 1.0;                                                               \
    const double l21 = fma(-l20, l10, p21) * i1, l31 = fma(-l30, l10, p31) * i1;                                      \
    const double d2 = fma(-l21, l21, fma(-l20, l20, p22));                                                            \
    const double i2 = d2 > 0 ? rsqrt_refined(d2) : 1.0;                                                               \
    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * i2;                                                      \
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));                                            \
    const double i3 = d3 > 0 ? rsqrt_refined(d3) : 1.0;                                                               \
    if (!(d0 > 0) || !(d1 > 0) || !(d2 > 0) || !(d3 > 0)) fail = 1;                                                   \
    /* this lane's row against the block */                                                                           \
    const double x0 = a0 * i0;                                                                                        \
    const double x1 = fma(-x0, l10, a1) * i1;                                                                         \
    const double x2 = fma(-x1, l21, fma(-x0, l20, a2)) * i2;                                                          \
    const double x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, a3))) * i3;                                           \
    double out = (lk == 0) ? x0 : ((lk == 1) ? x1 : ((lk == 2) ? x2 : x3));                                           \
    const int cpos = j + lk;                                                                                          \
    if (R == cpos) {                                                                                                  \
      const double dd = (lk == 0) ? d0 : ((lk == 1) ? d1 : ((lk == 2) ? d2 : d3));                                    \
      const double ii = (lk == 0) ? i0 : ((lk == 1) ? i1 : ((lk == 2) ? i2 : i3));                                    \
      out = dd * ii; /* sqrt(d) */                                                                                    \
      dinv[R] = ii;                                                                                                   \
      ldiag[R] = out;                                                                                                 \
    }                                                                                                                 \
    if (R < cpos) out = 0.0; /* above the diagonal */                                                                 \
    Lc[R][lk] = out;                                                                                                  \
    PT[M] = out;                                                                                                      \
  }
    // 3. for an owned tile itself: rows / columns up to j+3 are final -> zero operands
#define GSX_OWN_UPD(PT, R)                                                                                            \
  {                                                                                                                   \
    const int rc = jb + li;                                                                                           \
    double a = Lc[rc][lk], b = Lc[R][lk];                                                                             \
    a = (rc >= j + 4) ? -a : 0.0;                                                                                     \
    b = (R >= j + 4) ? b : 0.0;                                                                                       \
    PT = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, PT, 0, 0, 0);                                                     \
  }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = jb + 4 * m;
      if (j < fw) {
        if (own0) P[r0][lk] = pt0[m];
        if (own1) P[r1][lk] = pt1[m];
        lds_bar();
        BST_ADD(2)
        if (own0) GSX_PHASE2(pt0, r0, m)
        if (own1) GSX_PHASE2(pt1, r1, m)
        BST_ADD(3)
        lds_bar();
        BST_ADD(4)
        // rank-4 updates: D[col][row] -= sum_k L[col][j+k] L[row][j+k]
        if (m < 3) {
          if (own0) GSX_OWN_UPD(pt0, r0)
          if (own1) GSX_OWN_UPD(pt1, r1)
        }
        // tiles right of the column (tile column > tc: nothing of them is final yet, no masks): the slots from s0 on
#pragma unroll
        for (int k = 0; k < MAXS; ++k)
          if (s0 <= k)
            acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lc[16 * tj[k] + li][lk], Lc[16 * ti[k] + li][lk], acc[k], 0, 0,
                                                          0);
        BST_ADD(5)
      }
    }
#undef GSX_PHASE2
#undef GSX_OWN_UPD
    // the finished tiles: L inside a diagonal 32-tile goes to the square (and to LDS for the inverse), the rest to
    // the L-panel area
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (!(o == 0 ? own0 : own1)) continue;
      const int r = (o == 0) ? r0 : r1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = jb + 4 * q + lk;
        if (r >= fw || cc >= fw || r < cc) continue;
        const bool same = (r >> 5) == (cc >> 5);
        const double v = (o == 0) ? pt0[q] : pt1[q];
        (same ? A : X)[(c0 + r) + (i64)(c0 + cc) * n] = v;
        if (same) tiles[((r >> 5) * T + (r & 31)) * (T + 1) + (cc & 31)] = v;
      }
    }
    BST_ADD(6)
  }
  if (fail) sfail = 1;
  lds_bar();

  // ---- (L^-1)' of every diagonal 32-tile into its strictly upper triangle, 1 / L_cc into the L-panel area --------------
  for (int c = tid; c < fw; c += NW * 64) X[(c0 + c) + (i64)(c0 + c) * n] = dinv[c];
  // 1. the 16 x 16 diagonal blocks: thread (b, c) solves L_bb x = e_c right-looking (the partial sums of all later rows
  //    advance together: independent FMAs)
  if (tid < 16 * nt16) {
    const int b = tid >> 4, c = tid & 15;
    const int w16 = min(16, fw - 16 * b);
    const double* Lt = tiles + ((size_t)(b >> 1) * T + (b & 1) * 16) * (T + 1) + (b & 1) * 16;
    double* xd = Xd + (size_t)b * 16 * 17;
    double sum[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double di = (k < w16) ? dinv[16 * b + k] : 0.0;
      const double xk = (k == c) ? di : ((k > c) ? -sum[k] * di : 0.0);
      xd[k * 17 + c] = xk;
      if (k > c && k < w16) A[(c0 + 16 * b + c) + (i64)(c0 + 16 * b + k) * n] = xk;
#pragma unroll
      for (int i = k + 1; i < 16; ++i) sum[i] = fma((i < w16) ? Lt[i * (T + 1) + k] : 0.0, xk, sum[i]);
    }
  }
  lds_bar();
  // 2. the off-diagonal block of a 32-tile: X10 = -X11 (L10 X00), one wave per tile, two products of four matrix-core
  //    instructions; the first product's accumulator IS the second one's operand (same lane layout)
  for (int kb = wv; kb < nt32; kb += NW) {
    if (fw - kb * T <= 16) continue;  // a single 16-block
    const double* L10 = tiles + ((size_t)kb * T + 16) * (T + 1);
    const double* X00 = Xd + (size_t)(2 * kb) * 16 * 17;
    const double* X11 = Xd + (size_t)(2 * kb + 1) * 16 * 17;
    const int w1 = min(16, fw - kb * T - 16);
    v4d t1 = {0.0, 0.0, 0.0, 0.0}, t2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // T[i][jj] = sum_k L10[i][k] X00[k][jj]; lane: t1[q] = T[4 q + lk][li]
      const double a = (li < w1) ? L10[li * (T + 1) + 4 * s + lk] : 0.0;
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X00[(4 * s + lk) * 17 + li], t1, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)    // X10[i][jj] = -sum_k X11[i][k] T[k][jj]: operand T[4 s + lk][li] = t1[s]
      t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-X11[li * 17 + 4 * s + lk], t1[s], t2, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // (L^-1)[16 + 4 q + lk][li] -> transposed position (row li, column 16 + 4 q + lk)
      const int i = 4 * q + lk;
      if (i < w1) A[(c0 + kb * T + li) + (i64)(c0 + kb * T + 16 + i) * n] = t2[q];
    }
  }
  BST_ADD(7)
  if (tid == 0) {
    int bad = sfail;
    if (c0 + fw == F) {  // conditioning test on the last two pivots (cholesky.cpp:145-158)
      int e1, e2;
      (void)frexp(ldiag[fw - 1], &e1);
      if (F >= 2) {
        const double p2 = (fw >= 2) ? ldiag[fw - 2] : A[(F - 2) + (i64)(F - 2) * n];
        (void)frexp(p2, &e2);
        if (!(e2 - e1 < 12)) bad = 1;
      } else if (!(e1 > -12)) {
        bad = 1;
      }
    }
    if (bad) report_failure(status, d.front);
  }
}

// lane roles of the 256-thread kernels: wave wv owns the 16 x 16 quadrant (row half wv & 1, column half wv >> 1) of
// a 32 x 32 tile; entry q of a lane = (row r0 + li, column cq0 + 4 q + lk)
struct Quad {
  int li, lk, r0, cq0;
  __device__ __forceinline__ Quad() {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    li = lane & 15;
    lk = lane >> 4;
    r0 = 16 * (wv & 1);
    cq0 = 16 * (wv >> 1);
  }
};

// ---- B. rows of L21 (and the rhs row): X_i = A_i L11^-T, 32 rows per workgroup --------------------------------------
//   for k = 0 .. : T = A_ik - sum_{p<k} X_ip L_kp';  X_ik = T (L_kk^-1)'      (tile inverse left by big_diag)
__global__ void __launch_bounds__(256) big_rows_kernel(const BigDesc* descs, int c0, int chunk, double* arena) {
  extern __shared__ double dyn[];
  const BigDesc d = descs[blockIdx.y];
  const int n = d.N, F = d.F;
  if (c0 >= F) return;
  const int fw = min(chunk, F - c0), base = c0 + fw;
  const int ri = base + T * blockIdx.x;
  if (ri >= n) return;
  const int hi = min(T, n - ri);
  const int nk = (fw + T - 1) / T;
  const int ldx = T * nk + 2;           // row stride = 2 mod 32 doubles: the operand reads below are conflict-free
  double* Xrow = dyn;                   // [32][ldx]: the solved row block so far
  double(*Tt)[T + 2] = (double(*)[T + 2])(dyn + (size_t)T * ldx);
  double* A = arena + d.off;
  double* X = arena + d.xoff;
  const Quad L;
  const int row = L.r0 + L.li;
  for (int k = 0; k < nk; ++k) {
    const int wk = min(T, fw - k * T), ck = c0 + k * T;
    v4d acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = L.cq0 + 4 * q + L.lk;
      acc[q] = (row < hi && c < wk) ? A[(ri + row) + (i64)(ck + c) * n] : 0.0;
    }
    const int lrow = k * T + L.cq0 + L.li;  // row of L11 this lane feeds (inside the chunk)
    for (int p = 0; p < k; ++p) {
      const double* Lkp = X + (c0 + lrow) + (i64)(c0 + p * T) * n;
      const double* xr = Xrow + (size_t)row * ldx + p * T;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double a = (lrow < fw) ? Lkp[(i64)(4 * s + L.lk) * n] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, xr[4 * s + L.lk], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Tt[row][L.cq0 + 4 * q + L.lk] = acc[q];
    // (L_kk^-1)[c][kk], kk <= c: strictly upper triangle of the diagonal tile (transposed), 1 / L_cc in the L-panel area
    const int c = L.cq0 + L.li;
    double av[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int kk = 4 * s + L.lk;
      av[s] = 0.0;
      if (c < wk && kk < c) av[s] = A[(ck + kk) + (i64)(ck + c) * n];
      else if (c < wk && kk == c) av[s] = X[(ck + c) + (i64)(ck + c) * n];
    }
    lds_bar();
    v4d x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < 4 || L.cq0 != 0)  // columns of the first half only need kk < 16 (wave-uniform)
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], Tt[row][4 * s + L.lk], x, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = L.cq0 + 4 * q + L.lk;
      Xrow[(size_t)row * ldx + k * T + cc] = x[q];
      if (row < hi && cc < wk) X[(ri + row) + (i64)(ck + cc) * n] = x[q];
    }
    lds_bar();
  }
}

// ---- C. Schur complement: C_ij -= X_i X_j' over the lower 32 x 32 tile pairs of the rows below the chunk ------------
__global__ void __launch_bounds__(256) big_schur_kernel(const BigDesc* descs, int c0, int chunk, double* arena) {
  const BigDesc d = descs[blockIdx.y];
  const int n = d.N, F = d.F;
  if (c0 >= F) return;
  const int fw = min(chunk, F - c0), base = c0 + fw;
  const int ntile = (n - base + T - 1) / T;
  const int t = blockIdx.x;
  if (t >= ntile * (ntile + 1) / 2) return;
  int i = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  const int j = t - i * (i + 1) / 2;
  const int ri = base + i * T, rj = base + j * T;
  const int hi = min(T, n - ri), hj = min(T, n - rj);
  double* A = arena + d.off;
  const double* X = arena + d.xoff;
  const Quad L;
  const int row = L.r0 + L.li;
  double cv[4];
  bool live[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = L.cq0 + 4 * q + L.lk;
    live[q] = row < hi && c < hj && !(i == j && row < c);
    cv[q] = live[q] ? A[(ri + row) + (i64)(rj + c) * n] : 0.0;
  }
  const bool ain = L.cq0 + L.li < hj, bin = row < hi;
  const double* pa = X + (rj + L.cq0 + L.li) + (i64)(c0 + L.lk) * n;
  const double* pb = X + (ri + row) + (i64)(c0 + L.lk) * n;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  int kk = 0;
  for (; kk + 32 <= fw; kk += 32) {  // eight k-steps with all sixteen loads in flight
    double a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      a[s] = ain ? pa[(i64)(kk + 4 * s) * n] : 0.0;
      b[s] = bin ? pb[(i64)(kk + 4 * s) * n] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
  }
  for (; kk < fw; kk += 4) {
    const bool in = kk + L.lk < fw;
    const double a = (ain && in) ? pa[(i64)kk * n] : 0.0;
    const double b = (bin && in) ? pb[(i64)kk * n] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (live[q]) A[(ri + row) + (i64)(rj + L.cq0 + 4 * q + L.lk) * n] = cv[q] - acc[q];
}

}  // namespace

#ifdef GSX_STAMP
void big_stamp_dump(const char* what) {
  unsigned long long h[16];
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bstamp), sizeof(h));
  printf("[bstamp] %s: diag prologue %llu | column setup %llu | publish+bar1 %llu | phase2 %llu | bar2 %llu | updates %llu | "
         "column store %llu | inverse %llu  (cycles, wave 0 of block 0)\n", what, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_bstamp), z, sizeof(z));
}
#endif

// ---- host side -----------------------------------------------------------------------------------------------------
// Rounds of one launch group: every front of the group advances by `chunk` frontal columns per round.
void plan_big_group(const BigDesc* descs, int count, BigPlan& plan) {
  plan = BigPlan();
  int maxF = 0;
  for (int k = 0; k < count; ++k) maxF = std::max(maxF, descs[k].F);
  if (count == 0 || maxF == 0) return;
  const int nch = (maxF + kMaxChunk - 1) / kMaxChunk;
  plan.chunk = std::min(kMaxChunk, (((maxF + nch - 1) / nch + T - 1) / T) * T);
  for (int c0 = 0; c0 < maxF; c0 += plan.chunk) {
    int fw = 0, rb = 0, pairs = 0;
    for (int k = 0; k < count; ++k) {
      if (descs[k].F <= c0) continue;
      const int w = std::min(plan.chunk, descs[k].F - c0);
      const int nt = (descs[k].N - (c0 + w) + T - 1) / T;
      fw = std::max(fw, w);
      rb = std::max(rb, nt);
      pairs = std::max(pairs, nt * (nt + 1) / 2);
    }
    plan.fw.push_back(fw);
    plan.rb.push_back(rb);
    plan.pairs.push_back(pairs);
  }
}

void launch_big_diag(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, DevStatus* status,
                     hipStream_t st) {
  if (!count) return;
  static bool attr = false;
  const auto lds_for = [](int fw) {
    return (size_t)(((fw + T - 1) / T) * T * (T + 1) + ((fw + 15) / 16) * 16 * 17) * sizeof(double);
  };
  const int kTilesLds = (int)lds_for(kMaxChunk);
  if (!attr) {
    hipFuncSetAttribute((const void*)big_diag_kernel<8, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    hipFuncSetAttribute((const void*)big_diag_kernel<8, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    attr = true;
  }
  const int c0 = round * plan.chunk, fw = plan.fw[round];
  const size_t lds = lds_for(fw);
  const int nt16 = (fw + 15) / 16;
  // tiles: nt16 (nt16 + 1) / 2 over the waves; a column's tiles (<= nt16) over at most two slots of a wave
  if (nt16 <= 4) big_diag_kernel<4, 3><<<count, 256, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else if (nt16 <= 6) big_diag_kernel<8, 3><<<count, 512, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else if (nt16 <= 10) big_diag_kernel<8, 7><<<count, 512, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else big_diag_kernel<8, 10><<<count, 512, lds, st>>>(descs, c0, plan.chunk, arena, status);
}

void launch_big_rows(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, hipStream_t st) {
  if (!count || plan.rb[round] <= 0) return;
  static bool attr = false;
  const auto lds_for = [](int fw) { return (size_t)(T * (T * ((fw + T - 1) / T) + 2) + T * (T + 2)) * sizeof(double); };
  if (!attr) {
    hipFuncSetAttribute((const void*)big_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(kMaxChunk));
    attr = true;
  }
  big_rows_kernel<<<dim3(plan.rb[round], count), 256, lds_for(plan.fw[round]), st>>>(descs, round * plan.chunk, plan.chunk,
                                                                                   arena);
}

void launch_big_schur(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, hipStream_t st) {
  if (!count || plan.pairs[round] <= 0) return;
  big_schur_kernel<<<dim3(plan.pairs[round], count), 256, 0, st>>>(descs, round * plan.chunk, plan.chunk, arena);
}

}  // namespace gsx
