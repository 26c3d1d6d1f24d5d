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
#include <cstdlib>

#include "gsx_internal.h"
#include "kernels.h"

namespace gsx {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int T = kTile;          // 32: edge of the diagonal tiles of the stored layout
constexpr int kMaxChunk = 192;    // frontal columns one round factors (12 tile rows of 16: 78 tiles over 8 waves)

#ifdef GSX_STAMP
__device__ unsigned long long g_rstamp[16][8];   // big_diag_rows_kernel: [wave][phase] cycles, block 0
#define RST_BEGIN unsigned long long rst0__ = __builtin_amdgcn_s_memtime();
#define RST_ADD(slot)                                                              \
  {                                                                                \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();                         \
    if (lane == 0 && blockIdx.x == 0) g_rstamp[wv][slot] += t__ - rst0__;          \
    rst0__ = t__;                                                                  \
  }
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
#define RST_BEGIN
#define RST_ADD(slot)
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
// Lower 16 x 16 tiles (ti >= tj), column-major over the triangle, tile t -> wave t % NW, slot t / NW: a wave's slots are
// sorted by tile column and (NW >= tile rows) it owns at most one tile of any column.  A lane (li = lane & 15,
// lk = lane >> 4) holds of a tile the entries (row 16 ti + li, column 16 tj + 4 q + lk), q = 0..3 — the accumulator
// layout of v_mfma_f64_16x16x4 with the tile computed transposed: a rank-4 update of a tile is ONE instruction whose
// operands are the tile column's and the tile row's L values, and a tile's own four columns ARE the row-side operand.
// Per tile column tc (16 pivots), two workgroup barriers:
//   D. the wave that owns the diagonal tile factors it alone, pivot by pivot, with NO cross-lane data movement but one
//      v_readlane of the pivot: the lanes that hold column j scale it and ARE the (one non-zero k of the) operands of
//      the rank-1 update of the tile, one matrix-core instruction.  An identity tile carried through the same column
//      operations comes out as L_dd^-T: the inverse of the diagonal tile (published; it is also what the stored
//      32 x 32 tile inverses are assembled from).  Chain per pivot: readlane -> rsq + correction -> mul -> mfma.
//   O. the waves that own the other tiles of the column: L_tile = A_tile L_dd^-T, four matrix-core instructions whose
//      row-side operand is the tile as it stands in its registers; they publish their rows of L (a 16-column strip)
//   U. every tile to the right takes its rank-16 update, four matrix-core instructions with operands from the strip.
//      The next diagonal tile is the first slot of its owner's chain, and that wave goes straight on to its D: the
//      sequential part of column tc+1 runs beside the updates of column tc.
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {  // src_lane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

template <int NW, int MAXS>
__global__ void __launch_bounds__(NW * 64) big_diag_kernel(const BigDesc* descs, int c0, int chunk, double* arena,
                                                           DevStatus* status) {
  constexpr int LS = 18;                 // row stride (doubles) of the L strips: operand reads hit 32 distinct bank pairs
  __shared__ double Lc[2][kMaxChunk][LS];  // rows of L of a tile column (16 columns, tile rows below the diagonal tile), columns alternate
  __shared__ double dinv[kMaxChunk];     // 1 / L_cc
  __shared__ double ldiag[kMaxChunk];    // L_cc
  __shared__ int sfail;
  extern __shared__ double dyn[];        // the factored 32 x 32 diagonal tiles [kb][r][33], then the 16 x 16 inverses [b][i][17]
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
    if (t < ntile) {
      int c = 0, rem = t;
      while (rem >= nt16 - c) {
        rem -= nt16 - c;
        ++c;
      }
      tj[s] = c;
      ti[s] = c + rem;
    }
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

  // rank-16 update of slot k from the strip of L
#define GSX_UPD16(k, c)                                                           \
  {                                                                               \
    const double* pa = &Lc[(c)&1][16 * tj[k] + li][lk];                           \
    const double* pb = &Lc[(c)&1][16 * ti[k] + li][lk];                           \
    const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12];                 \
    const double b0 = pb[0], b1 = pb[4], b2 = pb[8], b3 = pb[12];                 \
    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, acc[k], 0, 0, 0);      \
    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, acc[k], 0, 0, 0);      \
    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a2, b2, acc[k], 0, 0, 0);      \
    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a3, b3, acc[k], 0, 0, 0);      \
  }
  int fail = 0;
  int defer_k = MAXS;  // first slot whose update by the PREVIOUS column is still to do (MAXS: none)
  BST_ADD(0)
  for (int tc = 0; tc < nt16; ++tc) {
    const int jb = 16 * tc;
    // this wave's tile of column tc: the column's tiles are t = start .. start + (nt16 - tc) - 1, tile row tc + (t - start)
    const int start = tc * nt16 - tc * (tc - 1) / 2;
    int off = (wv - start) % NW;
    off += off < 0 ? NW : 0;
    const bool own = off < nt16 - tc;
    const bool diag = own && off == 0;
    const int slot = (start + off - wv) / NW;
    const int r = 16 * (tc + off) + li;  // this lane's row of the owned tile
    // (the loop's back edge sits between the matrix-core updates and this read of their result: explicit wait states)
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    v4d pt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
      if (own && s == slot) pt = acc[s];
    const int s0 = max(0, (start + (nt16 - tc) - wv + NW - 1) / NW);  // slots of tile columns <= tc: finished after this column
    BST_ADD(1)

    if (diag) {
      // ---- D: the diagonal tile, alone.  Other waves may still be in U of the previous column on this SIMD's matrix
      //      core: this wave is the critical path of the whole front and issues first.  The loop is straight-line code:
      //      a branch between a matrix-core instruction and the v_readlane of its result hides the dependency from the
      //      compiler's hazard padding (measured: stale pivots) ----
      __builtin_amdgcn_s_setprio(3);
      v4d E;
#pragma unroll
      for (int q = 0; q < 4; ++q) E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, lkj = j & 3;
        __builtin_amdgcn_sched_barrier(0);  // keep the pivots apart: the chain is the schedule
        int lj = li;
        asm volatile("" : "+v"(lj));  // (the lane masks are two compares a pivot; hoisted out of the column loop they spill)
        const double dj = readlane_f64(pt[q], lkj * 16 + j);
        const double sj = rsqrt_refined(dj);  // (a non-positive pivot makes L_jj a NaN: caught when the column is stored)
        const bool colj = lk == lkj;          // the lanes that hold column j
        const double xj = pt[q] * sj;
        const double xm = (colj && lj >= j) ? xj : 0.0;  // L[li][j] (zero above the diagonal)
        const double ej = colj ? E[q] * sj : 0.0;
        pt[q] = colj ? xm : pt[q];
        E[q] = colj ? ej : E[q];
        if (j < 15) {  // rank-1 update of the columns right of j: the column's lanes are the k = lkj operands, all others zero
          const double xu = (lj > j) ? xm : 0.0;
          pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      // the diagonal of the tile: entry (li, li) sits in the lane with lk = li & 3, register li >> 2
      if (lk == (li & 3)) {
        const int qd = li >> 2;
        const double l = (qd == 0) ? pt[0] : ((qd == 1) ? pt[1] : ((qd == 2) ? pt[2] : pt[3]));
        ldiag[jb + li] = l;
        dinv[jb + li] = 1.0 / l;
        if (jb + li < fw && !(l > 0)) fail = 1;  // non-positive or non-finite pivot (Eigen::LLT NumericalIssue)
      }
      // E[row][col] = (L_dd^-1)[col][row]: published as Xd[tc][i][c] = (L_dd^-1)[i][c]; its strictly lower part inside the
      // same 32-tile also goes to the front (transposed: the strictly upper triangle of the diagonal tile)
      double* xd = Xd + (size_t)tc * 16 * 17;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 4 * q + lk, c = li;
        xd[i * 17 + c] = E[q];
        if (i > c && jb + i < fw) A[(c0 + jb + c) + (i64)(c0 + jb + i) * n] = E[q];
      }
    }
    BST_ADD(2)
    lds_bar();
    BST_ADD(3)
    // the updates by the previous column that this wave put off to start its D early, beside the other waves' O
#pragma unroll
    for (int k = 0; k < MAXS; ++k)
      if (k >= defer_k) GSX_UPD16(k, tc + 1)
    defer_k = MAXS;
    if (own && !diag) {
      // ---- O: a tile below the diagonal tile: L_tile[row][col] = sum_k A[row][k] (L_dd^-1)[col][k] ----
      const double* xd = Xd + (size_t)tc * 16 * 17 + li * 17 + lk;
      v4d nt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) nt = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[4 * s], pt[s], nt, 0, 0, 0);
      pt = nt;
#pragma unroll
      for (int q = 0; q < 4; ++q) Lc[tc & 1][r][4 * q + lk] = pt[q];
    }
    BST_ADD(4)
    // the finished tile: L inside a diagonal 32-tile goes to the square (and to LDS for the inverse), the rest to the
    // L-panel area
    if (own) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = jb + 4 * q + lk;
        if (r >= fw || cc >= fw || r < cc) continue;
        const bool same = (r >> 5) == (cc >> 5);
        (same ? A : X)[(c0 + r) + (i64)(c0 + cc) * n] = pt[q];
        if (same) tiles[((r >> 5) * T + (r & 31)) * (T + 1) + (cc & 31)] = pt[q];
      }
    }
    BST_ADD(5)
    lds_bar();
    BST_ADD(6)
    // ---- U: rank-16 update of every tile right of the column: D[col][row] -= sum_k L[col][k] L[row][k].  The wave
    //      that owns the next diagonal tile (its first unfinished slot) updates only that one now and the rest after its D ----
    {
      int offn = (wv - (start + nt16 - tc)) % NW;
      offn += offn < 0 ? NW : 0;
      const bool next_diag = tc + 1 < nt16 && offn == 0;
#pragma unroll
      for (int k = 0; k < MAXS; ++k)
        if (k >= s0 && (!next_diag || k == s0)) GSX_UPD16(k, tc)
      if (next_diag) defer_k = s0 + 1;
    }
    BST_ADD(7)
  }
#undef GSX_UPD16
  if (fail) sfail = 1;
  lds_bar();

  // ---- (L^-1)' of every diagonal 32-tile into its strictly upper triangle (its two 16 x 16 diagonal blocks are in place:
  //      D), 1 / L_cc into the L-panel area ----
  for (int c = tid; c < fw; c += NW * 64) X[(c0 + c) + (i64)(c0 + c) * n] = dinv[c];
  // the off-diagonal block of a 32-tile: X10 = -X11 (L10 X00), one wave per tile, two products of four matrix-core
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
  BST_ADD(8)
  if (tid == 0) {
    int bad = sfail;
    // conditioning test on the last two pivots (cholesky.cpp:145-158): here only for the dense entry (one clique, parent ==
    // kStandaloneFront); the fronts of a tree are tested per reference clique by cond_check_kernel
    if (d.parent == kStandaloneFront && c0 + fw == F) {
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

// ---- D, four pivots at a time --------------------------------------------------------------------------------------
// The pivot-by-pivot tile factorization above issues ~38 vector instructions and two matrix-core instructions per pivot
// (the lane masks, the reciprocal square root, two rank-1 updates), ~370 cycles on a lone wave — and that chain is the
// critical path of every blocked front.  Here the 16 x 16 tile goes four columns at a time: the 4 x 4 diagonal block is
// read out with ten lane reads and factored, together with its inverse, in wave-uniform registers (every lane the same
// numbers: 4 reciprocal square roots and ~30 multiply-adds); then ONE matrix-core instruction forms the block's four
// columns of L, X = A_b M' (M = L44^-1 sits in the sixteen lanes of the operand that face the block), and ONE applies the
// rank-4 update to the columns to the right; the inverse tile E = L_dd^-T follows with the same two instructions.
// pt: the tile, lane (li, lk) holds (row li, column 4 q + lk); only its lower triangle is read.  A non-positive pivot
// turns its column into NaNs (caught by the caller's test of the diagonal).
// MEASURED, NOT USED (compile with GSX_D_BLOCK4 to switch it in; parity-green): 125 instructions per four pivots against
// 180, and no faster — 5 400-7 600 cycles per tile against 5 900-8 600 (tools/bigfront_bench.hip, F = 192: 51.7 us
// against 50.0; F = 30: 15.7 against 18.0).  The twenty v_readlane of the 4 x 4 block cost ~30 cycles each, as much as
// the matrix-core and mask instructions they replace: the chain is bound by how fast ONE wave issues instructions
// that feed scalar registers, not by their number.
__device__ __forceinline__ void tile_potrf16_b4(v4d& pt, v4d& E, const int li, const int lk) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int base = 4 * b;
    const double d00 = readlane_f64(pt[b], base), d10 = readlane_f64(pt[b], base + 1), d20 = readlane_f64(pt[b], base + 2),
                 d30 = readlane_f64(pt[b], base + 3), d11 = readlane_f64(pt[b], 16 + base + 1),
                 d21 = readlane_f64(pt[b], 16 + base + 2), d31 = readlane_f64(pt[b], 16 + base + 3),
                 d22 = readlane_f64(pt[b], 32 + base + 2), d32 = readlane_f64(pt[b], 32 + base + 3),
                 d33 = readlane_f64(pt[b], 48 + base + 3);
    // L44 and M = L44^-1 (r_i = 1 / l_ii)
    const double r0 = rsqrt_refined(d00);
    const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
    const double r1 = rsqrt_refined(fma(-l10, l10, d11));
    const double l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
    const double r2 = rsqrt_refined(fma(-l21, l21, fma(-l20, l20, d22)));
    const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
    const double r3 = rsqrt_refined(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33))));
    const double m10 = -r1 * (l10 * r0), m21 = -r2 * (l21 * r1), m32 = -r3 * (l32 * r2);
    const double m20 = -r2 * fma(l21, m10, l20 * r0), m31 = -r3 * fma(l32, m21, l31 * r1);
    const double m30 = -r3 * fma(l32, m20, fma(l31, m10, l30 * r0));
    // the operand Linv16[li][base + lk]: M[li - base][lk] on the block's rows, zero elsewhere
    const int i = li - base;
    const double row0 = lk == 0 ? r0 : 0.0;
    const double row1 = lk == 0 ? m10 : (lk == 1 ? r1 : 0.0);
    const double row2 = lk == 0 ? m20 : (lk == 1 ? m21 : (lk == 2 ? r2 : 0.0));
    const double row3 = lk == 0 ? m30 : (lk == 1 ? m31 : (lk == 2 ? m32 : r3));
    const double aop = i == 0 ? row0 : (i == 1 ? row1 : (i == 2 ? row2 : (i == 3 ? row3 : 0.0)));
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d x = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, pt[b], z, 0, 0, 0);
    const v4d eb = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, E[b], z, 0, 0, 0);
    const bool below = li >= base + 4;
    const double xn = x[b];
    pt[b] = below ? xn : ((i >= 0 && lk <= i) ? xn : 0.0);   // (zero above the diagonal)
    E[b] = eb[b];
    if (b < 3) {
      const double xu = below ? xn : 0.0;
      pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
      E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, eb[b], E, 0, 0, 0);
    }
  }
}

// ---- A'. the same L11 = chol(A11[c0 .. c0 + fw)), ROW-OWNED and barrier-free --------------------------------------------
// Round 3.  In the kernel above every tile column costs two workgroup barriers, and the critical path D(tc) -> O of the
// tile under the diagonal -> U of the next diagonal tile -> D(tc + 1) crosses waves twice; measured 4.6 us per 16 pivots
// of which the D chain is 1.5-1.9.  Here wave r owns tile ROW r (tiles (r, 0..r), in registers) and nothing is a barrier:
//   * wave tc factors its diagonal tile (D, as above) and publishes L_dd^-T (flag);
//   * every wave r > tc then forms its tile of the column, L(r, tc) = A(r, tc) L_dd^-T (O), publishes its 16 rows of the
//     column's strip (a per-row flag), and updates its tiles (r, tc+1 .. r) with the strip rows of the waves above it as
//     they appear (U);
//   * wave tc + 1's only remaining tile is its diagonal one: O, one U, and it is in its own D — ONE cross-wave hand-off
//     (the inverse) per 16 pivots on the critical path, no wait for anybody's trailing updates.
// Strips live in three LDS buffers (column tc in buffer tc mod 3); a wave writes column tc only when every reader of
// column tc - 3 has reported done.  All waits are bounded polls of LDS words written by waves of this workgroup (always
// resident together); a poll that runs out marks the front as failed instead of hanging.
template <int NW>
__global__ void __launch_bounds__(NW * 64) big_diag_rows_kernel(const BigDesc* descs, int c0, int chunk, double* arena,
                                                                DevStatus* status) {
  constexpr int LS = 18;
  constexpr int kRows = NW * 16;
  __shared__ double Ls[3][kRows][LS];   // strips: rows of L of a tile column, 16 columns each
  __shared__ double dinv[kRows];
  __shared__ double ldiag[kRows];
  __shared__ int flagE[NW];             // column tc: the inverse of its diagonal tile is in Xd
  __shared__ int rowCols[NW];           // row r: number of columns whose strip rows it has published
  __shared__ int doneCnt[NW];           // column tc: waves that have finished reading its strip
  __shared__ int sfail;
  extern __shared__ double dyn[];       // Xd: the 16 x 16 inverses [tc][i][17]; then L10 of every 32-tile [kb][16][17]
  const BigDesc d = descs[blockIdx.x];
  const int n = d.N, F = d.F;
  if (c0 >= F) return;
  const int fw = min(chunk, F - c0);
  const int nt16 = (fw + 15) >> 4, nt32 = (fw + T - 1) / T;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* A = arena + d.off;
  double* X = arena + d.xoff;
  double* Xd = dyn;
  double* t10 = dyn + (size_t)nt16 * 16 * 17;
  if (tid < NW) {
    flagE[tid] = 0;
    rowCols[tid] = 0;
    doneCnt[tid] = 0;
  }
  if (tid == 0) sfail = 0;
  const int r = wv;                      // this wave's tile row
  const bool active = r < nt16;
  const int rr = 16 * r + li;            // this lane's row
  v4d acc[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = 16 * j + 4 * q + lk;
      double v = (rr == cc) ? 1.0 : 0.0;  // padding beyond fw: identity
      if (active && j <= r && rr < fw && cc < fw) {
        const int hi = max(rr, cc), lo = min(rr, cc);  // the diagonal tile is kept symmetric
        v = A[(c0 + hi) + (i64)(c0 + lo) * n];
      }
      acc[j][q] = v;
    }
  }
  lds_bar();   // (the flags are zero)
  RST_BEGIN
  int fail = 0;
  // bounded poll of an LDS word another wave of this workgroup will set
  auto wait_ge = [&](int* word, int want) {
    int polls = 0;
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++polls > (1 << 18)) {   // (tens of milliseconds: a logic error, not a slow wave)
        fail = 1;
        break;
      }
    }
  };
  // a finished tile (r, tc): L inside a diagonal 32-tile goes to the square, the rest to the L-panel area; the tile under
  // the diagonal inside a 32-tile also to LDS (the stored 32 x 32 inverses are assembled from it)
  auto store_tile = [&](const v4d& t, int tc) {
    const int jb = 16 * tc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = jb + 4 * q + lk;
      if (rr >= fw || cc >= fw || rr < cc) continue;
      const bool same = (rr >> 5) == (cc >> 5);
      (same ? A : X)[(c0 + rr) + (i64)(c0 + cc) * n] = t[q];
      if (same && r == tc + 1) t10[((size_t)(rr >> 5) * 16 + (rr & 15)) * 17 + (cc & 15)] = t[q];
    }
  };
  if (active) {
    v4d last = {0.0, 0.0, 0.0, 0.0};   // L(r, r - 1): stored after this row's D (the hand-off comes first)
    // ---- the columns left of the diagonal: O, then U of the tiles right of the column.  tc is a compile-time index (the
    //      tiles are registers); a wave leaves the chain of guards at its own row ----
#pragma unroll
    for (int tc = 0; tc < NW - 1; ++tc) {
      if (tc < r) {
        // the strip buffer of column tc was column tc - 3's: every reader of that one must be done
        if (tc >= 3) wait_ge(&doneCnt[tc - 3], nt16 - 1 - (tc - 3));
        RST_ADD(1)
        wait_ge(&flagE[tc], 1);
        RST_ADD(2)
        // O: L(r, tc) = A(r, tc) L_dd^-T
        const double* xd = Xd + (size_t)tc * 16 * 17 + li * 17 + lk;
        v4d nt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) nt = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[4 * s4], acc[tc][s4], nt, 0, 0, 0);
        if (tc + 1 == r) {
          // the critical hand-off: this row's diagonal tile takes its last update straight from the registers (both
          // operands are this wave's own L(r, tc)); everything else of the step waits until D is out
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc[tc + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-nt[s4], nt[s4], acc[tc + 1], 0, 0, 0);
          last = nt;
        }
        // this row's 16 rows of the column's strip
        double(*Lb)[LS] = Ls[tc % 3];
#pragma unroll
        for (int q = 0; q < 4; ++q) Lb[rr][4 * q + lk] = nt[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&rowCols[r], tc + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        RST_ADD(3)
        if (tc + 1 < r) {
          // U: (r, j) -= L(r, tc) L(j, tc)' for j = tc + 1 .. r, the rows above as they appear
          const double* pb = &Lb[rr][lk];
          const double b0 = pb[0], b1 = pb[4], b2 = pb[8], b3 = pb[12];
#pragma unroll
          for (int j = tc + 1; j < NW; ++j) {
            if (j <= r) {
              if (j < r) {
                RST_ADD(6)
                wait_ge(&rowCols[j], tc + 1);
                RST_ADD(5)
              }
              const double* pa = &Lb[16 * j + li][lk];
              const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12];
              acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a2, b2, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a3, b3, acc[j], 0, 0, 0);
            }
          }
          RST_ADD(6)
          store_tile(nt, tc);
          if (lane == 0) __hip_atomic_fetch_add(&doneCnt[tc], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          RST_ADD(4)
        }
      }
    }
    // ---- D: this row's diagonal tile, alone (straight-line: see big_diag_kernel) ----
    {
      const int jb = 16 * r;
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
      v4d pt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < NW; ++j)
        if (j == r) pt = acc[j];
      RST_ADD(7)
      __builtin_amdgcn_s_setprio(3);
      v4d E;
#pragma unroll
      for (int q = 0; q < 4; ++q) E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
#ifndef GSX_D_BLOCK4   // (the four-pivots-at-a-time variant below: measured no faster, see tile_potrf16_b4)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, lkj = j & 3;
        __builtin_amdgcn_sched_barrier(0);
        int lj = li;
        asm volatile("" : "+v"(lj));
        const double dj = readlane_f64(pt[q], lkj * 16 + j);
        const double sj = rsqrt_refined(dj);
        const bool colj = lk == lkj;
        const double xj = pt[q] * sj;
        const double xm = (colj && lj >= j) ? xj : 0.0;
        const double ej = colj ? E[q] * sj : 0.0;
        pt[q] = colj ? xm : pt[q];
        E[q] = colj ? ej : E[q];
        if (j < 15) {
          const double xu = (lj > j) ? xm : 0.0;
          pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
        }
      }
#else
      tile_potrf16_b4(pt, E, li, lk);
#endif
      __builtin_amdgcn_s_setprio(0);
      double* xd = Xd + (size_t)r * 16 * 17;
#pragma unroll
      for (int q = 0; q < 4; ++q) xd[(4 * q + lk) * 17 + li] = E[q];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&flagE[r], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      RST_ADD(0)
      // (off the critical path from here on)
      if (lk == (li & 3)) {
        const int qd = li >> 2;
        const double l = (qd == 0) ? pt[0] : ((qd == 1) ? pt[1] : ((qd == 2) ? pt[2] : pt[3]));
        ldiag[jb + li] = l;
        dinv[jb + li] = 1.0 / l;
        if (jb + li < fw && !(l > 0)) fail = 1;  // non-positive or non-finite pivot (Eigen::LLT NumericalIssue)
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // the inverse's strictly lower part inside its 32-tile, transposed
        const int i = 4 * q + lk, c = li;
        if (i > c && jb + i < fw) A[(c0 + jb + c) + (i64)(c0 + jb + i) * n] = E[q];
      }
      store_tile(pt, r);
      if (r > 0) {
        store_tile(last, r - 1);
        if (lane == 0) __hip_atomic_fetch_add(&doneCnt[r - 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      RST_ADD(4)
    }
  }
  if (fail) sfail = 1;
  lds_bar();

  // ---- (L^-1)' of every diagonal 32-tile into its strictly upper triangle, 1 / L_cc into the L-panel area ----
  for (int c = tid; c < fw; c += NW * 64) X[(c0 + c) + (i64)(c0 + c) * n] = dinv[c];
  for (int kb = wv; kb < nt32; kb += NW) {
    if (fw - kb * T <= 16) continue;  // a single 16-block
    const double* L10 = t10 + (size_t)kb * 16 * 17;
    const double* X00 = Xd + (size_t)(2 * kb) * 16 * 17;
    const double* X11 = Xd + (size_t)(2 * kb + 1) * 16 * 17;
    const int w1 = min(16, fw - kb * T - 16);
    v4d t1 = {0.0, 0.0, 0.0, 0.0}, t2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {  // T[i][jj] = sum_k L10[i][k] X00[k][jj]; lane: t1[q] = T[4 q + lk][li]
      const double a = (li < w1) ? L10[li * 17 + 4 * s4 + lk] : 0.0;
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X00[(4 * s4 + lk) * 17 + li], t1, 0, 0, 0);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-X11[li * 17 + 4 * s4 + lk], t1[s4], t2, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * q + lk;
      if (i < w1) A[(c0 + kb * T + li) + (i64)(c0 + kb * T + 16 + i) * n] = t2[q];
    }
  }
  if (tid == 0) {
    int bad = sfail;
    if (d.parent == kStandaloneFront && c0 + fw == F) {  // (the dense entry: one clique, tested here — cholesky.cpp:145-158)
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
// Right-looking along the row block:  X_ik = T_k (L_kk^-1)'  then  T_k' -= X_ik L_k'k'  for every k' > k.
// The 32 x 32 tiles T_k of the row block stay in matrix-core accumulators; the tiles of L11's column block k (the
// diagonal tile's inverse left by big_diag and the L tiles below it) are staged through LDS, double-buffered: the
// loads of column block k+1 are in flight during step k.  The dependent chain of a step is LDS -> 8 products -> LDS ->
// 8 products; nothing on it touches global memory.
static_assert(kMaxChunk / T == 6, "launch_big_rows dispatches on 1..6 column steps");
constexpr int TS = T * (T + 2);       // an LDS tile: row stride 34 doubles (operand reads conflict-free)

// NK = 32-column steps of the widest chunk of the launch group (a narrower chunk runs its missing steps on tiles of
// zeros: no run-time guards, the whole kernel is straight-line code over register arrays)
template <int NK>
__global__ void __launch_bounds__(256) big_rows_kernel(const BigDesc* descs, int c0, int chunk, double* arena) {
  extern __shared__ double dyn[];
  const BigDesc d = descs[blockIdx.y];
  const int n = d.N, F = d.F;
  if (c0 >= F) return;
  const int fw = min(chunk, F - c0), base = c0 + fw;
  const int nrb = (n - base + T - 1) / T;
  if ((int)blockIdx.x >= nrb) return;
  constexpr int nk = NK;
  double* Lb = dyn;                                   // [2][nk] tiles: column block k = inverse of L_kk, then L_k'k, k' > k
  double(*Tt)[T + 2] = (double(*)[T + 2])(dyn + (size_t)2 * nk * TS);
  double(*Xt)[T + 2] = (double(*)[T + 2])(dyn + (size_t)(2 * nk + 1) * TS);
  double* A = arena + d.off;
  double* X = arena + d.xoff;
  const Quad L;
  const int tid = threadIdx.x, row = L.r0 + L.li;
  // element e = tid + 256 q of a tile as (lo = e & 31, hi = e >> 5): the fast index of the global read
  const int elo = tid & 31, ehi = tid >> 5;

  // column block kc of L11 into registers (4 doubles a thread per tile) / from registers into LDS buffer kc & 1
  double stg[NK][4];
  auto fetch = [&](int kc) {
    const int ck = c0 + kc * T, wk = min(T, fw - kc * T);  // (wk <= 0 beyond a narrower chunk: tiles of zeros)
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // tile 0: (L_kk^-1)[c][kk], kk <= c — strictly upper triangle of the diagonal tile
      const int kk = elo, c = ehi + 8 * q;  //   (transposed), 1 / L_cc on the L-panel area's diagonal
      double v = 0.0;
      if (c < wk && kk < c) v = A[(ck + kk) + (i64)(ck + c) * n];
      else if (c < wk && kk == c) v = X[(ck + c) + (i64)(ck + c) * n];
      stg[0][q] = v;
    }
#pragma unroll
    for (int t = 1; t < NK; ++t) {  // tile t: L[rows of 32-block kc + t][columns of block kc]
      if (kc + t < nk) {
        const int rr = (kc + t) * T;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = elo, kk = ehi + 8 * q;
          stg[t][q] = (rr + c < fw && kk < wk) ? X[(c0 + rr + c) + (i64)(ck + kk) * n] : 0.0;
        }
      }
    }
  };
  auto stash = [&](int kc) {
    double* buf = Lb + (size_t)(kc & 1) * nk * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[(ehi + 8 * q) * (T + 2) + elo] = stg[0][q];  // [c][kk]
#pragma unroll
    for (int t = 1; t < NK; ++t) {
      if (kc + t < nk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) buf[(size_t)t * TS + elo * (T + 2) + ehi + 8 * q] = stg[t][q];  // [c][kk]
      }
    }
  };

  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int ri = base + T * rb, hi = min(T, n - ri);
    fetch(0);
    v4d tk[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int wk = min(T, fw - k * T);  // (<= 0 beyond the chunk: a tile of zeros)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = L.cq0 + 4 * q + L.lk;
        tk[k][q] = (row < hi && c < wk) ? A[(ri + row) + (i64)(c0 + k * T + c) * n] : 0.0;
      }
    }
    lds_bar();  // (a previous row block's reads of the buffers are done)
    stash(0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int wk = min(T, fw - k * T);
      const double* buf = Lb + (size_t)(k & 1) * nk * TS;
      if (k + 1 < nk) fetch(k + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) Tt[row][L.cq0 + 4 * q + L.lk] = tk[k][q];
      lds_bar();
      // X[r][c] = sum_{kk <= c} T[r][kk] Linv[c][kk]: columns of the first half only need kk < 16 (wave-uniform)
      v4d x = {0.0, 0.0, 0.0, 0.0};
      {
        const double* pa = buf + (L.cq0 + L.li) * (T + 2) + L.lk;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < 4 || L.cq0 != 0) x = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s], Tt[row][4 * s + L.lk], x, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = L.cq0 + 4 * q + L.lk;
        Xt[row][cc] = x[q];
        if (row < hi && cc < wk) X[(ri + row) + (i64)(c0 + k * T + cc) * n] = x[q];
      }
      lds_bar();
#pragma unroll
      for (int k2 = k + 1; k2 < NK; ++k2) {  // T_k2[r][c] -= sum_kk X[r][kk] L[32 k2 + c][32 k + kk]
        const double* pa = buf + (size_t)(k2 - k) * TS + (L.cq0 + L.li) * (T + 2) + L.lk;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          tk[k2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * s], Xt[row][4 * s + L.lk], tk[k2], 0, 0, 0);
      }
      if (k + 1 < nk) stash(k + 1);
    }
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
  // the operands of the whole chunk (up to 48 k-steps) are requested together, straight-line: the kernel is ONE memory
  // round trip (a loop that waits for its loads pays a round trip, > 1 us, per iteration)
  {
    constexpr int kSteps = kMaxChunk / 4;
    double a[kSteps], b[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const bool in = 4 * s + L.lk < fw;
      a[s] = (ain && in) ? pa[(i64)(4 * s) * n] : 0.0;
      b[s] = (bin && in) ? pb[(i64)(4 * s) * n] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < kSteps; ++s)
      if (4 * s < fw) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
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
  printf("[bstamp] %s: diag prologue %llu | column setup %llu | D %llu | bar1 %llu | O %llu | store %llu | bar2 %llu | U %llu | "
         "inverse %llu  (cycles, wave 0 of block 0)\n", what, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);

  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_bstamp), z, sizeof(z));
  unsigned long long rs[16][8];
  hipMemcpyFromSymbol(rs, HIP_SYMBOL(g_rstamp), sizeof(rs));
  for (int w = 0; w < 12; ++w)
    printf("[rstamp] wave %2d: D %8llu | wait buffer %8llu | wait E %8llu | O+publish %8llu | store %8llu | wait rows %8llu | U %8llu | top %8llu\n",
           w, rs[w][0], rs[w][1], rs[w][2], rs[w][3], rs[w][4], rs[w][5], rs[w][6], rs[w][7]);
  unsigned long long rz[16][8] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(g_rstamp), rz, sizeof(rz));
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
  if (!attr) {  // (static + dynamic LDS exceeds the 64 KB default)
    hipFuncSetAttribute((const void*)big_diag_kernel<12, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    hipFuncSetAttribute((const void*)big_diag_kernel<12, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    hipFuncSetAttribute((const void*)big_diag_kernel<8, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    hipFuncSetAttribute((const void*)big_diag_kernel<8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    hipFuncSetAttribute((const void*)big_diag_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kTilesLds);
    attr = true;
  }
  const int c0 = round * plan.chunk, fw = plan.fw[round];
  const size_t lds = lds_for(fw);
  const int nt16 = (fw + 15) / 16;
  static const bool v1 = std::getenv("GSX_DIAG_V1") != nullptr;
  if (!v1) {   // the row-owned, barrier-free kernel
    static bool attr2 = false;
    const auto lds2_for = [](int w) { return (size_t)(((w + 15) / 16) * 16 * 17 + ((w + T - 1) / T) * 16 * 17) * sizeof(double); };
    if (!attr2) {
      hipFuncSetAttribute((const void*)big_diag_rows_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2_for(kMaxChunk));
      hipFuncSetAttribute((const void*)big_diag_rows_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2_for(kMaxChunk));
      hipFuncSetAttribute((const void*)big_diag_rows_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2_for(kMaxChunk));
      attr2 = true;
    }
    const size_t lds2 = lds2_for(fw);
    if (nt16 <= 4) big_diag_rows_kernel<4><<<count, 256, lds2, st>>>(descs, c0, plan.chunk, arena, status);
    else if (nt16 <= 8) big_diag_rows_kernel<8><<<count, 512, lds2, st>>>(descs, c0, plan.chunk, arena, status);
    else big_diag_rows_kernel<12><<<count, 768, lds2, st>>>(descs, c0, plan.chunk, arena, status);
    return;
  }
  // tiles: nt16 (nt16 + 1) / 2 over the waves, at least as many waves as tile rows
  if (nt16 <= 4) big_diag_kernel<4, 3><<<count, 256, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else if (nt16 <= 6) big_diag_kernel<8, 3><<<count, 512, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else if (nt16 <= 8) big_diag_kernel<8, 5><<<count, 512, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else if (nt16 <= 10) big_diag_kernel<12, 5><<<count, 768, lds, st>>>(descs, c0, plan.chunk, arena, status);
  else big_diag_kernel<12, 7><<<count, 768, lds, st>>>(descs, c0, plan.chunk, arena, status);
}

template <int NK>
static void launch_rows_nk(const BigDesc* descs, int count, int gx, int c0, int chunk, double* arena, hipStream_t st) {
  static bool attr = false;
  constexpr size_t lds = (size_t)(2 * NK + 2) * TS * sizeof(double);
  if (!attr) {
    hipFuncSetAttribute((const void*)big_rows_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  big_rows_kernel<NK><<<dim3(gx, count), 256, lds, st>>>(descs, c0, chunk, arena);
}

void launch_big_rows(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, hipStream_t st) {
  if (!count || plan.rb[round] <= 0) return;
  // one workgroup per 32-row block; with many fronts in the group a workgroup takes several row blocks of its front in
  // turn (the column blocks of L11 fill most of the LDS: about one workgroup per CU)
  int gx = plan.rb[round];
  if ((long)gx * count > 1024) gx = std::max(1, std::min(gx, 1024 / count));
  const int c0 = round * plan.chunk;
  switch ((plan.fw[round] + T - 1) / T) {
    case 1: launch_rows_nk<1>(descs, count, gx, c0, plan.chunk, arena, st); break;
    case 2: launch_rows_nk<2>(descs, count, gx, c0, plan.chunk, arena, st); break;
    case 3: launch_rows_nk<3>(descs, count, gx, c0, plan.chunk, arena, st); break;
    case 4: launch_rows_nk<4>(descs, count, gx, c0, plan.chunk, arena, st); break;
    case 5: launch_rows_nk<5>(descs, count, gx, c0, plan.chunk, arena, st); break;
    default: launch_rows_nk<6>(descs, count, gx, c0, plan.chunk, arena, st); break;
  }
}

void launch_big_schur(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, hipStream_t st) {
  if (!count || plan.pairs[round] <= 0) return;
  big_schur_kernel<<<dim3(plan.pairs[round], count), 256, 0, st>>>(descs, round * plan.chunk, plan.chunk, arena);
}

// ---- back-substitution of the blocked fronts of a level ----------------------------------------------------------------------
// OptimizeClique (gtsam/linear/linearAlgorithms-inst.h:49-117) for a front kept in the blocked layout:
//   L11' x_F = d - L21' x_S.
// One workgroup per front.  Everything the sequential part needs is brought on chip ONCE, with all loads in flight
// together: the strictly lower 32 x 32 tiles of L11 and the inverses of its diagonal tiles (left by big_diag) go to LDS
// while the waves form y = d - L21' x_S column by column straight from the L panel (one pass over L21, the bulk of the
// bytes).  The chain itself — for the column blocks from last to first: x_p = (L_pp^-1)' y_p, then y_k -= L_pk' x_p
// for every k < p — then runs out of LDS: two barriers and ~64 multiply-adds per thread a block.
namespace {
constexpr int kBsThreads = 1024;
constexpr int kBsMaxSep = 320;  // separator rows (+ rhs) the kernel's straight-line GEMV covers
__global__ void __launch_bounds__(kBsThreads) backsolve_big_kernel(DevSymbolic S, const int* ids, const double* arena,
                                                                   double* delta, DevStatus* status) {
  extern __shared__ double sm[];
  const int f = ids[blockIdx.x];
  if (wildfire_skip(S, f, threadIdx.x == 0)) return;  // a clique no change reaches
  const int n = S.fr_N[f], F = S.fr_F[f];
  const double* A = arena + S.fr_off[f];
  const double* X = A + big_panel_offset(n);
  const int* gi = S.gidx + S.gidx_ptr[f];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = (F + T - 1) / T;
  // LDS: xs[n] (solution: frontal + separator), y[F rounded], inverse tiles [nk] packed by rows ((L^-1)[r][c], r >= c, at
  // r (r + 1) / 2 + c), strictly lower tiles [(p, k), k < p][32][32] (L[32 p + r][32 k + c])
  double* xs = sm;
  double* y = xs + ((n + 1) & ~1);
  double* xp = y + nk * T;        // the current block's solution (32)
  double* inv = xp + T;
  constexpr int kTri = T * (T + 1) / 2;
  double* low = inv + (size_t)nk * kTri;
  for (int r = F + tid; r < n - 1; r += kBsThreads) xs[r] = delta[gi[r]];
  // tiles -> LDS, every load of a thread in flight before its first LDS store (a load -> store loop would pay one memory
  // round trip per tile).  The inverse sits transposed in the strictly upper triangle of the diagonal tile
  // ((L^-1)[r][c] at row c, column r): the fast index of its read is c.
  {
    constexpr int kMaxK = kMaxChunk / T, kMaxLow = kMaxK * (kMaxK - 1) / 2;
    double vi[kMaxK], vl[kMaxLow];
#pragma unroll
    for (int q = 0; q < kMaxK; ++q) {  // thread tid: element (c = tid & 31, r = tid >> 5) of diagonal tile q
      const int c = tid & 31, r = tid >> 5;
      const int gr = q * T + r, gc = q * T + c;
      vi[q] = 0.0;
      if (q < nk && r >= c && gr < F) vi[q] = (r > c) ? A[gc + (i64)gr * n] : X[gr + (i64)gr * n];  // (diagonal: 1 / L_cc)
    }
    int tp = 1, tk = 0;  // tile (p, k), k < p, in the order t = p (p - 1) / 2 + k
#pragma unroll
    for (int t = 0; t < kMaxLow; ++t) {  // element (r = tid & 31, c = tid >> 5): rows below a diagonal tile, L-panel area
      const int r = tid & 31, c = tid >> 5;
      const int gr = tp * T + r, gc = tk * T + c;
      vl[t] = (t < nk * (nk - 1) / 2 && gr < F) ? X[gr + (i64)gc * n] : 0.0;
      if (++tk == tp) ++tp, tk = 0;
    }
#pragma unroll
    for (int q = 0; q < kMaxK; ++q) {
      const int c = tid & 31, r = tid >> 5;
      if (q < nk && r >= c) inv[(size_t)q * kTri + r * (r + 1) / 2 + c] = vi[q];
    }
#pragma unroll
    for (int t = 0; t < kMaxLow; ++t)
      if (t < nk * (nk - 1) / 2) low[(size_t)t * T * T + (tid & 31) * T + (tid >> 5)] = vl[t];
  }
  __syncthreads();  // xs (separator part) is in place
  // y[c] = d[c] - sum_{r >= F} L[r][c] xs[r]: wave w takes the columns w, w + 16, ...; lanes over the rows.  Straight-line
  // code: the loads of six columns (up to 30 a lane) are all in flight before the first use — a loop that waits for its
  // load costs a memory round trip (> 1 us) per iteration.
  {
    constexpr int kRC = kBsMaxSep / 64;  // row chunks of 64
    double xr[kRC];
#pragma unroll
    for (int i = 0; i < kRC; ++i) {
      const int r = F + lane + 64 * i;
      xr[i] = (r < n - 1) ? xs[r] : 0.0;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double v[6][kRC], dv[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c = wave + 16 * (6 * half + j);
        const double* col = X + (i64)c * n;
#pragma unroll
        for (int i = 0; i < kRC; ++i) {
          const int r = F + lane + 64 * i;
          v[j][i] = (c < F && r < n - 1) ? col[r] : 0.0;
        }
        dv[j] = (c < F && lane == 0) ? col[n - 1] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c = wave + 16 * (6 * half + j);
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < kRC; ++i) acc += v[j][i] * xr[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0 && c < nk * T) y[c] = dv[j] - acc;   // (columns beyond F: 0)
      }
    }
  }
  __syncthreads();
  for (int p = nk - 1; p >= 0; --p) {
    // x_p = (L_pp^-1)' y_p:  x[c] = sum_{r >= c} (L^-1)[r][c] y[r]   (operands first, then the sum: LDS latency once)
    if (tid < T) {
      const double* iv = inv + (size_t)p * kTri;
      double a[T], b[T];
#pragma unroll
      for (int r = 0; r < T; ++r) {
        a[r] = (r >= tid) ? iv[r * (r + 1) / 2 + tid] : 0.0;
        b[r] = y[p * T + r];
      }
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < T; ++r) acc = fma(a[r], b[r], acc);
      xp[tid] = acc;
      if (p * T + tid < F) xs[p * T + tid] = acc;
    }
    lds_bar();
    // y_k -= L_pk' x_p for every k < p: thread c of the 32 p columns left of the block
    if (tid < p * T) {
      const int k = tid >> 5, c = tid & 31;
      const double* lt = low + (size_t)(p * (p - 1) / 2 + k) * T * T;
      double a[T], b[T];
#pragma unroll
      for (int r = 0; r < T; ++r) {
        a[r] = lt[r * T + c];
        b[r] = xp[r];
      }
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < T; ++r) acc = fma(a[r], b[r], acc);
      y[tid] -= acc;
    }
    lds_bar();
  }
  int bad = 0;
  for (int r = tid; r < F; r += kBsThreads) {
    const double x = xs[r];
    delta[gi[r]] = x;
    if (!isfinite(x)) bad = 1;
  }
  if (bad) atomicAdd(&status->n_nonfinite, 1);
}
}  // namespace

// The LDS-class fronts (n <= kSmallMaxN) with at most 32 frontal columns — every level of a pose graph below the
// blocked fronts: a level used to cost 20-25 us, a chain of dependent memory round trips (gather, then a loop of dot
// products, then a substitution with a division per pivot).  Here every global load of the front — indices, the
// triangle, the whole L21 panel, the right-hand side — is issued before anything waits (two round trips: the indices,
// then the parents' solution), the dot products are 18 multiply-adds per thread out of registers, and the 32 pivots
// are a straight-line chain of lane reads with the reciprocals taken beforehand.
namespace {
constexpr int kBsSmallF = 32, kBsSmallSep = 144;
// COH (the tree kernel below): the parents' solution is read, and this clique's is written, with agent-scope accesses —
// the value may have been stored a moment ago by a workgroup on another CU / XCD inside the same launch
template <bool COH>
__device__ __forceinline__ double delta_load(const double* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH>
__device__ __forceinline__ void delta_store(double* p, double v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// LDS pool of one clique (doubles): xs[kBsSmallSep] | part[2][64] | three 32 x 33 tiles; the panel loop of a tall
// (medium) clique keeps its kMedMaxN solution entries where the second and third tile would be
constexpr int kBsTile = kBsSmallF * (kBsSmallF + 1);
constexpr int kBsPool = kBsSmallSep + 4 * kBsSmallF + 3 * kBsTile;
static_assert(2 * kBsTile >= kMedMaxN, "backsolve_panels_body: xs of a medium clique");
template <bool COH>
__device__ __forceinline__ void backsolve_small_body(const DevSymbolic& S, const int f, const double* arena, double* delta,
                                                     DevStatus* status, double* pool) {
  double* xs = pool;
  double(*part)[kBsSmallF] = (double(*)[kBsSmallF])(pool + kBsSmallSep);
  double(*tile)[kBsSmallF + 1] = (double(*)[kBsSmallF + 1])(pool + kBsSmallSep + 4 * kBsSmallF);
  const int n = S.fr_N[f], F = S.fr_F[f], nS = n - 1 - F;
  const double* A = arena + S.fr_off[f];
  const int* gi = S.gidx + S.gidx_ptr[f];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid < nS ? gi[F + tid] : -1;
  const int gf = tid < F ? gi[tid] : -1;
  double tv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, r = e & 31, c = e >> 5;
    tv[q] = (r < F && c <= r) ? A[r + (i64)c * n] : (r == c ? 1.0 : 0.0);
  }
  // L21: the wave pair (wave & 1) takes 16 columns, the pair (wave >> 1) every other group of four rows; a lane reads
  // four consecutive rows' worth with its three neighbours (lane >> 4)
  constexpr int kSteps = kBsSmallSep / 8;
  const int cg = 16 * (wave & 1) + (lane & 15), h = wave >> 1, rq = lane >> 4;
  double lv[kSteps];
  const double* col = A + (i64)cg * n + F + rq;
#pragma unroll
  for (int k = 0; k < kSteps; ++k) {
    const int r0 = 4 * (h + 2 * k);
    lv[k] = (cg < F && r0 + rq < nS) ? col[r0] : 0.0;
  }
  const double dv = tid < F ? A[(n - 1) + (i64)tid * n] : 0.0;
  if (tid < kBsSmallSep) xs[tid] = g >= 0 ? delta_load<COH>(delta + g) : 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q;
    tile[e & 31][e >> 5] = tv[q];
  }
  lds_bar();
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kSteps; ++k) acc = fma(lv[k], xs[4 * (h + 2 * k) + rq], acc);
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (rq == 0) part[h][cg] = acc;
  lds_bar();
  if (wave != 0) return;
  const int l = lane & 31;
  double lc[kBsSmallF];
#pragma unroll
  for (int c = 0; c < kBsSmallF; ++c) lc[c] = tile[c][l];
  const double dinv = 1.0 / tile[l][l];
  double yr = lane < F ? dv - part[0][l] - part[1][l] : 0.0;
#pragma unroll
  for (int c = kBsSmallF - 1; c >= 0; --c) {
    const double xc = readlane_f64(yr, c) * readlane_f64(dinv, c);
    yr = lane == c ? xc : (lane < c ? fma(-lc[c], xc, yr) : yr);
  }
  if (lane < F) {
    delta_store<COH>(delta + gf, yr);
    if (!isfinite(yr)) atomicAdd(&status->n_nonfinite, 1);
  }
}
__global__ void __launch_bounds__(256) backsolve_small_kernel(DevSymbolic S, const int* ids, const double* arena,
                                                              double* delta, DevStatus* status) {
  __shared__ double pool[kBsSmallSep + 4 * kBsSmallF + kBsTile];
  const int f = ids[blockIdx.x];
  if (wildfire_skip(S, f, threadIdx.x == 0)) return;  // a clique no change reaches
  backsolve_small_body<false>(S, f, arena, delta, status, pool);
}
// The same for 32 < F <= 64 (the unamalgamated nested-dissection leaves of a pose graph run to ~60 frontal scalars): two
// column blocks.  L21 is read in two passes of 32 columns, the chain solves block 1, takes its part out of block 0's
// right-hand side through the 32 x 32 off-diagonal tile, and solves block 0.
template <bool COH>
__device__ __forceinline__ void backsolve_small2_body(const DevSymbolic& S, const int f, const double* arena, double* delta,
                                                      DevStatus* status, double* pool) {
  constexpr int B = kBsSmallF;
  double* xs = pool;
  double(*part)[2 * B] = (double(*)[2 * B])(pool + kBsSmallSep);
  double(*t00)[B + 1] = (double(*)[B + 1])(pool + kBsSmallSep + 4 * B);
  double(*t11)[B + 1] = (double(*)[B + 1])(pool + kBsSmallSep + 4 * B + kBsTile);
  double(*t10)[B + 1] = (double(*)[B + 1])(pool + kBsSmallSep + 4 * B + 2 * kBsTile);   // t10[r][c] = L[32 + r][c]
  const int n = S.fr_N[f], F = S.fr_F[f], nS = n - 1 - F;
  const double* A = arena + S.fr_off[f];
  const int* gi = S.gidx + S.gidx_ptr[f];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid < nS ? gi[F + tid] : -1;
  const int gf = tid < F ? gi[tid] : -1;
  double v00[4], v11[4], v10[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, r = e & 31, c = e >> 5;
    v00[q] = (r < F && c <= r) ? A[r + (i64)c * n] : (r == c ? 1.0 : 0.0);
    v11[q] = (B + r < F && c <= r) ? A[(B + r) + (i64)(B + c) * n] : (r == c ? 1.0 : 0.0);
    v10[q] = (B + r < F) ? A[(B + r) + (i64)c * n] : 0.0;
  }
  constexpr int kSteps = kBsSmallSep / 8;
  const int cl = 16 * (wave & 1) + (lane & 15), h = wave >> 1, rq = lane >> 4;
  double lv[2][kSteps];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int cg = B * p + cl;
    const double* col = A + (i64)cg * n + F + rq;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
      const int r0 = 4 * (h + 2 * k);
      lv[p][k] = (cg < F && r0 + rq < nS) ? col[r0] : 0.0;
    }
  }
  const double dv = tid < F ? A[(n - 1) + (i64)tid * n] : 0.0;   // (F <= 64: wave 0, lane = column)
  if (tid < kBsSmallSep) xs[tid] = g >= 0 ? delta_load<COH>(delta + g) : 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q;
    t00[e & 31][e >> 5] = v00[q];
    t11[e & 31][e >> 5] = v11[q];
    t10[e & 31][e >> 5] = v10[q];
  }
  lds_bar();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < kSteps; ++k) acc = fma(lv[p][k], xs[4 * (h + 2 * k) + rq], acc);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (rq == 0) part[h][B * p + cl] = acc;
  }
  lds_bar();
  if (wave != 0) return;
  // lane c (< 64) holds the right-hand side of column c; the two 32-pivot chains run on the lanes of their block
  const int l = lane & 31;
  double y = lane < F ? dv - part[0][lane] - part[1][lane] : 0.0;
  double lc[B];
  {  // block 1: lanes 32..63
#pragma unroll
    for (int c = 0; c < B; ++c) lc[c] = t11[c][l];
    const double dinv = 1.0 / t11[l][l];
#pragma unroll
    for (int c = B - 1; c >= 0; --c) {
      const double xc = readlane_f64(y, B + c) * readlane_f64(dinv, c);
      y = lane == B + c ? xc : ((lane >= B && lane < B + c) ? fma(-lc[c], xc, y) : y);
    }
  }
  // y_0 -= L10' x_1: lane c < 32 sums over the 32 rows of the off-diagonal tile
  {
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < B; ++r) acc = fma(t10[r][l], readlane_f64(y, B + r), acc);
    if (lane < B) y -= acc;
  }
  {  // block 0: lanes 0..31
#pragma unroll
    for (int c = 0; c < B; ++c) lc[c] = t00[c][l];
    const double dinv = 1.0 / t00[l][l];
#pragma unroll
    for (int c = B - 1; c >= 0; --c) {
      const double xc = readlane_f64(y, c) * readlane_f64(dinv, c);
      y = lane == c ? xc : (lane < c ? fma(-lc[c], xc, y) : y);
    }
  }
  if (lane < F) {
    delta_store<COH>(delta + gf, y);
    if (!isfinite(y)) atomicAdd(&status->n_nonfinite, 1);
  }
}
__global__ void __launch_bounds__(256) backsolve_small2_kernel(DevSymbolic S, const int* ids, const double* arena,
                                                               double* delta, DevStatus* status) {
  __shared__ double pool[kBsPool];
  const int f = ids[blockIdx.x];
  if (wildfire_skip(S, f, threadIdx.x == 0)) return;  // a clique no change reaches
  backsolve_small2_body<false>(S, f, arena, delta, status, pool);
}
// Any LDS-class clique (F <= kSmallMaxN): the 32-column panel loop of backsolve_kernel (kernels.hip) on the same pool
template <bool COH>
__device__ __forceinline__ void backsolve_panels_body(const DevSymbolic& S, const int f, const double* arena, double* delta,
                                                      DevStatus* status, double* pool) {
  constexpr int B = kBsSmallF;
  double* y = pool + kBsSmallSep;                      // B
  double(*tile)[B + 1] = (double(*)[B + 1])(pool + kBsSmallSep + 4 * B);
  double* xs = pool + kBsSmallSep + 4 * B + kBsTile;   // n - 1 <= kMedMaxN entries: frontal + separator
  const int n = S.fr_N[f], F = S.fr_F[f];
  const double* A = arena + S.fr_off[f];
  const int* gi = S.gidx + S.gidx_ptr[f];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  for (int r = F + tid; r < n - 1; r += nt) xs[r] = delta_load<COH>(delta + gi[r]);
  lds_bar();
  const int nblk = (F + B - 1) / B;
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int c0 = kb * B, w = min(B, F - c0);
    double treg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * 256, r = e & 31, c = e >> 5;
      treg[q] = (r < w && c <= r) ? A[(c0 + r) + (i64)(c0 + c) * n] : (r == c ? 1.0 : 0.0);
    }
    for (int c = 2 * wave; c < w; c += 2 * nw) {  // two columns per wave: independent loads and reductions
      const double* col0 = A + (i64)(c0 + c) * n;
      const bool two = c + 1 < w;
      const double* col1 = two ? col0 + n : col0;
      double acc0 = 0, acc1 = 0;
      for (int r = c0 + w + lane; r < n - 1; r += 64) {
        const double x = xs[r];
        acc0 += col0[r] * x;
        acc1 += col1[r] * x;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        acc0 += __shfl_down(acc0, o, 64);
        acc1 += __shfl_down(acc1, o, 64);
      }
      if (lane == 0) {
        y[c] = col0[n - 1] - acc0;
        if (two) y[c + 1] = col1[n - 1] - acc1;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * 256;
      tile[e & 31][e >> 5] = treg[q];
    }
    lds_bar();
    if (wave == 0) {  // solve tile' x = y backwards; lane r holds y_r
      double yr = (lane < w) ? y[lane] : 0.0;
      for (int c = w - 1; c >= 0; --c) {
        const double xc = __shfl(yr, c, 64) / tile[c][c];
        if (lane == c) yr = xc;
        else if (lane < c) yr -= tile[c][lane] * xc;
      }
      if (lane < w) xs[c0 + lane] = yr;
    }
    lds_bar();
  }
  int bad = 0;
  for (int r = tid; r < F; r += nt) {
    const double x = xs[r];
    delta_store<COH>(delta + gi[r], x);
    if (!isfinite(x)) bad = 1;
  }
  if (bad) atomicAdd(&status->n_nonfinite, 1);
}

// ---------------------------------------------------------------------------------------------
// backsolve_tree: the tree fronts (Symbolic::tree_*) of ALL levels in one launch, top-down (the pre-order of
// gtsam/linear/linearAlgorithms-inst.h:49-117 without level barriers).  A clique may be solved as soon as its parent is:
// the workgroup that has solved a clique goes on with its first tree child itself and publishes the others in a ready
// list; a workgroup without work takes the next TICKET (in order: 0, 1, 2, ...), waits until the list entry of that
// ticket exists, and solves it.  Tickets below n_roots are the cliques whose parent is not a tree front (solved by the
// level launches before this kernel); the number of tickets = n_roots + the number of non-first children, known to the
// host.  Why this cannot hang:
// tickets are handed out in order to running workgroups only, so every published entry is claimed by a workgroup that is
// running, and a workgroup that is solving never waits — if all running workgroups waited, every published clique would
// be solved already and all its children published, i.e. the list would be complete.  (A bounded number of polls guards
// against a corrupt list all the same: the kernel then gives up and reports a non-finite solution.)
// Solutions cross workgroups inside the launch: agent-scope stores / loads of delta (COH above), entries and tickets
// through agent-scope atomics.  An entry carries the launch's epoch, so the list needs no clearing between launches.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) backsolve_tree_kernel(DevSymbolic S, BacksolveTreeArgs Q, const double* arena,
                                                             double* delta, DevStatus* status) {
  __shared__ double pool[kBsPool];
  __shared__ int s_f;
  const int tid = threadIdx.x;
  for (;;) {
    if (tid == 0) {
      const int t = atomicAdd(Q.head, 1);
      int f = -1;
      if (t < Q.n_roots) {
        f = Q.roots[t];
      } else if (t < Q.n_tickets) {
        const unsigned long long want = (unsigned long long)Q.epoch << 32;
        const unsigned long long* slot = Q.ready + (t - Q.n_roots);
        int polls = 0;
        for (;;) {
          const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((v >> 32 << 32) == want) {
            f = (int)(v & 0xffffffffu);
            break;
          }
          if (++polls > (1 << 20)) {  // (~a second: the list is corrupt — give up loudly)
            atomicAdd(&status->n_nonfinite, 1);
            f = -2;
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      s_f = f;
    }
    __syncthreads();
    int f = s_f;
    if (f < 0) return;   // (uniform: the list is exhausted)
    while (f >= 0) {
      const int F = S.fr_F[f];
      const bool low = S.fr_N[f] - 2 <= kBsSmallSep;   // (the two straight-line kernels cover 144 separator rows)
      if (low && F <= kBsSmallF) backsolve_small_body<true>(S, f, arena, delta, status, pool);
      else if (low && F <= 2 * kBsSmallF) backsolve_small2_body<true>(S, f, arena, delta, status, pool);
      else backsolve_panels_body<true>(S, f, arena, delta, status, pool);
      __syncthreads();  // every store of the solution has drained (s_waitcnt vmcnt(0) before the barrier)
      // the first tree child (the deepest subtree: the host sorted them) is solved by this workgroup right away, the
      // others are published: one claim of a range of entries, then one entry per lane
      const int c0 = Q.child_ptr[f], nc = Q.child_ptr[f + 1] - c0;
      if (nc > 1 && tid < 64) {
        int base = 0;
        if (tid == 0) base = atomicAdd(Q.tail, nc - 1);
        base = __shfl(base, 0, 64);
        for (int k = 1 + tid; k < nc; k += 64)
          __hip_atomic_store(Q.ready + base + k - 1, ((unsigned long long)Q.epoch << 32) | (unsigned)Q.children[c0 + k],
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      f = nc > 0 ? Q.children[c0] : -1;
    }
  }
}
}  // namespace

void launch_backsolve_tree(const DevSymbolic& S, const BacksolveTreeArgs& Q, const double* arena, double* delta,
                           DevStatus* status, hipStream_t st) {
  if (Q.n_tickets <= 0) return;
  const int grid = std::min(Q.n_tickets, 256 * 5);
  backsolve_tree_kernel<<<grid, 256, 0, st>>>(S, Q, arena, delta, status);
}

bool backsolve_small_fits(int max_n, int max_F) { return max_F <= 2 * kBsSmallF && max_n <= kSmallMaxN && max_n - 2 <= kBsSmallSep; }

void launch_backsolve_small(const DevSymbolic& S, const int* ids, int count, int max_F, const double* arena, double* delta,
                            DevStatus* status, hipStream_t st) {
  if (!count) return;
  if (max_F <= kBsSmallF) backsolve_small_kernel<<<count, 256, 0, st>>>(S, ids, arena, delta, status);
  else backsolve_small2_kernel<<<count, 256, 0, st>>>(S, ids, arena, delta, status);
}

// LDS the kernel needs for a front with F frontal columns and n rows; 0 when it does not fit (the caller keeps the
// generic back-substitution kernel for such a level)
size_t backsolve_big_lds(int max_n, int max_F, int max_sep_rows) {
  if (max_sep_rows > kBsMaxSep || max_F > kMaxChunk) return 0;
  const size_t nk = (size_t)(max_F + T - 1) / T;
  const size_t doubles = (((size_t)max_n + 1) & ~(size_t)1) + nk * T + T + nk * (T * (T + 1) / 2) + nk * (nk - 1) / 2 * T * T;
  const size_t bytes = doubles * sizeof(double);
  return bytes <= 160 * 1024 - 256 ? bytes : 0;
}

void launch_backsolve_big(const DevSymbolic& S, const int* ids, int count, int max_n, int max_F, const double* arena,
                          double* delta, DevStatus* status, hipStream_t st) {
  if (!count) return;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)backsolve_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr = true;
  }
  backsolve_big_kernel<<<count, kBsThreads, backsolve_big_lds(max_n, max_F, 0), st>>>(S, ids, arena, delta, status);
}

}  // namespace gsx
