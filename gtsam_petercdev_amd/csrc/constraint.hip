// constraint.hip — hard constraints inside the multifrontal factorization (product; gfx950).
//
// The reference sends a clique that holds a zero-sigma row through EliminateQR with Constrained::QR
// (gtsam/linear/HessianFactor.cpp:538-551, JacobianFactor.cpp:804-842, NoiseModel.cpp:503-620): column by column, a
// constraint row with an entry in the column becomes the conditional of that scalar — x_j = (d - sum c_o x_o) / c_j,
// sigma 0 — and is substituted into every other row; the other columns are weighted Gram-Schmidt steps.
//
// Here the fronts are Hessians (J'J) and the elimination is a blocked Cholesky, so the constraint rows of a front are
// turned into an UNCONSTRAINED front with equivalent conditionals (the same solution for the frontal variables given the
// separator) and the same Schur complement, which the blocked kernels then factor as they factor every other front.  With y = (x, -1) (the rhs row of a front is its last row), M the front's
// augmented Hessian and C y = 0 the rows:
//   1. constraint_reduce_kernel: Gauss-Jordan on C with pivots in FRONTAL columns only (complete pivoting among them; an
//      entry counts if it exceeds 1e-9 — check_if_constraint, NoiseModel.cpp:483-501).  Pivot columns P, the others
//      O (separator and rhs included):  y_P = -R y_O.  Rows that found no frontal pivot go to the front where the first
//      of their variables is frontal (symbolic.cpp).
//   2. the quadratic restricted to the constraint is  1/2 y_O' (E'ME) y_O  with E = [-R; I].  The front written back is
//          M' = [ I    R          ]      i.e.   1/2 y_O' (E'ME) y_O + 1/2 |y_P + R y_O|^2 :
//               [ R'   R'R + E'ME ]
//      an ordinary positive definite front whose minimiser over y_P is the constraint, and whose Schur complement
//      after ANY elimination order of the frontal columns is that of the constrained problem.  constraint_apply_kernel
//      writes it (lower triangle), from B = M_P., Z = M_PP R + R - B:   M'_OO = M_OO + R'Z - B'R.
// The penalty rows mu c'c that the constraint rows also leave in J'J (problem.cpp) are constant on C y = 0: E'(c'c)E = 0.
// Back-substitution, the rhs and the error functions need nothing new.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

#include "kernels.h"

namespace gsx {

namespace {

constexpr int kConMaxRows = 1024;

__device__ __forceinline__ double sym_at(const double* A, int n, int a, int b) {
  return a >= b ? A[a + (i64)b * n] : A[b + (i64)a * n];
}

// One workgroup per constrained front.
__global__ void __launch_bounds__(256) constraint_reduce_kernel(const ConDesc* descs, int first,
                                                                const int* own_col_ptr, const int* own_cols,
                                                                const i64* own_jac, const int* own_m, const int* child_list,
                                                                const int* fwd_map, const double* jac, double* arena,
                                                                double* work, int* iwork, DevStatus* status) {
  const ConDesc d = descs[first + blockIdx.x];
  const int n = d.N, K = d.K, tid = threadIdx.x, nt = blockDim.x;
  double* C = work + d.work;                  // K x n, row-major
  double* Bm = C + (i64)K * n;                // p x n
  double* Zm = Bm + (i64)K * n;               // p x n
  int* colpiv = iwork + d.iwork;              // n: pivot index of a column, -1
  int* prow = colpiv + n;                     // K: row of pivot k
  int* pcol = prow + K;                       // K: column of pivot k
  int* np_out = pcol + K;                     // 1: number of pivots
  const double* A = arena + d.off;
  __shared__ double colj[kConMaxRows];
  __shared__ unsigned char used[kConMaxRows];
  __shared__ int s_np, s_nfwd, s_fail;
  __shared__ double s_red[256];
  __shared__ int s_ri[256], s_rc[256];

  // ---- the rows: the front's own (from the Jacobian store), then the children's leftovers ----
  for (i64 e = tid; e < (i64)K * n; e += nt) C[e] = 0;
  for (int c = tid; c < n; c += nt) colpiv[c] = -1;
  for (int i = tid; i < K; i += nt) used[i] = 0;
  if (tid == 0) s_np = 0, s_nfwd = 0, s_fail = 0;
  __syncthreads();
  int row = 0;
  for (int r = d.own_begin; r < d.own_end; ++r, ++row) {
    const int c0 = own_col_ptr[r], c1 = own_col_ptr[r + 1], m = own_m[r];
    const double* src = jac + own_jac[r];
    for (int c = c0 + tid; c < c1; c += nt) C[(i64)row * n + own_cols[c]] = src[(i64)(c - c0) * m];
  }
  for (int q = d.child_begin; q < d.child_end; ++q) {
    const ConDesc ch = descs[child_list[q]];
    const int s1 = ch.N - ch.F;   // separator + rhs
    const int* map = fwd_map + ch.fwd_map;
    const double* src = work + ch.fwd;
    for (int i = 0; i < ch.n_fwd; ++i, ++row)
      for (int u = tid; u < s1; u += nt)
        if (map[u] >= 0) C[(i64)row * n + map[u]] = src[(i64)i * s1 + u];
  }
  __syncthreads();

  // ---- Gauss-Jordan, pivots in the frontal columns.  COMPLETE pivoting: each step takes the largest entry left among
  //      the unused rows and the unused frontal columns (> 1e-9, the threshold of check_if_constraint).  Constrained::QR
  //      walks the columns in order and takes the best row for each; a row it leaves without a frontal pivot that way
  //      still finds one among the later columns of its staggered [R d] — here the frontal columns are all a row can use
  //      before it must go to another front, so the choice must not strand a row that the frontal block could absorb
  //      (a Pose3 between-constraint: its rotation rows are zero in the translation columns).  Any complete set of
  //      pivots gives the same constrained minimiser. ----
  const int max_steps = min(K, d.F);
  for (int step = 0; step < max_steps; ++step) {
    // the largest eligible entry: (value, row, column), ties to the smaller row, then column
    double bv = 1e-9;
    int bi = -1, bc = -1;
    for (i64 e = tid; e < (i64)K * d.F; e += nt) {
      const int i = (int)(e / d.F), c = (int)(e % d.F);
      if (used[i] || colpiv[c] >= 0) continue;
      const double a = fabs(C[(i64)i * n + c]);
      if (a > bv) {
        bv = a;
        bi = i;
        bc = c;
      }
    }
    s_red[tid] = bv;
    s_ri[tid] = bi;
    s_rc[tid] = bc;
    __syncthreads();
    for (int w = nt >> 1; w > 0; w >>= 1) {
      if (tid < w) {
        const double ov = s_red[tid + w];
        const int oi = s_ri[tid + w], oc = s_rc[tid + w];
        const bool better = oi >= 0 && (s_ri[tid] < 0 || ov > s_red[tid] ||
                                        (ov == s_red[tid] && (oi < s_ri[tid] || (oi == s_ri[tid] && oc < s_rc[tid]))));
        if (better) {
          s_red[tid] = ov;
          s_ri[tid] = oi;
          s_rc[tid] = oc;
        }
      }
      __syncthreads();
    }
    const int best = s_ri[0], j = s_rc[0];
    __syncthreads();
    if (best < 0) break;   // (uniform)
    for (int i = tid; i < K; i += nt) colj[i] = C[(i64)i * n + j];
    if (tid == 0) {
      used[best] = 1;
      colpiv[j] = s_np;
      prow[s_np] = best;
      pcol[s_np] = j;
      s_np = s_np + 1;
    }
    __syncthreads();
    const double inv = 1.0 / colj[best];
    for (int c = tid; c < n; c += nt) {
      const double pr = C[(i64)best * n + c] * inv;
      for (int i = 0; i < K; ++i) {
        if (i == best) continue;
        const double a = colj[i];
        if (a != 0.0) C[(i64)i * n + c] -= a * pr;
      }
      C[(i64)best * n + c] = pr;
    }
    __syncthreads();
  }
  const int p = s_np;

  // ---- the rows without a frontal pivot: to the parent when they still say something about the separator ----
  {
    const int s1 = n - d.F;
    double* fwd = work + d.fwd;
    for (i64 e = tid; e < (i64)d.n_fwd * s1; e += nt) fwd[e] = 0;
    __syncthreads();
    for (int i = 0; i < K; ++i) {
      if (used[i]) continue;
      double mx = 0;
      for (int c = d.F + tid; c < n - 1; c += nt) mx = fmax(mx, fabs(C[(i64)i * n + c]));
      s_red[tid] = mx;
      __syncthreads();
      for (int w = nt >> 1; w > 0; w >>= 1) {
        if (tid < w) s_red[tid] = fmax(s_red[tid], s_red[tid + w]);
        __syncthreads();
      }
      const bool live = s_red[0] > 1e-9;
      __syncthreads();
      if (!live) continue;   // a redundant row (or one no variable is left in): nothing to enforce
      const int slot = s_nfwd;
      __syncthreads();
      if (slot >= d.n_fwd) {
        if (tid == 0) s_fail = 1;
      } else {
        for (int u = tid; u < s1; u += nt) fwd[(i64)slot * s1 + u] = C[(i64)i * n + d.F + u];
      }
      if (tid == 0) s_nfwd = slot + 1;
      __syncthreads();
    }
  }
  if (tid == 0) {
    *np_out = p;
    if (s_fail) {  // a constraint row this clique cannot absorb and its parent does not expect
      atomicAdd(&status->n_fail, 1);
      atomicMin(&status->first_front, d.front);
    }
  }

  // ---- R (the pivot rows, zero in the pivot columns), B = M_P., Z = M_PP R + R - B ----
  for (int k = 0; k < p; ++k) {
    const int r = prow[k], jk = pcol[k];
    for (int c = tid; c < n; c += nt) {
      const double rv = colpiv[c] >= 0 ? 0.0 : C[(i64)r * n + c];
      C[(i64)r * n + c] = rv;
      Bm[(i64)k * n + c] = sym_at(A, n, jk, c);
    }
  }
  __syncthreads();
  for (int k = 0; k < p; ++k) {
    const int jk = pcol[k];
    for (int c = tid; c < n; c += nt) {
      double g = 0;
      for (int k2 = 0; k2 < p; ++k2) g += sym_at(A, n, jk, pcol[k2]) * C[(i64)prow[k2] * n + c];
      Zm[(i64)k * n + c] = g + C[(i64)prow[k] * n + c] - Bm[(i64)k * n + c];
    }
  }
}

// The front written back (lower triangle): grid (column blocks of 64, fronts).
__global__ void __launch_bounds__(256) constraint_apply_kernel(const ConDesc* descs, double* arena, const double* work,
                                                               const int* iwork) {
  const ConDesc d = descs[blockIdx.y];
  const int n = d.N, K = d.K;
  const double* C = work + d.work;
  const double* Bm = C + (i64)K * n;
  const double* Zm = Bm + (i64)K * n;
  const int* colpiv = iwork + d.iwork;
  const int* prow = colpiv + n;
  const int p = prow[2 * K];
  if (p == 0) return;
  double* A = arena + d.off;
  for (int b = blockIdx.x * 64; b < n; b += gridDim.x * 64) {
    const int bw = min(64, n - b);
    // entries (a, b + bb), a >= b + bb: threads over (a, bb) with a fastest
    for (i64 e = threadIdx.x; e < (i64)(n - b) * bw; e += blockDim.x) {
      const int bb = (int)(e / (n - b)), a = b + (int)(e % (n - b));
      const int col = b + bb;
      if (a < col) continue;
      const int pa = colpiv[a], pb = colpiv[col];
      double v;
      if (pa >= 0 && pb >= 0) {
        v = a == col ? 1.0 : 0.0;
      } else if (pa >= 0) {
        v = C[(i64)prow[pa] * n + col];
      } else if (pb >= 0) {
        v = C[(i64)prow[pb] * n + a];
      } else {
        v = A[a + (i64)col * n];
        for (int k = 0; k < p; ++k) {
          const i64 o = (i64)k * n;
          const double ra = C[(i64)prow[k] * n + a], rb = C[(i64)prow[k] * n + col];
          v += ra * Zm[o + col] - Bm[o + a] * rb;
        }
      }
      A[a + (i64)col * n] = v;
    }
  }
}

// diag(J'J) as the reference forms it: a constraint row enters with its UNWHITENED entries (JacobianFactor::
// hessianDiagonalAdd whitens a column with the model, and Constrained::whiten leaves a row of sigma 0 as it is —
// JacobianFactor.cpp:539-564, NoiseModel.cpp:395-410), while the stored row carries sqrt(mu).  One thread per tangent
// scalar a constraint row touches, its terms in a fixed order.
__global__ void __launch_bounds__(64) constraint_hdiag_kernel(int n, const int* tan, const int* ptr, const i64* jidx,
                                                              const double* w, const double* jac, double* hdiag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k) s += w[k] * jac[jidx[k]] * jac[jidx[k]];
  hdiag[tan[i]] -= s;
}

}  // namespace

void launch_constraint_hdiag(int n, const int* tan, const int* ptr, const i64* jidx, const double* w, const double* jac,
                             double* hdiag, hipStream_t st) {
  if (n > 0) constraint_hdiag_kernel<<<(n + 63) / 64, 64, 0, st>>>(n, tan, ptr, jidx, w, jac, hdiag);
}

void launch_constraint_fronts(const DevSymbolic& S, const ConTables& T, int first, int count, int max_n, const double* jac,
                              double* arena, DevStatus* status, hipStream_t st) {
  if (count <= 0) return;
  constraint_reduce_kernel<<<count, 256, 0, st>>>(T.descs, first, T.own_col_ptr, T.own_cols, T.own_jac, T.own_m,
                                                  T.child_list, T.fwd_map, jac, arena, T.work, T.iwork, status);
  const int bx = std::max(1, std::min((max_n + 63) / 64, 256));
  constraint_apply_kernel<<<dim3(bx, count), 256, 0, st>>>(T.descs + first, arena, T.work, T.iwork);
}

}  // namespace gsx
