// kernels.h — device-side tables and the launch functions of kernels.hip (product; gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace gsx {

typedef long long i64;

// Everything a per-factor kernel needs before it touches the state, in ONE 32-byte record (graph order): the chain
// factor id -> key pointer -> variable ids -> state offsets -> state was five dependent loads deep.
struct FactorRec {
  i64 jac_off;
  int s0, s1;                 // state offsets of the first two variables (-1: none)
  int meas_off, noise_off;    // doubles into meas / noise
  int type_kind;              // f_type | noise kind (with the robust bits) << 8 | type of the first variable << 24
  int rows_dim;               // rows | dimension of the first variable (<= 255) << 8 | columns of [A b] << 16
};
// Immutable problem tables on the device (graph order).
struct DevProblem {
  int n_vars, n_factors;
  const int *var_type, *var_dim, *var_state_off, *var_tan_off;
  const int *f_type, *f_rows, *f_key_ptr, *f_vars, *f_noise_kind, *f_cols;
  const i64 *f_meas_off, *f_noise_off, *f_jac_off;
  const double *meas, *noise;
  const FactorRec* frec;  // per factor
  const int* f_active;  // factors this rank evaluates in the error kernels (nullptr: all n_active = n_factors)
  int n_active;
};

// One H-assembly term: panel[dst + i, j] += sum_r J[r, colB + i] * J[r, colA + j] (32-byte record).
struct TermRec {
  i64 jac;
  int m, colA, colB, dB, dst, pad;
};

// Packed per-variable / per-child records (32 bytes, fetched through the scalar path).
struct VarRec {
  i64 h_off, hmap_off;
  int dA, rows, loc, toff;
};
// Everything the leaf kernels need about one leaf clique, in schedule order (80 bytes through the scalar path: the
// dependent-load chain of a 64-thread workgroup is record -> data instead of id -> front tables -> variable tables ->
// data).  The H-panel fields describe the FIRST frontal variable (the only one of a BAL landmark clique).
struct LeafRec {
  i64 off, h_off, hmap_off, gidx_ptr;
  int n, F, nfv, lean;
  int dA, rows, loc, toff;
  int fvar_ptr, front, pad0, pad1;
};
struct ChildRec {
  i64 src0, cmap_off;   // arena offset of the child's Schur complement (top-left), offset of its row map
  int nc, s1, pad0, pad1;
};
// One LDS-class front: what front_small needs before its first data load, in one 32-byte record (the chain is
// id -> record -> variable records (stored per front: fvar_recs) -> data)
struct FrontRec {
  i64 off;
  int n, F, nfv, fvar_ptr, child_ptr, nchild;
};

// Symbolic tables on the device.
struct DevSymbolic {
  int n_fronts;
  // fronts
  const i64* fr_off;            // arena offset
  const int *fr_N, *fr_F, *fr_nfv, *fr_fvar_ptr, *fvars, *fr_parent, *fr_child_ptr, *children;
  const int* fr_lean;           // 1: lean leaf — only its n x F L panel is stored, no Schur complement
  const i64 *cmap_ptr, *gidx_ptr;
  const int *cmap, *gidx;
  // H panels
  const i64 *h_off, *hmap_ptr;
  const int *h_rows, *hmap, *h_loc;
  // H assembly terms
  const i64* term_ptr;
  const TermRec* terms;
  const VarRec* var_recs;
  const ChildRec* child_recs;
  const FrontRec* front_recs;   // per front
  const VarRec* fvar_recs;      // var_recs in the order of fvars (a front's frontal variables are contiguous)
  // partial ("wildfire") back-substitution (gsx_backsubstitute_wildfire): when wf_dirty is set, a back-substitution
  // kernel first decides whether its clique is dirty — reached through a dirty parent, and re-eliminated (wf_replaced,
  // per front) or with a separator variable that changed (wf_changed, per variable) — records that in wf_dirty and
  // leaves a clean clique alone (ISAM2Clique::isDirty, gtsam/nonlinear/ISAM2Clique.cpp:68-90; children are visited only
  // through a dirty parent, :261-287)
  unsigned char* wf_dirty;
  const unsigned char *wf_replaced, *wf_changed;
};
#ifdef __HIPCC__
// every thread of the clique's workgroup (or wave: `writer` = its first lane) evaluates the same rule on the same bytes
__device__ __forceinline__ bool wildfire_skip(const DevSymbolic& S, int f, bool writer) {
  if (S.wf_dirty == nullptr) return false;
  const int p = S.fr_parent[f];
  bool dirty = (p < 0) || S.wf_dirty[p];
  if (dirty && !S.wf_replaced[f]) {
    dirty = false;
    for (int q = S.fr_fvar_ptr[f] + S.fr_nfv[f]; q < S.fr_fvar_ptr[f + 1]; ++q)
      if (S.wf_changed[S.fvars[q]]) {
        dirty = true;
        break;
      }
  }
  if (writer) S.wf_dirty[f] = dirty ? 1 : 0;
  return !dirty;
}
#endif

// status words written by the factorization / back-substitution kernels
struct DevStatus {
  int n_fail;        // fronts whose partial Cholesky failed (non-positive pivot or exponent-gap test)
  int first_front;   // smallest failing front id (INT_MAX when none)
  int n_nonfinite;   // non-finite entries met in back-substitution
  int n_cheirality;  // SFM factors zeroed in the last linearize
  int n_backsub;     // frontal variables back-substituted by the last wildfire pass (lastBacksubVariableCount)
};

// scalar slots in the device scalar buffer
// sharded problems: the slots summed over the ranks before the host reads them (bit k = slot k; two more bits for the
// status counters of the factorization / of linearize), kXScalars doubles in the exchange vector
constexpr int kXScalars = 16;
constexpr unsigned kXFact = 1u << 12, kXLin = 1u << 13;
void launch_shard_pack(const double* scalars, const DevStatus* status, unsigned dirty, double* x, hipStream_t st);
void launch_shard_unpack(const double* x, unsigned dirty, double* scalars, DevStatus* status, hipStream_t st);
// states of `n` variables (packed one after the other in `src`, variable i at src_off[i]) into the values vector
void launch_scatter_states(const DevProblem& P, const int* vars, const int* src_off, int n, const double* src,
                           double* values, hipStream_t st);
void launch_mask_copy(const double* in, const unsigned char* mask, int64_t n, double* out, hipStream_t st);

enum { SC_LAMBDA = 0, SC_ERR = 1, SC_LIN0 = 2, SC_LIND = 3, SC_TRIAL_ERR = 4, SC_DOT0 = 5, SC_DOT1 = 6, SC_DOT2 = 7,
       SC_DOT3 = 8, SC_LIN0_LOCAL = 9 /* 1/2 sum |b|^2 of this rank's factors: never exchanged */, SC_COUNT = 12 };

struct BigDesc {   // one big front of a level
  i64 off, xoff;             // arena offsets of the n x n front and of its n x F L-panel area
  int N, F, front, parent;   // parent front id (-1 root; kStandaloneFront: the dense entry, not a front of a tree)
};
constexpr int kStandaloneFront = -3;
// choleskyPartial's conditioning test per reference clique, after the factorization (kernels.hip)
void launch_cond_check(int n, const i64* last, const i64* prev, const int* front, const double* arena, DevStatus* status,
                       hipStream_t st);
// a big front owns n x n doubles followed by its n x F L panel (rows below each diagonal tile)
__host__ __device__ inline i64 big_panel_offset(int n) { return ((i64)n * n + 1) & ~(i64)1; }

// ---- launches (all asynchronous on `st`) ---------------------------------------------------------
void launch_linearize(const DevProblem& P, const int* const type_lists[6], const int type_counts[6],
                      const double* values, double* jac, DevStatus* status, hipStream_t st);
// the graph error over the type lists of launch_linearize (a kernel per list, the factor type a compile-time constant)
void launch_error(const DevProblem& P, const int* const type_lists[6], const int type_counts[6], const double* values,
                  double* partials, int n_partials_cap, double* scalars, int slot, hipStream_t st);
// slice > 0: the staged kernel (doubles of LDS per wave = the largest [A b] range of 64 consecutive factors); 0: the direct one
void launch_linear_error(const DevProblem& P, const double* jac, const double* delta, double* partials,
                         int n_partials_cap, double* scalars, int slice, hipStream_t st);
// the linearized errors of an LM trial from the right-hand sides and the solved step (kernels.hip: lin0_kernel)
void launch_lin0(const DevProblem& P, const double* jac, double* partials, int cap, double* scalars, hipStream_t st);
void launch_model_error(const DevProblem& P, const DevSymbolic& S, const int* vars, int nvars, const double* H,
                        const double* delta, const double* damp, double* partials, int cap, double* scalars, hipStream_t st);
void launch_retract(const DevProblem& P, const double* values, const double* delta, double* out, hipStream_t st);
void launch_assemble_h_group(const DevProblem& P, const DevSymbolic& S, const int* vars, int count, int threads,
                             int lds_bytes, bool global, const double* jac, double* H, hipStream_t st);
// star variables in bundles (first position in vars, count) of one shape and <= 64 factors: a wave per bundle
void launch_assemble_h_star_bundles(const DevProblem& P, const DevSymbolic& S, const int* vars, const int2* bundles,
                                    int count, const double* jac, double* H, hipStream_t st);
// the diagonal-panel variables and the star bundles in one launch (they write different panels)
void launch_assemble_h_diag_star(const DevProblem& P, const DevSymbolic& S, const int* dvars, int dcount, const int* svars,
                                 const int2* bundles, int scount, const double* jac, double* H, hipStream_t st);
void launch_hessian_diag(const DevProblem& P, const DevSymbolic& S, const double* H, double* diag, hipStream_t st);
void launch_make_damping(int n, const double* hdiag, int diagonal, double mind, double maxd, double* damp,
                         hipStream_t st);
// small fronts of one size class: ids[0..count), each eliminated by one workgroup of `threads`
// threads with (max_n^2) doubles of LDS
void launch_front_small(const DevProblem& P, const DevSymbolic& S, const int* ids, int count, int max_n,
                        int threads, const double* H, const double* damp, const double* scalars, double* arena,
                        DevStatus* status, hipStream_t st);
// the LDS-class fronts of one tier, whole subtrees in one launch (kernels.hip: front_tree_kernel)
struct TreeArgs {
  const int* start;   // fronts without an unfinished child in the tier
  int nstart;
  int* cursor;        // start entries claimed so far (zeroed by launch_begin_factorization)
  int* pending;       // per front: children of the tier still to arrive (restores itself to npend)
  const int *up, *npend;
};
void launch_front_tree(const DevProblem& P, const DevSymbolic& S, const TreeArgs& T, int max_n, int threads, const double* H,
                       const double* damp, const double* scalars, double* arena, DevStatus* status, hipStream_t st);
// MEDIUM fronts (gsx_internal.h), a workgroup each; LDS = (max_panel + max_n) doubles
void launch_front_medium(const DevProblem& P, const DevSymbolic& S, const int* ids, int count, int max_panel, int max_n,
                         const double* H, const double* damp, const double* scalars, double* arena, DevStatus* status,
                         hipStream_t st);
// the tree tier of the medium fronts (and of the LDS fronts above them)
void launch_front_tree_med(const DevProblem& P, const DevSymbolic& S, const TreeArgs& T, size_t lds_bytes, const double* H,
                           const double* damp, const double* scalars, double* arena, DevStatus* status, hipStream_t st);
void launch_big_init(const DevProblem& P, const DevSymbolic& S, const BigDesc* descs, int count, int max_n,
                     int max_nfv, const double* H, const double* damp, const double* scalars, double* arena,
                     hipStream_t st);
// Partial Cholesky of a group of big fronts (bigfront.hip): rounds of `chunk` frontal columns, three launches each —
// diag (L11 of the chunk, one workgroup per front), rows (L21 = A21 L11^-T, one per 32 rows), schur (A22 -= L21 L21').
struct BigPlan {
  int chunk = 0;
  std::vector<int> fw, rb, pairs;  // per round: widest chunk, most 32-row blocks below it, most lower tile pairs
  int rounds() const { return (int)fw.size(); }
};
void plan_big_group(const BigDesc* host_descs, int count, BigPlan& plan);
void launch_big_diag(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, DevStatus* status,
                     hipStream_t st);
#ifdef GSX_STAMP
void big_stamp_dump(const char* what);
#endif
void launch_big_rows(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, hipStream_t st);
void launch_big_schur(const BigDesc* descs, int count, const BigPlan& plan, int round, double* arena, hipStream_t st);
void launch_front_leaf(const DevProblem& P, const DevSymbolic& S, const LeafRec* recs, int count, int max_panel, int threads,
                       const double* H, const double* damp, const double* scalars, double* arena, DevStatus* status,
                       hipStream_t st);
// Deterministic extend-add into big parents (big_gather).
// A source is either a block of a child's stored Schur complement (F = 0: entry (i, j) at off + i + j ld) or, for a
// lean leaf child, the product form -W_b W_a' with W_b = rows off.., W_a = rows off + d2.. of the child's n x F L panel.
struct GatherSrc {   // 16 bytes
  i64 off;
  int d2;            // product form: offset of the second row block relative to the first
  int ldF;           // ld | F << 24
};
struct GatherSeg {   // 32 bytes: one wave's work — up to kGatherChunk consecutive sources of one destination block
  i64 dst;           // arena offset of the destination block (top-left entry)
  i64 src;           // first source record
  int n, ld, dims;   // number of sources; destination ld; dB | dA << 8 | diag << 16
  int slot;          // -1: add straight into dst; else scratch slot of a split task
};
struct GatherArgs {
  const GatherSeg* segs;
  const GatherSrc* srcs;
  const i64* gt_dst;                       // per task (combine pass)
  const int *gt_ld, *gt_dims, *gm_task, *gm_slot, *gm_nslots;
  double* scratch;  // slots x 256 doubles (a 16 x 16 accumulator tile in matrix-core register layout)
};
void launch_big_gather(const GatherArgs& G, int seg0, int nseg, int m0, int nm, double* arena, hipStream_t st);
void launch_backsolve(const DevSymbolic& S, const int* ids, int count, int threads, int max_n, const double* arena,
                      double* delta, DevStatus* status, hipStream_t st);
// blocked fronts of a level whose L11 tiles fit in LDS (bigfront.hip); backsolve_big_lds = 0: they do not, use the kernel above
size_t backsolve_big_lds(int max_n, int max_F, int max_sep_rows);
void launch_backsolve_big(const DevSymbolic& S, const int* ids, int count, int max_n, int max_F, const double* arena,
                          double* delta, DevStatus* status, hipStream_t st);
// LDS-class fronts with <= 64 frontal columns (bigfront.hip): all loads of a front in flight together
bool backsolve_small_fits(int max_n, int max_F);
void launch_backsolve_small(const DevSymbolic& S, const int* ids, int count, int max_F, const double* arena, double* delta,
                            DevStatus* status, hipStream_t st);
// the tree fronts of all levels in one launch, top-down (bigfront.hip: backsolve_tree_kernel)
struct BacksolveTreeArgs {
  const int* roots;            // tree fronts whose parent is not a tree front
  int n_roots, n_tickets;      // ... their number; n_tickets = n_roots + the tree fronts that are not a first child
  const int *child_ptr, *children;   // per front: its tree children (CSR over all fronts), the deepest subtree first
  unsigned long long* ready;   // n_tickets - n_roots entries: epoch << 32 | front
  int *head, *tail;            // tickets handed out; entries published (zeroed by the host before the launch)
  unsigned epoch;              // differs from launch to launch (never 0)
};
void launch_backsolve_tree(const DevSymbolic& S, const BacksolveTreeArgs& Q, const double* arena, double* delta,
                           DevStatus* status, hipStream_t st);
// ISAM2's partial ("wildfire") back-substitution, the pass after a level's kernels: a dirty clique's new frontal solution
// is compared with the old one — a change of at least `threshold` in the infinity norm (or a re-eliminated clique) marks
// its frontal variables as changed, a smaller one is undone (valuesChanged / restoreFromOriginals,
// gtsam/nonlinear/ISAM2Clique.cpp:175-201).
struct WildfireArgs {
  const unsigned char *replaced, *dirty;  // per front
  unsigned char* changed;                 // per variable
  const double* old_delta;                // the solution before the pass
  DevStatus* status;                      // n_backsub += frontal variables back-substituted
  double threshold;
};
// flags[ids[k]] = 1 (the cliques a partial re-elimination has just redone)
void launch_mark_fronts(const int* ids, int count, unsigned char* flags, hipStream_t st);
void launch_wildfire_post(const DevSymbolic& S, const int* ids, int count, bool wave_per_clique, const WildfireArgs& W,
                          double* delta, hipStream_t st);
// leaf cliques of a level, a wave per clique
void launch_backsolve_leaf(const DevSymbolic& S, const LeafRec* recs, int count, int max_F, const double* arena,
                           double* delta, DevStatus* status, hipStream_t st);
// Dogleg (gtsam/nonlinear/DoglegOptimizerImpl.*): gradient g = A'b out of the H panels' rhs rows, |A x|^2 over the
// linearized graph, dot products and out = alpha a + beta b on tangent vectors (deterministic two-stage reductions)
void launch_gradient(const DevProblem& P, const DevSymbolic& S, const double* H, double* g, hipStream_t st);
void launch_ax_sqnorm(const DevProblem& P, const double* jac, const double* x, double* partials, int cap, double* scalars,
                      int slot, hipStream_t st);
void launch_vec_dot(const double* a, const double* b, int64_t n, double* partials, int cap, double* scalars, int slot,
                    hipStream_t st);
void launch_vec_axpby(double* out, double alpha, const double* a, double beta, const double* b, int64_t n, hipStream_t st);
// Marginal covariance block of one variable from the factorization: forward substitution of the
// variable's unit columns up the path of cliques from its own clique to the root, Sigma_vv = sum over the path of y_F' y_F.
// path[0..npath): front ids, child to root; (loc, dA): the variable's rows inside path[0]; lds_doubles = 2 max_n dA.
void launch_marginal_path(const DevSymbolic& S, const int* path, int npath, int loc, int dA, int max_n, const double* arena,
                          double* out, double* Y, int64_t n_tan, hipStream_t st);
void launch_joint_cross(const double* Ya, const double* Yb, int64_t n_tan, int dA, int dB, double* out, hipStream_t st);
void launch_set_scalar(double* scalars, int slot, double v, hipStream_t st);
// Hard constraints (constraint.hip): one record per constrained front (Symbolic::con_fronts order, children first)
struct ConDesc {
  i64 off;          // arena offset of the front's n x n panel
  i64 work, fwd;    // doubles into the work buffer: 3 K n for the rows / B / Z; n_fwd (n - F) rows handed to the parent
  i64 iwork;        // ints into the int work buffer: n + 2 K + 1
  int N, F, K, n_fwd;
  int own_begin, own_end;      // its own rows (ConTables::own_*)
  int child_begin, child_end;  // ConTables::child_list: the fronts below whose leftover rows arrive here
  int front;
  int fwd_map;                 // ConTables::fwd_map + this: per separator + rhs row, the row of the front its leftovers go to
};
struct ConTables {
  const ConDesc* descs;
  const int *own_col_ptr, *own_cols, *own_m;
  const i64* own_jac;
  const int* child_list;
  const int* fwd_map;
  double* work;
  int* iwork;
};
// the constrained fronts descs[first .. first + count), all of one level, after their gather and before their factorization
void launch_constraint_fronts(const DevSymbolic& S, const ConTables& T, int first, int count, int max_n, const double* jac,
                              double* arena, DevStatus* status, hipStream_t st);
// hdiag[tan[i]] -= sum over k in ptr[i] .. ptr[i + 1] of w[k] jac[jidx[k]]^2 (the constraint rows' share of diag(J'J) at
// the reference's weight)
void launch_constraint_hdiag(int n, const int* tan, const int* ptr, const i64* jidx, const double* w, const double* jac,
                             double* hdiag, hipStream_t st);
constexpr int kTreeCursors = 8;  // cursors of the tree kernels' start lists, zeroed with the status words
void launch_begin_factorization(double* scalars, double lambda, DevStatus* status, int* tree_cursors, hipStream_t st);
// dense unit kernel for gsx_cholesky_partial: in-place lower partial Cholesky of an n x n
// column-major matrix (lower triangle significant)
void launch_dense_partial(double* a, int n, int nf, DevStatus* status, hipStream_t st);
int max_dynamic_lds();

}  // namespace gsx
