// solver.hip — handle, device memory, level-scheduled multifrontal driver, LM/GN policy and the
// C-ABI of include/gsx.h.  Product code: no oracle, no CPU fallback — every numeric entry point
// returns GSX_E_NO_DEVICE when there is no usable GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "gsx_internal.h"
#include "kernels.h"

using namespace gsx;

namespace {

#define HIPCHK(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                     \
      return GSX_E_NO_DEVICE;                                                              \
    }                                                                                      \
  } while (0)

template <class Tp>
struct DevBuf {
  Tp* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc((void**)&p, count * sizeof(Tp));
  }
  hipError_t upload(const std::vector<Tp>& v, hipStream_t st) {
    hipError_t e = alloc(v.size() ? v.size() : 1);
    if (e != hipSuccess) return e;
    if (v.empty()) return hipSuccess;
    return hipMemcpyAsync(p, v.data(), v.size() * sizeof(Tp), hipMemcpyHostToDevice, st);
  }
  // reuse the allocation when it is large enough (scratch tables rebuilt at every call)
  hipError_t stage(const std::vector<Tp>& v, hipStream_t st) {
    if (v.size() > n || !p) {
      hipError_t e = alloc(std::max<size_t>(std::max<size_t>(v.size(), 2 * n), 64));
      if (e != hipSuccess) return e;
    }
    if (v.empty()) return hipSuccess;
    return hipMemcpyAsync(p, v.data(), v.size() * sizeof(Tp), hipMemcpyHostToDevice, st);
  }
  void release() {
    if (p) hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

struct SmallLaunch {
  int begin, count, max_n, threads;
  int max_panel = 0;  // leaf launches: largest n*F of the group (LDS doubles)
  int max_F = 0;      // rest launches: largest frontal dimension
  bool medium = false;  // MEDIUM fronts (gsx_internal.h): front_medium_kernel, max_panel = largest n*F
};
// an LDS-class launch group: the whole-front-in-LDS kernel, or the medium-front one
void launch_lds_group(const DevProblem& P, const DevSymbolic& S, const int* ids, const SmallLaunch& g, const double* H,
                      const double* damp, const double* scalars, double* arena, DevStatus* status, hipStream_t st) {
  if (g.medium) launch_front_medium(P, S, ids, g.count, g.max_panel, g.max_n, H, damp, scalars, arena, status, st);
  else launch_front_small(P, S, ids, g.count, g.max_n, g.threads, H, damp, scalars, arena, status, st);
}
struct BigLevel {
  int begin = 0, count = 0;
  BigPlan plan;  // rounds of the level's blocked fronts (bigfront.hip)
};

enum Phase { PH_LINEARIZE, PH_ASSEMBLE_H, PH_FACTORIZE, PH_BACKSOLVE, PH_LINERR, PH_RETRACT, PH_ERROR,
             PH_FACTOR_SMALL, PH_FACTOR_BIG, PH_FACTOR_LEAF, PH_K_SYRK, PH_K_TRSM, PH_K_POTRF0, PH_K_GATHER,
             PH_K_BACKSOLVE, PH_COUNT };
const char* kPhaseNames[PH_COUNT] = {"linearize", "assemble_hessian", "factorize", "backsolve", "linear_error",
                                     "retract", "error", "factor_small", "factor_big", "factor_leaf",
                                     "big_schur", "big_rows", "big_diag", "big_gather", "backsolve_launch"};

struct Timer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
  double ms = 0;
  int64_t count = 0;
};

}  // namespace

struct gsx_context {
  std::string err;
  int device = 0;
  bool has_device = false;
  hipStream_t stream = nullptr;
  HostProblem P;
  Symbolic S;
  bool has_symbolic = false;
  double relax = GSX_AMALGAMATION_AUTO;  // relaxed amalgamation (gsx_set_amalgamation); 0 = the reference's cliques, < 0 = the library's choice
  int relax_max_f = 128;
  std::vector<int> order;

  // device problem
  DevBuf<int> d_var_type, d_var_dim, d_var_state_off, d_var_tan_off, d_f_type, d_f_rows, d_f_key_ptr, d_f_vars,
      d_f_noise_kind, d_f_cols;
  DevBuf<i64> d_f_meas_off, d_f_noise_off, d_f_jac_off;
  DevBuf<double> d_meas, d_noise;
  DevBuf<FactorRec> d_frec;
  DevBuf<int> d_type_list[6];
  int type_count[6] = {0, 0, 0, 0, 0, 0};
  DevProblem DP{};
  // device symbolic
  DevBuf<i64> d_fr_off, d_cmap_ptr, d_gidx_ptr, d_h_off, d_hmap_ptr, d_term_ptr;
  DevBuf<TermRec> d_terms;
  DevBuf<VarRec> d_var_recs;
  DevBuf<ChildRec> d_child_recs;
  DevBuf<FrontRec> d_front_recs;
  DevBuf<i64> d_cond_last, d_cond_prev;   // conditioning test per reference clique (Symbolic::cond_*)
  DevBuf<int> d_cond_front;
  DevBuf<VarRec> d_fvar_recs;
  DevBuf<LeafRec> d_leaf_recs;   // leaf-kernel cliques of level 0, in schedule order
  int leaf_base = 0;             // schedule position of d_leaf_recs[0]
  int lin_err_slice = 0;         // LDS doubles per wave of the staged linear-error kernel (0: blocks too large, direct kernel)
  DevBuf<int> d_fr_N, d_fr_F, d_fr_nfv, d_fr_fvar_ptr, d_fvars, d_fr_parent, d_fr_lean, d_fr_child_ptr, d_children, d_cmap,
      d_gidx, d_h_rows, d_hmap, d_h_loc, d_sched, d_hvars;
  DevBuf<BigDesc> d_big;
  DevSymbolic DS{};
  // schedule
  std::vector<std::vector<SmallLaunch>> small_launch;  // per level
  std::vector<std::vector<SmallLaunch>> leaf_launch;   // per level (panel-only leaf kernel)
  std::vector<int> leaf_max_F;                         // per level: largest F among its leaf-kernel fronts
  // the full factorization runs the tree fronts (Symbolic::tree_*) in one launch per tier and only the other LDS-class
  // fronts level by level (rest_launch: ranges of d_rest_ids); small_launch keeps ALL of them for the filtered plans of
  // gsx_relinearize_partial
  std::vector<std::vector<SmallLaunch>> rest_launch;
  DevBuf<int> d_usched, d_rest_ids, d_tree_start, d_tree_up, d_tree_npend, d_tree_pending, d_tree_cursor;
  // back-substitution of the tree fronts in one launch (bigfront.hip: backsolve_tree_kernel)
  DevBuf<int> d_bst_roots, d_bst_child_ptr, d_bst_children, d_bst_counters;
  DevBuf<unsigned long long> d_bst_ready;
  int bst_roots = 0, bst_total = 0;
  unsigned bst_epoch = 0;
  struct TreeTier {
    int start0, nstart, max_n, threads;
    size_t med_lds;   // > 0: the tier of the medium fronts (front_tree_med_kernel), bytes of LDS
  };
  std::vector<TreeTier> tree_tiers;
  DevBuf<i64> d_gt_dst;  // gather tasks / segments for big parents
  DevBuf<GatherSeg> d_gsegs;
  DevBuf<GatherSrc> d_gsrcs;
  DevBuf<int> d_gt_ld, d_gt_dims, d_gm_task, d_gm_slot, d_gm_nslots;
  DevBuf<double> d_gscratch;
  GatherArgs GA{};
  std::vector<BigLevel> big_level;
  // hard constraints (Symbolic::con_*, constraint.hip): descs in (upper level, front) order
  struct ConLevel {
    int first = 0, count = 0, max_n = 0;
  };
  std::vector<ConLevel> con_level;      // per upper level
  std::vector<int> con_desc_of_front;   // front -> its record, -1
  DevBuf<ConDesc> d_con_descs;
  DevBuf<int> d_con_own_col_ptr, d_con_own_cols, d_con_own_m, d_con_child, d_con_fwd_map, d_con_iwork;
  DevBuf<i64> d_con_own_jac;
  DevBuf<double> d_con_work;
  ConTables CT{};
  // the constraint rows' entries by tangent scalar (constraint_hdiag_kernel): built with the problem tables
  int con_hd_n = 0;
  DevBuf<int> d_con_hd_tan, d_con_hd_ptr;
  DevBuf<i64> d_con_hd_jidx;
  DevBuf<double> d_con_hd_w;
  bool constrained() const { return !S.con_fronts.empty(); }
  std::vector<BigDesc> big_descs;
  int big_max_n = 0, big_max_nfv = 0;
  struct HGroup {
    int begin, count, threads, lds;
    bool global;
  };
  std::vector<HGroup> hgroups;
  DevBuf<int2> d_star_bundles;   // bundles of the star group (positions relative to the group's variable list)
  int n_star_bundles = 0;
  // numeric buffers
  DevBuf<double> d_values, d_trial, d_delta, d_udelta, d_jac, d_H, d_arena, d_hdiag, d_damp, d_partials, d_scalars;
  DevBuf<double> d_dlu, d_dld;  // Dogleg: steepest-descent point, dog-leg point (allocated on first use)
  DevBuf<DevStatus> d_status;
  double* h_scalars = nullptr;  // pinned
  DevStatus* h_status = nullptr;
  static constexpr int kPartials = 4096;
  bool values_set = false, linearized = false, h_ready = false, hdiag_ready = false, solved = false, damp_ready = false;
  bool lin0_ready = false;   // scalars[SC_LIN0_LOCAL] = 1/2 sum |b|^2 of the current [A b] blocks (cleared whenever they change)
  bool fact_valid = false;   // the arena holds a SUCCESSFUL factorization of the current linearization and tree ...
  double fact_lambda = 0.0;  // ... for this lambda
  bool fact_pending = false; // a factorization is queued whose status the host has not read yet (readback decides)
  int damp_kind = -1;
  double damp_min = 0, damp_max = 0;
  // LM state
  double lm_lambda = 0, lm_factor = 0, lm_error = 0;
  int lm_iterations = 0, lm_inner = 0;
  // stats
  Timer timers[PH_COUNT];
  int profiling = -1;  // -1: no event timers (default), 0: phase timers, 1: + every front launch (gsx_set_profiling)
  int64_t n_cheirality = 0;
  // host copies of the launch tables, for the filtered plans of gsx_relinearize_partial
  std::vector<LeafRec> h_leaf_recs;
  std::vector<GatherSeg> h_gsegs;
  std::vector<int> hv_list, hv_group_of_var, hv_pos;
  std::vector<int> fr_sched_pos, fr_group;        // front -> position in the schedule; launch group / big-desc index
  std::vector<int> fr_seg_ptr, fr_segs, seg_level; // destination front -> its gather segments; segment -> gather group
  std::vector<int> fr_gm_ptr, fr_gms, gm_level;    // ... and its multi-segment combine entries
  struct PartialScratch {                          // device tables of gsx_relinearize_partial, reused between calls
    DevBuf<int> marked, src_off, lists[6], hv, ids, gm_task, gm_slot, gm_nslots;
    DevBuf<double> states;
    DevBuf<LeafRec> leaf;
    DevBuf<BigDesc> big;
    DevBuf<GatherSeg> segs;
  } ps;
  // the SIDE work of a factorization (Symbolic::side_*): a low-priority queue of its own
  hipStream_t bulk = nullptr;
  hipEvent_t bulk_go = nullptr, bulk_done = nullptr;
  size_t leaf_side_group0 = 0;   // leaf_launch[0][leaf_side_group0 ..) are the side leaves' launches
  bool bulk_pending = false;     // side work of the factorization in flight that the main queue has not waited for yet
  // sharding (gsx_set_shard)
  int shard_rank = 0, shard_world = 1;
  gsx_allreduce_fn shard_cb = nullptr;
  void* shard_user = nullptr;
  bool shard_failed = false;    // an exchange callback failed since the last readback
  bool values_synced = true;    // every rank holds every variable's current value
  unsigned sc_dirty = 0;        // scalar slots / status counters written since the last exchange (kernels.h)
  DevBuf<int> d_f_active;
  DevBuf<unsigned char> d_own_tan, d_sched_tan, d_own_state;
  DevBuf<double> d_xscal;
  std::vector<double> jac_stage;   // host staging of gsx_set_block_jacobians
  bool sharded() const { return shard_world > 1; }
  // partial ("wildfire") back-substitution (gsx_backsubstitute_wildfire)
  // "replaced" = re-eliminated since the last complete undamped solution: all cliques (host flag: a full factorization
  // costs nothing here), or the ones gsx_relinearize_partial marked in the device flags; the flags are cleared lazily
  bool wf_all_replaced = true, wf_clear_pending = true;
  bool wf_delta_valid = false;              // d_delta holds a complete undamped solution on the current tree
  DevBuf<unsigned char> d_wf_replaced, d_wf_dirty, d_wf_changed;
  DevBuf<double> d_wf_old;
  DevBuf<int> d_wf_ids;
  // (device flags per front, allocated on first use for the current tree, zeroed when a clear is pending)
  hipError_t wf_flags_ready(hipStream_t st) {
    const size_t nf = (size_t)std::max(S.n_fronts, 1);
    if (d_wf_replaced.n != nf) {
      hipError_t e = d_wf_replaced.alloc(nf);
      if (e != hipSuccess) return e;
      wf_clear_pending = true;
    }
    if (wf_clear_pending) {
      hipError_t e = hipMemsetAsync(d_wf_replaced.p, 0, nf, st);
      if (e != hipSuccess) return e;
      wf_clear_pending = false;
    }
    return hipSuccess;
  }
};

namespace {

void timer_begin(gsx_context* c, int ph) {
  if (c->profiling < 0) return;  // (an event record costs ~5 us of stream time: gsx_set_profiling(h, -1) switches them off)
  Timer& t = c->timers[ph];
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!t.pool.empty()) {
    ev = t.pool.back();
    t.pool.pop_back();
  } else {
    hipEventCreate(&ev.first);
    hipEventCreate(&ev.second);
  }
  hipEventRecord(ev.first, c->stream);
  t.pending.push_back(ev);
}
void timer_end(gsx_context* c, int ph) {
  if (c->profiling < 0) return;
  hipEventRecord(c->timers[ph].pending.back().second, c->stream);
}
void timers_resolve(gsx_context* c) {
  hipStreamSynchronize(c->stream);
  for (int ph = 0; ph < PH_COUNT; ++ph) {
    Timer& t = c->timers[ph];
    for (auto& ev : t.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        t.ms += ms;
        t.count++;
      }
      t.pool.push_back(ev);
    }
    t.pending.clear();
  }
}

// factor lists of the linearize kernels (one per factor family) and of the error kernels; `owned` (may be null: all)
// restricts them to the factors this rank evaluates
gsx_status upload_factor_lists(gsx_context* c, const std::vector<char>* owned) {
  const HostProblem& P = c->P;
  hipStream_t st = c->stream;
  std::vector<int> lists[6], active;
  for (int f = 0; f < P.n_factors; ++f) {
    if (owned && !(*owned)[f]) continue;
    active.push_back(f);
    const int t = P.f_type[f];
    const int vt = P.types[P.f_vars[P.f_key_ptr[f]]];
    if (t == GSX_F_SFM) lists[0].push_back(f);
    else if (t == GSX_F_PROJECTION) lists[4].push_back(f);
    else if (t == GSX_F_BEARINGRANGE) lists[5].push_back(f);
    else if (t == GSX_F_BETWEEN && vt == GSX_VAR_POSE2) lists[1].push_back(f);
    else if (t == GSX_F_BETWEEN && vt == GSX_VAR_POSE3) lists[2].push_back(f);
    else if (t != GSX_F_LINEAR) lists[3].push_back(f);
  }
  for (int k = 0; k < 6; ++k) {
    c->type_count[k] = (int)lists[k].size();
    HIPCHK(c, c->d_type_list[k].upload(lists[k], st));
  }
  if (owned) {
    HIPCHK(c, c->d_f_active.upload(active, st));
    c->DP.f_active = c->d_f_active.p;
  } else {
    c->DP.f_active = nullptr;
  }
  c->DP.n_active = (int)active.size();
  HIPCHK(c, hipStreamSynchronize(st));
  return GSX_OK;
}

// [A b] of a GSX_F_LINEAR factor with its noise model folded in: data preparation of a GIVEN linear factor
// (JacobianFactor::whiten, gtsam/linear/JacobianFactor.cpp), on the host
void whiten_linear_factor(const HostProblem& P, int f, const double* src, double* J) {
  const int m = P.f_rows[f], nc = P.f_cols[f];
  std::copy(src, src + (size_t)m * nc, J);
  const double* np = P.noise.data() + P.f_noise_ptr[f];
  const int kind = P.f_noise_kind[f];
  for (int cidx = 0; cidx < nc; ++cidx) {
    if (kind == GSX_NOISE_ISOTROPIC)
      for (int r = 0; r < m; ++r) J[cidx * m + r] *= 1.0 / np[0];
    else if (kind == GSX_NOISE_DIAGONAL)
      for (int r = 0; r < m; ++r) J[cidx * m + r] *= 1.0 / np[r];
    else if (kind == GSX_NOISE_GAUSSIAN)
      for (int r = 0; r < m; ++r) {
        double sacc = 0;
        for (int k = r; k < m; ++k) sacc += np[r * m + k] * J[cidx * m + k];
        J[cidx * m + r] = sacc;
      }
  }
}

gsx_status upload_problem(gsx_context* c) {
  const HostProblem& P = c->P;
  hipStream_t st = c->stream;
  HIPCHK(c, c->d_var_type.upload(P.types, st));
  HIPCHK(c, c->d_var_dim.upload(P.dims, st));
  HIPCHK(c, c->d_var_state_off.upload(P.state_off, st));
  HIPCHK(c, c->d_var_tan_off.upload(P.tan_off, st));
  HIPCHK(c, c->d_f_type.upload(P.f_type, st));
  HIPCHK(c, c->d_f_rows.upload(P.f_rows, st));
  HIPCHK(c, c->d_f_key_ptr.upload(P.f_key_ptr, st));
  HIPCHK(c, c->d_f_vars.upload(P.f_vars, st));
  HIPCHK(c, c->d_f_noise_kind.upload(P.f_noise_kind, st));
  HIPCHK(c, c->d_f_cols.upload(P.f_cols, st));
  std::vector<i64> mo(P.f_meas_ptr.begin(), P.f_meas_ptr.end()), no(P.f_noise_ptr.begin(), P.f_noise_ptr.end()),
      jo(P.f_jac_off.begin(), P.f_jac_off.end());
  HIPCHK(c, c->d_f_meas_off.upload(mo, st));
  HIPCHK(c, c->d_f_noise_off.upload(no, st));
  HIPCHK(c, c->d_f_jac_off.upload(jo, st));
  {
    // the largest [A b] range of 64 consecutive factors (the blocks are laid out in factor order)
    int64_t worst = 0;
    bool ordered = true;
    for (int i0 = 0; i0 < P.n_factors; i0 += 64) {
      const int l = std::min(P.n_factors, i0 + 64) - 1;
      worst = std::max<int64_t>(worst, P.f_jac_off[l] + (int64_t)P.f_rows[l] * P.f_cols[l] - P.f_jac_off[i0]);
    }
    for (int f = 1; f < P.n_factors; ++f) ordered = ordered && P.f_jac_off[f] >= P.f_jac_off[f - 1];
    worst = (worst + 1) & ~(int64_t)1;
    c->lin_err_slice = (ordered && worst > 0 && 2 * worst * (int64_t)sizeof(double) <= 96 * 1024) ? (int)worst : 0;
  }
  HIPCHK(c, c->d_meas.upload(P.meas, st));
  HIPCHK(c, c->d_noise.upload(P.noise, st));
  c->con_hd_n = 0;
  if (!P.con_factor.empty()) {
    struct E {
      int tan;
      i64 jidx;
      double w;
    };
    std::vector<E> es;
    for (size_t i = 0; i < P.con_factor.size(); ++i) {
      const int f = P.con_factor[i], m = P.f_rows[f];
      int col = 0;
      for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q)
        for (int d = 0; d < P.dims[P.f_vars[q]]; ++d, ++col)
          es.push_back(E{P.tan_off[P.f_vars[q]] + d, P.f_jac_off[f] + (i64)col * m + P.con_row[i], 1.0 - 1.0 / P.con_mu[i]});
    }
    std::stable_sort(es.begin(), es.end(), [](const E& a, const E& b) { return a.tan < b.tan; });
    std::vector<int> tan, ptr;
    std::vector<i64> jidx;
    std::vector<double> w;
    for (const E& e : es) {
      if (tan.empty() || tan.back() != e.tan) {
        tan.push_back(e.tan);
        ptr.push_back((int)jidx.size());
      }
      jidx.push_back(e.jidx);
      w.push_back(e.w);
    }
    ptr.push_back((int)jidx.size());
    c->con_hd_n = (int)tan.size();
    HIPCHK(c, c->d_con_hd_tan.upload(tan, st));
    HIPCHK(c, c->d_con_hd_ptr.upload(ptr, st));
    HIPCHK(c, c->d_con_hd_jidx.upload(jidx, st));
    HIPCHK(c, c->d_con_hd_w.upload(w, st));
  }
  gsx_status fl = upload_factor_lists(c, nullptr);
  if (fl != GSX_OK) return fl;
  HIPCHK(c, c->d_values.alloc(std::max<int64_t>(P.state_size, 1)));
  HIPCHK(c, c->d_trial.alloc(std::max<int64_t>(P.state_size, 1)));
  HIPCHK(c, c->d_delta.alloc(std::max<int64_t>(P.tan_size, 1)));
  HIPCHK(c, c->d_udelta.alloc(std::max<int64_t>(P.tan_size, 1)));
  HIPCHK(c, c->d_hdiag.alloc(std::max<int64_t>(P.tan_size, 1)));
  HIPCHK(c, c->d_damp.alloc(std::max<int64_t>(P.tan_size, 1)));
  HIPCHK(c, c->d_jac.alloc(std::max<int64_t>(P.jac_size, 1)));
  HIPCHK(c, c->d_partials.alloc(gsx_context::kPartials));
  HIPCHK(c, c->d_scalars.alloc(SC_COUNT));
  HIPCHK(c, c->d_status.alloc(1));
  if (!c->h_scalars) HIPCHK(c, hipHostMalloc((void**)&c->h_scalars, SC_COUNT * sizeof(double)));  // (gsx_update comes here again)
  if (!c->h_status) HIPCHK(c, hipHostMalloc((void**)&c->h_status, sizeof(DevStatus)));
  HIPCHK(c, hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * sizeof(double), st));
  HIPCHK(c, hipMemsetAsync(c->d_delta.p, 0, std::max<int64_t>(P.tan_size, 1) * sizeof(double), st));
  // LINEAR factors: their (whitened) [A b] as given at creation (gsx_set_block_jacobians refreshes them in place)
  {
    std::vector<double> jac;
    bool any = false;
    for (int f = 0; f < P.n_factors; ++f) any = any || P.f_type[f] == GSX_F_LINEAR;
    if (any) {
      jac.assign(P.jac_size, 0.0);
      for (int f = 0; f < P.n_factors; ++f)
        if (P.f_type[f] == GSX_F_LINEAR) whiten_linear_factor(P, f, P.meas.data() + P.f_meas_ptr[f], jac.data() + P.f_jac_off[f]);
      HIPCHK(c, hipMemcpyAsync(c->d_jac.p, jac.data(), jac.size() * sizeof(double), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipStreamSynchronize(st));
    }
  }
  {
    // one 32-byte record per factor (kernels.h: FactorRec)
    if (P.meas.size() >= (size_t)INT32_MAX || P.noise.size() >= (size_t)INT32_MAX || P.state_size >= INT32_MAX) {
      c->err = "problem too large for 32-bit measurement / noise / state offsets";
      return GSX_E_INVALID;
    }
    std::vector<FactorRec> fr(std::max(P.n_factors, 1));
    for (int f = 0; f < P.n_factors; ++f) {
      if (P.f_cols[f] >= 32768 || P.f_rows[f] > 255) {
        c->err = "factor too wide for the packed factor record";
        return GSX_E_INVALID;
      }
      const int kp = P.f_key_ptr[f], nk = P.f_key_ptr[f + 1] - kp;
      const int v0 = nk > 0 ? P.f_vars[kp] : -1, v1 = nk > 1 ? P.f_vars[kp + 1] : -1;
      fr[f] = FactorRec{(i64)P.f_jac_off[f], v0 >= 0 ? P.state_off[v0] : -1, v1 >= 0 ? P.state_off[v1] : -1,
                        (int)P.f_meas_ptr[f], (int)P.f_noise_ptr[f],
                        P.f_type[f] | (P.f_noise_kind[f] << 8) | ((v0 >= 0 ? P.types[v0] : 0) << 24),
                        P.f_rows[f] | (std::min(v0 >= 0 ? P.dims[v0] : 0, 255) << 8) | (P.f_cols[f] << 16)};
    }
    HIPCHK(c, c->d_frec.upload(fr, st));
  }
  DevProblem& D = c->DP;
  D.n_vars = P.n_vars;
  D.n_factors = P.n_factors;
  D.frec = c->d_frec.p;
  D.var_type = c->d_var_type.p; D.var_dim = c->d_var_dim.p;
  D.var_state_off = c->d_var_state_off.p; D.var_tan_off = c->d_var_tan_off.p;
  D.f_type = c->d_f_type.p; D.f_rows = c->d_f_rows.p; D.f_key_ptr = c->d_f_key_ptr.p; D.f_vars = c->d_f_vars.p;
  D.f_noise_kind = c->d_f_noise_kind.p; D.f_cols = c->d_f_cols.p;
  D.f_meas_off = c->d_f_meas_off.p; D.f_noise_off = c->d_f_noise_off.p; D.f_jac_off = c->d_f_jac_off.p;
  D.meas = c->d_meas.p; D.noise = c->d_noise.p;
  HIPCHK(c, hipStreamSynchronize(st));
  return GSX_OK;
}

int small_threads_for(int n) {
  // (twice the threads the row-per-thread panel needed: the assembly and the stores of a front are a handful of memory
  //  round trips that more lanes shorten, and the tile phases of the factorization split over the waves; measured -2 %
  //  on the 100 000-pose graphs, nothing beyond 2x; GSX_SMALL_THREADS_SHIFT overrides for experiments)
  static const int shift = std::getenv("GSX_SMALL_THREADS_SHIFT") ? std::atoi(std::getenv("GSX_SMALL_THREADS_SHIFT")) : 1;
  int t = 512;
  if (n <= 48) t = 64;
  else if (n <= 72) t = 128;
  else if (n <= 100) t = 256;
  return std::min(512, t << shift);
}

// Build the launch plan (host) and upload the symbolic tables.
gsx_status upload_symbolic(gsx_context* c) {
  const Symbolic& S = c->S;
  const HostProblem& P = c->P;
  hipStream_t st = c->stream;
  HIPCHK(c, c->d_fr_off.upload(std::vector<i64>(S.off.begin(), S.off.end()), st));
  HIPCHK(c, c->d_fr_N.upload(S.N, st));
  HIPCHK(c, c->d_fr_F.upload(S.F, st));
  HIPCHK(c, c->d_fr_nfv.upload(S.nfrontal_vars, st));
  HIPCHK(c, c->d_fr_fvar_ptr.upload(S.fvar_ptr, st));
  HIPCHK(c, c->d_fvars.upload(S.fvars, st));
  HIPCHK(c, c->d_fr_parent.upload(S.parent, st));
  {
    std::vector<int> flags(S.n_fronts);  // bit 0: lean leaf, bit 1: blocked (big) layout
    for (int f = 0; f < S.n_fronts; ++f) flags[f] = (S.lean[f] ? 1 : 0) | (S.cls[f] == 2 ? 2 : 0);
    HIPCHK(c, c->d_fr_lean.upload(flags, st));
  }
  if (c->sharded()) {
    // what this rank evaluates, and the masks that complete a vector across the ranks: `own` = this rank's subtree
    // variables (+ the cap's on rank 0): each scalar owned exactly once; `sched` = every variable this rank assembles
    gsx_status fl = upload_factor_lists(c, &S.f_owned);
    if (fl != GSX_OK) return fl;
    std::vector<unsigned char> own_tan(std::max<int64_t>(P.tan_size, 1), 0), sched_tan(own_tan.size(), 0),
        own_state(std::max<int64_t>(P.state_size, 1), 0);
    for (int v = 0; v < P.n_vars; ++v) {
      const int o = S.owner[S.front_of_var[v]];
      const bool own = o < 0 ? c->shard_rank == 0 : o == c->shard_rank;
      const bool sched = o < 0 || o == c->shard_rank;
      for (int d = 0; d < P.dims[v]; ++d) {
        own_tan[P.tan_off[v] + d] = own;
        sched_tan[P.tan_off[v] + d] = sched;
      }
      const int sd = (v + 1 < P.n_vars ? P.state_off[v + 1] : (int)P.state_size) - P.state_off[v];
      for (int d = 0; d < sd; ++d) own_state[P.state_off[v] + d] = own;
    }
    HIPCHK(c, c->d_own_tan.upload(own_tan, st));
    HIPCHK(c, c->d_sched_tan.upload(sched_tan, st));
    HIPCHK(c, c->d_own_state.upload(own_state, st));
    HIPCHK(c, c->d_xscal.alloc(kXScalars));
    HIPCHK(c, hipMemsetAsync(c->d_delta.p, 0, std::max<int64_t>(P.tan_size, 1) * sizeof(double), st));
  }
  HIPCHK(c, c->d_fr_child_ptr.upload(S.child_ptr, st));
  HIPCHK(c, c->d_children.upload(S.children, st));
  HIPCHK(c, c->d_cmap_ptr.upload(std::vector<i64>(S.cmap_ptr.begin(), S.cmap_ptr.end()), st));
  HIPCHK(c, c->d_gidx_ptr.upload(std::vector<i64>(S.gidx_ptr.begin(), S.gidx_ptr.end()), st));
  HIPCHK(c, c->d_cmap.upload(S.cmap, st));
  HIPCHK(c, c->d_gidx.upload(S.gidx, st));
  HIPCHK(c, c->d_h_off.upload(std::vector<i64>(S.h_off.begin(), S.h_off.end()), st));
  HIPCHK(c, c->d_hmap_ptr.upload(std::vector<i64>(S.hmap_ptr.begin(), S.hmap_ptr.end()), st));
  HIPCHK(c, c->d_h_rows.upload(S.h_rows, st));
  HIPCHK(c, c->d_hmap.upload(S.hmap, st));
  HIPCHK(c, c->d_h_loc.upload(S.h_loc, st));
  HIPCHK(c, c->d_term_ptr.upload(std::vector<i64>(S.term_ptr.begin(), S.term_ptr.end()), st));
  {
    std::vector<TermRec> terms(S.t_jac.size());
    for (size_t i = 0; i < terms.size(); ++i)
      terms[i] = TermRec{(i64)S.t_jac[i], S.t_m[i], S.t_colA[i], S.t_colB[i], S.t_dB[i], S.t_dst[i], 0};
    HIPCHK(c, c->d_terms.upload(terms, st));
    std::vector<VarRec> vrec(P.n_vars);
    for (int v = 0; v < P.n_vars; ++v)
      vrec[v] = VarRec{(i64)S.h_off[v], (i64)S.hmap_ptr[v], P.dims[v], S.h_rows[v], S.h_loc[v], P.tan_off[v]};
    std::vector<ChildRec> crec(S.children.size());
    for (size_t i = 0; i < crec.size(); ++i) {
      const int ch = S.children[i];
      crec[i] = ChildRec{(i64)S.off[ch] + (i64)S.F[ch] * S.N[ch] + S.F[ch], (i64)S.cmap_ptr[ch], S.N[ch],
                         S.N[ch] - S.F[ch], 0, 0};
    }
    std::vector<VarRec> fvrec(S.fvars.size());
    for (size_t i = 0; i < fvrec.size(); ++i) fvrec[i] = vrec[S.fvars[i]];
    std::vector<FrontRec> frec(S.n_fronts);
    for (int f = 0; f < S.n_fronts; ++f)
      frec[f] = FrontRec{(i64)S.off[f], S.N[f], S.F[f], S.nfrontal_vars[f], S.fvar_ptr[f], S.child_ptr[f],
                         S.child_ptr[f + 1] - S.child_ptr[f]};
    HIPCHK(c, c->d_var_recs.upload(vrec, st));
    HIPCHK(c, c->d_child_recs.upload(crec, st));
    HIPCHK(c, c->d_front_recs.upload(frec, st));
    HIPCHK(c, c->d_cond_last.upload(std::vector<i64>(S.cond_last.begin(), S.cond_last.end()), st));
    HIPCHK(c, c->d_cond_prev.upload(std::vector<i64>(S.cond_prev.begin(), S.cond_prev.end()), st));
    HIPCHK(c, c->d_cond_front.upload(S.cond_front, st));
    HIPCHK(c, c->d_fvar_recs.upload(fvrec, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  HIPCHK(c, c->d_sched.upload(S.sched, st));
  {
    // leaf-kernel cliques are childless, hence all at level 0, first in its schedule
    const int b = S.n_levels ? S.lvl_ptr[0] : 0, e = S.n_levels ? S.lvl_leaf_end[0] : 0;
    std::vector<LeafRec> lr((size_t)std::max(e - b, 0));
    for (int k = b; k < e; ++k) {
      const int f = S.sched[k];
      const int v0 = S.fvars[S.fvar_ptr[f]];
      lr[k - b] = LeafRec{(i64)S.off[f], (i64)S.h_off[v0], (i64)S.hmap_ptr[v0], (i64)S.gidx_ptr[f],
                          S.N[f], S.F[f], S.nfrontal_vars[f], (int)S.lean[f],
                          P.dims[v0], S.h_rows[v0], S.h_loc[v0], P.tan_off[v0],
                          S.fvar_ptr[f], f, 0, 0};
    }
    c->leaf_base = b;
    c->h_leaf_recs = lr;
    HIPCHK(c, c->d_leaf_recs.upload(lr, st));
  }
  HIPCHK(c, c->d_H.alloc(std::max<int64_t>(S.h_size, 1)));
  HIPCHK(c, c->d_arena.alloc(std::max<int64_t>(S.arena_size, 1)));
  // ---- H assembly groups: variables bucketed by panel size (LDS) and term count -----------------
  {
    struct VI {
      int v, psize;
      int64_t terms;
    };
    std::vector<VI> light, heavy, huge, diag, star, tile;
    // a "tile" variable: at most 64 terms, all from NARROW factors (<= 16 columns with the rhs, <= 8 rows): one matrix-core
    // product per factor (assemble_h_tile_kernel).  GSX_H_TILE_OFF keeps the generic kernel (measurements).
    static const bool tile_off = std::getenv("GSX_H_TILE_OFF") != nullptr;
    auto is_tile = [&](int v) {
      const int64_t tb = S.term_ptr[v], te = S.term_ptr[v + 1];
      if (tile_off || te - tb < 2 || te - tb > 64 || (int64_t)S.h_rows[v] * P.dims[v] * 8 * 4 > 48 * 1024) return false;
      int64_t t = tb;
      while (t < te) {
        int64_t e = t + 1;
        while (e < te && S.t_jac[e] == S.t_jac[t]) ++e;
        // the factor's last term is its rhs term: column index = number of columns - 1
        if (S.t_m[t] > 8 || S.t_dB[e - 1] != 1 || S.t_dst[e - 1] != S.h_rows[v] - 1 || S.t_colB[e - 1] + 1 > 16) return false;
        t = e;
      }
      return true;
    };
    // a "star" variable: all its factors are binary with a later-eliminated partner, all of one shape (three terms
    // each: own block, partner block, rhs — with equal rows, columns and partner dimension)
    std::vector<int> star_dst;
    auto is_star = [&](int v) {
      const int64_t tb = S.term_ptr[v], te = S.term_ptr[v + 1];
      if (te - tb < 3 || (te - tb) % 3 != 0 || P.dims[v] > 7) return false;
      star_dst.clear();
      for (int64_t t = tb; t < te; t += 3) {
        star_dst.push_back(S.t_dst[t + 1]);
        if (S.t_dst[t] != 0 || S.t_dst[t + 2] != S.h_rows[v] - 1 || S.t_dB[t + 2] != 1) return false;
        if (S.t_dst[t + 1] == 0 || S.t_dst[t + 1] == S.h_rows[v] - 1) return false;
        if (S.t_m[t] != S.t_m[tb] || S.t_colA[t] != S.t_colA[tb] || S.t_colB[t + 1] != S.t_colB[tb + 1] ||
            S.t_dB[t + 1] != S.t_dB[tb + 1] || S.t_colB[t + 2] != S.t_colB[tb + 2] || S.t_jac[t] != S.t_jac[t + 1] ||
            S.t_jac[t] != S.t_jac[t + 2])
          return false;
      }
      // every partner block is written by exactly one factor (two factors to the same partner would have to be summed)
      std::sort(star_dst.begin(), star_dst.end());
      return std::adjacent_find(star_dst.begin(), star_dst.end()) == star_dst.end();
    };
    for (int v = 0; v < P.n_vars; ++v) {
      if (!S.scheduled[S.front_of_var[v]]) continue;  // another rank's subtree
      const int psize = S.h_rows[v] * P.dims[v];
      const int64_t terms = S.term_ptr[v + 1] - S.term_ptr[v];
      if (is_star(v)) {
        star.push_back({v, psize, terms});
        continue;
      }
      // no later neighbour through any factor (panel = own block + rhs) and many factors: the matrix-core kernel
      if (terms >= 64 && S.h_rows[v] == P.dims[v] + 1 && P.dims[v] <= 15) diag.push_back({v, psize, terms});
      else if (is_tile(v)) tile.push_back({v, psize, terms});
      else if (terms >= 96 && (int64_t)psize * 4 * 8 <= 48 * 1024) heavy.push_back({v, psize, terms});
      else if ((int64_t)psize * 8 <= 48 * 1024) light.push_back({v, psize, terms});
      else huge.push_back({v, psize, terms});
    }
    auto by_psize = [](const VI& a, const VI& b) { return a.psize < b.psize; };
    std::stable_sort(light.begin(), light.end(), by_psize);
    std::stable_sort(heavy.begin(), heavy.end(), by_psize);
    std::vector<int> hv;
    c->hgroups.clear();
    auto emit = [&](std::vector<VI>& L, int threads, int copies, bool global) {
      size_t i = 0;
      while (i < L.size()) {
        // group = run whose largest panel is at most 2x (and at least 1 KB) of its smallest
        size_t j = i;
        const int lo = std::max(L[i].psize, 128);
        while (j < L.size() && L[j].psize <= 2 * lo) ++j;
        const int maxp = L[j - 1].psize;
        c->hgroups.push_back({(int)hv.size(), (int)(j - i), threads, global ? 0 : maxp * copies * 8, global});
        for (size_t k = i; k < j; ++k) hv.push_back(L[k].v);
        i = j;
      }
    };
    std::stable_sort(tile.begin(), tile.end(), by_psize);
    emit(tile, -2, 4, false);   // (four waves a workgroup, a panel each)
    emit(light, 64, 1, false);
    emit(heavy, 256, 4, false);
    emit(huge, 64, 1, true);
    c->n_star_bundles = 0;
    if (!star.empty()) {  // threads == -1 marks the group
      c->hgroups.push_back({(int)hv.size(), (int)star.size(), -1, 0, false});
      for (const VI& x : star) hv.push_back(x.v);
      // bundles for the one-wave-per-bundle kernel: consecutive variables of one shape, <= 5 of them (the own entries of
      // a bundle fit a wave: 5 x (3*3 + 3) lanes), <= 64 factors; a variable too big or of another shape is alone
      auto shape = [&](int v) {
        const int64_t t = S.term_ptr[v];
        return std::array<int, 6>{P.dims[v], S.t_m[t], S.t_colA[t], S.t_colB[t + 1], S.t_dB[t + 1], S.t_colB[t + 2]};
      };
      std::vector<int2> bundles;
      size_t i = 0;
      while (i < star.size()) {
        const auto sh = shape(star[i].v);
        const int nown = sh[0] * sh[0] + sh[0];
        int nf = (int)(star[i].terms / 3), nv = 1;
        if (nf > 64 || nown > 64) {  // (the bundle kernel holds a factor per lane)
          bundles.clear();
          break;
        }
        while (i + nv < star.size() && nv < 5 && (nv + 1) * nown <= 64 && shape(star[i + nv].v) == sh &&
               nf + (int)(star[i + nv].terms / 3) <= 64) {
          nf += (int)(star[i + nv].terms / 3);
          ++nv;
        }
        bundles.push_back(int2{(int)i, nv});
        i += nv;
      }
      c->n_star_bundles = (int)bundles.size();
      if (c->n_star_bundles) HIPCHK(c, c->d_star_bundles.upload(bundles, st));
    }
    if (!diag.empty()) {  // threads == 0 marks the group for launch_assemble_h_group
      c->hgroups.push_back({(int)hv.size(), (int)diag.size(), 0, 0, false});
      for (const VI& x : diag) hv.push_back(x.v);
    }
    c->hv_list = hv;
    c->hv_group_of_var.assign(P.n_vars, -1);
    c->hv_pos.assign(P.n_vars, -1);
    for (size_t g = 0; g < c->hgroups.size(); ++g)
      for (int k = c->hgroups[g].begin; k < c->hgroups[g].begin + c->hgroups[g].count; ++k) {
        c->hv_group_of_var[hv[k]] = (int)g;
        c->hv_pos[hv[k]] = k;
      }
    HIPCHK(c, c->d_hvars.upload(hv, st));
  }
  // ---- factorization launch plan ---------------------------------------------------------------------
  c->small_launch.assign(S.n_levels, {});
  c->big_descs.clear();
  c->big_max_n = c->big_max_nfv = 0;
  c->leaf_launch.assign(S.n_levels, {});
  c->leaf_max_F.assign(S.n_levels, 0);
  std::vector<int> rest_ids;
  for (int l = 0; l < S.n_levels; ++l) {
    int i = S.lvl_ptr[l];
    for (int k = i; k < S.lvl_leaf_end[l]; ++k) c->leaf_max_F[l] = std::max(c->leaf_max_F[l], S.F[S.sched[k]]);
    // leaf-kernel fronts: sorted by (F, N); a launch = same F, panel size within 1.5x — or more, see below
    const int le_all = S.lvl_leaf_end[l];
    if (l == 0) c->leaf_side_group0 = (size_t)-1;
    for (int part = 0; part < 2; ++part) {   // the leaves launched with the level, then (level 0 only) the side leaves
    const int le = (l == 0 && part == 0) ? S.leaf_side_begin : le_all;
    if (l == 0 && part == 1) c->leaf_side_group0 = c->leaf_launch[l].size();
    while (i < le) {
      const int F0 = S.F[S.sched[i]], n0 = std::max(S.N[S.sched[i]], 8);
      int j = i, maxp = 0, maxn = 0;
      // (when every leaf of this width is lean and narrow — the landmarks under blocked camera fronts: panel only, no
      //  Schur complement to form — they go in ONE launch of one-wave workgroups whatever their height: four launches by
      //  panel height took 178 us on BAL-1723, one takes 155.  Otherwise the 1.5x rule and the thread classes: the outer
      //  product of a stored complement wants a thread per row, and the Pose2 leaves got slower in one launch.)
      bool narrow = F0 <= 4;
      for (int k = i; narrow && k < le && S.F[S.sched[k]] == F0; ++k) narrow = S.lean[S.sched[k]] != 0;
      // (few leaves — every level of a pose graph: ONE launch whatever their width and height, with the thread class of
      //  the tallest; the launches by (F, height) were each bound by the latency of one leaf: pose2_100k 11 x 18 us ->
      //  38 us, pose3_100k 4 x 21 -> 30)
      const bool all = !narrow && (le - S.lvl_ptr[l]) <= 16384;
      // (experiment, off: GSX_LEAF_SPLIT=<doubles> puts the narrow cliques with a panel above the limit in a launch of their
      //  own, so that launch_front_leaf can pack the others four to a workgroup — GSX_LEAF_PACK, measured: no gain)
      static const int pack_limit = std::getenv("GSX_LEAF_SPLIT") ? std::atoi(std::getenv("GSX_LEAF_SPLIT")) : (1 << 30);
      auto packable = [&](int k) { return S.N[S.sched[k]] * S.F[S.sched[k]] <= pack_limit; };
      while (j < le && (all || (S.F[S.sched[j]] == F0 &&
                                ((narrow && packable(j) == packable(i)) || (!narrow && S.N[S.sched[j]] * 2 <= n0 * 3))))) {
        maxp = std::max(maxp, S.N[S.sched[j]] * S.F[S.sched[j]]);
        maxn = std::max(maxn, S.N[S.sched[j]]);
        ++j;
      }
      // (the outer product of a stored complement runs a thread per trailing row: never fewer threads than n - F)
      int rows_below = 0;
      for (int k = i; k < j; ++k) rows_below = std::max(rows_below, S.N[S.sched[k]] - S.F[S.sched[k]]);
      c->leaf_launch[l].push_back({i, j - i, maxn, narrow ? 64 : (maxn <= 72 && rows_below <= 64 ? 64 : (maxn <= 110 ? 128 : 256)), maxp});
      i = j;
    }
    }
    const int se = S.lvl_small_end[l];
    int se_lds = se;   // the medium fronts (n > kSmallMaxN) sit at the end of the level's LDS-class range (sorted by n)
    while (se_lds > i && S.med[S.sched[se_lds - 1]]) --se_lds;
    while (i < se_lds) {
      const int thr = small_threads_for(S.N[S.sched[i]]);
      int j = i, maxn = 0;
      // same thread class, and rows within 1.6x of the smallest member (LDS footprint within 2.6x: in effect one group per
      // thread class — measured on the 100 000-pose graphs: 4 launches per level beat 7 finer ones (1.3x) by 3 %)
      const int n0 = std::max(S.N[S.sched[i]], 12);
      while (j < se_lds && small_threads_for(S.N[S.sched[j]]) == thr && S.N[S.sched[j]] * 10 <= n0 * 16) {
        maxn = std::max(maxn, S.N[S.sched[j]]);
        ++j;
      }
      c->small_launch[l].push_back({i, j - i, maxn, thr});
      i = j;
    }
    // A launch of few fronts is bound by the latency of ONE front (20-60 us), not by occupancy: merge the
    // size groups of a level when that costs no residency — all 64-thread groups if they total <= 2048
    // fronts (48^2 doubles = 18 KB LDS each: 8 per CU), all wider groups if they total <= 256 fronts.
    {
      std::vector<SmallLaunch>& G = c->small_launch[l];
      std::vector<SmallLaunch> merged;
      size_t k = 0;
      while (k < G.size()) {
        const bool narrow = G[k].threads == 64;
        size_t e = k;
        int total = 0, maxn = 0, thr = 0;
        while (e < G.size() && (G[e].threads == 64) == narrow) {
          total += G[e].count;
          maxn = std::max(maxn, G[e].max_n);
          thr = std::max(thr, G[e].threads);
          ++e;
        }
        if (total <= (narrow ? 2048 : 256)) {
          merged.push_back({G[k].begin, total, maxn, thr});
        } else {
          for (size_t q = k; q < e; ++q) merged.push_back(G[q]);
        }
        k = e;
      }
      G.swap(merged);
      // ... and when the whole level has at most 256 small fronts (one workgroup per CU even at the full LDS size) a
      // single launch: the level then costs the latency of its slowest front once, not once per size group
      int total = 0, maxn = 0;
      for (const SmallLaunch& g : G) {
        total += g.count;
        maxn = std::max(maxn, g.max_n);
      }
      if (G.size() > 1 && total <= 256) {
        const SmallLaunch one{G[0].begin, total, maxn, std::max(256, small_threads_for(maxn))};
        G.assign(1, one);
      }
      if (se_lds < se) {   // the level's medium fronts: one group
        SmallLaunch m{se_lds, se - se_lds, 0, 512};
        m.medium = true;
        for (int k = se_lds; k < se; ++k) {
          m.max_n = std::max(m.max_n, S.N[S.sched[k]]);
          m.max_panel = std::max(m.max_panel, S.N[S.sched[k]] * S.F[S.sched[k]]);
        }
        G.push_back(m);
      }
    }
  }
  // ---- the UPPER schedule (Symbolic::ulevel): what the level loop of the full factorization / back-substitution runs —
  //      per upper level the LDS-class fronts that are not tree fronts (whole-front-in-LDS ones, then medium ones: two
  //      launches at most) and the blocked fronts ----
  c->big_level.assign(S.n_ulevels, BigLevel());
  c->rest_launch.assign(S.n_ulevels, {});
  for (int l = 0; l < S.n_ulevels; ++l) {
    const int b0 = S.ulvl_ptr[l], se = S.ulvl_small_end[l];
    for (int pass = 0; pass < 2; ++pass) {
      SmallLaunch r{(int)rest_ids.size(), 0, 0, 0};
      r.medium = pass == 1;
      for (int k = b0; k < se; ++k) {
        const int f = S.usched[k];
        if ((S.med[f] != 0) != r.medium) continue;
        rest_ids.push_back(f);
        r.count++;
        r.max_n = std::max(r.max_n, S.N[f]);
        r.max_F = std::max(r.max_F, S.F[f]);
        r.max_panel = std::max(r.max_panel, S.N[f] * S.F[f]);
      }
      r.threads = r.medium ? 512 : small_threads_for(r.max_n);
      if (r.count) c->rest_launch[l].push_back(r);
    }
    BigLevel& B = c->big_level[l];
    B.begin = (int)c->big_descs.size();
    for (int k = se; k < S.ulvl_ptr[l + 1]; ++k) {
      const int f = S.usched[k];
      c->big_descs.push_back(BigDesc{(i64)S.off[f], (i64)S.off[f] + big_panel_offset(S.N[f]), S.N[f], S.F[f], f,
                                     S.parent[f]});
      c->big_max_n = std::max(c->big_max_n, S.N[f]);
      c->big_max_nfv = std::max(c->big_max_nfv, S.nfrontal_vars[f]);
    }
    B.count = (int)c->big_descs.size() - B.begin;
    plan_big_group(c->big_descs.data() + B.begin, B.count, B.plan);
  }
  HIPCHK(c, c->d_usched.upload(S.usched, st));
  HIPCHK(c, c->d_rest_ids.upload(rest_ids, st));
  // ---- hard constraints: the constrained fronts by upper level, their rows and work areas ----
  c->con_level.assign(S.n_ulevels, gsx_context::ConLevel());
  c->con_desc_of_front.assign(S.n_fronts, -1);
  c->CT = ConTables{};
  if (!S.con_fronts.empty()) {
    const int nc = (int)S.con_fronts.size();
    std::vector<int> perm(nc);   // desc position -> index in S.con_fronts
    for (int k = 0; k < nc; ++k) perm[k] = k;
    std::stable_sort(perm.begin(), perm.end(),
                     [&](int a, int b) { return S.ulevel[S.con_fronts[a]] < S.ulevel[S.con_fronts[b]]; });
    std::vector<int> where(nc);
    for (int k = 0; k < nc; ++k) where[perm[k]] = k;
    std::vector<ConDesc> descs(nc);
    std::vector<int> child_list;
    i64 work = 0, iwork = 0;
    for (int k = 0; k < nc; ++k) {
      const int q = perm[k], f = S.con_fronts[q], l = S.ulevel[f];
      ConDesc& d = descs[k];
      d.off = S.off[f];
      d.N = S.N[f];
      d.F = S.F[f];
      d.K = S.con_in[q];
      d.n_fwd = S.con_fwd[q];
      d.work = work;
      work += 3 * (i64)d.K * d.N;
      d.fwd = work;
      work += (i64)d.n_fwd * (d.N - d.F);
      d.iwork = iwork;
      iwork += d.N + 2 * d.K + 1;
      d.own_begin = S.con_own_ptr[q];
      d.own_end = S.con_own_ptr[q + 1];
      d.child_begin = (int)child_list.size();
      for (int e = S.con_child_ptr[q]; e < S.con_child_ptr[q + 1]; ++e) child_list.push_back(where[S.con_child[e]]);
      d.child_end = (int)child_list.size();
      d.front = f;
      d.fwd_map = S.con_fwd_map_ptr[q];
      c->con_desc_of_front[f] = k;
      gsx_context::ConLevel& L = c->con_level[l];
      if (!L.count) L.first = k;
      L.count++;
      L.max_n = std::max(L.max_n, d.N);
    }
    if (child_list.empty()) child_list.push_back(0);
    HIPCHK(c, c->d_con_descs.upload(descs, st));
    HIPCHK(c, c->d_con_own_col_ptr.upload(S.con_own_col_ptr, st));
    HIPCHK(c, c->d_con_own_cols.upload(S.con_own_cols, st));
    HIPCHK(c, c->d_con_own_m.upload(S.con_own_m, st));
    std::vector<i64> oj(S.con_own_jac.begin(), S.con_own_jac.end());
    if (oj.empty()) oj.push_back(0);
    HIPCHK(c, c->d_con_own_jac.upload(oj, st));
    HIPCHK(c, c->d_con_child.upload(child_list, st));
    std::vector<int> fmap = S.con_fwd_map;
    if (fmap.empty()) fmap.push_back(-1);
    HIPCHK(c, c->d_con_fwd_map.upload(fmap, st));
    HIPCHK(c, c->d_con_work.alloc(std::max<i64>(work, 1)));
    HIPCHK(c, c->d_con_iwork.alloc(std::max<i64>(iwork, 1)));
    c->CT = ConTables{c->d_con_descs.p, c->d_con_own_col_ptr.p, c->d_con_own_cols.p, c->d_con_own_m.p, c->d_con_own_jac.p,
                      c->d_con_child.p, c->d_con_fwd_map.p, c->d_con_work.p, c->d_con_iwork.p};
  }
  {
    c->tree_tiers.clear();
    const int nb = (int)S.tree_bounds.size();
    const int ntier = std::min<int>((int)S.tree_start_ptr.size() - 1, kTreeCursors);
    for (int t = 0; t < ntier; ++t) {
      int maxn = 0;
      size_t lds = 0;   // the medium tier: doubles of LDS of its most demanding front
      for (int f = 0; f < S.n_fronts; ++f)
        if (S.tree_tier[f] == t) {
          maxn = std::max(maxn, S.N[f]);
          lds = std::max(lds, S.med[f] ? (size_t)S.N[f] * S.F[f] + S.N[f] : (size_t)S.N[f] * S.N[f] + S.N[f]);
        }
      c->tree_tiers.push_back({S.tree_start_ptr[t], S.tree_start_ptr[t + 1] - S.tree_start_ptr[t], maxn,
                               t < nb ? (S.tree_threads[t] > 0 ? S.tree_threads[t] : small_threads_for(maxn)) : 512,
                               t >= nb ? lds * sizeof(double) : (size_t)0});
    }
    HIPCHK(c, c->d_tree_start.upload(S.tree_start, st));
    HIPCHK(c, c->d_tree_up.upload(S.tree_up, st));
    HIPCHK(c, c->d_tree_npend.upload(S.tree_npend, st));
    HIPCHK(c, c->d_tree_pending.upload(S.tree_npend, st));
    HIPCHK(c, c->d_tree_cursor.upload(std::vector<int>(kTreeCursors, 0), st));
    // top-down: roots = tree fronts whose parent is not a tree front; per front its tree children, the deepest subtree
    // first (a workgroup goes on with that one itself); roots by depth as well: the long chains start first
    std::vector<int> roots, cptr(S.n_fronts + 1, 0), cidx, height(S.n_fronts, 0);
    for (int f = 0; f < S.n_fronts; ++f)   // children have smaller ids
      if (S.tree_tier[f] >= 0 && S.parent[f] >= 0 && S.tree_tier[S.parent[f]] >= 0)
        height[S.parent[f]] = std::max(height[S.parent[f]], height[f] + 1);
    int total = 0;
    for (int f = 0; f < S.n_fronts; ++f) {
      const size_t b = cidx.size();
      for (int k = S.child_ptr[f]; k < S.child_ptr[f + 1]; ++k)
        if (S.tree_tier[f] >= 0 && S.tree_tier[S.children[k]] >= 0) cidx.push_back(S.children[k]);
      std::stable_sort(cidx.begin() + b, cidx.end(), [&](int x, int y) { return height[x] > height[y]; });
      cptr[f + 1] = (int)cidx.size();
      if (S.tree_tier[f] < 0) continue;
      total += (int)(cidx.size() - b) > 1 ? (int)(cidx.size() - b) - 1 : 0;   // published entries
      if (S.parent[f] < 0 || S.tree_tier[S.parent[f]] < 0) roots.push_back(f);
    }
    std::stable_sort(roots.begin(), roots.end(), [&](int x, int y) { return height[x] > height[y]; });
    total += (int)roots.size();   // = tickets
    c->bst_roots = (int)roots.size();
    c->bst_total = total;
    HIPCHK(c, c->d_bst_roots.upload(roots, st));
    HIPCHK(c, c->d_bst_child_ptr.upload(cptr, st));
    HIPCHK(c, c->d_bst_children.upload(cidx, st));
    HIPCHK(c, c->d_bst_counters.upload(std::vector<int>(2, 0), st));
    HIPCHK(c, c->d_bst_ready.upload(std::vector<unsigned long long>((size_t)std::max(total - (int)roots.size(), 1), 0ull), st));
  }
  // front -> where it sits in the launch plan (for the filtered plans of gsx_relinearize_partial)
  c->fr_sched_pos.assign(S.n_fronts, -1);
  c->fr_group.assign(S.n_fronts, -1);
  for (size_t k = 0; k < S.sched.size(); ++k) c->fr_sched_pos[S.sched[k]] = (int)k;
  for (int l = 0; l < S.n_levels; ++l) {
    for (size_t g = 0; g < c->leaf_launch[l].size(); ++g)
      for (int k = c->leaf_launch[l][g].begin; k < c->leaf_launch[l][g].begin + c->leaf_launch[l][g].count; ++k)
        c->fr_group[S.sched[k]] = (int)g;
    for (size_t g = 0; g < c->small_launch[l].size(); ++g)
      for (int k = c->small_launch[l][g].begin; k < c->small_launch[l][g].begin + c->small_launch[l][g].count; ++k)
        c->fr_group[S.sched[k]] = (int)g;
  }
  for (size_t k = 0; k < c->big_descs.size(); ++k) c->fr_group[c->big_descs[k].front] = (int)k;
  {
    auto csr = [&](const std::vector<int>& task_of, const std::vector<int>& lvl_ptr, std::vector<int>& ptr,
                   std::vector<int>& items, std::vector<int>& level_of) {
      ptr.assign(S.n_fronts + 1, 0);
      level_of.assign(task_of.size(), 0);   // (the gather GROUP: an upper level, or n_ulevels = the side group)
      for (int l = 0; l + 1 < (int)lvl_ptr.size(); ++l)
        for (int i = lvl_ptr[l]; i < lvl_ptr[l + 1]; ++i) level_of[i] = l;
      for (int t : task_of) ptr[S.gt_front[t] + 1]++;
      for (int f = 0; f < S.n_fronts; ++f) ptr[f + 1] += ptr[f];
      items.resize(task_of.size());
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (size_t i = 0; i < task_of.size(); ++i) items[fill[S.gt_front[task_of[i]]]++] = (int)i;
    };
    csr(S.gseg_task, S.gseg_lvl_ptr, c->fr_seg_ptr, c->fr_segs, c->seg_level);
    csr(S.gm_task, S.gm_lvl_ptr, c->fr_gm_ptr, c->fr_gms, c->gm_level);
  }
  HIPCHK(c, c->d_big.upload(c->big_descs, st));
  {
    // gather sources as absolute arena offsets + leading dimension (no dependent metadata loads in the kernel)
    std::vector<GatherSrc> srcs(S.gs_child.size());
    for (size_t i = 0; i < S.gs_child.size(); ++i) {
      const int ch = S.gs_child[i];
      if (S.N[ch] >= (1 << 24)) {
        c->err = "front with more than 2^24 rows";
        return GSX_E_INVALID;
      }
      const bool lean = S.gs_loc2[i] >= 0;
      srcs[i] = GatherSrc{(i64)S.off[ch] + S.gs_loc[i], lean ? S.gs_loc2[i] - S.gs_loc[i] : 0,
                          S.N[ch] | (lean ? S.F[ch] << 24 : 0)};
    }
    std::vector<GatherSeg> segs(S.gseg_task.size());
    for (size_t i = 0; i < segs.size(); ++i) {
      const int t = S.gseg_task[i];
      segs[i] = GatherSeg{(i64)S.gt_dst[t], (i64)S.gseg_begin[i], (int)(S.gseg_end[i] - S.gseg_begin[i]), S.gt_ld[t],
                          S.gt_dims[t], S.gseg_slot[i]};
    }
    HIPCHK(c, c->d_gt_dst.upload(std::vector<i64>(S.gt_dst.begin(), S.gt_dst.end()), st));
    HIPCHK(c, c->d_gt_ld.upload(S.gt_ld, st));
    HIPCHK(c, c->d_gt_dims.upload(S.gt_dims, st));
    HIPCHK(c, c->d_gsrcs.upload(srcs, st));
    c->h_gsegs = segs;
    HIPCHK(c, c->d_gsegs.upload(segs, st));
    HIPCHK(c, c->d_gm_task.upload(S.gm_task, st));
    HIPCHK(c, c->d_gm_slot.upload(S.gm_slot, st));
    HIPCHK(c, c->d_gm_nslots.upload(S.gm_nslots, st));
    HIPCHK(c, c->d_gscratch.alloc((size_t)std::max(S.g_max_slots, 1) * 256));
    HIPCHK(c, hipStreamSynchronize(st));
    c->GA = GatherArgs{c->d_gsegs.p, c->d_gsrcs.p, c->d_gt_dst.p, c->d_gt_ld.p, c->d_gt_dims.p, c->d_gm_task.p,
                       c->d_gm_slot.p, c->d_gm_nslots.p, c->d_gscratch.p};
  }
  DevSymbolic& D = c->DS;
  D.n_fronts = S.n_fronts;
  D.fr_off = c->d_fr_off.p; D.fr_N = c->d_fr_N.p; D.fr_F = c->d_fr_F.p; D.fr_nfv = c->d_fr_nfv.p;
  D.fr_fvar_ptr = c->d_fr_fvar_ptr.p; D.fvars = c->d_fvars.p; D.fr_parent = c->d_fr_parent.p;
  D.fr_lean = c->d_fr_lean.p;
  D.fr_child_ptr = c->d_fr_child_ptr.p; D.children = c->d_children.p;
  D.cmap_ptr = c->d_cmap_ptr.p; D.gidx_ptr = c->d_gidx_ptr.p; D.cmap = c->d_cmap.p; D.gidx = c->d_gidx.p;
  D.h_off = c->d_h_off.p; D.hmap_ptr = c->d_hmap_ptr.p; D.h_rows = c->d_h_rows.p; D.hmap = c->d_hmap.p;
  D.h_loc = c->d_h_loc.p;
  D.term_ptr = c->d_term_ptr.p; D.terms = c->d_terms.p;
  D.var_recs = c->d_var_recs.p; D.child_recs = c->d_child_recs.p;
  D.front_recs = c->d_front_recs.p; D.fvar_recs = c->d_fvar_recs.p;
  HIPCHK(c, hipStreamSynchronize(st));
  // a new tree: the arena and H were re-allocated for it — nothing computed for the old one may be reused
  c->h_ready = false;
  c->solved = false;
  c->fact_valid = c->fact_pending = false;
  c->wf_all_replaced = true;
  c->wf_delta_valid = false;
  c->damp_ready = c->hdiag_ready = false;   // (sharded: the damping weights carry the ownership mask)
  if (c->sharded()) c->linearized = false;  // the owned factor set changed with the partition
  return GSX_OK;
}

// in-place sum of device memory over the ranks of a sharded problem (gsx_set_shard); a failure is latched and reported
// by the next readback
void shard_allreduce(gsx_context* c, double* dptr, int64_t n) {
  if (!c->sharded() || n <= 0) return;
  // A rank whose own stream has failed still ENTERS the collective — its peers are in it, and skipping it would leave
  // them blocked until the backend's timeout — with its share poisoned by NaNs, so that every rank sees the failure in
  // the sum (non-finite pivots / solution => GSX_E_INDETERMINATE there, the latched failure here).
  if (hipStreamSynchronize(c->stream) != hipSuccess) {
    c->shard_failed = true;
    (void)hipGetLastError();
    (void)hipMemset(dptr, 0xFF, (size_t)n * sizeof(double));   // (all-ones bytes are a NaN)
  }
  if (c->shard_cb(c->shard_user, dptr, n) != 0) c->shard_failed = true;
}

// ---- device pipeline pieces (all asynchronous) -------------------------------------------------------
void dev_linearize(gsx_context* c) {
  timer_begin(c, PH_LINEARIZE);
  hipMemsetAsync(&c->d_status.p->n_cheirality, 0, sizeof(int), c->stream);
  const int* lists[6] = {c->d_type_list[0].p, c->d_type_list[1].p, c->d_type_list[2].p, c->d_type_list[3].p,
                         c->d_type_list[4].p, c->d_type_list[5].p};
  launch_linearize(c->DP, lists, c->type_count, c->d_values.p, c->d_jac.p, c->d_status.p, c->stream);
  timer_end(c, PH_LINEARIZE);
  c->sc_dirty |= kXLin;
  c->linearized = true;
  c->lin0_ready = false;
  c->fact_valid = false;
  c->h_ready = false;
  if (c->damp_kind != 0) c->damp_ready = false;  // (lambda I weights do not depend on the linearization: keep them)
  c->solved = false;
}

void dev_assemble_h(gsx_context* c) {
  timer_begin(c, PH_ASSEMBLE_H);
  // the star bundles (landmarks) and the diagonal-panel variables (cameras) share one launch when both exist
  const gsx_context::HGroup *gs = nullptr, *gd = nullptr;
  for (const auto& g : c->hgroups) {
    if (g.threads == -1 && c->n_star_bundles) gs = &g;
    if (g.threads == 0 && g.count > 0) gd = &g;
  }
  const bool fused = gs && gd;
  if (fused)
    launch_assemble_h_diag_star(c->DP, c->DS, c->d_hvars.p + gd->begin, gd->count, c->d_hvars.p + gs->begin,
                                c->d_star_bundles.p, c->n_star_bundles, c->d_jac.p, c->d_H.p, c->stream);
  for (const auto& g : c->hgroups) {
    if (fused && (&g == gs || &g == gd)) continue;
    if (g.threads == -1 && c->n_star_bundles) {  // the whole star group, a wave per bundle of variables
      launch_assemble_h_star_bundles(c->DP, c->DS, c->d_hvars.p + g.begin, c->d_star_bundles.p, c->n_star_bundles,
                                     c->d_jac.p, c->d_H.p, c->stream);
      continue;
    }
    launch_assemble_h_group(c->DP, c->DS, c->d_hvars.p + g.begin, g.count, g.threads, g.lds, g.global, c->d_jac.p,
                            c->d_H.p, c->stream);
  }
  timer_end(c, PH_ASSEMBLE_H);
  c->h_ready = true;
  c->hdiag_ready = false;  // diag(H) is extracted on demand: lambda * I damping never needs it
}

void dev_hessian_diag(gsx_context* c) {
  if (c->hdiag_ready) return;
  launch_hessian_diag(c->DP, c->DS, c->d_H.p, c->d_hdiag.p, c->stream);
  if (c->con_hd_n)
    launch_constraint_hdiag(c->con_hd_n, c->d_con_hd_tan.p, c->d_con_hd_ptr.p, c->d_con_hd_jidx.p, c->d_con_hd_w.p,
                            c->d_jac.p, c->d_hdiag.p, c->stream);
  if (c->sharded()) {
    // a subtree variable's diagonal lives on its owner, a cap variable's is the sum of every rank's terms
    launch_mask_copy(c->d_hdiag.p, c->d_sched_tan.p, c->P.tan_size, c->d_hdiag.p, c->stream);
    shard_allreduce(c, c->d_hdiag.p, c->P.tan_size);
  }
  c->hdiag_ready = true;
}

void dev_damping(gsx_context* c, int diagonal, double mind, double maxd) {
  if (c->damp_ready && c->damp_kind == diagonal && (diagonal == 0 || (c->damp_min == mind && c->damp_max == maxd))) return;
  if (diagonal) dev_hessian_diag(c);
  launch_make_damping((int)c->P.tan_size, c->d_hdiag.p, diagonal, mind, maxd, c->d_damp.p, c->stream);
  // (a cap variable is damped once: by rank 0's share of the exchange)
  if (c->sharded()) launch_mask_copy(c->d_damp.p, c->d_own_tan.p, c->P.tan_size, c->d_damp.p, c->stream);
  c->damp_ready = true;
  c->damp_kind = diagonal;
  c->damp_min = mind;
  c->damp_max = maxd;
}

// the blocked fronts of one launch group: per round of frontal columns L11 (one workgroup per front), the rows of
// L21 (one per 32 rows), the Schur complement (one per lower tile pair) — bigfront.hip
void dev_big_factor(gsx_context* c, const BigDesc* descs, int count, const BigPlan& plan, hipStream_t st, bool prof) {
  for (int r = 0; r < plan.rounds(); ++r) {
    if (prof) timer_begin(c, PH_K_POTRF0);  // (one HIP-event pair per kernel launch)
    launch_big_diag(descs, count, plan, r, c->d_arena.p, c->d_status.p, st);
    if (prof) timer_end(c, PH_K_POTRF0);
    if (prof) timer_begin(c, PH_K_TRSM);
    launch_big_rows(descs, count, plan, r, c->d_arena.p, st);
    if (prof) timer_end(c, PH_K_TRSM);
    if (prof) timer_begin(c, PH_K_SYRK);
    launch_big_schur(descs, count, plan, r, c->d_arena.p, st);
    if (prof) timer_end(c, PH_K_SYRK);
  }
}

// GSX_DEBUG_LAUNCH=1: synchronise after the launches of the dependency-driven kernels and say what came back
void debug_sync(gsx_context* c, const char* what, int a, int b) {
  static const bool on = std::getenv("GSX_DEBUG_LAUNCH") != nullptr;
  if (!on) return;
  fprintf(stderr, "[gsx] %s (%d, %d) ...", what, a, b);
  fflush(stderr);
  const hipError_t e = hipStreamSynchronize(c->stream);
  const hipError_t e2 = hipGetLastError();
  fprintf(stderr, " %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
  fflush(stderr);
}

// the low-priority queue of the side work (created on first use)
bool bulk_ready(gsx_context* c) {
  if (c->bulk) return true;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;
  if (hipStreamCreateWithPriority(&c->bulk, hipStreamNonBlocking, least) != hipSuccess) {
    c->bulk = nullptr;
    return false;
  }
  if (hipEventCreateWithFlags(&c->bulk_go, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->bulk_done, hipEventDisableTiming) != hipSuccess)
    return false;
  return true;
}

void dev_factorize(gsx_context* c, double lambda) {
  const Symbolic& S = c->S;
  c->fact_valid = false;   // becomes true when the read-back shows no failed front (readback)
  c->fact_pending = true;
  c->fact_lambda = lambda;
  c->wf_all_replaced = true;  // every clique re-eliminated
  c->sc_dirty |= kXFact;
  timer_begin(c, PH_FACTORIZE);
  launch_begin_factorization(c->d_scalars.p, lambda, c->d_status.p, c->d_tree_cursor.p, c->stream);
  if (!c->big_descs.empty())
    launch_big_init(c->DP, c->DS, c->d_big.p, (int)c->big_descs.size(), c->big_max_n, c->big_max_nfv, c->d_H.p,
                    c->d_damp.p, c->d_scalars.p, c->d_arena.p, c->stream);
  // ---- the leaf-kernel cliques (all childless: one group of launches) ----
  // The SIDE work (Symbolic::side_*): the lean leaves whose parents sit above the first blocked level, and their
  // product-form gather, on the low-priority queue — beside the latency-bound blocked chain of the lower levels
  // instead of in front of it.  (With per-launch timers everything stays on the one timed queue.)
  const int sg = S.n_ulevels;   // the side gather group
  const bool have_side = S.side_level0 >= 0;
  const bool use_bulk = have_side && c->profiling <= 0 && bulk_ready(c);
  if (S.n_levels > 0) {
    const std::vector<SmallLaunch>& LL = c->leaf_launch[0];
    const size_t n_main_leaf = c->leaf_side_group0 != (size_t)-1 ? c->leaf_side_group0 : LL.size();
    if (use_bulk) {
      // (Forked right behind the zeroing / H terms of the blocked fronts.  Measured on BAL-1723, ms per LM iteration: no
      //  second queue 1.924; forked here 1.880; forked behind the main queue's own leaves and gather 1.904 — the side work then
      //  runs beside the first blocked level only, whose 11-workgroup big_rows launch goes from 63 to 153 us under it.  The
      //  low stream priority does not keep the side kernels off the CUs: what is gained is the blocked chain starting
      //  earlier, what is lost is every kernel running beside them running slower.)
      hipEventRecord(c->bulk_go, c->stream);
      hipStreamWaitEvent(c->bulk, c->bulk_go, 0);
      for (size_t g = n_main_leaf; g < LL.size(); ++g) {
        const SmallLaunch& sl = LL[g];
        launch_front_leaf(c->DP, c->DS, c->d_leaf_recs.p + (sl.begin - c->leaf_base), sl.count, sl.max_panel, sl.threads,
                          c->d_H.p, c->d_damp.p, c->d_scalars.p, c->d_arena.p, c->d_status.p, c->bulk);
      }
      launch_big_gather(c->GA, S.gseg_lvl_ptr[sg], S.gseg_lvl_ptr[sg + 1] - S.gseg_lvl_ptr[sg], S.gm_lvl_ptr[sg],
                        S.gm_lvl_ptr[sg + 1] - S.gm_lvl_ptr[sg], c->d_arena.p, c->bulk);
      hipEventRecord(c->bulk_done, c->bulk);
      c->bulk_pending = true;
    }
    for (size_t g = 0; g < (use_bulk ? n_main_leaf : LL.size()); ++g) {
      const SmallLaunch& sl = LL[g];
      if (c->profiling > 0) timer_begin(c, PH_FACTOR_LEAF);
      launch_front_leaf(c->DP, c->DS, c->d_leaf_recs.p + (sl.begin - c->leaf_base), sl.count, sl.max_panel, sl.threads, c->d_H.p,
                        c->d_damp.p, c->d_scalars.p, c->d_arena.p, c->d_status.p, c->stream);
      if (c->profiling > 0) timer_end(c, PH_FACTOR_LEAF);
    }
  }
  // ---- the tree fronts of every level: one launch per tier (their leaf-kernel children were the launches above) ----
  for (size_t t = 0; t < c->tree_tiers.size(); ++t) {
    const gsx_context::TreeTier& tt = c->tree_tiers[t];
    if (!tt.nstart) continue;
    if (c->profiling > 0) timer_begin(c, PH_FACTOR_SMALL);
    const TreeArgs ta{c->d_tree_start.p + tt.start0, tt.nstart, c->d_tree_cursor.p + t, c->d_tree_pending.p,
                      c->d_tree_up.p, c->d_tree_npend.p};
    if (tt.med_lds)
      launch_front_tree_med(c->DP, c->DS, ta, tt.med_lds, c->d_H.p, c->d_damp.p, c->d_scalars.p, c->d_arena.p,
                            c->d_status.p, c->stream);
    else
      launch_front_tree(c->DP, c->DS, ta, tt.max_n, tt.threads, c->d_H.p, c->d_damp.p, c->d_scalars.p, c->d_arena.p,
                        c->d_status.p, c->stream);
    debug_sync(c, tt.med_lds ? "front_tree_med" : "front_tree", tt.nstart, (int)tt.med_lds);
    if (c->profiling > 0) timer_end(c, PH_FACTOR_SMALL);
  }
  // ---- the upper levels (Symbolic::ulevel): everything below them is complete ----
  for (int l = 0; l < S.n_ulevels; ++l) {
    for (const SmallLaunch& sl : c->rest_launch[l]) {
      if (c->profiling > 0) timer_begin(c, PH_FACTOR_SMALL);
      launch_lds_group(c->DP, c->DS, c->d_rest_ids.p + sl.begin, sl, c->d_H.p, c->d_damp.p, c->d_scalars.p, c->d_arena.p,
                       c->d_status.p, c->stream);
      debug_sync(c, sl.medium ? "front_medium (level)" : "front_small (level)", sl.count, sl.medium ? sl.max_panel : sl.max_n);
      if (c->profiling > 0) timer_end(c, PH_FACTOR_SMALL);
    }
    // the fronts above the first blocked level take the side leaves' contributions: those must have landed (and they
    // come first in every destination block's sum, as in the one-queue order)
    if (have_side && l == S.side_level0 + 1 && c->bulk_pending) {
      hipStreamWaitEvent(c->stream, c->bulk_done, 0);
      c->bulk_pending = false;
    }
    const BigLevel& B = c->big_level[l];
    const bool prof = c->profiling > 0;
    if (prof && B.count) timer_begin(c, PH_FACTOR_BIG);
    // deterministic extend-add into this level's blocked fronts: the children are complete.  (Group 0 also holds the
    // product-form contributions of the lean leaves — of all levels, or of the levels up to the first blocked one when
    // the others are side work.)
    if (S.gseg_lvl_ptr[l + 1] > S.gseg_lvl_ptr[l]) {
      if (prof) timer_begin(c, PH_K_GATHER);
      launch_big_gather(c->GA, S.gseg_lvl_ptr[l], S.gseg_lvl_ptr[l + 1] - S.gseg_lvl_ptr[l], S.gm_lvl_ptr[l],
                        S.gm_lvl_ptr[l + 1] - S.gm_lvl_ptr[l], c->d_arena.p, c->stream);
      if (prof) timer_end(c, PH_K_GATHER);
    }
    if (l == 0 && have_side && !use_bulk && S.gseg_lvl_ptr[sg + 1] > S.gseg_lvl_ptr[sg]) {   // (no second queue: in line)
      if (prof) timer_begin(c, PH_K_GATHER);
      launch_big_gather(c->GA, S.gseg_lvl_ptr[sg], S.gseg_lvl_ptr[sg + 1] - S.gseg_lvl_ptr[sg], S.gm_lvl_ptr[sg],
                        S.gm_lvl_ptr[sg + 1] - S.gm_lvl_ptr[sg], c->d_arena.p, c->stream);
      if (prof) timer_end(c, PH_K_GATHER);
    }
    if (B.count) {
      // sharded: every rank's share of the cap (H terms, damping, its subtrees' Schur complements) is in; their sum
      // is the assembled cap, which all ranks now factor alike
      if (l == S.cap_ulevel0) shard_allreduce(c, c->d_arena.p + S.cap_begin, S.cap_end - S.cap_begin);
      // hard constraints: the assembled fronts that hold constraint rows become unconstrained fronts with the same
      // conditionals and Schur complement (constraint.hip)
      if (c->con_level[l].count)
        launch_constraint_fronts(c->DS, c->CT, c->con_level[l].first, c->con_level[l].count, c->con_level[l].max_n,
                                 c->d_jac.p, c->d_arena.p, c->d_status.p, c->stream);
      dev_big_factor(c, c->d_big.p + B.begin, B.count, B.plan, c->stream, prof);
      if (prof) timer_end(c, PH_FACTOR_BIG);
    }
  }
  // choleskyPartial's conditioning test, per clique of the reference tree (cholesky.cpp:145-158)
  launch_cond_check((int)S.cond_last.size(), c->d_cond_last.p, c->d_cond_prev.p, c->d_cond_front.p, c->d_arena.p,
                    c->d_status.p, c->stream);
  timer_end(c, PH_FACTORIZE);
}

// The level-by-level back-substitution over ALL fronts.  wf: ISAM2's partial back-substitution — the kernels skip the
// cliques no change reaches (DS.wf_*), and a bookkeeping pass follows every level (nullptr: all cliques).
void dev_backsolve_levels(gsx_context* c, const WildfireArgs* wf) {
  const Symbolic& S = c->S;
  timer_begin(c, PH_BACKSOLVE);
  c->wf_delta_valid = false;  // (set again by the callers that leave a complete undamped solution behind)
  for (int l = S.n_levels - 1; l >= 0; --l) {
    const int se = S.lvl_small_end[l];
    const struct {
      int count;
    } B{S.lvl_ptr[l + 1] - se};
    const int le = S.lvl_leaf_end[l], n_rest = S.lvl_ptr[l + 1] - le;
    // the blocked fronts of the level: their own kernel when L11's tiles fit in LDS (bigfront.hip)
    int big_maxn = 0, big_maxF = 0, big_maxS = 0;
    for (int k = se; k < S.lvl_ptr[l + 1]; ++k) {
      big_maxn = std::max(big_maxn, S.N[S.sched[k]]);
      big_maxF = std::max(big_maxF, S.F[S.sched[k]]);
      big_maxS = std::max(big_maxS, S.N[S.sched[k]] - S.F[S.sched[k]]);
    }
    const bool big_own = B.count > 0 && backsolve_big_lds(big_maxn, big_maxF, big_maxS) > 0;
    if (big_own) {
      if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
      launch_backsolve_big(c->DS, c->d_sched.p + se, S.lvl_ptr[l + 1] - se, big_maxn, big_maxF, c->d_arena.p, c->d_delta.p,
                           c->d_status.p, c->stream);
      if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
    }
    // the LDS-class fronts: their own kernels when every one of them has at most 64 frontal columns
    const int* small_ids = c->d_sched.p + le;
    const int n_small = se - le;
    int small_maxn = 0, small_maxF = 0;
    for (int k = le; k < se; ++k) {
      small_maxn = std::max(small_maxn, S.N[S.sched[k]]);
      small_maxF = std::max(small_maxF, S.F[S.sched[k]]);
    }
    const bool small_own = n_small > 0 && backsolve_small_fits(small_maxn, small_maxF);
    if (small_own) {
      if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
      launch_backsolve_small(c->DS, small_ids, n_small, small_maxF, c->d_arena.p, c->d_delta.p, c->d_status.p,
                             c->stream);
      if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
    }
    const bool big_left = B.count > 0 && !big_own, small_left = n_small > 0 && !small_own;
    if (big_left && small_left && n_rest <= 512 && n_rest > B.count) {
      // few fronts: small and big ones of the level in ONE launch of the generic kernel
      int maxn = 0;
      for (int k = le; k < S.lvl_ptr[l + 1]; ++k) maxn = std::max(maxn, S.N[S.sched[k]]);
      if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
      launch_backsolve(c->DS, c->d_sched.p + le, n_rest, 1024, maxn, c->d_arena.p, c->d_delta.p, c->d_status.p, c->stream);
      if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
    } else {
      if (big_left) {
        int maxn = 0;
        for (int k = se; k < S.lvl_ptr[l + 1]; ++k) maxn = std::max(maxn, S.N[S.sched[k]]);
        if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
        launch_backsolve(c->DS, c->d_sched.p + se, S.lvl_ptr[l + 1] - se, 1024, maxn, c->d_arena.p, c->d_delta.p,
                         c->d_status.p, c->stream);
        if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
      }
      if (small_left) {
        // the size groups of the factorization (split by LDS footprint) mean nothing here — the generic kernel keeps
        // only n - 1 doubles per clique in LDS: all the level's LDS-class cliques go in ONE launch
        if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
        launch_backsolve(c->DS, small_ids, n_small, small_maxn <= 48 ? 64 : 256, small_maxn, c->d_arena.p,
                         c->d_delta.p, c->d_status.p, c->stream);
        if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
      }
    }
    // all leaf-kernel cliques of the level in one launch (a wave each)
    if (S.lvl_leaf_end[l] > S.lvl_ptr[l])
      launch_backsolve_leaf(c->DS, c->d_leaf_recs.p + (S.lvl_ptr[l] - c->leaf_base), S.lvl_leaf_end[l] - S.lvl_ptr[l],
                            c->leaf_max_F[l], c->d_arena.p, c->d_delta.p, c->d_status.p, c->stream);
    if (wf) {  // the leaf cliques (F <= 16) a thread each, the others a wave each
      launch_wildfire_post(c->DS, c->d_sched.p + S.lvl_ptr[l], le - S.lvl_ptr[l], false, *wf, c->d_delta.p, c->stream);
      launch_wildfire_post(c->DS, c->d_sched.p + le, n_rest, true, *wf, c->d_delta.p, c->stream);
    }
  }
  timer_end(c, PH_BACKSOLVE);
}


// OptimizeClique top-down (gtsam/linear/linearAlgorithms-inst.h:49-117): the upper fronts level by level (Symbolic::ulevel),
// then the tree fronts of all levels in one launch, then the leaf-kernel cliques.
void dev_backsolve(gsx_context* c, const WildfireArgs* wf = nullptr) {
  static const bool levels_only = std::getenv("GSX_BS_TREE_OFF") != nullptr;
  if (wf != nullptr || levels_only) {
    dev_backsolve_levels(c, wf);
    return;
  }
  const Symbolic& S = c->S;
  timer_begin(c, PH_BACKSOLVE);
  c->wf_delta_valid = false;  // (set again by the callers that leave a complete undamped solution behind)
  for (int l = S.n_ulevels - 1; l >= 0; --l) {
    const int b0 = S.ulvl_ptr[l], se = S.ulvl_small_end[l], e = S.ulvl_ptr[l + 1];
    // the blocked fronts of the level: their own kernel when L11's tiles fit in LDS (bigfront.hip)
    if (e > se) {
      int big_maxn = 0, big_maxF = 0, big_maxS = 0;
      for (int k = se; k < e; ++k) {
        big_maxn = std::max(big_maxn, S.N[S.usched[k]]);
        big_maxF = std::max(big_maxF, S.F[S.usched[k]]);
        big_maxS = std::max(big_maxS, S.N[S.usched[k]] - S.F[S.usched[k]]);
      }
      if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
      if (backsolve_big_lds(big_maxn, big_maxF, big_maxS) > 0)
        launch_backsolve_big(c->DS, c->d_usched.p + se, e - se, big_maxn, big_maxF, c->d_arena.p, c->d_delta.p, c->d_status.p,
                             c->stream);
      else
        launch_backsolve(c->DS, c->d_usched.p + se, e - se, 1024, big_maxn, c->d_arena.p, c->d_delta.p, c->d_status.p,
                         c->stream);
      if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
    }
    // the LDS-class fronts of the level that are not tree fronts (consecutive in d_usched)
    if (se > b0) {
      int maxn = 0, maxF = 0;
      for (int k = b0; k < se; ++k) {
        maxn = std::max(maxn, S.N[S.usched[k]]);
        maxF = std::max(maxF, S.F[S.usched[k]]);
      }
      if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
      if (backsolve_small_fits(maxn, maxF))
        launch_backsolve_small(c->DS, c->d_usched.p + b0, se - b0, maxF, c->d_arena.p, c->d_delta.p, c->d_status.p, c->stream);
      else
        launch_backsolve(c->DS, c->d_usched.p + b0, se - b0, maxn <= 48 ? 64 : 256, maxn, c->d_arena.p, c->d_delta.p,
                         c->d_status.p, c->stream);
      if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
    }
  }
  if (c->bst_total > 0) {
    // every upper front is solved: the tree fronts of all levels, one launch
    hipMemsetAsync(c->d_bst_counters.p, 0, 2 * sizeof(int), c->stream);
    if (++c->bst_epoch == 0) c->bst_epoch = 1;
    if (c->profiling > 0) timer_begin(c, PH_K_BACKSOLVE);
    launch_backsolve_tree(c->DS,
                          BacksolveTreeArgs{c->d_bst_roots.p, c->bst_roots, c->bst_total, c->d_bst_child_ptr.p,
                                            c->d_bst_children.p, c->d_bst_ready.p, c->d_bst_counters.p,
                                            c->d_bst_counters.p + 1, c->bst_epoch},
                          c->d_arena.p, c->d_delta.p, c->d_status.p, c->stream);
    if (c->profiling > 0) timer_end(c, PH_K_BACKSOLVE);
    debug_sync(c, "backsolve_tree", c->bst_roots, c->bst_total);
  }
  // all leaf-kernel cliques (childless: level 0) in one launch, a wave each
  if (S.n_levels > 0 && S.lvl_leaf_end[0] > S.lvl_ptr[0])
    launch_backsolve_leaf(c->DS, c->d_leaf_recs.p + (S.lvl_ptr[0] - c->leaf_base), S.lvl_leaf_end[0] - S.lvl_ptr[0],
                          c->leaf_max_F[0], c->d_arena.p, c->d_delta.p, c->d_status.p, c->stream);
  timer_end(c, PH_BACKSOLVE);
}

// solved_step: d_delta is the solution of the damped system just factored (the LM trial) — the two linearized errors then
// come from the right-hand sides and the step (kernels.hip: lin0_kernel / model_error_kernel), no pass over [A b];
// otherwise (an arbitrary point: Dogleg, gsx_linear_error) the direct evaluation
void dev_linear_error(gsx_context* c, bool solved_step = false) {
  // (hard constraints: the step solves the KKT system, not (H + lambda D) delta = g — the identity below does not hold)
  static const bool direct_env = std::getenv("GSX_LINERR_DIRECT") != nullptr;
  const bool direct_only = direct_env || c->constrained();
  timer_begin(c, PH_LINERR);
  if (solved_step && !direct_only && c->h_ready) {
    if (!c->lin0_ready) {
      launch_lin0(c->DP, c->d_jac.p, c->d_partials.p, gsx_context::kPartials, c->d_scalars.p, c->stream);
      c->lin0_ready = true;
    }
    launch_model_error(c->DP, c->DS, c->d_hvars.p, (int)c->hv_list.size(), c->d_H.p, c->d_delta.p, c->d_damp.p,
                       c->d_partials.p, gsx_context::kPartials, c->d_scalars.p, c->stream);
  } else {
    launch_linear_error(c->DP, c->d_jac.p, c->d_delta.p, c->d_partials.p, gsx_context::kPartials, c->d_scalars.p,
                        c->lin_err_slice, c->stream);
  }
  c->sc_dirty |= (1u << SC_LIN0) | (1u << SC_LIND);
  timer_end(c, PH_LINERR);
}
void dev_retract(gsx_context* c, const double* d_delta) {
  timer_begin(c, PH_RETRACT);
  launch_retract(c->DP, c->d_values.p, d_delta, c->d_trial.p, c->stream);
  timer_end(c, PH_RETRACT);
}
void dev_error(gsx_context* c, const double* d_vals, int slot) {
  timer_begin(c, PH_ERROR);
  const int* lists[6] = {c->d_type_list[0].p, c->d_type_list[1].p, c->d_type_list[2].p, c->d_type_list[3].p,
                         c->d_type_list[4].p, c->d_type_list[5].p};
  launch_error(c->DP, lists, c->type_count, d_vals, c->d_partials.p, gsx_context::kPartials, c->d_scalars.p, slot, c->stream);
  c->sc_dirty |= 1u << slot;
  timer_end(c, PH_ERROR);
}
gsx_status readback(gsx_context* c) {
  if (c->sharded() && c->sc_dirty) {
    // partial sums and status counters of this rank's share -> the whole problem's, identical on every rank
    launch_shard_pack(c->d_scalars.p, c->d_status.p, c->sc_dirty, c->d_xscal.p, c->stream);
    shard_allreduce(c, c->d_xscal.p, kXScalars);
    launch_shard_unpack(c->d_xscal.p, c->sc_dirty, c->d_scalars.p, c->d_status.p, c->stream);
  }
  c->sc_dirty = 0;
  if (c->shard_failed) {
    c->shard_failed = false;
    c->err = "the all-reduce callback of a sharded handle failed";
    return GSX_E_NO_DEVICE;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->d_scalars.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_status.p, sizeof(DevStatus), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->n_cheirality = c->h_status->n_cheirality;
  // the factorization in the arena is only trusted (solve at the same lambda, marginals, partial re-elimination) once its
  // status has been seen clean
  const bool bad = c->h_status->n_fail > 0 || c->h_status->n_nonfinite > 0;
  if (c->fact_pending) c->fact_valid = !bad;
  else if (bad) c->fact_valid = false;
  c->fact_pending = false;
  return GSX_OK;
}

bool factorization_ok(const gsx_context* c) { return c->h_status->n_fail == 0 && c->h_status->n_nonfinite == 0; }

uint64_t failing_key(gsx_context* c) {
  const DevStatus& s = *c->h_status;
  if (s.n_fail > 0 && s.first_front >= 0 && s.first_front < c->S.n_fronts)
    return c->P.keys[c->S.fvars[c->S.fvar_ptr[s.first_front]]];
  return c->P.n_vars ? c->P.keys[c->S.order.empty() ? 0 : c->S.order[0]] : 0;
}

gsx_status need_device(gsx_context* c) {
  if (!c->has_device) {
    c->err = "no usable gfx950 device (the hot path has no CPU fallback)";
    return GSX_E_NO_DEVICE;
  }
  return GSX_OK;
}

// ---- LM policy (host scalars only) — LevenbergMarquardtOptimizer.cpp:121-308 ---------------------------
struct Trace {
  gsx_lm_result* r;
  void push(double err, double lambda, int acc) {
    if (!r) return;
    if (r->trace_len < r->trace_cap) {
      if (r->trace_error) r->trace_error[r->trace_len] = err;
      if (r->trace_lambda) r->trace_lambda[r->trace_len] = lambda;
      if (r->trace_accepted) r->trace_accepted[r->trace_len] = acc;
    }
    r->trace_len++;
  }
};

// One damped trial on the device + the controller's verdict on it (csrc/lm_policy.cpp: gsx_lm_decide).
gsx_status lm_try_lambda(gsx_context* c, const gsx_lm_params& p, Trace& tr, gsx_lm_result* res, bool* done) {
  dev_damping(c, p.diagonal_damping, p.min_diagonal, p.max_diagonal);
  dev_factorize(c, c->lm_lambda);
  dev_backsolve(c);
  dev_linear_error(c, true);
  dev_retract(c, c->d_delta.p);
  dev_error(c, c->d_trial.p, SC_TRIAL_ERR);
  gsx_status st = readback(c);
  if (st != GSX_OK) return st;
  const bool solved = factorization_ok(c);
  if (!solved && res) res->n_solve_failures++;
  gsx_lm_state ctl{c->lm_lambda, c->lm_factor, c->lm_error, c->lm_iterations, c->lm_inner};
  gsx_lm_decision d;
  gsx_lm_decide(&p, &ctl, solved, c->h_scalars[SC_LIN0], c->h_scalars[SC_LIND], c->h_scalars[SC_TRIAL_ERR], &d);
  if (p.verbosity >= 1)
    std::printf("%4d %12.6g %12.2e %10.2e %6d\n", c->lm_iterations, d.trial_cost, d.cost_change, d.lambda_tried, d.solved);
  if (d.verdict == GSX_LM_TAKE) {
    std::swap(c->d_values.p, c->d_trial.p);  // the trial values become the current ones
    c->values_synced = false;
    c->linearized = false;
    c->h_ready = false;
    c->solved = false;
    c->fact_valid = false;
  }
  c->lm_lambda = ctl.lambda;
  c->lm_factor = ctl.factor;
  c->lm_error = ctl.cost;
  c->lm_iterations = ctl.outer_iterations;
  c->lm_inner = ctl.inner_iterations;
  tr.push(d.trial_cost, d.lambda_tried, d.verdict == GSX_LM_TAKE ? 1 : (solved ? 0 : -1));
  *done = d.verdict != GSX_LM_RETRY;
  return GSX_OK;
}

gsx_status lm_iterate(gsx_context* c, const gsx_lm_params& p, Trace& tr, gsx_lm_result* res) {
  dev_linearize(c);
  dev_assemble_h(c);
  bool done = false;
  while (!done) {
    gsx_status st = lm_try_lambda(c, p, tr, res, &done);
    if (st != GSX_OK) return st;
  }
  return GSX_OK;
}

bool check_convergence(double relTol, double absTol, double errTol, double currentError, double newError) {
  if (newError <= errTol) return true;
  const double absoluteDecrease = currentError - newError;
  const double relativeDecrease = absoluteDecrease / currentError;
  return (relTol && (relativeDecrease <= relTol)) || (absoluteDecrease <= absTol);
}

gsx_status compute_error_sync(gsx_context* c, double* out) {
  dev_error(c, c->d_values.p, SC_ERR);
  gsx_status st = readback(c);
  if (st != GSX_OK) return st;
  *out = c->h_scalars[SC_ERR];
  return GSX_OK;
}

gsx_status ensure_ready(gsx_context* c, bool need_values, bool need_symbolic) {
  gsx_status st = need_device(c);
  if (st != GSX_OK) return st;
  if (need_values && !c->values_set && c->P.state_size > 0) {
    c->err = "values not set";
    return GSX_E_STATE;
  }
  if (need_symbolic && !c->has_symbolic) {
    c->err = "ordering not set";
    return GSX_E_STATE;
  }
  return GSX_OK;
}

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

const char* gsx_version(void) { return "gsx 0.1.0 (gfx950)"; }

int32_t gsx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

gsx_status gsx_create(const gsx_problem_desc* desc, int32_t device, gsx_handle* out) {
  if (!out) return GSX_E_INVALID;
  *out = nullptr;
  gsx_context* c = new gsx_context();
  gsx_status st = lower_problem(desc, c->P, c->err);
  if (st != GSX_OK) {
    std::fprintf(stderr, "gsx_create: %s\n", c->err.c_str());
    delete c;
    return st;
  }
  c->device = device;
  int n = 0;
  if (hipGetDeviceCount(&n) == hipSuccess && n > 0 && device >= 0 && device < n && hipSetDevice(device) == hipSuccess &&
      hipStreamCreate(&c->stream) == hipSuccess) {
    c->has_device = true;
    st = upload_problem(c);
    if (st != GSX_OK) {
      std::fprintf(stderr, "gsx_create: %s\n", c->err.c_str());
      delete c;
      return st;
    }
  }
  *out = c;
  return GSX_OK;
}

gsx_status gsx_destroy(gsx_handle h) {
  if (!h) return GSX_OK;
  if (h->has_device) {
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (auto& t : h->timers) {
      for (auto& ev : t.pending) t.pool.push_back(ev);
      for (auto& ev : t.pool) {
        hipEventDestroy(ev.first);
        hipEventDestroy(ev.second);
      }
    }
    if (h->bulk) {
      hipStreamSynchronize(h->bulk);
      hipStreamDestroy(h->bulk);
      if (h->bulk_go) hipEventDestroy(h->bulk_go);
      if (h->bulk_done) hipEventDestroy(h->bulk_done);
    }
    if (h->h_scalars) hipHostFree(h->h_scalars);
    if (h->h_status) hipHostFree(h->h_status);
  }
  hipStream_t st = h->stream;
  const bool dev = h->has_device;
  delete h;
  if (dev && st) hipStreamDestroy(st);
  return GSX_OK;
}

const char* gsx_last_error(gsx_handle h) { return h ? h->err.c_str() : ""; }

gsx_status gsx_set_ordering(gsx_handle h, const uint64_t* keys, int32_t n) {
  if (!h || (!keys && n > 0)) return GSX_E_INVALID;
  if (n != h->P.n_vars) {
    h->err = "ordering size differs from the number of variables";
    return GSX_E_BAD_ORDERING;
  }
  std::map<uint64_t, int> idx;
  for (int v = 0; v < h->P.n_vars; ++v) idx[h->P.keys[v]] = v;
  std::vector<int> ord(n);
  for (int i = 0; i < n; ++i) {
    auto it = idx.find(keys[i]);
    if (it == idx.end()) {
      h->err = "ordering contains a key that is not a variable of the graph";
      return GSX_E_BAD_ORDERING;
    }
    ord[i] = it->second;
  }
  gsx_status st = symbolic_analysis(h->P, ord, h->relax, h->relax_max_f, h->shard_rank, h->shard_world, h->S, h->err);
  if (st != GSX_OK) return st;
  h->order = ord;
  h->has_symbolic = true;
  if (h->has_device) {
    hipSetDevice(h->device);
    st = upload_symbolic(h);
    if (st != GSX_OK) return st;
  }
  return GSX_OK;
}

gsx_status gsx_compute_ordering(gsx_handle h, int32_t kind, uint64_t* keys_out) {
  if (!h || !keys_out || kind < 0 || kind > 4) return GSX_E_INVALID;
  std::vector<int> ord;
  compute_ordering(h->P, kind, ord);
  for (int i = 0; i < h->P.n_vars; ++i) keys_out[i] = h->P.keys[ord[i]];
  return GSX_OK;
}

gsx_status gsx_set_amalgamation(gsx_handle h, double relax, int32_t max_frontal_dim) {
  if (!h || relax != relax || (relax >= 0.0 && max_frontal_dim < 1)) return GSX_E_INVALID;
  h->relax = relax;
  h->relax_max_f = max_frontal_dim;
  return GSX_OK;
}

gsx_status gsx_set_shard(gsx_handle h, int32_t rank, int32_t world, gsx_allreduce_fn allreduce, void* user) {
  if (!h || world < 1 || rank < 0 || rank >= world || (world > 1 && !allreduce)) return GSX_E_INVALID;
  if (h->has_symbolic) {
    h->err = "gsx_set_shard must precede gsx_set_ordering";
    return GSX_E_STATE;
  }
  h->shard_rank = rank;
  h->shard_world = world;
  h->shard_cb = allreduce;
  h->shard_user = user;
  return GSX_OK;
}

gsx_status gsx_get_shard(gsx_handle h, gsx_shard_info* info, int32_t* front_owner, int32_t* factor_owned) {
  if (!h || !h->has_symbolic) return GSX_E_STATE;
  const Symbolic& S = h->S;
  if (info) {
    info->rank = S.shard_rank;
    info->world = S.shard_world;
    info->n_cap_fronts = S.n_cap;
    info->n_own_fronts = 0;
    for (int f = 0; f < S.n_fronts; ++f) info->n_own_fronts += S.owner[f] == S.shard_rank;
    info->cap_level0 = S.cap_level0;
    info->n_own_factors = 0;
    for (char o : S.f_owned) info->n_own_factors += o != 0;
    info->cap_doubles = S.cap_end - S.cap_begin;
    info->cap_flops = S.cap_cost;
    info->own_flops = S.own_cost;
    info->total_flops = S.total_cost;
  }
  if (front_owner)
    for (int f = 0; f < S.n_fronts; ++f) front_owner[f] = S.owner[f];
  if (factor_owned)
    for (int f = 0; f < h->P.n_factors; ++f) factor_owned[f] = S.f_owned[f];
  return GSX_OK;
}

gsx_status gsx_get_front_classes(gsx_handle h, int32_t* classes) {
  if (!h || !classes) return GSX_E_INVALID;
  if (!h->has_symbolic) return GSX_E_STATE;
  const Symbolic& S = h->S;
  for (int f = 0; f < S.n_fronts; ++f)
    classes[f] = (S.med[f] ? 3 : S.cls[f]) | (S.tree_tier[f] >= 0 ? 4 : 0) | (S.lean[f] ? 8 : 0);
  return GSX_OK;
}

gsx_status gsx_scratch_buffer(gsx_handle h, double** device_ptr, int64_t* count) {
  if (!h || !device_ptr || !count) return GSX_E_INVALID;
  gsx_status st = need_device(h);
  if (st != GSX_OK) return st;
  *device_ptr = h->d_partials.p;   // (rewritten from scratch by every reduction that uses it)
  *count = gsx_context::kPartials;
  return GSX_OK;
}

gsx_status gsx_get_ordering(gsx_handle h, uint64_t* keys_out) {
  if (!h || !h->has_symbolic) return GSX_E_STATE;
  for (int i = 0; i < h->P.n_vars; ++i) keys_out[i] = h->P.keys[h->order[i]];
  return GSX_OK;
}

gsx_status gsx_get_tree(gsx_handle h, int32_t* n_fronts, int64_t* n_sep_total, int32_t* parent, int32_t* frontal_ptr,
                        int32_t* frontal_vars, int32_t* sep_ptr, int32_t* sep_vars) {
  if (!h || !h->has_symbolic) return GSX_E_STATE;
  const Symbolic& S = h->S;
  if (n_fronts) *n_fronts = S.n_fronts;
  if (n_sep_total) *n_sep_total = (int64_t)S.fvars.size() - h->P.n_vars;
  if (!parent) return GSX_OK;
  int fp = 0, sp = 0;
  for (int f = 0; f < S.n_fronts; ++f) {
    parent[f] = S.parent[f];
    frontal_ptr[f] = fp;
    sep_ptr[f] = sp;
    for (int k = S.fvar_ptr[f]; k < S.fvar_ptr[f + 1]; ++k) {
      if (k - S.fvar_ptr[f] < S.nfrontal_vars[f]) frontal_vars[fp++] = S.fvars[k];
      else sep_vars[sp++] = S.fvars[k];
    }
  }
  frontal_ptr[S.n_fronts] = fp;
  sep_ptr[S.n_fronts] = sp;
  return GSX_OK;
}

int64_t gsx_state_size(gsx_handle h) { return h ? h->P.state_size : 0; }
int64_t gsx_tangent_size(gsx_handle h) { return h ? h->P.tan_size : 0; }
int64_t gsx_jacobian_size(gsx_handle h) { return h ? h->P.jac_size : 0; }

gsx_status gsx_set_values(gsx_handle h, const double* packed, int64_t n) {
  if (!h || n != h->P.state_size || (!packed && n > 0)) return GSX_E_INVALID;
  gsx_status st = need_device(h);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  if (n > 0) {
    HIPCHK(h, hipMemcpyAsync(h->d_values.p, packed, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  h->values_set = true;
  h->values_synced = true;
  h->linearized = h->h_ready = h->solved = false;
  return GSX_OK;
}

gsx_status gsx_get_values(gsx_handle h, double* packed, int64_t n) {
  if (!h || n != h->P.state_size || (!packed && n > 0)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  if (h->sharded() && !h->values_synced && h->has_symbolic) {
    // every variable from its owner: the ranks only kept their own subtrees (and the cap) current
    launch_mask_copy(h->d_values.p, h->d_own_state.p, n, h->d_trial.p, h->stream);
    shard_allreduce(h, h->d_trial.p, n);
    std::swap(h->d_values.p, h->d_trial.p);
    h->values_synced = true;
    if (h->shard_failed) {
      h->shard_failed = false;
      h->err = "the all-reduce callback of a sharded handle failed";
      return GSX_E_NO_DEVICE;
    }
  }
  if (n > 0) {
    HIPCHK(h, hipMemcpyAsync(packed, h->d_values.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return GSX_OK;
}

gsx_status gsx_error(gsx_handle h, double* out) {
  if (!h || !out) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  return compute_error_sync(h, out);
}

gsx_status gsx_linearize(gsx_handle h) {
  if (!h) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  dev_linearize(h);
  return readback(h);
}

gsx_status gsx_get_jacobians(gsx_handle h, double* out, int64_t n) {
  if (!h || n != h->P.jac_size || (!out && n > 0)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  if (!h->linearized) {
    h->err = "linearize first";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  if (n > 0) {
    HIPCHK(h, hipMemcpyAsync(out, h->d_jac.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return GSX_OK;
}

gsx_status gsx_hessian_diagonal(gsx_handle h, double* out, int64_t n) {
  if (!h || n != h->P.tan_size || (!out && n > 0)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (!h->linearized) {
    h->err = "linearize first";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  if (!h->h_ready) dev_assemble_h(h);
  dev_hessian_diag(h);
  if (n > 0) {
    HIPCHK(h, hipMemcpyAsync(out, h->d_hdiag.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return GSX_OK;
}

gsx_status gsx_solve(gsx_handle h, double lambda, int32_t diagonal_damping, double min_diagonal, double max_diagonal,
                     double* delta_out, int64_t n, uint64_t* bad_key) {
  if (!h || (delta_out && n != h->P.tan_size)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (!h->linearized) {
    h->err = "linearize first";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  if (!h->h_ready) dev_assemble_h(h);
  // (the undamped factorization of this very linearization may already be resident: after gsx_relinearize_partial,
  // a marginal query or a previous lambda = 0 solve; then only the back-substitution runs)
  if (!(lambda == 0.0 && h->fact_valid && h->fact_lambda == 0.0)) {
    dev_damping(h, diagonal_damping, min_diagonal, max_diagonal);
    dev_factorize(h, lambda);
  } else {
    launch_begin_factorization(h->d_scalars.p, 0.0, h->d_status.p, nullptr, h->stream);  // status reset for the substitution
    h->sc_dirty |= kXFact;
  }
  dev_backsolve(h);
  st = readback(h);
  if (st != GSX_OK) return st;
  if (h->h_status->n_fail > 0 || h->h_status->n_nonfinite > 0) {
    if (bad_key) *bad_key = failing_key(h);
    h->solved = false;
    h->fact_valid = false;
    h->err = "indeterminate linear system";
    return GSX_E_INDETERMINATE;
  }
  h->solved = true;
  if (lambda == 0.0 && !h->sharded()) {  // a complete undamped solution: what a later wildfire pass starts from
    h->wf_delta_valid = true;
    h->wf_all_replaced = false;
    h->wf_clear_pending = true;
  }
  if (delta_out && n > 0) {
    const double* src = h->d_delta.p;
    if (h->sharded()) {
      launch_mask_copy(h->d_delta.p, h->d_own_tan.p, n, h->d_udelta.p, h->stream);
      shard_allreduce(h, h->d_udelta.p, n);
      src = h->d_udelta.p;
    }
    HIPCHK(h, hipMemcpyAsync(delta_out, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return GSX_OK;
}

// ISAM2's partial ("wildfire") back-substitution (include/gsx.h; gtsam/nonlinear/ISAM2-impl.cpp:48-77,
// ISAM2Clique.cpp:203-287) on the resident undamped factorization.
gsx_status gsx_backsubstitute_wildfire(gsx_handle h, double threshold, double* delta_out, int64_t n,
                                       int64_t* n_vars_solved, uint64_t* bad_key) {
  if (!h || (delta_out && n != h->P.tan_size)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (h->sharded()) {
    h->err = "the partial back-substitution is not available on a sharded handle";
    return GSX_E_STATE;
  }
  if (!h->linearized || !h->fact_valid || h->fact_lambda != 0.0) {
    h->err = "gsx_backsubstitute_wildfire needs the resident undamped factorization of the current linearization "
             "(gsx_solve with lambda = 0, or gsx_relinearize_partial after one, first)";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  const Symbolic& S = h->S;
  launch_begin_factorization(h->d_scalars.p, 0.0, h->d_status.p, nullptr, h->stream);  // status reset for the substitution
  h->sc_dirty |= kXFact;
  // (DeltaImpl::UpdateGaussNewtonDelta: threshold <= 0 = all cliques; so is a pass with every clique replaced)
  const bool full = !(threshold > 0.0) || !h->wf_delta_valid || h->wf_all_replaced;
  if (full) {
    dev_backsolve(h);
  } else {
    const size_t nf = (size_t)std::max(S.n_fronts, 1), nv = (size_t)std::max(h->P.n_vars, 1), nt = (size_t)h->P.tan_size;
    HIPCHK(h, h->wf_flags_ready(h->stream));
    if (h->d_wf_dirty.n < nf) HIPCHK(h, h->d_wf_dirty.alloc(nf));
    if (h->d_wf_changed.n < nv) HIPCHK(h, h->d_wf_changed.alloc(nv));
    if (h->d_wf_old.n < nt) HIPCHK(h, h->d_wf_old.alloc(std::max<size_t>(nt, 1)));
    HIPCHK(h, hipMemsetAsync(h->d_wf_changed.p, 0, nv, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_wf_old.p, h->d_delta.p, nt * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    const WildfireArgs W{h->d_wf_replaced.p, h->d_wf_dirty.p, h->d_wf_changed.p, h->d_wf_old.p, h->d_status.p, threshold};
    h->DS.wf_dirty = h->d_wf_dirty.p;
    h->DS.wf_replaced = h->d_wf_replaced.p;
    h->DS.wf_changed = h->d_wf_changed.p;
    dev_backsolve(h, &W);
    h->DS.wf_dirty = nullptr;
  }
  st = readback(h);
  if (st != GSX_OK) return st;
  if (h->h_status->n_fail > 0 || h->h_status->n_nonfinite > 0) {
    if (bad_key) *bad_key = failing_key(h);
    h->solved = false;
    h->fact_valid = false;
    h->err = "indeterminate linear system";
    return GSX_E_INDETERMINATE;
  }
  h->solved = true;
  h->wf_delta_valid = true;
  h->wf_all_replaced = false;
  h->wf_clear_pending = true;
  if (n_vars_solved) *n_vars_solved = full ? (int64_t)h->P.n_vars : (int64_t)h->h_status->n_backsub;
  if (delta_out && n > 0) {
    HIPCHK(h, hipMemcpyAsync(delta_out, h->d_delta.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return GSX_OK;
}

gsx_status gsx_linear_error(gsx_handle h, double* e0, double* ed) {
  if (!h) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  if (!h->linearized || (ed && !h->solved)) {
    h->err = "linearize (and solve) first";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  dev_linear_error(h);
  st = readback(h);
  if (st != GSX_OK) return st;
  if (e0) *e0 = h->h_scalars[SC_LIN0];
  if (ed) *ed = h->h_scalars[SC_LIND];
  return GSX_OK;
}

gsx_status gsx_retract(gsx_handle h, const double* delta, int64_t n, int32_t commit, double* trial_error) {
  if (!h) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  const double* dd = h->d_delta.p;
  if (delta) {
    if (n != h->P.tan_size) return GSX_E_INVALID;
    if (n > 0) HIPCHK(h, hipMemcpyAsync(h->d_udelta.p, delta, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    dd = h->d_udelta.p;
  } else if (!h->solved) {
    h->err = "no delta: solve first";
    return GSX_E_STATE;
  }
  dev_retract(h, dd);
  if (trial_error) dev_error(h, h->d_trial.p, SC_TRIAL_ERR);
  st = readback(h);
  if (st != GSX_OK) return st;
  if (trial_error) *trial_error = h->h_scalars[SC_TRIAL_ERR];
  if (commit) {
    std::swap(h->d_values.p, h->d_trial.p);
    h->values_synced = false;
    h->linearized = h->h_ready = h->solved = false;
  }
  return GSX_OK;
}

void gsx_lm_params_legacy(gsx_lm_params* p) {
  // LevenbergMarquardtParams::SetLegacyDefaults — LevenbergMarquardtParams.h:69-82
  *p = gsx_lm_params{100, 1e-5, 1e-5, 0.0, 1e-5, 10.0, 1e5, 0.0, 1e-3, 0, 1, 1e-6, 1e32, 0};
}
void gsx_lm_params_ceres(gsx_lm_params* p) {
  // LevenbergMarquardtParams::SetCeresDefaults — LevenbergMarquardtParams.h:85-98
  *p = gsx_lm_params{50, 1e-6, 0.0, 0.0, 1e-4, 2.0, 1e32, 1e-16, 1e-3, 1, 0, 1e-6, 1e32, 0};
}

gsx_status gsx_lm_reset(gsx_handle h, const gsx_lm_params* p) {
  if (!h || !p) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, false);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  st = compute_error_sync(h, &h->lm_error);
  if (st != GSX_OK) return st;
  h->lm_lambda = p->lambda_initial;
  h->lm_factor = p->lambda_factor;
  h->lm_iterations = 0;
  h->lm_inner = 0;
  return GSX_OK;
}

gsx_status gsx_lm_iterate(gsx_handle h, const gsx_lm_params* p, double* error, double* lambda) {
  if (!h || !p) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  Trace tr{nullptr};
  st = lm_iterate(h, *p, tr, nullptr);
  if (st != GSX_OK) return st;
  if (error) *error = h->lm_error;
  if (lambda) *lambda = h->lm_lambda;
  return GSX_OK;
}

gsx_status gsx_lm_trial(gsx_handle h, int32_t relinearize, double lambda, int32_t diagonal_damping, double min_diagonal,
                        double max_diagonal, double* lin0, double* lind, double* trial_error) {
  if (!h || !(lambda >= 0.0)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (!relinearize && !h->linearized) {
    h->err = "linearize first";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  if (relinearize) dev_linearize(h);
  if (!h->h_ready) dev_assemble_h(h);
  dev_damping(h, diagonal_damping, min_diagonal, max_diagonal);
  dev_factorize(h, lambda);
  dev_backsolve(h);
  dev_linear_error(h, true);
  dev_retract(h, h->d_delta.p);
  dev_error(h, h->d_trial.p, SC_TRIAL_ERR);
  st = readback(h);
  if (st != GSX_OK) return st;
  if (h->h_status->n_fail > 0 || h->h_status->n_nonfinite > 0) {
    h->solved = false;
    h->fact_valid = false;
    h->err = "indeterminate linear system";
    return GSX_E_INDETERMINATE;
  }
  h->solved = true;
  if (lin0) *lin0 = h->h_scalars[SC_LIN0];
  if (lind) *lind = h->h_scalars[SC_LIND];
  if (trial_error) *trial_error = h->h_scalars[SC_TRIAL_ERR];
  return GSX_OK;
}

// NonlinearOptimizer::defaultOptimize — gtsam/nonlinear/NonlinearOptimizer.cpp:62-117
gsx_status gsx_lm_optimize(gsx_handle h, const gsx_lm_params* p, gsx_lm_result* r) {
  if (!h || !p) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  st = gsx_lm_reset(h, p);
  if (st != GSX_OK) return st;
  Trace tr{r};
  if (r) {
    r->initial_error = h->lm_error;
    r->trace_len = 0;
    r->n_solve_failures = 0;
  }
  double currentError = h->lm_error;
  if (!(currentError <= p->error_tol) && !(h->lm_iterations >= p->max_iterations)) {
    double newError = currentError;
    do {
      currentError = newError;
      st = lm_iterate(h, *p, tr, r);
      if (st != GSX_OK) return st;
      newError = h->lm_error;
    } while (h->lm_iterations < p->max_iterations &&
             !check_convergence(p->relative_error_tol, p->absolute_error_tol, p->error_tol, currentError, newError) &&
             std::isfinite(currentError));
  }
  if (r) {
    r->final_error = h->lm_error;
    r->final_lambda = h->lm_lambda;
    r->iterations = h->lm_iterations;
    r->inner_iterations = h->lm_inner;
  }
  return GSX_OK;
}

// GaussNewtonOptimizer::iterate — gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66
gsx_status gsx_gn_optimize(gsx_handle h, int32_t max_iterations, double relTol, double absTol, double errTol,
                           gsx_lm_result* r) {
  if (!h) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  st = compute_error_sync(h, &h->lm_error);
  if (st != GSX_OK) return st;
  h->lm_iterations = 0;
  Trace tr{r};
  if (r) {
    r->initial_error = h->lm_error;
    r->trace_len = 0;
    r->n_solve_failures = 0;
  }
  double currentError = h->lm_error;
  if (!(currentError <= errTol) && max_iterations > 0) {
    double newError = currentError;
    do {
      currentError = newError;
      dev_linearize(h);
      dev_assemble_h(h);
      dev_damping(h, 0, 0, 0);
      dev_factorize(h, 0.0);
      dev_backsolve(h);
      dev_retract(h, h->d_delta.p);
      dev_error(h, h->d_trial.p, SC_TRIAL_ERR);
      st = readback(h);
      if (st != GSX_OK) return st;
      if (h->h_status->n_fail > 0 || h->h_status->n_nonfinite > 0) {
        h->err = "indeterminate linear system";
        return GSX_E_INDETERMINATE;
      }
      std::swap(h->d_values.p, h->d_trial.p);
      h->values_synced = false;
      h->linearized = h->h_ready = h->solved = false;
      h->lm_error = h->h_scalars[SC_TRIAL_ERR];
      h->lm_iterations++;
      newError = h->lm_error;
      tr.push(newError, 0.0, 1);
    } while (h->lm_iterations < max_iterations && !check_convergence(relTol, absTol, errTol, currentError, newError) &&
             std::isfinite(currentError));
  }
  if (r) {
    r->final_error = h->lm_error;
    r->final_lambda = 0;
    r->iterations = h->lm_iterations;
    r->inner_iterations = h->lm_iterations;
  }
  return GSX_OK;
}

// DoglegOptimizerImpl::ComputeDoglegPoint coefficients: dx_d = cu dx_u + cn dx_n  (DoglegOptimizerImpl.cpp:26-86)
static void dogleg_coefficients(double delta, double uu, double nn, double un, double* cu, double* cn) {
  const double deltaSq = delta * delta;
  if (deltaSq < uu) {
    *cu = std::sqrt(deltaSq / uu);
    *cn = 0.0;
  } else if (deltaSq < nn) {
    const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - deltaSq;
    const double sq = std::sqrt(b * b - 4 * a * c);
    const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
    const double eps = std::numeric_limits<double>::epsilon();
    const double tau = (-eps <= tau1 && tau1 <= 1.0 + eps) ? tau1 : tau2;
    *cu = 1. - tau;
    *cn = tau;
  } else {
    *cu = 0.0;
    *cn = 1.0;
  }
}

gsx_status gsx_dogleg_point(double delta, const double* dx_u, const double* dx_n, int64_t n, double* out) {
  if (!dx_u || !dx_n || !out || n < 0 || !(delta >= 0)) return GSX_E_INVALID;
  double uu = 0, nn = 0, un = 0;
  for (int64_t i = 0; i < n; ++i) {
    uu += dx_u[i] * dx_u[i];
    nn += dx_n[i] * dx_n[i];
    un += dx_u[i] * dx_n[i];
  }
  double cu, cn;
  dogleg_coefficients(delta, uu, nn, un, &cu, &cn);
  for (int64_t i = 0; i < n; ++i) out[i] = (cn == 0.0 ? cu * dx_u[i] : (cu == 0.0 ? dx_n[i] : cu * dx_u[i] + cn * dx_n[i]));
  return GSX_OK;
}

// DoglegOptimizer::iterate (gtsam/nonlinear/DoglegOptimizer.cpp:84-121) with DoglegOptimizerImpl::Iterate in
// ONE_STEP_PER_ITERATION mode (DoglegOptimizerImpl.h:137-252), inside NonlinearOptimizer::defaultOptimize.
// The reference evaluates the model M on the Bayes tree [R S d]; M(0) - M(dx) is the same number on the linearized
// graph it was eliminated from (the two differ by a constant), which is what the device has.
gsx_status gsx_dogleg_optimize(gsx_handle h, double delta_initial, int32_t max_iterations, double relTol, double absTol,
                               double errTol, gsx_lm_result* r) {
  if (!h || !(delta_initial >= 0)) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (h->sharded()) {
    h->err = "Dogleg is not available on a sharded handle";
    return GSX_E_STATE;
  }
  if (h->constrained()) {
    // (the steepest-descent leg of the dog leg knows nothing of the constraint rows: a blended step would violate them)
    h->err = "Dogleg is not available on a problem with hard constraints";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  const int64_t nt = h->P.tan_size;
  if (!h->d_dlu.p) {
    HIPCHK(h, h->d_dlu.alloc(std::max<int64_t>(nt, 1)));
    HIPCHK(h, h->d_dld.alloc(std::max<int64_t>(nt, 1)));
  }
  st = compute_error_sync(h, &h->lm_error);
  if (st != GSX_OK) return st;
  h->lm_iterations = 0;
  double delta = delta_initial;
  Trace tr{r};
  if (r) {
    r->initial_error = h->lm_error;
    r->trace_len = 0;
    r->n_solve_failures = 0;
  }
  double currentError = h->lm_error;
  if (!(currentError <= errTol) && max_iterations > 0) {
    double newError = currentError;
    do {
      currentError = newError;
      dev_linearize(h);
      dev_assemble_h(h);
      dev_damping(h, 0, 0, 0);
      dev_factorize(h, 0.0);
      dev_backsolve(h);  // d_delta = Newton point dx_n
      // steepest-descent point: grad = -A'b = -g, dx_u = -(grad'grad / |A grad|^2) grad = (g'g / |A g|^2) g
      launch_gradient(h->DP, h->DS, h->d_H.p, h->d_dlu.p, h->stream);
      launch_vec_dot(h->d_dlu.p, h->d_dlu.p, nt, h->d_partials.p, gsx_context::kPartials, h->d_scalars.p, SC_DOT0, h->stream);
      launch_ax_sqnorm(h->DP, h->d_jac.p, h->d_dlu.p, h->d_partials.p, gsx_context::kPartials, h->d_scalars.p, SC_DOT1,
                       h->stream);
      launch_vec_dot(h->d_delta.p, h->d_delta.p, nt, h->d_partials.p, gsx_context::kPartials, h->d_scalars.p, SC_DOT2,
                     h->stream);
      launch_vec_dot(h->d_dlu.p, h->d_delta.p, nt, h->d_partials.p, gsx_context::kPartials, h->d_scalars.p, SC_DOT3,
                     h->stream);
      st = readback(h);
      if (st != GSX_OK) return st;
      if (h->h_status->n_fail > 0 || h->h_status->n_nonfinite > 0) {
        h->err = "indeterminate linear system";
        return GSX_E_INDETERMINATE;
      }
      const double gg = h->h_scalars[SC_DOT0], ag2 = h->h_scalars[SC_DOT1];
      const double step = gg / ag2;
      const double uu = step * step * gg, nn = h->h_scalars[SC_DOT2], un = step * h->h_scalars[SC_DOT3];
      const double f_error = h->lm_error;
      double result_f = f_error, cu = 0, cn = 0;
      bool stay = true, zero_step = false;
      while (stay) {
        dogleg_coefficients(delta, uu, nn, un, &cu, &cn);
        launch_vec_axpby(h->d_dld.p, cu * step, h->d_dlu.p, cn, h->d_delta.p, nt, h->stream);  // dx_d (d_dlu holds g)
        dev_retract(h, h->d_dld.p);
        dev_error(h, h->d_trial.p, SC_TRIAL_ERR);
        launch_linear_error(h->DP, h->d_jac.p, h->d_dld.p, h->d_partials.p, gsx_context::kPartials, h->d_scalars.p,
                            h->lin_err_slice, h->stream);
        st = readback(h);
        if (st != GSX_OK) return st;
        result_f = h->h_scalars[SC_TRIAL_ERR];
        const double M_error = h->h_scalars[SC_LIN0], new_M = h->h_scalars[SC_LIND];
        const double rho = (std::abs(f_error - result_f) < 1e-15 || std::abs(M_error - new_M) < 1e-15)
                               ? 0.5
                               : (f_error - result_f) / (M_error - new_M);
        if (rho >= 0.75) {
          const double dnorm = std::sqrt(cu * cu * uu + 2 * cu * cn * un + cn * cn * nn);
          delta = std::max(delta, 3.0 * dnorm);
          stay = false;
        } else if (rho >= 0.25) {
          stay = false;
        } else if (rho >= 0.0) {
          if (delta > 1e-5) delta *= 0.5;
          stay = false;
        } else {  // f increased (NaN lands here too): shrink the region until it does not
          if (delta > 1e-5) {
            delta *= 0.5;
            stay = true;
          } else {
            zero_step = true;  // do not allow the error to increase
            result_f = f_error;
            stay = false;
          }
        }
      }
      if (!zero_step) std::swap(h->d_values.p, h->d_trial.p);
      h->linearized = h->h_ready = h->solved = false;
      h->lm_error = result_f;
      h->lm_iterations++;
      newError = h->lm_error;
      tr.push(newError, delta, 1);  // the trace's "lambda" column carries the trust-region radius
    } while (h->lm_iterations < max_iterations && !check_convergence(relTol, absTol, errTol, currentError, newError) &&
             std::isfinite(currentError));
  }
  if (r) {
    r->final_error = h->lm_error;
    r->final_lambda = delta;
    r->iterations = h->lm_iterations;
    r->inner_iterations = h->lm_iterations;
  }
  return GSX_OK;
}

// Marginals::marginalCovariance(key) — gtsam/nonlinear/Marginals.cpp:107-136 — from the undamped factorization of the
// current linearization (the block of H^-1 in the variable's tangent space), out: dA x dA column-major.
// ---- partial relinearization and re-elimination on a fixed graph ---------------------------------------------------
// The step at the heart of iSAM2's update (gtsam/nonlinear/ISAM2.cpp:419-484 relinearize the marked variables' factors,
// :725-783 re-eliminate the top of the tree that contains them), on a fixed structure: the Bayes tree, its
// factorization and every clean subtree's Schur complement stay resident in HBM; only what the moved variables touch
// is redone — their factors' Jacobians, the H panels of those factors' variables, and the cliques holding them plus all
// ancestors (a re-done clique is re-assembled from H and from ALL its children, whose stored contributions are intact).
namespace {
// The three stages of a partial update (gsx_relinearize_partial, gsx_update), each driven by explicit dirty lists.
// 1. Jacobians of the listed factors (graph order inside each family, as in the full lists)
gsx_status partial_linearize(gsx_handle h, std::vector<int>& dfac) {
  const HostProblem& P = h->P;
  hipStream_t sm = h->stream;
  gsx_context::PartialScratch& ps = h->ps;
  {
    std::sort(dfac.begin(), dfac.end());
    std::vector<int> lists[6];
    for (int f : dfac) {
      const int t = P.f_type[f];
      const int vt = P.types[P.f_vars[P.f_key_ptr[f]]];
      if (t == GSX_F_SFM) lists[0].push_back(f);
      else if (t == GSX_F_PROJECTION) lists[4].push_back(f);
      else if (t == GSX_F_BEARINGRANGE) lists[5].push_back(f);
      else if (t == GSX_F_BETWEEN && vt == GSX_VAR_POSE2) lists[1].push_back(f);
      else if (t == GSX_F_BETWEEN && vt == GSX_VAR_POSE3) lists[2].push_back(f);
      else if (t != GSX_F_LINEAR) lists[3].push_back(f);
    }
    const int* lp[6];
    int cnt[6];
    for (int k = 0; k < 6; ++k) {
      HIPCHK(h, ps.lists[k].stage(lists[k], sm));
      lp[k] = ps.lists[k].p;
      cnt[k] = (int)lists[k].size();
    }
    timer_begin(h, PH_LINEARIZE);
    launch_linearize(h->DP, lp, cnt, h->d_values.p, h->d_jac.p, h->d_status.p, sm);
    timer_end(h, PH_LINEARIZE);
  }
  h->lin0_ready = false;   // (some [A b] blocks changed)
  return GSX_OK;
}
// 2. the H panels of the listed variables, group by group with the groups' own launch shapes
gsx_status partial_assemble(gsx_handle h, std::vector<int>& dvar) {
  hipStream_t sm = h->stream;
  gsx_context::PartialScratch& ps = h->ps;
  {
    std::sort(dvar.begin(), dvar.end(), [&](int a, int b) { return h->hv_pos[a] < h->hv_pos[b]; });
    std::vector<std::pair<int, int>> ranges(h->hgroups.size(), {0, 0});
    for (size_t i = 0; i < dvar.size(); ++i) {
      std::pair<int, int>& r = ranges[h->hv_group_of_var[dvar[i]]];
      if (!r.second) r.first = (int)i;
      r.second++;
    }
    HIPCHK(h, ps.hv.stage(dvar, sm));
    timer_begin(h, PH_ASSEMBLE_H);
    for (size_t g = 0; g < h->hgroups.size(); ++g)
      if (ranges[g].second)
        launch_assemble_h_group(h->DP, h->DS, ps.hv.p + ranges[g].first, ranges[g].second, h->hgroups[g].threads,
                                h->hgroups[g].lds, h->hgroups[g].global, h->d_jac.p, h->d_H.p, sm);
    timer_end(h, PH_ASSEMBLE_H);
    h->hdiag_ready = false;
  }
  return GSX_OK;
}
// 3. the listed cliques, level by level, in the order and with the launch shapes of the full schedule (a re-done clique is
//    re-assembled from H and from ALL its children, whose Schur complements / L panels are resident in the arena)
gsx_status partial_factor(gsx_handle h, std::vector<int>& dfr) {
  if (!h->wf_all_replaced) {  // these cliques are "replaced" for the next wildfire pass
    HIPCHK(h, h->wf_flags_ready(h->stream));
    HIPCHK(h, h->d_wf_ids.stage(dfr, h->stream));
    launch_mark_fronts(h->d_wf_ids.p, (int)dfr.size(), h->d_wf_replaced.p, h->stream);
  }
  const Symbolic& S = h->S;
  hipStream_t sm = h->stream;
  gsx_context::PartialScratch& ps = h->ps;
  gsx_status st = GSX_OK;
  {
    std::sort(dfr.begin(), dfr.end(), [&](int a, int b) { return h->fr_sched_pos[a] < h->fr_sched_pos[b]; });
    std::vector<LeafRec> leaf;
    std::vector<int> ids;
    std::vector<BigDesc> big;
    std::vector<GatherSeg> segs;
    std::vector<int> gm_task, gm_slot, gm_nslots;
    struct LevelPlan {
      std::vector<std::array<int, 4>> leaf;   // begin, count, max_panel, threads
      std::vector<SmallLaunch> small;         // ranges of `ids`, with the launch shape of the full schedule's group
      int big_begin = 0, big_count = 0;
      std::vector<int> big_fronts;             // the level's dirty blocked fronts
      // ... by UPPER level: a blocked front is factored with the plan (chunk width, kernel shapes) of its launch group in
      // the full schedule — the group is an upper level — so that the same front takes the same rounds and the same
      // kernels whatever else is dirty (found at config-5 size: a plan made for the dirty subset alone picked another
      // chunk width, and the step differed from the full path's in the last bits)
      std::vector<std::array<int, 3>> big_sub; // begin, count, upper level
      // the gather segments of the level's dirty blocked fronts by gather group (launched one group after the other, in
      // the full schedule's order: the side / lean group before the stored children's — same sums, same bits)
      struct GroupPlan {
        int group;
        std::vector<int> seg_idx, gm_idx;
        int seg0 = 0, nseg = 0, m0 = 0, nm = 0;
      };
      std::vector<GroupPlan> groups;
      GroupPlan& group(int g) {
        for (GroupPlan& gp : groups)
          if (gp.group == g) return gp;
        groups.push_back(GroupPlan{g, {}, {}});
        return groups.back();
      }
    };
    std::vector<LevelPlan> plan(S.n_levels);
    int big_max_n = 0, big_max_nfv = 0;
    {
      int last_level = -1, last_cls = -1, last_group = -1;
      for (int f : dfr) {  // schedule order: level, then class, then launch group
        const int l = S.level[f], cls = S.cls[f], g = h->fr_group[f];
        LevelPlan& L = plan[l];
        const bool same = l == last_level && cls == last_cls && (cls == 2 || g == last_group);
        if (cls == 0) {
          const SmallLaunch& sl = h->leaf_launch[l][g];
          if (!same) L.leaf.push_back({(int)leaf.size(), 0, sl.max_panel, sl.threads});
          L.leaf.back()[1]++;
          leaf.push_back(h->h_leaf_recs[h->fr_sched_pos[f] - h->leaf_base]);
        } else if (cls == 1) {
          const SmallLaunch& sl = h->small_launch[l][g];
          if (!same) {
            L.small.push_back(sl);
            L.small.back().begin = (int)ids.size();
            L.small.back().count = 0;
          }
          L.small.back().count++;
          ids.push_back(f);
        } else {
          const BigDesc& d = h->big_descs[g];
          L.big_fronts.push_back(f);
          big_max_n = std::max(big_max_n, d.N);
          big_max_nfv = std::max(big_max_nfv, S.nfrontal_vars[f]);
          // its gather segments (sources: ALL its children, clean or not), at the level the full schedule runs them
          for (int k = h->fr_seg_ptr[f]; k < h->fr_seg_ptr[f + 1]; ++k)
            L.group(h->seg_level[h->fr_segs[k]]).seg_idx.push_back(h->fr_segs[k]);
          for (int k = h->fr_gm_ptr[f]; k < h->fr_gm_ptr[f + 1]; ++k)
            L.group(h->gm_level[h->fr_gms[k]]).gm_idx.push_back(h->fr_gms[k]);
        }
        last_level = l;
        last_cls = cls;
        last_group = g;
      }
    }
    for (int l = 0; l < S.n_levels; ++l) {
      LevelPlan& L = plan[l];
      std::stable_sort(L.big_fronts.begin(), L.big_fronts.end(), [&](int a, int b) { return S.ulevel[a] < S.ulevel[b]; });
      L.big_begin = (int)big.size();
      L.big_count = (int)L.big_fronts.size();
      for (int f : L.big_fronts) {
        const int ul = S.ulevel[f];
        if (L.big_sub.empty() || L.big_sub.back()[2] != ul) L.big_sub.push_back({(int)big.size(), 0, ul});
        L.big_sub.back()[1]++;
        big.push_back(h->big_descs[h->fr_group[f]]);
      }
      // group order: the side group (index n_ulevels) and group 0 hold the lean leaves' sums — first, as in the full
      // schedule; inside a group no particular order is needed (every segment adds into its own destination block or its
      // own scratch slot)
      std::stable_sort(L.groups.begin(), L.groups.end(), [&](const LevelPlan::GroupPlan& a, const LevelPlan::GroupPlan& b) {
        const int ka = a.group == S.n_ulevels ? -1 : a.group, kb = b.group == S.n_ulevels ? -1 : b.group;
        return ka < kb;
      });
      for (LevelPlan::GroupPlan& G : L.groups) {
        G.seg0 = (int)segs.size();
        for (int i : G.seg_idx) segs.push_back(h->h_gsegs[i]);
        G.nseg = (int)G.seg_idx.size();
        G.m0 = (int)gm_task.size();
        for (int i : G.gm_idx) {
          gm_task.push_back(S.gm_task[i]);
          gm_slot.push_back(S.gm_slot[i]);
          gm_nslots.push_back(S.gm_nslots[i]);
        }
        G.nm = (int)G.gm_idx.size();
      }
    }
    HIPCHK(h, ps.leaf.stage(leaf, sm));
    HIPCHK(h, ps.ids.stage(ids, sm));
    HIPCHK(h, ps.big.stage(big, sm));
    HIPCHK(h, ps.segs.stage(segs, sm));
    HIPCHK(h, ps.gm_task.stage(gm_task, sm));
    HIPCHK(h, ps.gm_slot.stage(gm_slot, sm));
    HIPCHK(h, ps.gm_nslots.stage(gm_nslots, sm));
    GatherArgs GA = h->GA;
    GA.segs = ps.segs.p;
    GA.gm_task = ps.gm_task.p;
    GA.gm_slot = ps.gm_slot.p;
    GA.gm_nslots = ps.gm_nslots.p;
    h->sc_dirty |= kXFact;
    timer_begin(h, PH_FACTORIZE);
    launch_begin_factorization(h->d_scalars.p, 0.0, h->d_status.p, nullptr, sm);
    dev_damping(h, 0, 0, 0);
    if (!big.empty())
      launch_big_init(h->DP, h->DS, ps.big.p, (int)big.size(), big_max_n, big_max_nfv, h->d_H.p, h->d_damp.p,
                      h->d_scalars.p, h->d_arena.p, sm);
    for (int l = 0; l < S.n_levels; ++l) {
      const LevelPlan& L = plan[l];
      for (const auto& g : L.leaf)
        launch_front_leaf(h->DP, h->DS, ps.leaf.p + g[0], g[1], g[2], g[3], h->d_H.p, h->d_damp.p, h->d_scalars.p,
                          h->d_arena.p, h->d_status.p, sm);
      for (const SmallLaunch& g : L.small)
        launch_lds_group(h->DP, h->DS, ps.ids.p + g.begin, g, h->d_H.p, h->d_damp.p, h->d_scalars.p, h->d_arena.p,
                         h->d_status.p, sm);
      if (L.big_count) {
        for (const LevelPlan::GroupPlan& G : L.groups)
          if (G.nseg) launch_big_gather(GA, G.seg0, G.nseg, G.m0, G.nm, h->d_arena.p, sm);
        if (h->constrained())   // (the clean children's leftover rows are still in the work area)
          for (int k = L.big_begin; k < L.big_begin + L.big_count; ++k) {
            const int cd = h->con_desc_of_front[big[k].front];
            if (cd >= 0)
              launch_constraint_fronts(h->DS, h->CT, cd, 1, big[k].N, h->d_jac.p, h->d_arena.p, h->d_status.p, sm);
          }
        for (const auto& sg : L.big_sub)
          dev_big_factor(h, ps.big.p + sg[0], sg[1], h->big_level[sg[2]].plan, sm, false);
      }
    }
    // (all cliques: the clean ones' pivots are resident and unchanged, the test is two loads a clique)
    launch_cond_check((int)S.cond_last.size(), h->d_cond_last.p, h->d_cond_prev.p, h->d_cond_front.p, h->d_arena.p,
                      h->d_status.p, sm);
    timer_end(h, PH_FACTORIZE);
  }
  h->solved = false;
  st = readback(h);  // (also keeps the temporary tables alive until the launches have run)
  if (st != GSX_OK) return st;
  if (h->h_status->n_fail > 0) {
    h->fact_valid = false;
    h->err = "indeterminate linear system";
    return GSX_E_INDETERMINATE;
  }
  return GSX_OK;
}
}  // namespace

static gsx_status gsx_relinearize_partial_impl(gsx_handle h, const uint64_t* keys, int32_t n_keys, const double* states,
                                   int64_t n_states, gsx_partial_stats* out) {
  if (!h || (n_keys > 0 && !keys) || n_keys < 0) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (h->sharded()) {
    h->err = "partial re-elimination is not available on a sharded handle";
    return GSX_E_STATE;
  }
  if (!h->linearized || !h->h_ready || !h->fact_valid || h->fact_lambda != 0.0) {
    h->err = "gsx_relinearize_partial needs the resident undamped factorization of the current linearization "
             "(gsx_linearize + gsx_solve with lambda = 0 first)";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  const HostProblem& P = h->P;
  const Symbolic& S = h->S;
  hipStream_t sm = h->stream;
  // marked variables, and their new states
  std::vector<int> marked(n_keys), src_off(n_keys);
  std::vector<char> is_marked(P.n_vars, 0);
  int64_t need = 0;
  for (int k = 0; k < n_keys; ++k) {
    auto it = std::lower_bound(P.keys.begin(), P.keys.end(), keys[k]);
    if (it == P.keys.end() || *it != keys[k]) {
      h->err = "gsx_relinearize_partial: a key is not a variable of the graph";
      return GSX_E_INVALID;
    }
    const int v = (int)(it - P.keys.begin());
    if (is_marked[v]) return GSX_E_INVALID;
    is_marked[v] = 1;
    marked[k] = v;
    src_off[k] = (int)need;
    need += (v + 1 < P.n_vars ? P.state_off[v + 1] : (int)P.state_size) - P.state_off[v];
  }
  if (states && n_states != need) return GSX_E_INVALID;
  gsx_context::PartialScratch& ps = h->ps;
  if (states && n_keys > 0) {
    HIPCHK(h, ps.marked.stage(marked, sm));
    HIPCHK(h, ps.src_off.stage(src_off, sm));
    HIPCHK(h, ps.states.stage(std::vector<double>(states, states + need), sm));
    launch_scatter_states(h->DP, ps.marked.p, ps.src_off.p, n_keys, ps.states.p, h->d_values.p, sm);
    h->values_synced = true;
  }
  // what is dirty: factors touching a marked variable; the H panels of all their variables; those variables' cliques and
  // every ancestor.  (Everything below is driven by the dirty lists, not by the size of the graph.)
  std::vector<char> f_dirty(P.n_factors, 0), v_dirty(P.n_vars, 0), fr_dirty(S.n_fronts, 0);
  std::vector<int> dfac, dvar, dfr;
  for (int v : marked)
    for (int k = S.vf_ptr[v]; k < S.vf_ptr[v + 1]; ++k) {
      const int f = S.vf[k];
      if (f_dirty[f]) continue;
      f_dirty[f] = 1;
      dfac.push_back(f);
      for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q) {
        const int u = P.f_vars[q];
        if (!v_dirty[u]) {
          v_dirty[u] = 1;
          dvar.push_back(u);
        }
      }
    }
  for (int v : dvar)
    for (int f = S.front_of_var[v]; f >= 0 && !fr_dirty[f]; f = S.parent[f]) {
      fr_dirty[f] = 1;
      dfr.push_back(f);
    }
  if (dfac.empty()) {
    if (out) *out = gsx_partial_stats{0, 0, 0, S.n_fronts};
    return GSX_OK;
  }
  {
    // When most of the extend-add work of the tree is dirty anyway (the big cliques near the root carry most gather
    // segments), filtering costs more on the host than it saves on the device: take the full path — same bits.
    int64_t dseg = 0;
    for (int f : dfr) dseg += h->fr_seg_ptr[f + 1] - h->fr_seg_ptr[f];
    if ((dseg * 10 > (int64_t)h->h_gsegs.size() * 3 && (int64_t)dfr.size() * 20 > S.n_fronts) ||
        (int64_t)dfr.size() * 100 > (int64_t)S.n_fronts * 15) {
      if (out) *out = gsx_partial_stats{P.n_factors, P.n_vars, S.n_fronts, S.n_fronts};
      dev_linearize(h);
      dev_assemble_h(h);
      dev_damping(h, 0, 0, 0);
      dev_factorize(h, 0.0);
      st = readback(h);
      if (st != GSX_OK) return st;
      if (h->h_status->n_fail > 0) {
        h->fact_valid = false;
        h->err = "indeterminate linear system";
        return GSX_E_INDETERMINATE;
      }
      return GSX_OK;
    }
  }
  if (out) *out = gsx_partial_stats{(int)dfac.size(), (int)dvar.size(), (int)dfr.size(), S.n_fronts};
  st = partial_linearize(h, dfac);
  if (st != GSX_OK) return st;
  st = partial_assemble(h, dvar);
  if (st != GSX_OK) return st;
  return partial_factor(h, dfr);
}

// ISAM2::update's structural part on a live handle (include/gsx.h).  The numeric state that survives: the values of the
// kept variables and the [A b] blocks of the kept factors (iSAM2's fixed linearization point).
static gsx_status gsx_update_impl(gsx_handle h, const gsx_problem_desc* desc, const int32_t* factor_origin, const double* new_values,
                      int64_t n_new_values, gsx_update_stats* out) {
  if (!h || !desc || (desc->n_factors > 0 && !factor_origin)) return GSX_E_INVALID;
  gsx_status st = need_device(h);
  if (st != GSX_OK) return st;
  if (h->sharded()) {
    h->err = "gsx_update is not available on a sharded handle";
    return GSX_E_STATE;
  }
  if (!h->has_symbolic || !h->values_set) {
    h->err = "gsx_update needs a handle with values and an ordering (gsx_set_values, gsx_set_ordering first)";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  HostProblem P2;
  st = lower_problem(desc, P2, h->err);
  if (st != GSX_OK) return st;
  const HostProblem& P1 = h->P;
  auto state_len = [](const HostProblem& P, int v) {
    return (v + 1 < P.n_vars ? (int64_t)P.state_off[v + 1] : P.state_size) - P.state_off[v];
  };
  // variables: kept by key
  std::vector<int> old_of(P2.n_vars, -1), new_of(P1.n_vars, -1);
  int64_t need = 0;
  int n_added = 0;
  for (int v = 0; v < P2.n_vars; ++v) {
    auto it = std::lower_bound(P1.keys.begin(), P1.keys.end(), P2.keys[v]);
    if (it != P1.keys.end() && *it == P2.keys[v]) {
      const int o = (int)(it - P1.keys.begin());
      if (P1.types[o] != P2.types[v] || P1.dims[o] != P2.dims[v]) {
        h->err = "gsx_update: a kept variable changed its type or dimension";
        return GSX_E_INVALID;
      }
      old_of[v] = o;
      new_of[o] = v;
    } else {
      need += state_len(P2, v);
      ++n_added;
    }
  }
  if (need != n_new_values || (need > 0 && !new_values)) {
    h->err = "gsx_update: new_values does not hold exactly the states of the new variables";
    return GSX_E_INVALID;
  }
  // factors: kept by origin (same shape), and which variables the change touches
  std::vector<char> used(P1.n_factors, 0), affected(P2.n_vars, 0);
  int n_fadd = 0;
  for (int f = 0; f < P2.n_factors; ++f) {
    const int o = factor_origin[f];
    if (o < -1 || o >= P1.n_factors || (o >= 0 && used[o])) {
      h->err = "gsx_update: factor_origin is not an injective map into the current factors";
      return GSX_E_INVALID;
    }
    if (o >= 0) {
      bool same = P1.f_rows[o] == P2.f_rows[f] && P1.f_cols[o] == P2.f_cols[f] && P1.f_type[o] == P2.f_type[f] &&
                  P1.f_key_ptr[o + 1] - P1.f_key_ptr[o] == P2.f_key_ptr[f + 1] - P2.f_key_ptr[f];
      for (int q = 0; same && q < P2.f_key_ptr[f + 1] - P2.f_key_ptr[f]; ++q)
        same = old_of[P2.f_vars[P2.f_key_ptr[f] + q]] == P1.f_vars[P1.f_key_ptr[o] + q];
      if (!same) {
        h->err = "gsx_update: a kept factor changed its shape or its variables";
        return GSX_E_INVALID;
      }
      used[o] = 1;
    } else {
      ++n_fadd;
      for (int q = P2.f_key_ptr[f]; q < P2.f_key_ptr[f + 1]; ++q) affected[P2.f_vars[q]] = 1;
    }
  }
  int n_frem = 0, n_vrem = 0;
  for (int o = 0; o < P1.n_factors; ++o)
    if (!used[o]) {
      ++n_frem;
      for (int q = P1.f_key_ptr[o]; q < P1.f_key_ptr[o + 1]; ++q)
        if (new_of[P1.f_vars[q]] >= 0) affected[new_of[P1.f_vars[q]]] = 1;
    }
  for (int o = 0; o < P1.n_vars; ++o) n_vrem += new_of[o] < 0;
  for (int v = 0; v < P2.n_vars; ++v)
    if (old_of[v] < 0) affected[v] = 1;
  // the states: current ones of the kept variables + the given ones of the new variables
  std::vector<double> vals1(std::max<int64_t>(P1.state_size, 1)), vals2(std::max<int64_t>(P2.state_size, 1));
  {
    double tmp = 0;
    st = gsx_get_values(h, P1.state_size > 0 ? vals1.data() : &tmp, P1.state_size);
    if (st != GSX_OK) return st;
    int64_t src = 0;
    for (int v = 0; v < P2.n_vars; ++v) {
      const int64_t len = state_len(P2, v);
      if (old_of[v] >= 0) std::copy_n(vals1.data() + P1.state_off[old_of[v]], len, vals2.data() + P2.state_off[v]);
      else {
        std::copy_n(new_values + src, len, vals2.data() + P2.state_off[v]);
        src += len;
      }
    }
  }
  const bool keep_jac = h->linearized;
  // elimination order: unaffected variables in their current relative order, then the affected ones by static degree
  std::vector<int> ord;
  ord.reserve(P2.n_vars);
  for (int pos = 0; pos < P1.n_vars; ++pos) {
    const int v = new_of[h->S.order[pos]];
    if (v >= 0 && !affected[v]) ord.push_back(v);
  }
  {
    std::vector<int> deg(P2.n_vars, 0), aff;
    for (int f = 0; f < P2.n_factors; ++f) {
      const int nk = P2.f_key_ptr[f + 1] - P2.f_key_ptr[f];
      for (int q = P2.f_key_ptr[f]; q < P2.f_key_ptr[f + 1]; ++q) deg[P2.f_vars[q]] += nk - 1;
    }
    for (int v = 0; v < P2.n_vars; ++v)
      if (affected[v]) aff.push_back(v);
    std::stable_sort(aff.begin(), aff.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    ord.insert(ord.end(), aff.begin(), aff.end());
  }
  // the new tree, on the host, before anything of the handle is touched: a failure up to here leaves the handle as it was
  const double relax = h->S.relax;
  const int relax_max_f = h->S.relax_max_f;
  const auto ta = std::chrono::steady_clock::now();
  Symbolic S2;
  st = symbolic_analysis(P2, ord, relax, relax_max_f, 0, 1, S2, h->err);
  if (st != GSX_OK) return st;
  const double t_symbolic = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
  // switch the handle over.  From here on a failure (an allocation at config-5 sizes, a copy) leaves device tables of two
  // different graphs behind: every state flag is dropped first and set again only after the last upload, so that a
  // failed update makes every later numeric call answer GSX_E_STATE until gsx_set_values + gsx_set_ordering rebuild the
  // handle — never kernels on mismatched tables.
  h->has_symbolic = h->values_set = h->linearized = h->h_ready = h->solved = false;
  h->fact_valid = h->fact_pending = h->hdiag_ready = h->damp_ready = h->lin0_ready = false;
  h->wf_delta_valid = false;
  h->wf_all_replaced = true;
  HostProblem P_old = std::move(h->P);
  DevBuf<double> jac_old;
  std::swap(jac_old.p, h->d_jac.p);
  std::swap(jac_old.n, h->d_jac.n);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const auto t0 = std::chrono::steady_clock::now();
  h->P = std::move(P2);
  const HostProblem& P = h->P;
  st = upload_problem(h);
  if (st != GSX_OK) return st;
  if (std::getenv("GSX_INJECT_UPDATE_FAILURE")) {   // (tests/test_gpu_update.py: a failure in the middle of the switch)
    h->err = "gsx_update: injected failure";
    return GSX_E_NOMEM;
  }
  HIPCHK(h, hipMemcpyAsync(h->d_values.p, vals2.data(), (size_t)P.state_size * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (keep_jac) {
    // kept [A b] blocks, device to device, contiguous runs merged
    int64_t src0 = 0, dst0 = 0, len0 = 0;
    auto flush = [&]() -> hipError_t {
      if (!len0) return hipSuccess;
      const hipError_t e = hipMemcpyAsync(h->d_jac.p + dst0, jac_old.p + src0, (size_t)len0 * sizeof(double),
                                          hipMemcpyDeviceToDevice, h->stream);
      len0 = 0;
      return e;
    };
    for (int f = 0; f < P.n_factors; ++f) {
      const int o = factor_origin[f];
      if (o < 0) continue;
      const int64_t s0 = P_old.f_jac_off[o], d0 = P.f_jac_off[f], len = (int64_t)P.f_rows[f] * P.f_cols[f];
      if (len0 && s0 == src0 + len0 && d0 == dst0 + len0) len0 += len;
      else {
        HIPCHK(h, flush());
        src0 = s0;
        dst0 = d0;
        len0 = len;
      }
    }
    HIPCHK(h, flush());
  }
  const auto t1 = std::chrono::steady_clock::now();
  h->relax = relax;
  h->relax_max_f = relax_max_f;
  h->S = std::move(S2);
  h->order = ord;
  st = upload_symbolic(h);
  if (st != GSX_OK) return st;
  const auto t2 = std::chrono::steady_clock::now();
  h->has_symbolic = true;   // (partial_linearize below needs the tables; the other flags follow at the very end)
  h->values_set = true;
  h->values_synced = true;
  bool relinearized = false;
  if (keep_jac) {
    std::vector<int> dfac;
    for (int f = 0; f < P.n_factors; ++f)
      if (factor_origin[f] < 0) dfac.push_back(f);
    hipMemsetAsync(&h->d_status.p->n_cheirality, 0, sizeof(int), h->stream);
    st = partial_linearize(h, dfac);
    if (st != GSX_OK) {
      h->has_symbolic = h->values_set = false;
      return st;
    }
    h->sc_dirty |= kXLin;
    relinearized = true;  // every factor has its [A b]: the kept ones at their old linearization point
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess) {
    h->has_symbolic = h->values_set = false;
    h->err = "gsx_update: device failure";
    return GSX_E_NO_DEVICE;
  }
  h->linearized = relinearized;
  const auto t3 = std::chrono::steady_clock::now();
  if (out) {
    int n_aff = 0;
    for (char a : affected) n_aff += a;
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    *out = gsx_update_stats{n_added, n_vrem, n_fadd, n_frem, n_aff, h->S.n_fronts, t_symbolic + sec(t1, t2), sec(t0, t1) + sec(t2, t3)};
  }
  return GSX_OK;
}

namespace {
// the undamped factorization of the current linearization, resident in the arena (shared by the marginal entry points)
gsx_status marginals_prepare(gsx_handle h) {
  gsx_status st = ensure_ready(h, true, true);
  if (st != GSX_OK) return st;
  if (h->sharded()) {
    h->err = "marginals are not available on a sharded handle";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  if (!h->linearized) dev_linearize(h);
  if (!h->h_ready) dev_assemble_h(h);
  if (!h->fact_valid || h->fact_lambda != 0.0) {
    dev_damping(h, 0, 0, 0);
    dev_factorize(h, 0.0);
    st = readback(h);
    if (st != GSX_OK) return st;
    if (h->h_status->n_fail > 0) {
      h->fact_valid = false;
      h->err = "indeterminate linear system";
      return GSX_E_INDETERMINATE;
    }
    h->solved = false;  // the back-substitution of a previous solve no longer matches the arena
  }
  return GSX_OK;
}
int find_var(gsx_handle h, uint64_t key) {
  auto it = std::lower_bound(h->P.keys.begin(), h->P.keys.end(), key);
  return (it == h->P.keys.end() || *it != key) ? -1 : (int)(it - h->P.keys.begin());
}
}  // namespace

// Marginals::jointMarginalCovariance — gtsam/nonlinear/Marginals.cpp:138-189 (there: eliminate down to the joint, take
// the information, invert).  Here: the columns (L^-1)_{:,v} of every requested variable are kept, and block (a, b) of
// the joint covariance is their product.
gsx_status gsx_joint_marginal_covariance(gsx_handle h, const uint64_t* keys, int32_t n_keys, double* out, int64_t n_out) {
  if (!h || !keys || !out || n_keys < 1 || n_keys > 64) return GSX_E_INVALID;
  gsx_status st = marginals_prepare(h);
  if (st != GSX_OK) return st;
  const Symbolic& S = h->S;
  if (h->constrained()) {
    // (the rewritten fronts give a constraint pivot unit variance where the reference's conditional has sigma 0)
    h->err = "marginal covariances are not available on a problem with hard constraints";
    return GSX_E_STATE;
  }
  // column groups: a variable's unit columns go through the path kernel at most 16 at a time (fewer when the cliques on its
  // path to the root are tall: the kernel keeps two (rows x columns) panels in LDS) — Marginals.cpp:107-136 has no limit on
  // the dimension of a variable, and neither has this entry point
  struct Group {
    int var, col0, width, off;   // off: first row / column of the group in the joint matrix
  };
  std::vector<Group> groups;
  std::vector<int> vars(n_keys);
  int D = 0;
  constexpr size_t kLds = 160 * 1024 - 4096 - 4096;
  for (int k = 0; k < n_keys; ++k) {
    vars[k] = find_var(h, keys[k]);
    if (vars[k] < 0) {
      h->err = "joint marginal of a key that is not a variable of the graph";
      return GSX_E_INVALID;
    }
    for (int q = 0; q < k; ++q)
      if (vars[q] == vars[k]) return GSX_E_INVALID;
    int max_n = 0;
    for (int f = S.front_of_var[vars[k]]; f >= 0; f = S.parent[f]) max_n = std::max(max_n, S.N[f]);
    const int fit = (int)std::min<size_t>(16, kLds / ((size_t)2 * max_n * sizeof(double)));
    if (fit < 1) {
      h->err = "marginal: the cliques on the path to the root are too large for the one-workgroup kernel";
      return GSX_E_NOMEM;
    }
    const int d = h->P.dims[vars[k]];
    for (int c0 = 0; c0 < d; c0 += fit) groups.push_back({vars[k], c0, std::min(fit, d - c0), D + c0});
    D += d;
  }
  if (n_out != (int64_t)D * D) return GSX_E_INVALID;
  const int64_t nt = std::max<int64_t>(h->P.tan_size, 1);
  DevBuf<double> d_Y, d_blk, d_sig;
  DevBuf<int> d_path;
  HIPCHK(h, d_Y.alloc((size_t)nt * D));
  HIPCHK(h, d_blk.alloc((size_t)16 * 16));
  HIPCHK(h, d_sig.alloc(256));
  HIPCHK(h, hipMemsetAsync(d_Y.p, 0, (size_t)nt * D * sizeof(double), h->stream));
  for (const Group& g : groups) {
    std::vector<int> path;
    int max_n = 0;
    for (int f = S.front_of_var[g.var]; f >= 0; f = S.parent[f]) {
      path.push_back(f);
      max_n = std::max(max_n, S.N[f]);
    }
    HIPCHK(h, d_path.upload(path, h->stream));
    launch_marginal_path(h->DS, d_path.p, (int)path.size(), S.h_loc[g.var] + g.col0, g.width, max_n, h->d_arena.p, d_sig.p,
                         d_Y.p + (size_t)g.off * nt, nt, h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));  // (d_path is reused)
  }
  std::vector<double> blk((size_t)16 * 16);
  for (size_t a = 0; a < groups.size(); ++a)
    for (size_t b = a; b < groups.size(); ++b) {
      const int dA = groups[a].width, dB = groups[b].width, oa = groups[a].off, ob = groups[b].off;
      launch_joint_cross(d_Y.p + (size_t)oa * nt, d_Y.p + (size_t)ob * nt, nt, dA, dB, d_blk.p, h->stream);
      HIPCHK(h, hipMemcpyAsync(blk.data(), d_blk.p, (size_t)dA * dB * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (int i = 0; i < dA; ++i)
        for (int j = 0; j < dB; ++j) {
          const double x = blk[i + (size_t)j * dA];
          out[(size_t)(oa + i) * D + ob + j] = x;
          out[(size_t)(ob + j) * D + oa + i] = x;
        }
    }
  HIPCHK(h, hipGetLastError());
  return GSX_OK;
}

gsx_status gsx_marginal_covariance(gsx_handle h, uint64_t key, double* out, int64_t n_out) {
  if (!h || !out) return GSX_E_INVALID;
  const int v = find_var(h, key);
  if (v < 0) {
    h->err = "marginal of a key that is not a variable of the graph";
    return GSX_E_INVALID;
  }
  const int dA = h->P.dims[v];
  if (n_out != (int64_t)dA * dA) return GSX_E_INVALID;
  gsx_status st = marginals_prepare(h);
  if (st != GSX_OK) return st;
  const Symbolic& S = h->S;
  if (h->constrained()) {
    h->err = "marginal covariances are not available on a problem with hard constraints";
    return GSX_E_STATE;
  }
  std::vector<int> path;
  int max_n = 0;
  for (int f = S.front_of_var[v]; f >= 0; f = S.parent[f]) {
    path.push_back(f);
    max_n = std::max(max_n, S.N[f]);
  }
  // a wide variable, or tall cliques on its path: the variable's columns in groups (the joint entry point splits them)
  if (dA > 16 || (size_t)2 * max_n * dA * sizeof(double) > 160 * 1024 - 4096 - 4096)
    return gsx_joint_marginal_covariance(h, &key, 1, out, n_out);
  DevBuf<int> d_path;
  DevBuf<double> d_out;
  HIPCHK(h, d_path.upload(path, h->stream));
  HIPCHK(h, d_out.alloc((size_t)dA * dA));
  launch_marginal_path(h->DS, d_path.p, (int)path.size(), S.h_loc[v], dA, max_n, h->d_arena.p, d_out.p, nullptr, 0,
                       h->stream);
  HIPCHK(h, hipMemcpyAsync(out, d_out.p, (size_t)dA * dA * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  return GSX_OK;
}

static gsx_status gsx_set_block_jacobians_impl(gsx_handle h, int32_t first_factor, int32_t n_factors, const double* values,
                                   int64_t n_values) {
  if (!h || !values || first_factor < 0 || n_factors < 0 || (int64_t)first_factor + (int64_t)n_factors > (int64_t)h->P.n_factors) return GSX_E_INVALID;
  gsx_status st = need_device(h);
  if (st != GSX_OK) return st;
  const HostProblem& P = h->P;
  int64_t need = 0;
  for (int f = first_factor; f < first_factor + n_factors; ++f) {
    if (P.f_type[f] != GSX_F_LINEAR) {
      h->err = "gsx_set_block_jacobians: not a GSX_F_LINEAR factor";
      return GSX_E_INVALID;
    }
    need += (int64_t)P.f_rows[f] * P.f_cols[f];
  }
  if (need != n_values) return GSX_E_INVALID;
  if (n_factors == 0) return GSX_OK;
  hipSetDevice(h->device);
  // the factors' blocks are consecutive in the Jacobian pool when the factors are: one staging buffer, one copy per run
  std::vector<double>& stage = h->jac_stage;
  stage.resize((size_t)need);
  int64_t src = 0;
  for (int f = first_factor; f < first_factor + n_factors; ++f) {
    whiten_linear_factor(P, f, values + src, stage.data() + src);
    src += (int64_t)P.f_rows[f] * P.f_cols[f];
  }
  src = 0;
  for (int f = first_factor; f < first_factor + n_factors;) {
    int e = f;
    int64_t len = 0;
    while (e < first_factor + n_factors && P.f_jac_off[e] == P.f_jac_off[f] + len) {
      len += (int64_t)P.f_rows[e] * P.f_cols[e];
      ++e;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_jac.p + P.f_jac_off[f], stage.data() + src, (size_t)len * sizeof(double),
                             hipMemcpyHostToDevice, h->stream));
    src += len;
    f = e;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));  // (the staging buffer is reused by the next call)
  h->h_ready = false;      // H must be re-assembled from the new blocks
  h->lin0_ready = false;
  h->hdiag_ready = false;
  h->fact_valid = false;
  h->solved = false;
  if (h->damp_kind != 0) h->damp_ready = false;
  return GSX_OK;
}

gsx_status gsx_solve_gfg_h(gsx_handle h, const double* blocks, int64_t n_blocks, double* delta_out, int64_t n,
                           uint64_t* bad_key) {
  if (!h || !delta_out) return GSX_E_INVALID;
  for (int f = 0; f < h->P.n_factors; ++f)
    if (h->P.f_type[f] != GSX_F_LINEAR) {
      h->err = "gsx_solve_gfg_h: the handle must hold GSX_F_LINEAR factors only";
      return GSX_E_INVALID;
    }
  gsx_status st = GSX_OK;
  if (blocks) st = gsx_set_block_jacobians(h, 0, h->P.n_factors, blocks, n_blocks);
  if (st != GSX_OK) return st;
  if (!h->values_set) {
    std::vector<double> zeros(h->P.state_size, 0.0);
    st = gsx_set_values(h, zeros.data(), h->P.state_size);
    if (st != GSX_OK) return st;
  }
  h->linearized = true;  // a linear graph IS its linearization
  return gsx_solve(h, 0.0, 0, 0, 0, delta_out, n, bad_key);
}

static gsx_status gsx_get_conditional_impl(gsx_handle h, int32_t front, int32_t* n_frontal, int32_t* n_cols, double* out,
                               int64_t n_out) {
  if (!h) return GSX_E_INVALID;
  gsx_status st = ensure_ready(h, false, true);
  if (st != GSX_OK) return st;
  const Symbolic& S = h->S;
  if (front < 0 || front >= S.n_fronts) return GSX_E_INVALID;
  const int F = S.F[front], N = S.N[front];  // N = F + separator + 1 (rhs)
  if (n_frontal) *n_frontal = F;
  if (n_cols) *n_cols = N;
  if (!out) return GSX_OK;
  if (n_out != (int64_t)F * N) return GSX_E_INVALID;
  if (!h->fact_valid) {
    h->err = "gsx_get_conditional: no factorization resident (solve first)";
    return GSX_E_STATE;
  }
  if (h->sharded() && !S.scheduled[front]) {
    h->err = "gsx_get_conditional: the clique belongs to another rank";
    return GSX_E_STATE;
  }
  hipSetDevice(h->device);
  // the front keeps L = [R S d]' column by column: L[r][c] at r + c N; blocked (big) fronts keep the rows below each
  // 32 x 32 diagonal tile in their L-panel area (kernels.h: BigDesc)
  const bool big = S.cls[front] == 2;
  const i64 cols = (i64)N * F;
  std::vector<double> sq((size_t)cols), xp;
  HIPCHK(h, hipMemcpyAsync(sq.data(), h->d_arena.p + S.off[front], (size_t)cols * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
  if (big) {
    xp.resize((size_t)cols);
    HIPCHK(h, hipMemcpyAsync(xp.data(), h->d_arena.p + S.off[front] + big_panel_offset(N), (size_t)cols * sizeof(double),
                             hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int c = 0; c < F; ++c)        // column c of L = row c of [R S d]
    for (int r = 0; r < N; ++r) {
      double v = 0.0;
      if (r >= c) {
        const bool in_tile = big && r < F && (r / kTile) == (c / kTile);
        v = (big && !in_tile) ? xp[(size_t)r + (size_t)c * N] : sq[(size_t)r + (size_t)c * N];
      }
      out[(size_t)c + (size_t)r * F] = v;  // F x N column-major: entry (row c, column r)
    }
  return GSX_OK;
}

gsx_status gsx_solve_gfg(const gsx_problem_desc* desc, const uint64_t* ordering, int32_t device, double* delta_out,
                         int64_t n, uint64_t* bad_key) {
  if (!desc || !delta_out) return GSX_E_INVALID;
  for (int f = 0; f < desc->n_factors; ++f)
    if (desc->f_type[f] != GSX_F_LINEAR) return GSX_E_INVALID;
  gsx_handle h = nullptr;
  gsx_status st = gsx_create(desc, device, &h);
  if (st != GSX_OK) return st;
  std::vector<uint64_t> ord(desc->n_vars);
  if (ordering) std::copy(ordering, ordering + desc->n_vars, ord.begin());
  else st = gsx_compute_ordering(h, GSX_ORDER_MINDEGREE, ord.data());
  if (st == GSX_OK) st = gsx_set_ordering(h, ord.data(), desc->n_vars);
  std::vector<double> zeros(h->P.state_size, 0.0);
  if (st == GSX_OK) st = gsx_set_values(h, zeros.data(), h->P.state_size);
  if (st == GSX_OK) st = gsx_linearize(h);
  if (st == GSX_OK) st = gsx_solve(h, 0.0, 0, 0, 0, delta_out, n, bad_key);
  gsx_destroy(h);
  return st;
}

gsx_status gsx_cholesky_partial(double* abc, int32_t n, int32_t nfrontal, int32_t device, int32_t* ok) {
  if (!abc || n <= 0 || nfrontal < 0 || nfrontal > n || !ok) return GSX_E_INVALID;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device >= nd || hipSetDevice(device) != hipSuccess)
    return GSX_E_NO_DEVICE;
  *ok = 1;
  if (nfrontal == 0) return GSX_OK;
  // lower (ours) = transpose of upper (reference)
  std::vector<double> a((size_t)n * n);
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < n; ++r) a[(size_t)c * n + r] = (r >= c) ? abc[(size_t)r * n + c] : 0.0;
  double* d = nullptr;
  DevStatus* ds = nullptr;
  hipStream_t st;
  if (hipStreamCreate(&st) != hipSuccess) return GSX_E_NO_DEVICE;
  if (hipMalloc((void**)&d, a.size() * sizeof(double)) != hipSuccess) return GSX_E_NOMEM;
  hipMalloc((void**)&ds, sizeof(DevStatus));
  DevStatus init{0, INT_MAX, 0, 0};
  hipMemcpyAsync(ds, &init, sizeof(init), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, st);
  launch_dense_partial(d, n, nfrontal, ds, st);
  hipMemcpyAsync(a.data(), d, a.size() * sizeof(double), hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(&init, ds, sizeof(init), hipMemcpyDeviceToHost, st);
  hipError_t e = hipStreamSynchronize(st);
  hipFree(d);
  hipFree(ds);
  hipStreamDestroy(st);
  if (e != hipSuccess) return GSX_E_NO_DEVICE;
  *ok = init.n_fail == 0 ? 1 : 0;
  for (int c = 0; c < n; ++c)
    for (int r = 0; r <= c; ++r) abc[(size_t)c * n + r] = a[(size_t)r * n + c];
  return GSX_OK;
}


// No exception crosses the C boundary (include/gsx.h): the entry points whose work is sized by caller input (host vectors
// of the plans, staging buffers) answer an allocation failure with a status, like the readers in io.cpp.
#define GSX_GUARD(h, call)                          \
  try {                                             \
    return (call);                                  \
  } catch (const std::bad_alloc&) {                 \
    if (h) (h)->err = "out of host memory";         \
    return GSX_E_NOMEM;                             \
  } catch (const std::exception& e) {               \
    if (h) (h)->err = e.what();                     \
    return GSX_E_INVALID;                           \
  }
gsx_status gsx_set_block_jacobians(gsx_handle h, int32_t first_factor, int32_t n_factors, const double* values,
                                   int64_t n_values) {
  GSX_GUARD(h, gsx_set_block_jacobians_impl(h, first_factor, n_factors, values, n_values));
}
gsx_status gsx_update(gsx_handle h, const gsx_problem_desc* desc, const int32_t* factor_origin, const double* new_values,
                      int64_t n_new_values, gsx_update_stats* out) {
  GSX_GUARD(h, gsx_update_impl(h, desc, factor_origin, new_values, n_new_values, out));
}
gsx_status gsx_get_conditional(gsx_handle h, int32_t front, int32_t* n_frontal, int32_t* n_cols, double* out,
                               int64_t n_out) {
  GSX_GUARD(h, gsx_get_conditional_impl(h, front, n_frontal, n_cols, out, n_out));
}
gsx_status gsx_relinearize_partial(gsx_handle h, const uint64_t* keys, int32_t n_keys, const double* states,
                                   int64_t n_states, gsx_partial_stats* out) {
  GSX_GUARD(h, gsx_relinearize_partial_impl(h, keys, n_keys, states, n_states, out));
}

gsx_status gsx_get_stats(gsx_handle h, gsx_stats* out) {
  if (!h || !out) return GSX_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  if (h->has_symbolic) {
    const Symbolic& S = h->S;
    out->n_fronts = S.n_fronts;
    out->n_levels = S.n_levels;
    out->max_front_dim = S.max_F;
    out->max_front_rows = S.max_rows;
    out->n_small_fronts = S.n_small;
    out->n_big_fronts = S.n_big;
    out->n_upper_levels = S.n_ulevels;
    out->n_medium_fronts = 0;
    for (int f = 0; f < S.n_fronts; ++f) {
      out->n_medium_fronts += S.med[f] && S.cls[f] == 1;
      out->n_tree_fronts += S.tree_tier[f] >= 0;
    }
    out->n_constrained_fronts = (int64_t)S.con_fronts.size();
    out->factor_flops = S.flops;
    out->front_bytes = S.front_bytes;
    out->lpanel_bytes = S.lpanel_bytes;
    out->hessian_bytes = 8.0 * (double)S.h_size;
  }
  out->n_constraint_rows = (int64_t)h->P.con_factor.size();
  out->jacobian_bytes = 8.0 * (double)h->P.jac_size;
  out->total_dim = (double)h->P.tan_size;
  if (h->has_device) {
    hipSetDevice(h->device);
    timers_resolve(h);
  }
  out->ms_linearize = h->timers[PH_LINEARIZE].ms;
  out->ms_assemble_hessian = h->timers[PH_ASSEMBLE_H].ms;
  out->ms_factorize = h->timers[PH_FACTORIZE].ms;
  out->ms_backsolve = h->timers[PH_BACKSOLVE].ms;
  out->ms_linear_error = h->timers[PH_LINERR].ms;
  out->ms_retract = h->timers[PH_RETRACT].ms;
  out->ms_error = h->timers[PH_ERROR].ms;
  out->n_linearize = h->timers[PH_LINEARIZE].count;
  out->n_factorize = h->timers[PH_FACTORIZE].count;
  out->n_backsolve = h->timers[PH_BACKSOLVE].count;
  out->n_error = h->timers[PH_ERROR].count;
  out->n_cheirality = h->n_cheirality;
  out->amalgamation_relax = h->S.relax;
  out->amalgamation_max_frontal_dim = h->S.relax_max_f;
  return GSX_OK;
}

gsx_status gsx_reset_stats(gsx_handle h) {
  if (!h) return GSX_E_INVALID;
  if (h->has_device) {
    hipSetDevice(h->device);
    timers_resolve(h);
  }
  for (auto& t : h->timers) {
    t.ms = 0;
    t.count = 0;
  }
  return GSX_OK;
}

gsx_status gsx_synchronize(gsx_handle h) {
  if (!h) return GSX_E_INVALID;
  gsx_status st = need_device(h);
  if (st != GSX_OK) return st;
  hipSetDevice(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GSX_OK;
}

gsx_status gsx_set_profiling(gsx_handle h, int32_t level) {
  if (!h) return GSX_E_INVALID;
  h->profiling = level;
  return GSX_OK;
}

gsx_status gsx_kernel_time(gsx_handle h, const char* name, double* avg_ms, int64_t* launches) {
  if (!h || !name || !avg_ms) return GSX_E_INVALID;
  if (h->has_device) {
    hipSetDevice(h->device);
    timers_resolve(h);
  }
  for (int ph = 0; ph < PH_COUNT; ++ph)
    if (std::strcmp(name, kPhaseNames[ph]) == 0) {
      *avg_ms = h->timers[ph].count ? h->timers[ph].ms / (double)h->timers[ph].count : 0.0;
      if (launches) *launches = h->timers[ph].count;
      return GSX_OK;
    }
  return GSX_E_INVALID;
}

}  // extern "C"
