// gsx_internal.h — host-side data structures of libgsx (product; never includes oracle/).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gsx.h"

namespace gsx {

// Size classes of fronts: S* are eliminated inside LDS by one workgroup, BIG by the blocked
// global-memory path.  n = frontal + separator + 1 (rhs) scalar rows.
constexpr int kSmallMaxN = 140;     // 140^2 * 8 B = 156.8 KB <= 160 KB LDS
constexpr int kTile = 32;           // tile edge of the blocked big-front path
constexpr int kGatherChunk = 64;    // sources per gather segment: one wave, one source record per lane
constexpr int kLeafMaxF = 16;      // leaf cliques with at most this many frontal scalars use the panel-only kernel
constexpr int kLeafMaxPanel = 8192; // doubles of LDS (n x F) a lean leaf may use: 64 KB
// MEDIUM fronts: more rows than fit LDS as a square, but whose n x F frontal panel does.  One workgroup eliminates such a
// front with the panel in LDS and the trailing block in the arena (front_medium_body, kernels.hip); storage and class
// (cls 1) are those of an LDS front, so parents, back-substitution, marginals read them unchanged.
constexpr int kMedMaxN = 512;       // rows (two staged row maps of that many ints)
constexpr int kMedMaxPanel = 18432; // doubles of LDS for the panel: 144 KB
constexpr int kMedMaxLeafKids = 8;  // a front with more childless small children than this stays blocked: its leaves are
                                    // LEAN only under a blocked parent (every BAL camera front: hundreds of landmarks)

struct HostProblem {
  int n_vars = 0, n_factors = 0;
  std::vector<uint64_t> keys;
  std::vector<int> types, dims, state_off, tan_off;
  int64_t state_size = 0, tan_size = 0, jac_size = 0;
  // factors (graph order)
  std::vector<int> f_type, f_rows, f_key_ptr, f_vars, f_noise_kind;
  std::vector<int64_t> f_meas_ptr, f_noise_ptr, f_jac_off;
  std::vector<double> meas, noise;
  std::vector<int> f_cols;           // sum d + 1
  // hard-constraint rows (zero sigma), by (factor, row) in graph order, with the weight mu of the error functions; on the
  // device they are rows of sigma 1 / sqrt(mu) (problem.cpp) that the factorization eliminates by constraint pivots
  std::vector<int> con_factor, con_row;
  std::vector<double> con_mu;
};

// Symbolic factorization for one ordering.
struct Symbolic {
  double relax = 0.0;                // relaxed amalgamation threshold (0 = the reference's cliques)
  int relax_max_f = 128;             // ... and the largest merged frontal dimension
  int n_fronts = 0;
  std::vector<int> order;            // position -> var
  std::vector<int> pos;              // var -> position
  std::vector<int> front_of_var;     // var -> front in which it is frontal
  std::vector<int> parent;           // front -> parent front (-1 root)
  std::vector<int> fvar_ptr, fvars;  // CSR: front -> local variable list (frontals first, then separator by position)
  std::vector<int> nfrontal_vars;    // front -> number of frontal variables
  std::vector<int> F, S, N;          // frontal dim, separator dim, n = F+S+1
  std::vector<int64_t> off;          // arena offset (doubles) of the n x n column-major front
  std::vector<int> level;            // 0 = leaves
  std::vector<char> cls, lean;       // size class (0 leaf kernel, 1 small, 2 big); lean leaf (no Schur complement stored)
  std::vector<char> med;             // cls 1 with n > kSmallMaxN: a MEDIUM front (panel in LDS, trailing block in the arena)
  std::vector<int> child_ptr, children;  // CSR front -> children (in the reference's child order)
  // scalar row maps
  std::vector<int64_t> cmap_ptr;     // front -> offset in cmap of its (S+1) update-row -> parent-row map
  std::vector<int> cmap;
  std::vector<int64_t> gidx_ptr;     // front -> offset in gidx of its (F+S) row -> global tangent index map
  std::vector<int> gidx;
  // H panels (per variable, in elimination order of rows)
  std::vector<int64_t> h_off;        // var -> offset of its panel in the H pool
  std::vector<int> h_rows;           // var -> panel rows (d_A + sum later-neighbour dims + 1)
  std::vector<int64_t> hmap_ptr;     // var -> offset in hmap of panel-row -> front-row map
  std::vector<int> hmap;
  std::vector<int> h_loc;            // var -> local scalar column offset of the variable in its front
  int64_t h_size = 0, arena_size = 0;
  // H assembly term lists: per variable, terms (one per (factor, destination block))
  std::vector<int64_t> term_ptr;     // var -> range in the term arrays
  std::vector<int64_t> t_jac;        // offset of the factor's [A b]
  std::vector<int> t_m, t_colA, t_colB, t_dB, t_dst;
  // schedule: fronts sorted by (level, class); per level the ranges of each class
  std::vector<int> sched;            // front ids
  std::vector<int> lvl_ptr;          // level -> range in sched (size n_levels+1)
  std::vector<int> lvl_small_end;    // level -> end of the small (LDS) fronts inside the level range
  std::vector<int> lvl_leaf_end;     // level 0 only: end of the leaf-kernel fronts (no children, F <= kLeafMaxF)
  int n_levels = 0;
  // The UPPER schedule of the full factorization / back-substitution: the fronts that are neither tree fronts nor leaf-kernel
  // cliques (every blocked front, and the LDS fronts above one) are levelled among THEMSELVES — ulevel = longest chain of
  // upper fronts below — because everything else is finished by the leaf launches and the tree kernels before the first
  // of them starts.  (The height-from-the-leaves levels above stay the order of gsx_relinearize_partial and of the wildfire
  // pass.)  pose3_100k: 17 levels with blocked fronts by height, 11 by ulevel.
  std::vector<int> ulevel;           // front -> upper level, -1: not an upper front
  std::vector<int> usched;           // upper fronts by (ulevel, LDS-class before blocked, n)
  std::vector<int> ulvl_ptr;         // ulevel -> range in usched (size n_ulevels + 1)
  std::vector<int> ulvl_small_end;   // ulevel -> end of the LDS-class fronts inside the range
  int n_ulevels = 0;
  int cap_ulevel0 = -1;              // first upper level of the cap, -1: no cap
  // SIDE work: lean leaves whose (blocked) parent sits above the first blocked level are needed late — their leaf
  // factorization and their product-form gather (gather group `n_levels`, scratch slots from side_slot0) run on a second,
  // low-priority queue beside the blocked chain of the lower levels; the main queue waits for them before level
  // side_level0 + 1.  side_level0 < 0: none.
  int side_level0 = -1;              // the first UPPER level with blocked fronts
  int leaf_side_begin = 0;           // position in sched where the side leaves of level 0 start (== lvl_leaf_end[0]: none)
  int side_slot0 = 0;                // first scratch slot of the side gather group
  std::vector<char> side;            // front -> side leaf
  // Subtree-fused elimination of the LDS-class fronts (front_tree_kernel): a front is a TREE front when it is LDS-class and
  // so is its whole subtree (leaf-kernel children aside).  Tree fronts are cut into tiers by the largest front of their
  // subtree (= the LDS a workgroup needs to climb it); one launch per tier, inside which a workgroup that finishes a front
  // takes the parent when it was the last child to arrive (ClusterTree-inst.h:219-318 post-order, without level barriers).
  std::vector<int> tree_bounds;      // tier -> largest n of its fronts
  std::vector<int> tree_threads;     // tier -> workgroup size (0: the host's default for that n)
  std::vector<int> tree_tier;        // front -> tier, -1: not a tree front
  std::vector<int> tree_up;          // front -> parent when that is a tree front of the same tier, else -1
  std::vector<int> tree_npend;       // front -> children that are tree fronts of the same tier
  std::vector<int> tree_start;       // per tier (tree_start_ptr): the fronts with tree_npend == 0, in schedule order
  std::vector<int> tree_start_ptr;
  // deterministic extend-add into BIG parents: one task per destination block (parent var pair),
  // with the list of source blocks (child Schur-complement blocks) in child order
  std::vector<int64_t> gt_dst;       // task -> arena offset of the destination block (top-left entry)
  std::vector<int> gt_ld, gt_dims;   // destination leading dimension; dB | dA << 8 | diag << 16
  std::vector<int64_t> gt_ptr;       // task -> range in the source arrays (size n_tasks + 1)
  std::vector<int> gs_child, gs_loc; // source: child front id, offset of the block inside the child's front
  std::vector<int> gs_loc2;          // lean child: row of the second factor in its L panel (gs_loc = row of the first); else -1
  std::vector<int> gt_lvl_ptr;       // upper level of the PARENT -> task range (size n_ulevels + 2: the side group last)
  // segments of the source lists (one wave each) and the multi-segment tasks that need a combine pass
  std::vector<int> gseg_task, gseg_slot;          // segment -> task, scratch slot (-1: adds straight to dst)
  std::vector<int64_t> gseg_begin, gseg_end;      // segment -> source range
  std::vector<int> gseg_lvl_ptr;                  // upper level -> segment range (size n_ulevels + 2: the side group last)
  std::vector<int> gm_task, gm_slot, gm_nslots;   // multi-segment task -> first slot, number of slots
  std::vector<int> gm_lvl_ptr;                    // upper level -> multi-task range (size n_ulevels + 2)
  int g_max_slots = 0;                            // scratch slots needed (x 128 doubles), reused per level
  // choleskyPartial's conditioning test per REFERENCE clique: arena offsets of the last / second-to-last (-1: single pivot)
  // diagonal entries of L, and the front that holds them (only the fronts this rank factors)
  std::vector<int64_t> cond_last, cond_prev;
  std::vector<int> cond_front;
  // for partial re-elimination (gsx_relinearize_partial): variable -> factors CSR, gather task -> destination front
  std::vector<int> vf_ptr, vf;
  std::vector<int> gt_front;
  // HARD CONSTRAINTS (HostProblem::con_*; EliminatePreferCholesky -> EliminateQR, HessianFactor.cpp:538-551,
  // Constrained::QR, NoiseModel.cpp:503-620).  A constraint row lives in the front where the first-eliminated variable of
  // its factor is frontal; there it is used as a PIVOT (one frontal scalar expressed by the others — no Cholesky step for
  // it), or, when the front has fewer frontal scalars than rows, handed to the parent.  A front that takes rows in is
  // always BLOCKED (cls 2): its assembled panel is in the arena, where the constraint kernels rewrite it into an
  // unconstrained front with equivalent conditionals and the same Schur complement (constraint.hip).
  std::vector<char> con;               // front -> takes constraint rows in
  std::vector<int> con_fronts;         // those fronts, children before parents
  std::vector<int> con_index;          // front -> index in con_fronts, -1
  std::vector<int> con_in, con_fwd;    // per constrained front: rows in (own + the children's), rows handed to the parent
  std::vector<int> con_own_ptr;        // constrained front -> its own rows (CSR into con_own_*)
  std::vector<int64_t> con_own_jac;    // own row -> offset of its first entry in the Jacobian store ([A b] + row)
  std::vector<int> con_own_m;          // ... rows of its factor (the column stride)
  std::vector<int> con_own_col_ptr;    // own row -> range in con_own_cols (size + 1)
  std::vector<int> con_own_cols;       // per column of the factor's [A b]: the front's row (n - 1: the rhs)
  std::vector<int> con_child_ptr, con_child;  // constrained front -> the constrained fronts below it whose leftover rows arrive here
  std::vector<int> con_fwd_map_ptr, con_fwd_map;  // constrained front -> per separator + rhs row: the row of the front its
                                                  // leftover rows go to (-1: a variable none of them holds)
  // sharding of ONE problem over the GPUs of a node (gsx_set_shard): the fronts whose subtree is cheaper than a share of
  // the whole tree form independent subtrees dealt to the ranks; the rest — the top of the tree, the "cap" — is
  // assembled from every rank's contributions with one all-reduce and factored by all ranks alike.
  int shard_rank = 0, shard_world = 1;
  std::vector<int> owner;            // front -> owning rank, -1 = cap (all ranks); all 0 when not sharded
  std::vector<char> scheduled;       // front -> this rank processes it (owner == rank or cap)
  std::vector<char> f_owned;         // factor -> this rank linearizes it and adds its J'J terms
  int cap_level0 = -1;               // first level of the cap (every cap front is at or above it), -1: no cap
  int64_t cap_begin = 0, cap_end = 0;  // the cap fronts' contiguous share of the arena (doubles)
  int n_cap = 0;
  double cap_cost = 0, own_cost = 0, total_cost = 0;  // flop estimates: cap, this rank's subtrees, whole tree
  // stats
  double flops = 0, front_bytes = 0, lpanel_bytes = 0;
  int64_t max_F = 0, max_rows = 0, n_small = 0, n_big = 0, n_medium = 0;
};

// host algorithms
gsx_status lower_problem(const gsx_problem_desc* d, HostProblem& P, std::string& err);
void compute_ordering(const HostProblem& P, int kind, std::vector<int>& order);
// multilevel nested dissection of the vertices `verts` of the graph `adj` (nd.cpp); sets of at most `leaf` vertices and
// the separators are ordered by `leaf_order` (appends to `out`)
typedef void (*NdLeafOrder)(const std::vector<std::vector<int>>& adj, const std::vector<int>& w,
                            const std::vector<int>& verts, std::vector<int>& out);
void multilevel_nested_dissection(const std::vector<std::vector<int>>& adj, const std::vector<int>& w,
                                  const std::vector<int>& verts, int leaf, NdLeafOrder leaf_order, std::vector<int>& out);
gsx_status symbolic_analysis(const HostProblem& P, const std::vector<int>& order, double relax, int relax_max_f,
                             int shard_rank, int shard_world, Symbolic& S, std::string& err);

}  // namespace gsx
