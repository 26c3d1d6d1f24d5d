// symbolic.cpp — symbolic analysis, done ONCE per (graph, ordering) and cached in the handle.
//
// Produces the same Bayes-tree cliques as the reference's EliminationTree constructor
// (gtsam/inference/EliminationTree-inst.h:78-156: a factor hangs off its first-eliminated
// variable, children are attached in factor order) followed by the JunctionTree constructor
// (gtsam/inference/JunctionTree-inst.h:51-153: symbolic elimination + the merge rule
// "myNrParents + myNrFrontals == child.nrParents", ClusterTree-inst.h:58-96) — but with its own
// data structures (flat arrays, path-compressed root finding, no recursion), and it goes on to
// build everything the device needs: per-front variable lists and scalar row maps, the
// per-variable Hessian panels with their assembly term lists, the level schedule and the arena
// layout.  The reference redoes this work at every solve (EliminateableFactorGraph-inst.h:123-146).
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <algorithm>
#include <iterator>
#include <numeric>

#include "gsx_internal.h"

namespace gsx {

namespace {
struct PhaseClock {   // GSX_TIME_SYMBOLIC=1: phase times of the analysis on stderr
  bool on = std::getenv("GSX_TIME_SYMBOLIC") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[symbolic] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t).count());
    t = now;
  }
};
}  // namespace

gsx_status symbolic_analysis(const HostProblem& P, const std::vector<int>& order, double relax, int relax_max_f,
                             int shard_rank, int shard_world, Symbolic& S, std::string& err) {
  const int n = P.n_vars, m = P.n_factors;
  if ((int)order.size() != n) {
    err = "ordering size differs from the number of variables";
    return GSX_E_BAD_ORDERING;
  }
  PhaseClock clk;
  S = Symbolic();
  S.relax = relax;
  S.relax_max_f = relax_max_f;
  S.shard_rank = shard_rank;
  S.shard_world = std::max(shard_world, 1);
  S.order = order;
  S.pos.assign(n, -1);
  for (int j = 0; j < n; ++j) {
    const int v = order[j];
    if (v < 0 || v >= n || S.pos[v] != -1) {
      err = "ordering is not a permutation of the variables";
      return GSX_E_BAD_ORDERING;
    }
    S.pos[v] = j;
  }
  // variable -> factors (ascending factor index), CSR
  std::vector<int> vf_ptr(n + 1, 0), vf;
  for (int k = 0; k < (int)P.f_vars.size(); ++k) vf_ptr[P.f_vars[k] + 1]++;
  for (int v = 0; v < n; ++v) vf_ptr[v + 1] += vf_ptr[v];
  vf.resize(P.f_vars.size());
  {
    std::vector<int> fill(vf_ptr.begin(), vf_ptr.end() - 1);
    for (int f = 0; f < m; ++f)
      for (int k = P.f_key_ptr[f]; k < P.f_key_ptr[f + 1]; ++k) vf[fill[P.f_vars[k]]++] = f;
  }
  S.vf_ptr = vf_ptr;
  S.vf = vf;
  clk.mark("variable-factor lists");
  // ---- elimination tree (nodes are elimination positions) -------------------------------------
  std::vector<int> eparent(n, -1), ancestor(n, -1), prevCol(m, -1);
  std::vector<int> ech_ptr(n + 1, 0);
  std::vector<std::pair<int, int>> child_edges;  // (parent j, child r) in attachment order
  child_edges.reserve(n);
  for (int j = 0; j < n; ++j) {
    const int v = order[j];
    for (int k = vf_ptr[v]; k < vf_ptr[v + 1]; ++k) {
      const int f = vf[k];
      if (prevCol[f] != -1) {
        int r = prevCol[f];
        // root of the current subtree containing r (path-compressed ancestor walk)
        int t = r;
        while (ancestor[t] != -1 && ancestor[t] != j) t = ancestor[t];
        const int root = t;
        t = r;
        while (t != root) {  // compress
          const int nx = ancestor[t];
          ancestor[t] = j;
          t = nx;
        }
        if (root != j) {
          if (ancestor[root] == -1) {
            ancestor[root] = j;
            eparent[root] = j;
            child_edges.push_back({j, root});
          }
        }
      }
      prevCol[f] = j;
    }
  }
  for (auto& e : child_edges) ech_ptr[e.first + 1]++;
  for (int j = 0; j < n; ++j) ech_ptr[j + 1] += ech_ptr[j];
  std::vector<int> ech(child_edges.size());
  {
    std::vector<int> fill(ech_ptr.begin(), ech_ptr.end() - 1);
    for (auto& e : child_edges) ech[fill[e.first]++] = e.second;  // attachment order preserved
  }
  clk.mark("elimination tree");
  // ---- symbolic column structures: struct[j] = later positions coupled to j, ascending ----------
  std::vector<std::vector<int>> st(n);
  std::vector<int> stamp(n, -1);
  for (int j = 0; j < n; ++j) {
    const int v = order[j];
    std::vector<int>& s = st[j];
    stamp[j] = j;
    for (int k = vf_ptr[v]; k < vf_ptr[v + 1]; ++k) {
      const int f = vf[k];
      for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q) {
        const int pj = S.pos[P.f_vars[q]];
        if (pj > j && stamp[pj] != j) {
          stamp[pj] = j;
          s.push_back(pj);
        }
      }
    }
    for (int c = ech_ptr[j]; c < ech_ptr[j + 1]; ++c)
      for (int pj : st[ech[c]])
        if (pj > j && stamp[pj] != j) {
          stamp[pj] = j;
          s.push_back(pj);
        }
    std::sort(s.begin(), s.end());
  }
  clk.mark("column structures");
  // ---- supernodes: the reference's merge rule ------------------------------------------------------
  // merged[j] = true when node j was merged into its etree parent's cluster.
  std::vector<char> merged(n, 0);
  std::vector<char> ref_merged(n, 0);  // ... by the reference's rule alone (the reference's cliques inside a relaxed one)
  std::vector<int> nfront_of(n, 1);  // frontal variable count of the cluster topped by node j
  // relaxed amalgamation (relax > 0; off = the reference's Bayes tree exactly): a child cluster is also merged
  // when the explicit zeros this adds to its columns are at most relax x its own L panel and the merged
  // frontal dimension stays moderate.  Fewer, larger cliques => fewer levels / kernel launches on the
  // latency-bound chains; the solution is unchanged (zeros are factored as zeros).
  std::vector<int64_t> fdim_of(n, 0), sdim_of(n, 0);  // scalar frontal / separator dims of the cluster topped by j
  std::vector<int> ref_nfront_of(n, 1);
  std::vector<int64_t> sdim_node(n, 0);
  for (int j = 0; j < n; ++j)
    for (int pj : st[j]) sdim_node[j] += P.dims[order[pj]];
  auto merge_pass = [&](double rx, int max_f) {
    for (int j = 0; j < n; ++j) {
      const size_t myNrParents = st[j].size();
      size_t myNrFrontals = 1;     // all frontal variables gathered so far
      size_t refNrFrontals = 1;    // ... counting only the reference's merges (its rule is evaluated on its own count,
                                   //     so that every reference clique survives whole inside a relaxed one)
      const int64_t sp = sdim_node[j];
      int64_t fp = P.dims[order[j]];
      for (int c = ech_ptr[j]; c < ech_ptr[j + 1]; ++c) {
        const int ch = ech[c];
        bool take = myNrParents + refNrFrontals == st[ch].size();
        ref_merged[ch] = take;
        if (take) refNrFrontals += ref_nfront_of[ch];
        if (!take && rx > 0) {
          const int64_t fc = fdim_of[ch], sc = sdim_of[ch];
          const int64_t extra = fc * (fp + sp - sc);
          take = (double)extra <= rx * (double)(fc * (fc + sc + 1)) && fc + fp <= max_f;
        }
        merged[ch] = take;
        if (take) {
          myNrFrontals += nfront_of[ch];
          fp += fdim_of[ch];
        }
      }
      nfront_of[j] = (int)myNrFrontals;
      ref_nfront_of[j] = (int)refNrFrontals;
      fdim_of[j] = fp;
      sdim_of[j] = sp;
    }
  };
  if (relax >= 0) {
    merge_pass(relax, relax_max_f);
  } else {
    // ---- the library's own choice (gsx_set_amalgamation(h, GSX_AMALGAMATION_AUTO, .)) ------------------------------
    // A candidate tree is priced by how it will run (microseconds, measured on MI355X: profiles/r03_*):
    //  * tree fronts (LDS-class with an all-LDS-class subtree; gsx_internal.h) run dependency-driven, one launch per
    //    tier: a front takes t = 4 + 1.9 children + 0.3 F + 0.04 n; a tier costs the larger of its longest chain of
    //    such fronts and its total front time over the workgroups resident at once (4 a CU up to 67 rows, 1 a CU beyond);
    //    their back-substitution is one launch: 7 per front, chain or total over 1280 workgroups;
    //  * everything else is a level schedule: a level with blocked fronts = gather + three launches (~55) + 0.33 per
    //    pivot of its widest chunk chain + its flops at ~10 TFLOP/s, + ~22 for its back-substitution; LDS fronts
    //    above blocked ones ~25 + 1.0 per pivot of the largest + flops at ~2 TFLOP/s.
    double best_cost = 0;
    double best_rx = 0;
    int best_mf = 128;
    bool first = true;
    std::vector<int> lvl(n), bigF, smallF;
    std::vector<double> bigFl, smallFl;
    // per node: what its (non-merged) child clusters hand up — folded into the cluster's top node
    std::vector<char> a_tree(n);        // all child clusters are tree fronts or leaf-kernel cliques
    std::vector<int> a_nch(n), a_subn(n), a_depth(n);
    std::vector<float> a_cp0(n), a_cp1(n);  // longest chain below, in tier 0 / tier 1
    const double cand_rx[] = {0.0, 0.125, 0.25, 0.5, 1.0, 2.0};
    const int cand_mf[] = {16, 24, 32, 48, 64, 96, 128, 160};
    for (double rx : cand_rx)
      for (int mf : cand_mf) {
        if (rx == 0.0 && mf != cand_mf[0]) continue;
        merge_pass(rx, mf);
        bigF.clear(), smallF.clear(), bigFl.clear(), smallFl.clear();
        std::fill(lvl.begin(), lvl.end(), 0);
        std::fill(a_tree.begin(), a_tree.end(), 1);
        std::fill(a_nch.begin(), a_nch.end(), 0);
        std::fill(a_subn.begin(), a_subn.end(), 0);
        std::fill(a_depth.begin(), a_depth.end(), 0);
        std::fill(a_cp0.begin(), a_cp0.end(), 0.f);
        std::fill(a_cp1.begin(), a_cp1.end(), 0.f);
        int nl = 0;
        double work[2] = {0, 0}, chain[2] = {0, 0}, n_tree = 0;
        int bs_depth = 0;
        for (int j = 0; j < n; ++j) {  // children before parents: everything of a cluster top is final when j is reached
          const int ep = eparent[j];
          if (merged[j]) {             // a merged node hands what its children gave it to the cluster's top
            if (ep >= 0) {
              lvl[ep] = std::max(lvl[ep], lvl[j]);
              a_tree[ep] = a_tree[ep] && a_tree[j];
              a_nch[ep] += a_nch[j];
              a_subn[ep] = std::max(a_subn[ep], a_subn[j]);
              a_depth[ep] = std::max(a_depth[ep], a_depth[j]);
              a_cp0[ep] = std::max(a_cp0[ep], a_cp0[j]);
              a_cp1[ep] = std::max(a_cp1[ep], a_cp1[j]);
            }
            continue;
          }
          const int l = lvl[j];
          const double F = (double)fdim_of[j], s1 = (double)sdim_of[j] + 1.0;
          const int nn = (int)(fdim_of[j] + sdim_of[j] + 1);
          const double fl = F * F * F / 3.0 + F * F * s1 + F * s1 * s1;
          const bool blocked = nn > kSmallMaxN;
          const bool leafk = !blocked && a_nch[j] == 0 && fdim_of[j] <= kLeafMaxF;   // the leaf kernel: one launch for all
          const bool tree = !blocked && !leafk && a_tree[j] && std::max(nn, a_subn[j]) <= kSmallMaxN;
          if (tree) {
            const int subn = std::max(nn, a_subn[j]);
            const int tier = subn <= 67 ? 0 : 1;
            const double t = 4.0 + 1.9 * a_nch[j] + 0.3 * F + 0.04 * nn;
            const double cp = t + (tier == 0 ? a_cp0[j] : a_cp1[j]);   // (a tier-1 front's tier-0 children ran in the launch before)
            work[tier] += t;
            chain[tier] = std::max(chain[tier], cp);
            n_tree += 1;
            const int depth = a_depth[j] + 1;
            bs_depth = std::max(bs_depth, depth);
            if (ep >= 0) {
              a_subn[ep] = std::max(a_subn[ep], subn);
              a_depth[ep] = std::max(a_depth[ep], depth);
              if (tier == 0) a_cp0[ep] = std::max(a_cp0[ep], (float)cp);
              else a_cp1[ep] = std::max(a_cp1[ep], (float)cp);
            }
          } else if (!leafk) {
            if (l >= nl) {
              nl = l + 1;
              bigF.resize(nl, 0), smallF.resize(nl, 0), bigFl.resize(nl, 0.0), smallFl.resize(nl, 0.0);
            }
            if (blocked) {
              bigF[l] = std::max(bigF[l], (int)fdim_of[j]);
              bigFl[l] += fl;
            } else {
              smallF[l] = std::max(smallF[l], (int)fdim_of[j]);
              smallFl[l] += fl;
            }
            if (ep >= 0) a_tree[ep] = 0;
          }
          if (ep >= 0) {
            a_nch[ep] += 1;
            if (!tree && !leafk) lvl[ep] = std::max(lvl[ep], l + 1);   // (the upper levels: gsx_internal.h)
          }
        }
        double cost = 40.0;  // (the leaf launches)
        for (int l = 0; l < nl; ++l) {
          if (bigF[l]) cost += 77.0 + 0.33 * bigF[l] + bigFl[l] / 1e7;
          else if (smallF[l]) cost += 50.0 + 1.0 * smallF[l] + smallFl[l] / 2e6;   // (beside blocked fronts they ride in their launches)
        }
        if (n_tree > 0) {
          cost += std::max(chain[0], work[0] / 1024.0) + 10.0;
          cost += std::max(chain[1], work[1] / 256.0) + 10.0;
          cost += std::max(7.0 * bs_depth, 7.0 * n_tree / 1280.0) + 10.0;
        }
        if (std::getenv("GSX_AMAL_TRACE"))
          fprintf(stderr, "[amalgamation] relax %.3f maxF %3d: cost %.0f us (levels %d, tree fronts %.0f, tiers %.0f|%.0f / %.0f|%.0f, depth %d)\n",
                  rx, mf, cost, nl, n_tree, chain[0], work[0] / 1024, chain[1], work[1] / 256, bs_depth);
        if (first || cost < best_cost) {
          first = false;
          best_cost = cost;
          best_rx = rx;
          best_mf = mf;
        }
      }
    merge_pass(best_rx, best_mf);
    S.relax = best_rx;
    S.relax_max_f = best_mf;
  }
  // top node of the cluster containing each node
  std::vector<int> top(n);
  for (int j = n - 1; j >= 0; --j) top[j] = merged[j] ? top[eparent[j]] : j;
  // fronts numbered by ascending top position (children before parents)
  std::vector<int> front_of_top(n, -1);
  int nfr = 0;
  for (int j = 0; j < n; ++j)
    if (!merged[j]) front_of_top[j] = nfr++;
  S.n_fronts = nfr;
  S.front_of_var.assign(n, -1);
  std::vector<int> nfv(nfr, 0);
  for (int j = 0; j < n; ++j) {
    const int fr = front_of_top[top[j]];
    S.front_of_var[order[j]] = fr;
    nfv[fr]++;
  }
  S.nfrontal_vars = nfv;
  S.parent.assign(nfr, -1);
  for (int j = 0; j < n; ++j)
    if (!merged[j] && eparent[j] != -1) S.parent[front_of_top[j]] = front_of_top[top[eparent[j]]];
  // children of fronts in the reference's order: walk the cluster's nodes; a non-merged etree child
  // of any node of the cluster is a child cluster.  Order: for the top node, children in attachment
  // order with merged ones replaced by their own (recursively expanded) children.
  S.child_ptr.assign(nfr + 1, 0);
  {
    std::vector<std::vector<int>> kids(nfr);
    // expanded child list per node, built bottom-up
    std::vector<std::vector<int>> expanded(n);
    for (int j = 0; j < n; ++j) {
      std::vector<int>& ex = expanded[j];
      for (int c = ech_ptr[j]; c < ech_ptr[j + 1]; ++c) {
        const int ch = ech[c];
        if (merged[ch]) {
          ex.insert(ex.end(), expanded[ch].begin(), expanded[ch].end());
          std::vector<int>().swap(expanded[ch]);
        } else {
          ex.push_back(front_of_top[ch]);
        }
      }
      if (!merged[j]) {
        kids[front_of_top[j]] = ex;
        std::vector<int>().swap(ex);
      }
    }
    for (int f = 0; f < nfr; ++f) S.child_ptr[f + 1] = S.child_ptr[f] + (int)kids[f].size();
    S.children.resize(S.child_ptr[nfr]);
    for (int f = 0; f < nfr; ++f) std::copy(kids[f].begin(), kids[f].end(), S.children.begin() + S.child_ptr[f]);
  }
  clk.mark("supernodes + amalgamation");
  // ---- per-front variable lists (frontals by position, then separator = struct of the top) -------
  S.fvar_ptr.assign(nfr + 1, 0);
  {
    std::vector<int> top_of_front(nfr);
    for (int j = 0; j < n; ++j)
      if (!merged[j]) top_of_front[front_of_top[j]] = j;
    for (int f = 0; f < nfr; ++f) S.fvar_ptr[f + 1] = S.fvar_ptr[f] + nfv[f] + (int)st[top_of_front[f]].size();
    S.fvars.resize(S.fvar_ptr[nfr]);
    std::vector<int> fill(S.fvar_ptr.begin(), S.fvar_ptr.end() - 1);
    for (int j = 0; j < n; ++j) {  // ascending position => frontals sorted by position
      const int f = front_of_top[top[j]];
      S.fvars[fill[f]++] = order[j];
    }
    for (int f = 0; f < nfr; ++f)
      for (int pj : st[top_of_front[f]]) S.fvars[fill[f]++] = order[pj];
  }
  std::vector<std::vector<int>>().swap(st);
  clk.mark("front variable lists");
  // ---- dims, arena layout, levels ---------------------------------------------------------------------
  S.F.assign(nfr, 0);
  S.S.assign(nfr, 0);
  S.N.assign(nfr, 0);
  S.off.assign(nfr + 1, 0);
  S.level.assign(nfr, 0);
  for (int f = 0; f < nfr; ++f) {
    int F = 0, Sd = 0;
    for (int k = 0; k < nfv[f]; ++k) F += P.dims[S.fvars[S.fvar_ptr[f] + k]];
    for (int k = S.fvar_ptr[f] + nfv[f]; k < S.fvar_ptr[f + 1]; ++k) Sd += P.dims[S.fvars[k]];
    S.F[f] = F;
    S.S[f] = Sd;
    S.N[f] = F + Sd + 1;
    const int64_t nn = (int64_t)S.N[f] * S.N[f];
    const double s1 = Sd + 1.0;
    S.flops += (double)F * F * F / 3.0 + (double)F * F * s1 + (double)F * s1 * s1;
    S.front_bytes += 8.0 * (double)nn;
    S.lpanel_bytes += 8.0 * (double)F * S.N[f];
    S.max_F = std::max<int64_t>(S.max_F, F);
    S.max_rows = std::max<int64_t>(S.max_rows, F + Sd);
  }
  for (int f = 0; f < nfr; ++f)  // children have smaller ids
    if (S.parent[f] >= 0) S.level[S.parent[f]] = std::max(S.level[S.parent[f]], S.level[f] + 1);
  clk.mark("dims, arena, levels");
  // ---- sharding: cap + subtrees dealt to the ranks (proportional mapping of the assembly tree) -------------------
  // sub[f] = flop estimate of f's whole subtree.  Fronts with sub[f] above a share of the total form the cap (an
  // upward-closed set: a parent's subtree contains its children's); what hangs below the cap is a forest of independent
  // subtrees, dealt to the ranks largest first onto the least loaded rank.  The share is chosen among a few candidates
  // to minimise (weighted) cap + heaviest rank.  All ranks compute the same partition from the same inputs.
  S.owner.assign(nfr, 0);
  std::vector<char> cap(nfr, 0);
  if (S.shard_world > 1) {
    std::vector<double> cost(nfr), sub(nfr);
    double total = 0;
    for (int f = 0; f < nfr; ++f) {
      const double F = S.F[f], s1 = S.S[f] + 1.0;
      cost[f] = F * F * F / 3.0 + F * F * s1 + F * s1 * s1 + 2000.0;  // (+ a constant: no front is free)
      sub[f] = cost[f];
    }
    for (int f = 0; f < nfr; ++f) {
      sub[f] += 0.0;
      if (S.parent[f] >= 0) sub[S.parent[f]] += sub[f];
      else total += sub[f];
    }
    // (children have smaller ids, so sub[f] is complete when f is reached as a child above)
    std::vector<int> roots, assign(nfr, -1), best_assign;
    std::vector<char> best_cap;
    double best = -1, best_capc = 0;
    std::vector<double> load(S.shard_world);
    for (double alpha : {0.51, 0.71, 1.0, 1.41, 2.0, 2.83, 4.0, 5.66, 8.0, 16.0, 32.0}) {  // (> 1/2: at least the root is cap, the work is split)
      const double tau = total / (S.shard_world * alpha);
      double capc = 0;
      roots.clear();
      for (int f = 0; f < nfr; ++f) {
        cap[f] = sub[f] > tau;
        if (cap[f]) capc += cost[f];
      }
      for (int f = 0; f < nfr; ++f)
        if (!cap[f] && (S.parent[f] < 0 || cap[S.parent[f]])) roots.push_back(f);
      std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return sub[a] > sub[b]; });
      std::fill(load.begin(), load.end(), 0.0);
      for (int r : roots) {
        int k = 0;
        for (int q = 1; q < S.shard_world; ++q)
          if (load[q] < load[k]) k = q;
        assign[r] = k;
        load[k] += sub[r];
      }
      const double worst = *std::max_element(load.begin(), load.end());
      // a cap flop costs more time than a subtree flop: the cap's few large fronts run one panel step per launch
      // (latency-bound), the subtrees' many fronts fill the GPU — 2x brackets the measured shares
      const double est = 2.0 * capc + worst;
      if (best < 0 || est < best) {
        best = est;
        best_capc = capc;
        best_cap = cap;
        best_assign = assign;
      }
    }
    cap = best_cap;
    for (int f = nfr - 1; f >= 0; --f) {  // parents first
      if (cap[f]) S.owner[f] = -1;
      else if (S.parent[f] < 0 || cap[S.parent[f]]) S.owner[f] = best_assign[f];
      else S.owner[f] = S.owner[S.parent[f]];
    }
    S.cap_cost = best_capc;
    S.total_cost = total;
    for (int f = 0; f < nfr; ++f) {
      if (S.owner[f] == S.shard_rank) S.own_cost += cost[f];
      S.n_cap += cap[f];
    }
    // every cap level above every subtree level: the cap's assembly (H terms + subtree contributions) is then complete,
    // on each rank for its share, before the first cap front is factored — ONE exchange per factorization
    int lbase = 0;
    for (int f = 0; f < nfr; ++f)
      if (!cap[f]) lbase = std::max(lbase, S.level[f] + 1);
    for (int f = 0; f < nfr; ++f)
      if (cap[f]) S.level[f] = lbase;
    for (int f = 0; f < nfr; ++f)
      if (cap[f] && S.parent[f] >= 0) S.level[S.parent[f]] = std::max(S.level[S.parent[f]], S.level[f] + 1);
    if (S.n_cap) S.cap_level0 = lbase;
  }
  S.scheduled.assign(nfr, 1);
  for (int f = 0; f < nfr; ++f) S.scheduled[f] = S.owner[f] < 0 || S.owner[f] == S.shard_rank;
  S.f_owned.assign(m, 1);
  if (S.shard_world > 1)
    for (int f = 0; f < m; ++f) {
      int first = -1;
      for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q)
        if (first < 0 || S.pos[P.f_vars[q]] < S.pos[first]) first = P.f_vars[q];
      const int o = first < 0 ? -1 : S.owner[S.front_of_var[first]];
      S.f_owned[f] = o < 0 ? S.shard_rank == 0 : o == S.shard_rank;  // (factors of cap variables only: rank 0)
    }
  // ---- hard constraints: which fronts take constraint rows in, how many they hand on -------------------------------
  S.con.assign(nfr, 0);
  S.con_index.assign(nfr, -1);
  S.con_fronts.clear();
  S.con_in.clear();
  S.con_fwd.clear();
  S.con_own_ptr.assign(1, 0);
  S.con_own_jac.clear();
  S.con_own_m.clear();
  S.con_own_col_ptr.assign(1, 0);
  S.con_own_cols.clear();
  S.con_child_ptr.assign(1, 0);
  S.con_child.clear();
  S.con_fwd_map_ptr.clear();
  S.con_fwd_map.clear();
  if (!P.con_factor.empty()) {
    if (S.shard_world > 1) {
      err = "hard constraints (zero-sigma rows) are not supported on a sharded handle";
      return GSX_E_INVALID;
    }
    std::vector<std::vector<int>> own(nfr);  // front -> constraint rows (indices into P.con_*)
    for (size_t i = 0; i < P.con_factor.size(); ++i) {
      const int f = P.con_factor[i];
      int first = -1;
      for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q)
        if (first < 0 || S.pos[P.f_vars[q]] < S.pos[first]) first = P.f_vars[q];
      own[S.front_of_var[first]].push_back((int)i);
    }
    // Symbolic pass, children before parents.  A row is a set of variables; in a front the frontal variables are taken
    // in order: of the rows that hold the variable, as many as it has scalars become pivots (generic rank), the others
    // take the union of those rows' variables without it.  What is left goes — all rows of the front together, each with
    // the union of their variables — to the front where the first of those variables is frontal: an ancestor whose front
    // holds all of them (they are a clique of the filled graph), possibly far up; the fronts in between never see the rows
    // (Constrained::QR leaves a row without an entry in the column alone, NoiseModel.cpp:566-590).
    struct Arrival {
      int src;                 // constrained-front index of the sender
      int rows;
      std::vector<int> vars;   // sorted by position
    };
    std::vector<std::vector<Arrival>> arrive(nfr);
    std::vector<int> loc(n, -1);
    for (int f = 0; f < nfr; ++f) {  // children have smaller ids
      if (own[f].empty() && arrive[f].empty()) continue;
      std::vector<std::vector<int>> rows;   // variable sets (by position order)
      for (int i : own[f]) {
        const int fac = P.con_factor[i];
        std::vector<int> vs(P.f_vars.begin() + P.f_key_ptr[fac], P.f_vars.begin() + P.f_key_ptr[fac + 1]);
        std::sort(vs.begin(), vs.end(), [&](int a, int b) { return S.pos[a] < S.pos[b]; });
        rows.push_back(std::move(vs));
      }
      for (const Arrival& a : arrive[f])
        for (int r = 0; r < a.rows; ++r) rows.push_back(a.vars);
      const int in = (int)rows.size();
      if (in > 1024) {
        err = "more than 1024 hard-constraint rows meet in one clique";
        return GSX_E_INVALID;
      }
      std::vector<char> used(in, 0);
      auto by_pos = [&](int a, int b) { return S.pos[a] < S.pos[b]; };
      for (int k = 0; k < nfv[f]; ++k) {
        const int v = S.fvars[S.fvar_ptr[f] + k];
        std::vector<int> touching, uni;
        for (int r = 0; r < in; ++r)
          if (!used[r] && std::binary_search(rows[r].begin(), rows[r].end(), v, by_pos)) {
            touching.push_back(r);
            std::vector<int> tmp;
            std::set_union(uni.begin(), uni.end(), rows[r].begin(), rows[r].end(), std::back_inserter(tmp), by_pos);
            uni.swap(tmp);
          }
        if (touching.empty()) continue;
        uni.erase(std::lower_bound(uni.begin(), uni.end(), v, by_pos));
        const int t = std::min<int>((int)touching.size(), P.dims[v]);
        for (int q = 0; q < (int)touching.size(); ++q) {
          if (q < t) used[touching[q]] = 1;
          else rows[touching[q]] = uni;
        }
      }
      int fwd = 0;
      std::vector<int> uni;
      for (int r = 0; r < in; ++r) {
        if (used[r] || rows[r].empty()) continue;   // (an empty row says 0 = d: nothing left to enforce)
        ++fwd;
        std::vector<int> tmp;
        std::set_union(uni.begin(), uni.end(), rows[r].begin(), rows[r].end(), std::back_inserter(tmp), by_pos);
        uni.swap(tmp);
      }
      const int me = (int)S.con_fronts.size();
      S.con[f] = 1;
      S.con_index[f] = me;
      S.con_fronts.push_back(f);
      S.con_in.push_back(in);
      S.con_fwd.push_back(fwd);
      int o = 0;
      for (int k = S.fvar_ptr[f]; k < S.fvar_ptr[f + 1]; ++k) {
        loc[S.fvars[k]] = o;
        o += P.dims[S.fvars[k]];
      }
      for (int i : own[f]) {
        const int fac = P.con_factor[i];
        S.con_own_jac.push_back(P.f_jac_off[fac] + P.con_row[i]);
        S.con_own_m.push_back(P.f_rows[fac]);
        for (int q = P.f_key_ptr[fac]; q < P.f_key_ptr[fac + 1]; ++q)
          for (int d = 0; d < P.dims[P.f_vars[q]]; ++d) S.con_own_cols.push_back(loc[P.f_vars[q]] + d);
        S.con_own_cols.push_back(S.N[f] - 1);
        S.con_own_col_ptr.push_back((int)S.con_own_cols.size());
      }
      S.con_own_ptr.push_back((int)S.con_own_jac.size());
      for (const Arrival& a : arrive[f]) S.con_child.push_back(a.src);
      S.con_child_ptr.push_back((int)S.con_child.size());
      // where the leftover rows go, and the map of this front's separator + rhs rows into that front's rows
      S.con_fwd_map_ptr.push_back((int)S.con_fwd_map.size());
      if (fwd > 0) {
        const int dest = S.front_of_var[uni.front()];
        arrive[dest].push_back(Arrival{me, fwd, uni});
        std::vector<int> dloc(0);
        int od = 0;
        std::vector<std::pair<int, int>> dvars;   // (variable, offset) of the destination front
        for (int k = S.fvar_ptr[dest]; k < S.fvar_ptr[dest + 1]; ++k) {
          dvars.push_back({S.fvars[k], od});
          od += P.dims[S.fvars[k]];
        }
        for (int k = S.fvar_ptr[f] + nfv[f]; k < S.fvar_ptr[f + 1]; ++k) {
          const int u = S.fvars[k];
          int at = -1;
          if (std::binary_search(uni.begin(), uni.end(), u, by_pos))
            for (const auto& dv : dvars)
              if (dv.first == u) at = dv.second;
          for (int d = 0; d < P.dims[u]; ++d) S.con_fwd_map.push_back(at < 0 ? -1 : at + d);
        }
        S.con_fwd_map.push_back(S.N[dest] - 1);   // the rhs
      }
    }
    S.con_fwd_map_ptr.push_back((int)S.con_fwd_map.size());
  }
  // Size class of a front: 0 = leaf kernel (no children, few frontal scalars: only the n x F panel lives in LDS),
  // 1 = small (whole front in LDS), 2 = big (blocked path in HBM; every cap front, whatever its size, because its
  // assembled state must exist in HBM for the exchange).  A leaf is LEAN when its parent is big and every
  // separator block fits a 16 x 16 matrix-core tile: its Schur complement is never materialised — the parent's
  // gather forms -L21 L21' block by block from the L panel — so it owns only n x F doubles of the arena and may
  // have any number of rows.
  S.lean.assign(nfr, 0);
  S.cls.assign(nfr, 1);
  S.med.assign(nfr, 0);
  // MEASURED AND SWITCHED OFF (GSX_MEDIUM=1 turns it on; tests/test_gpu_parity.py keeps the path honest): a medium front
  // takes ~53 us in its one workgroup (pose3_100k, mean n = 170, F = 45: children 13, panel 15, trailing update 18) and
  // the medium fronts of a pose graph form a chain a dozen deep, while the level schedule of the blocked path gives up
  // only six of its seventeen levels for them (the top fronts are wide in F): pose3_100k 2.94 -> 3.46 ms, pose2_100k
  // 1.65 -> 1.99 ms with the medium tier on.
  const bool med_off = std::getenv("GSX_MEDIUM") == nullptr;
  for (int f = 0; f < nfr; ++f) {   // medium fronts first: a leaf's leanness depends on its parent's class
    if (cap[f] || S.con[f] || S.N[f] <= kSmallMaxN || S.N[f] > kMedMaxN || (int64_t)S.N[f] * S.F[f] > kMedMaxPanel || med_off) continue;
    int leaf_kids = 0;
    for (int c = S.child_ptr[f]; c < S.child_ptr[f + 1]; ++c) {
      const int ch = S.children[c];
      leaf_kids += S.child_ptr[ch + 1] == S.child_ptr[ch] && S.F[ch] <= kLeafMaxF;
    }
    S.med[f] = leaf_kids <= kMedMaxLeafKids;
    S.n_medium += S.med[f];
  }
  std::vector<int64_t> fsize(nfr);
  for (int f = 0; f < nfr; ++f) {
    const bool childless = S.child_ptr[f + 1] == S.child_ptr[f];
    const int par = S.parent[f];
    bool lean = !cap[f] && !S.con[f] && childless && S.F[f] <= kLeafMaxF && S.F[f] > 0 && par >= 0 &&
                ((S.N[par] > kSmallMaxN && !S.med[par]) || cap[par] || S.con[par]) &&
                (int64_t)S.N[f] * S.F[f] <= kLeafMaxPanel;
    if (lean)
      for (int k = S.fvar_ptr[f] + nfv[f]; k < S.fvar_ptr[f + 1]; ++k) lean = lean && P.dims[S.fvars[k]] <= 16;
    S.lean[f] = lean;
    if (lean) S.cls[f] = 0;
    else if ((S.N[f] > kSmallMaxN && !S.med[f]) || cap[f] || S.con[f]) S.cls[f] = 2;
    else if (childless && S.F[f] <= kLeafMaxF) S.cls[f] = 0;
    if (S.cls[f] == 2) S.n_big++; else S.n_small++;
    int64_t sz = lean ? (int64_t)S.N[f] * S.F[f] : (int64_t)S.N[f] * S.N[f];
    if (S.cls[f] == 2) sz = ((sz + 1) & ~int64_t(1)) + (int64_t)S.N[f] * S.F[f];  // + L-panel area (kernels.h)
    fsize[f] = (sz + 1) & ~int64_t(1);  // 16-byte aligned fronts
  }
  // ---- tree fronts (gsx_internal.h): LDS-class fronts whose whole subtree is LDS-class, tiers by the subtree's largest n
  {
    S.tree_bounds.clear();
    S.tree_threads.clear();
    if (const char* e = std::getenv("GSX_TREE_TIERS")) {  // experiments: "45:128,79:256,140:512"; "0" switches the tree kernels off
      int v[2] = {0, 0}, k = 0;
      bool any = false;
      for (const char* q = e;; ++q) {
        if (*q >= '0' && *q <= '9') {
          v[k] = v[k] * 10 + (*q - '0');
          any = true;
        } else if (*q == ':') {
          k = 1;
        } else {
          if (any && v[0] > 0) {
            S.tree_bounds.push_back(std::min(v[0], kSmallMaxN));
            S.tree_threads.push_back(std::max(0, std::min(512, (v[1] + 63) / 64 * 64)));   // (the kernel's launch bound)
          }
          v[0] = v[1] = k = 0;
          any = false;
          if (!*q) break;
        }
      }
    } else {
      // two tiers: fronts of at most 67 rows (four to a CU's 160 KB of LDS) and the rest.  Swept on the 100 000-pose
      // graphs (tools/sweep_tree.sh): every further tier adds its own tail — the last few chains of a launch run alone
      // on the chip — and costs more than the LDS it frees: 45/79/140 +16 %, 53/98/140 +13 %, one tier +67 %.
      S.tree_bounds = {67, kSmallMaxN};
      S.tree_threads = {256, 512};
    }
    S.tree_tier.assign(nfr, -1);
    S.tree_up.assign(nfr, -1);
    S.tree_npend.assign(nfr, 0);
    const int nt = (int)S.tree_bounds.size();
    // (a last tier beyond the bounds holds the MEDIUM fronts and everything above them in their subtrees)
    std::vector<int> subn(nfr, 0);
    for (int f = 0; f < nfr && nt > 0; ++f) {  // children have smaller ids
      if (!S.scheduled[f] || S.cls[f] != 1) continue;
      bool ok = true;
      int mx = S.med[f] ? kSmallMaxN + 1 : S.N[f];   // (any medium front: the tier after the last bound)
      for (int c = S.child_ptr[f]; c < S.child_ptr[f + 1] && ok; ++c) {
        const int ch = S.children[c];
        if (S.cls[ch] == 0 && !S.lean[ch]) continue;  // done by the leaf launch before the tree kernels
        ok = S.tree_tier[ch] >= 0;
        mx = std::max(mx, subn[ch]);
      }
      if (!ok || (mx > S.tree_bounds.back() && mx <= kSmallMaxN)) continue;
      subn[f] = mx;
      int t = 0;
      while (t < nt && S.tree_bounds[t] < mx) ++t;
      S.tree_tier[f] = t;    // t == nt: the medium tier
    }
    for (int f = 0; f < nfr; ++f) {
      const int p = S.parent[f];
      if (S.tree_tier[f] >= 0 && p >= 0 && S.tree_tier[p] == S.tree_tier[f]) {
        S.tree_up[f] = p;
        S.tree_npend[p]++;
      }
    }
  }
  // ---- the upper levels (gsx_internal.h) ----
  S.ulevel.assign(nfr, -1);
  S.n_ulevels = 0;
  S.cap_ulevel0 = -1;
  {
    auto upper = [&](int f) { return S.scheduled[f] && S.cls[f] != 0 && S.tree_tier[f] < 0; };
    for (int f = 0; f < nfr; ++f)
      if (upper(f)) S.ulevel[f] = 0;
    for (int f = 0; f < nfr; ++f) {   // children have smaller ids
      const int p = S.parent[f];
      if (S.ulevel[f] >= 0 && p >= 0 && S.ulevel[p] >= 0) S.ulevel[p] = std::max(S.ulevel[p], S.ulevel[f] + 1);
    }
    if (S.n_cap) {   // every cap level above every subtree level: ONE exchange per factorization (see the sharding above)
      int ubase = 0;
      for (int f = 0; f < nfr; ++f)
        if (!cap[f] && S.ulevel[f] >= 0) ubase = std::max(ubase, S.ulevel[f] + 1);
      for (int f = 0; f < nfr; ++f)
        if (cap[f]) S.ulevel[f] = ubase;
      for (int f = 0; f < nfr; ++f)
        if (cap[f] && S.parent[f] >= 0) S.ulevel[S.parent[f]] = std::max(S.ulevel[S.parent[f]], S.ulevel[f] + 1);
      S.cap_ulevel0 = ubase;
    }
    for (int f = 0; f < nfr; ++f) S.n_ulevels = std::max(S.n_ulevels, S.ulevel[f] + 1);
  }
  {
    // A level's LDS fronts cost one launch bound by the latency of its slowest front (60-100 us for a 100-140 row front:
    // a barrier per pivot) however few they are.  Where the level has blocked (big) fronts anyway and only a handful of
    // LDS fronts, those join the blocked path: no launch of their own, and the level's panel-step chain is set by the
    // big fronts' wider frontal blocks in any case (measured on the 100 000-pose graphs, limit swept 256 / 1024 / 4096:
    // 1024 is best — 7.79 -> 7.13 ms and 4.56 -> 3.69 ms per iteration).
    // (levels = the upper levels: tree fronts and leaf-kernel cliques cost no launch of a level)
    std::vector<int> n_lds(S.n_ulevels + 1, 0), n_blk(S.n_ulevels + 1, 0);
    for (int f = 0; f < nfr; ++f) {
      if (S.ulevel[f] < 0) continue;
      if (S.cls[f] == 1) n_lds[S.ulevel[f]]++;
      else if (S.cls[f] == 2) n_blk[S.ulevel[f]]++;
    }
    for (int f = 0; f < nfr; ++f) {
      const int l = S.ulevel[f];
      if (l < 0 || S.cls[f] != 1 || n_blk[l] == 0 || n_lds[l] > 1024) continue;
      S.cls[f] = 2;
      if (S.med[f]) {
        S.med[f] = 0;
        S.n_medium--;
      }
      S.n_big++;
      S.n_small--;
      int64_t sz = (int64_t)S.N[f] * S.N[f];
      sz = ((sz + 1) & ~int64_t(1)) + (int64_t)S.N[f] * S.F[f];
      fsize[f] = (sz + 1) & ~int64_t(1);
    }
  }
  {
    // arena layout: the subtree fronts, then the cap fronts in one contiguous block (the exchange buffer)
    int64_t cursor = 0;
    for (int f = 0; f < nfr; ++f)
      if (!cap[f]) {
        S.off[f] = cursor;
        cursor += fsize[f];
      }
    S.cap_begin = cursor;
    for (int f = 0; f < nfr; ++f)
      if (cap[f]) {
        S.off[f] = cursor;
        cursor += fsize[f];
      }
    S.cap_end = cursor;
    S.off[nfr] = cursor;
    S.arena_size = cursor;
  }
  S.n_levels = 0;
  for (int f = 0; f < nfr; ++f) S.n_levels = std::max(S.n_levels, S.level[f] + 1);
  clk.mark("sharding");
  // ---- H panels and assembly terms (per variable) -----------------------------------------------------
  S.h_off.assign(n + 1, 0);
  S.h_rows.assign(n, 0);
  S.hmap_ptr.assign(n + 1, 0);
  S.h_loc.assign(n, 0);
  S.term_ptr.assign(n + 1, 0);
  std::vector<int> nb_ptr(n + 1, 0), nb;  // later neighbours per variable, ascending position
  {
    std::vector<int> tmp;
    std::fill(stamp.begin(), stamp.end(), -1);
    for (int v = 0; v < n; ++v) {
      tmp.clear();
      for (int k = vf_ptr[v]; k < vf_ptr[v + 1]; ++k) {
        const int f = vf[k];
        for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q) {
          const int u = P.f_vars[q];
          if (S.pos[u] > S.pos[v] && stamp[u] != v) {
            stamp[u] = v;
            tmp.push_back(u);
          }
        }
      }
      std::sort(tmp.begin(), tmp.end(), [&](int a, int b) { return S.pos[a] < S.pos[b]; });
      nb_ptr[v + 1] = nb_ptr[v] + (int)tmp.size();
      nb.insert(nb.end(), tmp.begin(), tmp.end());
      int rows = P.dims[v] + 1;
      for (int u : tmp) rows += P.dims[u];
      S.h_rows[v] = rows;
      S.h_off[v + 1] = S.h_off[v] + (((int64_t)rows * P.dims[v] + 1) & ~int64_t(1));
      S.hmap_ptr[v + 1] = S.hmap_ptr[v] + rows;
    }
    S.h_size = S.h_off[n];
  }
  // terms
  {
    std::vector<int> rowoff(n, -1);  // panel row offset of neighbour u in the current variable's panel
    int64_t nterms = 0;
    for (int v = 0; v < n; ++v) {
      int64_t c = 0;
      for (int k = vf_ptr[v]; k < vf_ptr[v + 1]; ++k) {
        const int f = vf[k];
        if (!S.f_owned[f]) continue;
        c += 2;  // diagonal + rhs
        for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q)
          if (S.pos[P.f_vars[q]] > S.pos[v]) ++c;
      }
      nterms += c;
      S.term_ptr[v + 1] = nterms;
    }
    S.t_jac.resize(nterms);
    S.t_m.resize(nterms);
    S.t_colA.resize(nterms);
    S.t_colB.resize(nterms);
    S.t_dB.resize(nterms);
    S.t_dst.resize(nterms);
    for (int v = 0; v < n; ++v) {
      int r = P.dims[v];
      for (int k = nb_ptr[v]; k < nb_ptr[v + 1]; ++k) {
        rowoff[nb[k]] = r;
        r += P.dims[nb[k]];
      }
      int64_t t = S.term_ptr[v];
      for (int k = vf_ptr[v]; k < vf_ptr[v + 1]; ++k) {
        const int f = vf[k];
        if (!S.f_owned[f]) continue;
        int colA = 0, col = 0;
        for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q) {
          if (P.f_vars[q] == v) colA = col;
          col += P.dims[P.f_vars[q]];
        }
        auto put = [&](int colB, int dB, int dst) {
          S.t_jac[t] = P.f_jac_off[f];
          S.t_m[t] = P.f_rows[f];
          S.t_colA[t] = colA;
          S.t_colB[t] = colB;
          S.t_dB[t] = dB;
          S.t_dst[t] = dst;
          ++t;
        };
        put(colA, P.dims[v], 0);
        col = 0;
        for (int q = P.f_key_ptr[f]; q < P.f_key_ptr[f + 1]; ++q) {
          const int u = P.f_vars[q];
          if (S.pos[u] > S.pos[v]) put(col, P.dims[u], rowoff[u]);
          col += P.dims[u];
        }
        put(col, 1, S.h_rows[v] - 1);
      }
    }
  }
  clk.mark("H panels and terms");
  // ---- scalar row maps ----------------------------------------------------------------------------------
  S.cmap_ptr.assign(nfr + 1, 0);
  S.gidx_ptr.assign(nfr + 1, 0);
  for (int f = 0; f < nfr; ++f) {
    S.cmap_ptr[f + 1] = S.cmap_ptr[f] + S.S[f] + 1;
    S.gidx_ptr[f + 1] = S.gidx_ptr[f] + S.F[f] + S.S[f];
  }
  S.cmap.assign(S.cmap_ptr[nfr], 0);
  S.gidx.assign(S.gidx_ptr[nfr], 0);
  S.hmap.assign(S.hmap_ptr[n], 0);
  {
    std::vector<int> loc(n, -1);
    for (int f = 0; f < nfr; ++f) {
      int o = 0;
      int64_t g = S.gidx_ptr[f];
      for (int k = S.fvar_ptr[f]; k < S.fvar_ptr[f + 1]; ++k) {
        const int v = S.fvars[k];
        loc[v] = o;
        for (int d = 0; d < P.dims[v]; ++d) S.gidx[g++] = P.tan_off[v] + d;
        o += P.dims[v];
      }
      // H panel maps of the frontal variables
      for (int k = 0; k < nfv[f]; ++k) {
        const int v = S.fvars[S.fvar_ptr[f] + k];
        S.h_loc[v] = loc[v];
        int64_t h = S.hmap_ptr[v];
        for (int d = 0; d < P.dims[v]; ++d) S.hmap[h++] = loc[v] + d;
        for (int q = nb_ptr[v]; q < nb_ptr[v + 1]; ++q) {
          const int u = nb[q];
          if (loc[u] < 0) {
            err = "internal: neighbour missing from front";
            return GSX_E_INVALID;
          }
          for (int d = 0; d < P.dims[u]; ++d) S.hmap[h++] = loc[u] + d;
        }
        S.hmap[h++] = S.N[f] - 1;
      }
      // update-row maps of the children
      for (int c = S.child_ptr[f]; c < S.child_ptr[f + 1]; ++c) {
        const int ch = S.children[c];
        int64_t cm = S.cmap_ptr[ch];
        for (int k = S.fvar_ptr[ch] + nfv[ch]; k < S.fvar_ptr[ch + 1]; ++k) {
          const int u = S.fvars[k];
          if (loc[u] < 0) {
            err = "internal: child separator variable missing from parent front";
            return GSX_E_INVALID;
          }
          for (int d = 0; d < P.dims[u]; ++d) S.cmap[cm++] = loc[u] + d;
        }
        S.cmap[cm++] = S.N[f] - 1;
      }
      for (int k = S.fvar_ptr[f]; k < S.fvar_ptr[f + 1]; ++k) loc[S.fvars[k]] = -1;
    }
  }
  // ---- choleskyPartial's conditioning test, clique by clique of the REFERENCE tree (cholesky.cpp:145-158: exponent gap
  //      < 12 between the last two pivots of a clique's frontal block; > -12 for a single pivot).  A relaxed front holds
  //      several reference cliques; the pivots are the same numbers wherever a clique is eliminated, so the test is a
  //      list of (second-to-last, last) diagonal entries of L per reference clique, read after the factorization.
  {
    std::vector<int> rtop(n), prev_in_clique(n, -1);
    for (int j = n - 1; j >= 0; --j) rtop[j] = ref_merged[j] ? rtop[eparent[j]] : j;
    for (int j = 0; j < n; ++j)  // ascending position: the last write is the largest position below the top
      if (rtop[j] != j) prev_in_clique[rtop[j]] = j;
    for (int t = 0; t < n; ++t) {
      if (rtop[t] != t) continue;
      const int v = order[t], f = S.front_of_var[v];
      if (!S.scheduled.empty() && !S.scheduled[f]) continue;
      // (a clique with constraint rows goes through EliminateQR in the reference: no conditioning test there, and the
      //  rewritten front mixes unit pivots with those of the soft rows)
      if (!S.con.empty() && S.con[f]) continue;
      const int d = P.dims[v];
      const int64_t ld = S.N[f];
      const int c1 = S.h_loc[v] + d - 1;
      int c2 = -1;
      if (d >= 2) c2 = c1 - 1;
      else if (prev_in_clique[t] >= 0) {
        const int u = order[prev_in_clique[t]];
        c2 = S.h_loc[u] + P.dims[u] - 1;
      }
      S.cond_last.push_back(S.off[f] + c1 + c1 * ld);
      S.cond_prev.push_back(c2 >= 0 ? S.off[f] + c2 + c2 * ld : -1);
      S.cond_front.push_back(f);
    }
  }
  clk.mark("row maps");
  // ---- side leaves (gsx_internal.h): lean leaves under blocked parents above the first blocked level ----
  S.side.assign(nfr, 0);
  S.side_level0 = -1;
  {
    static const bool side_off = std::getenv("GSX_SIDE_OFF") != nullptr;
    int l0 = -1;
    for (int f = 0; f < nfr; ++f)
      if (S.cls[f] == 2 && S.scheduled[f] && (l0 < 0 || S.ulevel[f] < l0)) l0 = S.ulevel[f];
    int64_t n_side = 0, n_lean = 0;
    if (l0 >= 0 && !side_off && S.shard_world == 1)
      for (int f = 0; f < nfr; ++f)
        if (S.lean[f] && S.scheduled[f]) {
          ++n_lean;
          if (S.ulevel[S.parent[f]] > l0) {
            S.side[f] = 1;
            ++n_side;
          }
        }
    // (worth a second queue only when it moves real work: a tenth of the lean leaves, and thousands of them)
    if (n_side >= 4096 && n_side * 10 >= n_lean) S.side_level0 = l0;
    else std::fill(S.side.begin(), S.side.end(), 0);
  }
  // ---- schedule: by level; inside a level: leaf-kernel fronts, other small (LDS) fronts by N, big fronts ----
  auto cls = [&](int f) { return (int)S.cls[f]; };
  S.sched.clear();
  for (int f = 0; f < nfr; ++f)
    if (S.scheduled[f]) S.sched.push_back(f);
  std::stable_sort(S.sched.begin(), S.sched.end(), [&](int a, int b) {
    if (S.level[a] != S.level[b]) return S.level[a] < S.level[b];
    const int ca = cls(a), cb = cls(b);
    if (ca != cb) return ca < cb;
    if (ca == 0 && S.side[a] != S.side[b]) return S.side[a] < S.side[b];   // the side leaves last among the leaves
    if (ca == 0 && S.F[a] != S.F[b]) return S.F[a] < S.F[b];
    return S.N[a] < S.N[b];
  });
  S.lvl_ptr.assign(S.n_levels + 1, 0);
  S.lvl_small_end.assign(S.n_levels, 0);
  S.lvl_leaf_end.assign(S.n_levels, 0);
  for (int f : S.sched) S.lvl_ptr[S.level[f] + 1]++;
  for (int l = 0; l < S.n_levels; ++l) S.lvl_ptr[l + 1] += S.lvl_ptr[l];
  for (int l = 0; l < S.n_levels; ++l) {
    int e = S.lvl_ptr[l];
    while (e < S.lvl_ptr[l + 1] && cls(S.sched[e]) == 0) ++e;
    S.lvl_leaf_end[l] = e;
    while (e < S.lvl_ptr[l + 1] && cls(S.sched[e]) == 1) ++e;
    S.lvl_small_end[l] = e;
  }
  S.leaf_side_begin = S.n_levels ? S.lvl_leaf_end[0] : 0;
  if (S.side_level0 >= 0)
    while (S.leaf_side_begin > S.lvl_ptr[0] && S.side[S.sched[S.leaf_side_begin - 1]]) --S.leaf_side_begin;
  {
    const int nt = (int)S.tree_bounds.size() + 1;   // (+ the medium tier)
    S.tree_start_ptr.assign(nt + 1, 0);
    S.tree_start.clear();
    for (int t = 0; t < nt; ++t) {
      for (int f : S.sched)
        if (S.tree_tier[f] == t && S.tree_npend[f] == 0) S.tree_start.push_back(f);
      S.tree_start_ptr[t + 1] = (int)S.tree_start.size();
    }
  }
  {
    S.usched.clear();
    for (int f = 0; f < nfr; ++f)
      if (S.ulevel[f] >= 0) S.usched.push_back(f);
    std::stable_sort(S.usched.begin(), S.usched.end(), [&](int a, int b) {
      if (S.ulevel[a] != S.ulevel[b]) return S.ulevel[a] < S.ulevel[b];
      if (S.cls[a] != S.cls[b]) return S.cls[a] < S.cls[b];
      return S.N[a] < S.N[b];
    });
    S.ulvl_ptr.assign(S.n_ulevels + 1, 0);
    S.ulvl_small_end.assign(S.n_ulevels, 0);
    for (int f : S.usched) S.ulvl_ptr[S.ulevel[f] + 1]++;
    for (int l = 0; l < S.n_ulevels; ++l) S.ulvl_ptr[l + 1] += S.ulvl_ptr[l];
    for (int l = 0; l < S.n_ulevels; ++l) {
      int e = S.ulvl_ptr[l];
      while (e < S.ulvl_ptr[l + 1] && S.cls[S.usched[e]] == 1) ++e;
      S.ulvl_small_end[l] = e;
    }
  }
  clk.mark("schedule");
  // ---- gather tasks for big parents ------------------------------------------------------------------------
  {
    // A task = one destination block of a big parent at one level; its sources in child order.  Tasks are wanted sorted
    // by (level, destination).  The parents' arena blocks are disjoint, so that order is: levels ascending, inside a level
    // the parents by arena offset, inside a parent the blocks column by column — each parent's contributions are
    // bucketed by (level, block) with a counting sort while they are generated (a global sort of the 48-byte
    // contribution records was 3/4 of the whole analysis at 10^6 landmarks).
    struct Local {
      int key, child, loc, loc2;   // key = level rank << 24 | column block * (nvp + 1) + row block
    };
    struct Src {
      int child, loc, loc2;
    };
    struct LevelOut {
      std::vector<int64_t> dst, ptr;   // per task: destination, first source
      std::vector<int> ld, dims, front;
      std::vector<Src> src;
    };
    std::vector<LevelOut> lout(S.n_ulevels + 1);   // (the upper levels + the side group)
    std::vector<int> parents;
    for (int p = 0; p < nfr; ++p)
      if (S.cls[p] == 2 && S.scheduled[p]) parents.push_back(p);
    std::sort(parents.begin(), parents.end(), [&](int x, int y) { return S.off[x] < S.off[y]; });
    std::vector<Local> loc, loc_sorted;
    std::vector<int> pvar_idx(n, -1), poffv, pdim, cidx, coff, bucket;
    for (int p : parents) {
      // parent-local index and scalar offset of each of its variables (+ the rhs as one more "variable")
      const int nvp = S.fvar_ptr[p + 1] - S.fvar_ptr[p];
      poffv.assign(nvp + 1, 0);
      pdim.assign(nvp + 1, 1);
      int o = 0;
      for (int k = 0; k < nvp; ++k) {
        const int u = S.fvars[S.fvar_ptr[p] + k];
        pvar_idx[u] = k;
        poffv[k] = o;
        pdim[k] = P.dims[u];
        o += P.dims[u];
      }
      poffv[nvp] = S.N[p] - 1;
      // the (at most three) levels its contributions run at: 0 for lean children, the first cap level for a subtree's
      // contribution to a cap front, its own level otherwise
      int lv[3] = {0, S.ulevel[p], S.ulevel[p]};
      if (S.owner[p] < 0) lv[1] = S.cap_ulevel0;
      // (a parent's lean children are all side leaves or none: sidedness is the parent's level)
      if (S.side_level0 >= 0 && S.ulevel[p] > S.side_level0) lv[0] = S.n_ulevels;
      loc.clear();
      for (int ci = S.child_ptr[p]; ci < S.child_ptr[p + 1]; ++ci) {
        const int ch = S.children[ci];
        if (!S.scheduled[ch]) continue;  // another rank's subtree: its contribution arrives with the exchange
        const int Fc = S.F[ch], Nc = S.N[ch];
        cidx.clear();
        coff.clear();
        int co = 0;
        for (int k = S.fvar_ptr[ch] + nfv[ch]; k < S.fvar_ptr[ch + 1]; ++k) {
          const int u = S.fvars[k];
          if (pvar_idx[u] < 0) {
            err = "internal: child separator variable missing from big parent";
            return GSX_E_INVALID;
          }
          cidx.push_back(pvar_idx[u]);
          coff.push_back(co);
          co += P.dims[u];
        }
        cidx.push_back(nvp);
        coff.push_back(co);
        const int nb = (int)cidx.size();
        // product-form sources (lean children) depend on the leaf kernel only: all of them, for every parent level, go
        // into gather group 0, launched once right after the leaves (throughput-bound), which leaves the per-level
        // gathers on the latency-bound chain with the few stored complements of big children
        int rank = S.lean[ch] ? 0 : ((S.owner[p] < 0 && S.owner[ch] >= 0) ? 1 : 2);
        // (two tasks for one destination block must never share a launch: sources whose group is the lean children's
        //  group join their task — one sum per block, in child order)
        if (rank != 0 && lv[rank] == lv[0]) rank = 0;
        for (int a = 0; a < nb; ++a)
          for (int b = a; b < nb; ++b) {  // block (row b, col a), b >= a
            Local c;
            c.key = (rank << 24) | (cidx[a] * (nvp + 1) + cidx[b]);
            c.child = ch;
            if (S.lean[ch]) {
              c.loc = Fc + coff[b];
              c.loc2 = Fc + coff[a];
            } else {
              c.loc = (Fc + coff[b]) + (Fc + coff[a]) * Nc;
              c.loc2 = -1;
            }
            loc.push_back(c);
          }
      }
      for (int k = 0; k < nvp; ++k) pvar_idx[S.fvars[S.fvar_ptr[p] + k]] = -1;
      if (loc.empty()) continue;
      // stable counting sort by key (3 ranks x (nvp + 1)^2 blocks)
      const int nblk = (nvp + 1) * (nvp + 1);
      bucket.assign((size_t)3 * nblk + 1, 0);
      auto slot = [&](int key) { return (key >> 24) * nblk + (key & 0xFFFFFF); };
      for (const Local& c : loc) bucket[slot(c.key) + 1]++;
      for (size_t k = 1; k < bucket.size(); ++k) bucket[k] += bucket[k - 1];
      loc_sorted.resize(loc.size());
      for (const Local& c : loc) loc_sorted[bucket[slot(c.key)]++] = c;
      for (size_t i = 0; i < loc_sorted.size(); ++i) {
        const Local& c = loc_sorted[i];
        const int rank = c.key >> 24, blk = c.key & 0xFFFFFF, pa = blk / (nvp + 1), pb = blk % (nvp + 1);
        LevelOut& L = lout[lv[rank]];
        if (i == 0 || loc_sorted[i - 1].key != c.key) {
          L.dst.push_back(S.off[p] + poffv[pb] + (int64_t)poffv[pa] * S.N[p]);
          L.ld.push_back(S.N[p]);
          L.dims.push_back(pdim[pb] | (pdim[pa] << 8) | ((pa == pb) ? (1 << 16) : 0));
          L.front.push_back(p);
          L.ptr.push_back((int64_t)L.src.size());
        }
        L.src.push_back({c.child, c.loc, c.loc2});
      }
    }
    clk.mark("  gather: contributions");
    const int ngl = S.n_ulevels + 1;   // gather groups: the upper levels + the side group
    S.gt_lvl_ptr.assign(ngl + 1, 0);
    size_t n_src = 0;
    for (const LevelOut& L : lout) n_src += L.src.size();
    S.gs_child.resize(n_src);
    S.gs_loc.resize(n_src);
    S.gs_loc2.resize(n_src);
    {
      size_t base = 0;
      for (int l = 0; l < ngl; ++l) {
        LevelOut& L = lout[l];
        for (size_t t = 0; t < L.dst.size(); ++t) {
          S.gt_dst.push_back(L.dst[t]);
          S.gt_ld.push_back(L.ld[t]);
          S.gt_dims.push_back(L.dims[t]);
          S.gt_front.push_back(L.front[t]);
          S.gt_ptr.push_back((int64_t)(base + L.ptr[t]));
        }
        S.gt_lvl_ptr[l + 1] = (int)L.dst.size();
        for (size_t i = 0; i < L.src.size(); ++i) {
          S.gs_child[base + i] = L.src[i].child;
          S.gs_loc[base + i] = L.src[i].loc;
          S.gs_loc2[base + i] = L.src[i].loc2;
        }
        base += L.src.size();
        L = LevelOut();
      }
    }
    S.gt_ptr.push_back((int64_t)n_src);
    for (int l = 0; l < ngl; ++l) S.gt_lvl_ptr[l + 1] += S.gt_lvl_ptr[l];
    clk.mark("  gather: tasks");
    // Segments: a task's source list is cut into chunks of at most kGatherChunk sources, one wave each.
    // Single-segment tasks add straight into the destination; multi-segment tasks write partial sums
    // to scratch slots which a second pass adds in slot order (fixed order => deterministic).
    S.gseg_lvl_ptr.assign(ngl + 1, 0);
    S.gm_lvl_ptr.assign(ngl + 1, 0);
    S.g_max_slots = 0;
    S.side_slot0 = 0;
    const int ntasks = (int)S.gt_dst.size();
    int t = 0;
    for (int l = 0; l < ngl; ++l) {
      // (the side group runs beside the level gathers: scratch slots of its own, behind theirs)
      if (l == S.n_ulevels) S.side_slot0 = S.g_max_slots;
      int slots = l == S.n_ulevels ? S.side_slot0 : 0;
      // (shorter chunks for the levels with few sources were measured: slower — more waves, more scratch)
      const int chunk = kGatherChunk;
      for (; t < S.gt_lvl_ptr[l + 1]; ++t) {
        const int64_t b = S.gt_ptr[t], e = S.gt_ptr[t + 1];
        const int dB = S.gt_dims[t] & 255, dA = (S.gt_dims[t] >> 8) & 255;
        const int64_t len = e - b;
        const bool split = len > chunk && dB <= 16 && dA <= 16;
        if (!split) {
          S.gseg_task.push_back(t);
          S.gseg_begin.push_back(b);
          S.gseg_end.push_back(e);
          S.gseg_slot.push_back(-1);
        } else {
          const int nseg = (int)((len + chunk - 1) / chunk);
          S.gm_task.push_back(t);
          S.gm_slot.push_back(slots);
          S.gm_nslots.push_back(nseg);
          for (int k = 0; k < nseg; ++k) {
            S.gseg_task.push_back(t);
            S.gseg_begin.push_back(b + (int64_t)k * chunk);
            S.gseg_end.push_back(std::min<int64_t>(e, b + (int64_t)(k + 1) * chunk));
            S.gseg_slot.push_back(slots + k);
          }
          slots += nseg;
        }
      }
      S.gseg_lvl_ptr[l + 1] = (int)S.gseg_task.size();
      S.gm_lvl_ptr[l + 1] = (int)S.gm_task.size();
      S.g_max_slots = std::max(S.g_max_slots, slots);
    }
    (void)ntasks;
  }
  clk.mark("gather plan + stats");
  return GSX_OK;
}

}  // namespace gsx
