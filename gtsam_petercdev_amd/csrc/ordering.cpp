// ordering.cpp — the library's OWN fill-reducing orderings for the stand-alone harness.
//
// In a drop-in run the ordering is an input (gsx_set_ordering): GTSAM computes it once at
// optimizer construction (gtsam/nonlinear/LevenbergMarquardtParams.h:112-117, Ordering::Create,
// gtsam/inference/Ordering.h:217-236, via CCOLAMD / METIS).  Neither CCOLAMD nor METIS is part
// of this library; these are independent implementations of the same ideas:
//   MINDEGREE  quotient-graph minimum degree with element absorption (approximate external degree)
//   ND         nested dissection by BFS level-set separators, min-degree at the leaves
//   SCHUR      3-D landmarks first, the rest (cameras) by MINDEGREE on the co-visibility graph
//   SCHUR_ND   same, cameras by ND (the analogue of the reference's METIS choice for BAL)
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <queue>
#include <set>

#include "gsx_internal.h"

namespace gsx {

typedef std::vector<std::vector<int>> Adj;

static Adj build_adjacency(const HostProblem& P) {
  Adj adj(P.n_vars);
  for (int f = 0; f < P.n_factors; ++f)
    for (int a = P.f_key_ptr[f]; a < P.f_key_ptr[f + 1]; ++a)
      for (int b = P.f_key_ptr[f]; b < P.f_key_ptr[f + 1]; ++b)
        if (a != b) adj[P.f_vars[a]].push_back(P.f_vars[b]);
  for (auto& l : adj) {
    std::sort(l.begin(), l.end());
    l.erase(std::unique(l.begin(), l.end()), l.end());
  }
  return adj;
}

// Minimum degree on the vertices `verts` of the graph `adj` (edges to vertices outside the set are
// ignored).  Appends the elimination order to `out`.
static void min_degree(const Adj& adj, const std::vector<int>& w, const std::vector<int>& verts,
                       std::vector<int>& out) {
  const int n = (int)verts.size();
  if (n == 0) return;
  std::vector<int> local(adj.size(), -1);
  for (int i = 0; i < n; ++i) local[verts[i]] = i;
  std::vector<std::vector<int>> A(n), E(n), Le(n);
  std::vector<int64_t> wt(n), deg(n);
  for (int i = 0; i < n; ++i) {
    wt[i] = w[verts[i]];
    for (int u : adj[verts[i]])
      if (local[u] >= 0) A[i].push_back(local[u]);
  }
  std::vector<char> state(n, 0);  // 0 active, 1 element, 2 absorbed
  std::vector<int64_t> lw(n, 0);  // weight of an element's variable list
  std::set<std::pair<int64_t, int>> pq;
  for (int i = 0; i < n; ++i) {
    int64_t d = 0;
    for (int u : A[i]) d += wt[u];
    deg[i] = d;
    pq.insert({d, i});
  }
  std::vector<int> stamp(n, -1), Lp;
  for (int step = 0; step < n; ++step) {
    const int p = pq.begin()->second;
    pq.erase(pq.begin());
    Lp.clear();
    stamp[p] = step;
    for (int u : A[p])
      if (state[u] == 0 && stamp[u] != step) {
        stamp[u] = step;
        Lp.push_back(u);
      }
    for (int e : E[p]) {
      if (state[e] != 1) continue;
      for (int u : Le[e])
        if (state[u] == 0 && stamp[u] != step) {
          stamp[u] = step;
          Lp.push_back(u);
        }
      state[e] = 2;
      std::vector<int>().swap(Le[e]);
    }
    state[p] = 1;
    std::vector<int>().swap(A[p]);
    std::vector<int>().swap(E[p]);
    int64_t lwp = 0;
    for (int u : Lp) lwp += wt[u];
    lw[p] = lwp;
    for (int i : Lp) {
      // prune variable neighbours now covered by element p
      auto& Ai = A[i];
      size_t k = 0;
      for (int u : Ai)
        if (state[u] == 0 && stamp[u] != step) Ai[k++] = u;
      Ai.resize(k);
      auto& Ei = E[i];
      k = 0;
      for (int e : Ei)
        if (state[e] == 1) Ei[k++] = e;
      Ei.resize(k);
      Ei.push_back(p);
      int64_t d = 0;
      for (int u : Ai) d += wt[u];
      for (int e : Ei) d += lw[e] - wt[i];
      pq.erase({deg[i], i});
      deg[i] = d;
      pq.insert({d, i});
    }
    Le[p] = Lp;
    out.push_back(verts[p]);
  }
}

// Nested dissection on the vertex set `verts`.
static void nested_dissection(const Adj& adj, const std::vector<int>& w, std::vector<int> verts,
                              std::vector<int>& label, int& next_label, int leaf, std::vector<int>& out) {
  if ((int)verts.size() <= leaf) {
    min_degree(adj, w, verts, out);
    return;
  }
  const int my = next_label++;
  for (int v : verts) label[v] = my;
  // connected components first
  std::vector<int> comp_of;  // reuse label with negative marks: do BFS using a visited stamp
  std::vector<int> bfs, lvl_start;
  auto run_bfs = [&](int start, int want_label, std::vector<int>& order, std::vector<int>& lstart, int mark) {
    order.clear();
    lstart.clear();
    order.push_back(start);
    label[start] = mark;
    size_t head = 0;
    lstart.push_back(0);
    while (head < order.size()) {
      const size_t end = order.size();
      for (; head < end; ++head)
        for (int u : adj[order[head]])
          if (label[u] == want_label) {
            label[u] = mark;
            order.push_back(u);
          }
      if (order.size() > end) lstart.push_back((int)end);
    }
    lstart.push_back((int)order.size());
  };
  // split into components
  std::vector<std::vector<int>> comps;
  {
    const int mark = next_label++;
    for (int v : verts) {
      if (label[v] != my) continue;
      std::vector<int> order, ls;
      run_bfs(v, my, order, ls, mark);
      comps.push_back(order);
    }
  }
  if (comps.size() > 1) {
    for (auto& c : comps) nested_dissection(adj, w, c, label, next_label, leaf, out);
    return;
  }
  // single component: pseudo-peripheral start by two sweeps
  const int l1 = next_label++;
  for (int v : verts) label[v] = l1;
  std::vector<int> order, ls;
  const int m1 = next_label++;
  run_bfs(verts[0], l1, order, ls, m1);
  int far = order.back();
  const int m2 = next_label++;
  run_bfs(far, m1, order, ls, m2);
  far = order.back();
  const int m3 = next_label++;
  run_bfs(far, m2, order, ls, m3);
  const int nl = (int)ls.size() - 1;
  if (nl < 3) {  // no usable level structure (dense blob)
    min_degree(adj, w, verts, out);
    return;
  }
  // choose the separator level: smallest level in the middle third (by cumulative count)
  const int total = (int)order.size();
  int best = -1;
  int64_t best_size = INT64_MAX;
  for (int l = 1; l < nl - 1; ++l) {
    const int before = ls[l];
    if (before < total / 3 || before > 2 * total / 3) continue;
    int64_t sz = 0;
    for (int k = ls[l]; k < ls[l + 1]; ++k) sz += w[order[k]];
    if (sz < best_size) {
      best_size = sz;
      best = l;
    }
  }
  if (best < 0) {
    // fall back to the level containing the median
    for (int l = 1; l < nl - 1; ++l)
      if (ls[l + 1] > total / 2) {
        best = l;
        break;
      }
    if (best < 0) best = nl / 2;
  }
  std::vector<int> part1(order.begin(), order.begin() + ls[best]);
  std::vector<int> sep, part2(order.begin() + ls[best + 1], order.end());
  {
    // thin the level-set separator: a vertex of the level without a neighbour in the next level does not
    // separate anything and goes to part 1 (labels: every vertex of this component carries m3 now)
    const int far_mark = next_label++;
    for (int v : part2) label[v] = far_mark;
    for (int k = ls[best]; k < ls[best + 1]; ++k) {
      const int v = order[k];
      bool touches = false;
      for (int u : adj[v])
        if (label[u] == far_mark) {
          touches = true;
          break;
        }
      (touches ? sep : part1).push_back(v);
    }
  }
  nested_dissection(adj, w, part1, label, next_label, leaf, out);
  nested_dissection(adj, w, part2, label, next_label, leaf, out);
  // order the separator itself by minimum degree restricted to the separator
  min_degree(adj, w, sep, out);
}

void compute_ordering(const HostProblem& P, int kind, std::vector<int>& order) {
  order.clear();
  order.reserve(P.n_vars);
  if (kind == GSX_ORDER_NATURAL) {
    order.resize(P.n_vars);
    std::iota(order.begin(), order.end(), 0);
    return;
  }
  Adj adj = build_adjacency(P);
  std::vector<int> all(P.n_vars);
  std::iota(all.begin(), all.end(), 0);
  if (kind == GSX_ORDER_MINDEGREE) {
    min_degree(adj, P.dims, all, order);
    return;
  }
  if (kind == GSX_ORDER_ND) {
    std::vector<int> label(P.n_vars, -1);
    int next_label = 0;
    nested_dissection(adj, P.dims, all, label, next_label, 48, order);
    return;
  }
  // SCHUR: landmarks = VECTOR(3) variables all of whose neighbours are cameras
  std::vector<char> is_lm(P.n_vars, 0);
  int n_lm = 0;
  for (int v = 0; v < P.n_vars; ++v) {
    if (P.types[v] != GSX_VAR_VECTOR || P.dims[v] != 3 || adj[v].empty()) continue;
    bool ok = true;
    for (int u : adj[v]) ok = ok && P.types[u] == GSX_VAR_CAMERA;
    if (ok) {
      is_lm[v] = 1;
      ++n_lm;
    }
  }
  if (n_lm == 0) {
    min_degree(adj, P.dims, all, order);
    return;
  }
  Adj red(P.n_vars);
  std::vector<int> rest;
  for (int v = 0; v < P.n_vars; ++v) {
    if (is_lm[v]) {
      order.push_back(v);
      for (int a : adj[v])
        for (int b : adj[v])
          if (a != b) red[a].push_back(b);
    } else {
      rest.push_back(v);
      for (int u : adj[v])
        if (!is_lm[u]) red[v].push_back(u);
    }
  }
  for (int v : rest) {
    auto& l = red[v];
    std::sort(l.begin(), l.end());
    l.erase(std::unique(l.begin(), l.end()), l.end());
  }
  if (kind == GSX_ORDER_SCHUR_ND) {  // nested dissection of the reduced (camera) graph: shallower tree
    std::vector<int> label(P.n_vars, -1);
    int next_label = 0;
    nested_dissection(red, P.dims, rest, label, next_label, 16, order);
  } else {
    min_degree(red, P.dims, rest, order);
  }
}

}  // namespace gsx
