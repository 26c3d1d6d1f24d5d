// ordering.cpp — the library's OWN fill-reducing orderings for the stand-alone harness.
//
// In a drop-in run the ordering is an input (gsx_set_ordering): GTSAM computes it once at
// optimizer construction (gtsam/nonlinear/LevenbergMarquardtParams.h:112-117, Ordering::Create,
// gtsam/inference/Ordering.h:217-236, via CCOLAMD / METIS).  Neither CCOLAMD nor METIS is part
// of this library; these are independent implementations of the same ideas:
//   MINDEGREE  quotient-graph minimum degree with element absorption (approximate external degree)
//   ND         multilevel nested dissection (nd.cpp: heavy-edge coarsening, FM-refined bisection, minimum-vertex-cover
//              separators), min-degree at the leaves and inside the separators
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

// (GSX_ND_LEAF overrides the leaf size of the dissection: a tuning knob of the stand-alone harness)
static int nd_leaf(int dflt) {
  const char* e = std::getenv("GSX_ND_LEAF");
  return e ? std::max(1, std::atoi(e)) : dflt;
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
    multilevel_nested_dissection(adj, P.dims, all, nd_leaf(24), min_degree, order);
    return;
  }
  // SCHUR: landmarks = VECTOR(3) variables all of whose neighbours are cameras (SFM) or poses (visual SLAM with a fixed
  // calibration: GenericProjectionFactor<Pose3, Point3>)
  std::vector<char> is_lm(P.n_vars, 0);
  int n_lm = 0;
  for (int v = 0; v < P.n_vars; ++v) {
    if (P.types[v] != GSX_VAR_VECTOR || P.dims[v] != 3 || adj[v].empty()) continue;
    bool ok = true;
    for (int u : adj[v]) ok = ok && (P.types[u] == GSX_VAR_CAMERA || P.types[u] == GSX_VAR_POSE3);
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
    multilevel_nested_dissection(red, P.dims, rest, nd_leaf(16), min_degree, order);
  } else {
    min_degree(red, P.dims, rest, order);
  }
}

}  // namespace gsx
