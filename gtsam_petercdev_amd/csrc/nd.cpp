// nd.cpp — multilevel nested dissection, the library's own analogue of what the reference delegates to METIS
// (Ordering::Metis -> METIS_NodeND, gtsam/inference/Ordering.cpp:211-251; adjacency as MetisIndex builds it,
// gtsam/inference/MetisIndex-inl.h:27-82).  Not METIS and not bit-compatible with it (METIS orderings are not even
// portable between platforms upstream: gtsam/inference/tests/testOrdering.cpp:306-336) — the same ingredients:
//   coarsening by heavy-edge matching  ->  greedy graph-growing bisection of the coarsest graph  ->
//   Fiduccia-Mattheyses boundary refinement on the way back up  ->  vertex separator = minimum vertex cover of
//   the cut edges (Koenig / Hopcroft-Karp) + separator refinement  ->  recursion, minimum degree at the leaves.
// Vertex weights are the variables' tangent dimensions: a separator is priced by the scalar columns it eliminates.
// Host only, deterministic (fixed-seed generator, no hashing, no addresses).
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <queue>

#include "gsx_internal.h"

namespace gsx {

namespace {

struct Graph {
  int n = 0;
  std::vector<int> xadj, adj, ew, vw;
  int64_t total_vw() const { return std::accumulate(vw.begin(), vw.end(), (int64_t)0); }
};

struct Rng {  // 64-bit LCG (Knuth's MMIX constants)
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 2862933555777941757ull + 3037000493ull) {}
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
  int below(int n) { return (int)(next() % (uint32_t)n); }
};

// ---- coarsening: heavy-edge matching ---------------------------------------------------------------------------------
void coarsen(const Graph& g, Rng& rng, std::vector<int>& cmap, Graph& gc) {
  const int n = g.n;
  std::vector<int> perm(n), match(n, -1);
  std::iota(perm.begin(), perm.end(), 0);
  for (int i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng.below(i + 1)]);
  cmap.assign(n, -1);
  int nc = 0;
  for (int v : perm) {
    if (match[v] >= 0) continue;
    int best = -1, bw = -1;
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
      const int u = g.adj[k];
      if (match[u] < 0 && u != v && (g.ew[k] > bw || (g.ew[k] == bw && g.vw[u] < g.vw[best]))) {
        best = u;
        bw = g.ew[k];
      }
    }
    match[v] = best >= 0 ? best : v;
    if (best >= 0) match[best] = v;
    cmap[v] = nc;
    if (best >= 0) cmap[best] = nc;
    ++nc;
  }
  gc = Graph();
  gc.n = nc;
  gc.vw.assign(nc, 0);
  gc.xadj.assign(nc + 1, 0);
  std::vector<int> first(nc, -1), second(nc, -1);
  for (int v = 0; v < n; ++v) {
    const int c = cmap[v];
    gc.vw[c] += g.vw[v];
    (first[c] < 0 ? first[c] : second[c]) = v;
  }
  std::vector<int> mark(nc, -1), pos(nc, 0);
  for (int c = 0; c < nc; ++c) {
    for (int v : {first[c], second[c]}) {
      if (v < 0) continue;
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int cu = cmap[g.adj[k]];
        if (cu == c) continue;
        if (mark[cu] != c) {
          mark[cu] = c;
          pos[cu] = (int)gc.adj.size();
          gc.adj.push_back(cu);
          gc.ew.push_back(g.ew[k]);
        } else {
          gc.ew[pos[cu]] += g.ew[k];
        }
      }
    }
    gc.xadj[c + 1] = (int)gc.adj.size();
  }
}

int64_t cut_of(const Graph& g, const std::vector<char>& part) {
  int64_t cut = 0;
  for (int v = 0; v < g.n; ++v)
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
      if (part[g.adj[k]] != part[v]) cut += g.ew[k];
  return cut / 2;
}

// ---- initial bisection of the coarsest graph: greedy graph growing, several seeds --------------------------------------
void grow_bisection(const Graph& g, Rng& rng, std::vector<char>& best_part) {
  const int n = g.n;
  const int64_t total = g.total_vw(), half = total / 2;
  int64_t best_cut = INT64_MAX;
  best_part.assign(n, 0);
  const int trials = std::min(n, 10);
  std::vector<char> part(n);
  std::vector<int64_t> gain(n);
  for (int t = 0; t < trials; ++t) {
    std::fill(part.begin(), part.end(), 0);
    std::fill(gain.begin(), gain.end(), 0);
    std::vector<char> in_front(n, 0);
    std::priority_queue<std::pair<int64_t, int>> pq;  // (gain, vertex), lazy
    int64_t w1 = 0;
    int seed = rng.below(n);
    while (w1 < half) {
      int v = -1;
      while (!pq.empty()) {
        auto [gn, u] = pq.top();
        pq.pop();
        if (part[u] == 0 && gn == gain[u]) {
          v = u;
          break;
        }
      }
      if (v < 0) {  // start, or the grown region exhausted its component: a fresh seed
        int tries = 0;
        while (part[seed] == 1 && tries++ < n) seed = (seed + 1) % n;
        if (part[seed] == 1) break;
        v = seed;
      }
      part[v] = 1;
      w1 += g.vw[v];
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int u = g.adj[k];
        if (part[u] == 1) continue;
        if (!in_front[u]) {  // gain of moving u into the region = (weight to region) - (weight to the rest)
          in_front[u] = 1;
          int64_t gsum = 0;
          for (int kk = g.xadj[u]; kk < g.xadj[u + 1]; ++kk) gsum += part[g.adj[kk]] == 1 ? g.ew[kk] : -g.ew[kk];
          gain[u] = gsum;
        } else {
          gain[u] += 2 * g.ew[k];
        }
        pq.push({gain[u], u});
      }
    }
    const int64_t cut = cut_of(g, part);
    if (cut < best_cut) {
      best_cut = cut;
      best_part = part;
    }
  }
}

// ---- Fiduccia-Mattheyses refinement of an edge bisection ------------------------------------------------------------------
void fm_refine(const Graph& g, std::vector<char>& part, double max_frac, int passes) {
  const int n = g.n;
  const int64_t total = g.total_vw();
  const int64_t max_w = (int64_t)(max_frac * (double)total) + 1;
  std::vector<int64_t> gain(n);
  std::vector<int> locked(n, -1);
  for (int pass = 0; pass < passes; ++pass) {
    int64_t w[2] = {0, 0};
    for (int v = 0; v < n; ++v) w[(int)part[v]] += g.vw[v];
    std::priority_queue<std::pair<int64_t, int>> pq[2];
    for (int v = 0; v < n; ++v) {
      int64_t ed = 0, id = 0;
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) (part[g.adj[k]] != part[v] ? ed : id) += g.ew[k];
      gain[v] = ed - id;
      if (ed > 0) pq[(int)part[v]].push({gain[v], v});  // boundary vertices only
    }
    std::vector<int> moved;
    int64_t cur = 0, best = 0;
    int best_len = 0, since_best = 0;
    const int limit = std::max(50, std::min(400, n / 20));
    while (since_best < limit) {
      // take from the side that may give (the heavier one must, when the other is at its limit)
      int v = -1, from = -1;
      for (int attempt = 0; attempt < 2 && v < 0; ++attempt) {
        int side;
        if (pq[0].empty() && pq[1].empty()) break;
        if (pq[0].empty()) side = 1;
        else if (pq[1].empty()) side = 0;
        else side = (pq[0].top().first > pq[1].top().first || (pq[0].top().first == pq[1].top().first && w[0] >= w[1])) ? 0 : 1;
        if (attempt == 1) side = 1 - side;
        while (!pq[side].empty()) {
          auto [gn, u] = pq[side].top();
          if (part[u] != side || locked[u] == pass || gn != gain[u]) {
            pq[side].pop();
            continue;
          }
          if (w[1 - side] + g.vw[u] > max_w) break;  // would unbalance: try the other side
          pq[side].pop();
          v = u;
          from = side;
          break;
        }
      }
      if (v < 0) break;
      part[v] = (char)(1 - from);
      locked[v] = pass;
      w[from] -= g.vw[v];
      w[1 - from] += g.vw[v];
      cur += gain[v];
      moved.push_back(v);
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int u = g.adj[k];
        if (locked[u] == pass) continue;
        gain[u] += (part[u] == part[v]) ? -2 * (int64_t)g.ew[k] : 2 * (int64_t)g.ew[k];
        pq[(int)part[u]].push({gain[u], u});
      }
      if (cur > best) {
        best = cur;
        best_len = (int)moved.size();
        since_best = 0;
      } else {
        ++since_best;
      }
    }
    for (int i = (int)moved.size() - 1; i >= best_len; --i) part[moved[i]] = (char)(1 - part[moved[i]]);  // roll back
    if (best <= 0) break;
  }
}

// ---- multilevel edge bisection -------------------------------------------------------------------------------------
void bisect(const Graph& g, Rng& rng, std::vector<char>& part) {
  std::vector<Graph> levels;
  std::vector<std::vector<int>> cmaps;
  const Graph* cur = &g;
  while (cur->n > 160) {
    Graph gc;
    std::vector<int> cmap;
    coarsen(*cur, rng, cmap, gc);
    if (gc.n > (int)(0.92 * cur->n)) break;  // matching no longer shrinks the graph (stars)
    levels.push_back(std::move(gc));
    cmaps.push_back(std::move(cmap));
    cur = &levels.back();
  }
  std::vector<char> p;
  grow_bisection(*cur, rng, p);
  fm_refine(*cur, p, 0.58, 6);
  for (int l = (int)levels.size() - 1; l >= 0; --l) {
    const Graph& fine = (l == 0) ? g : levels[l - 1];
    std::vector<char> pf(fine.n);
    for (int v = 0; v < fine.n; ++v) pf[v] = p[cmaps[l][v]];
    p.swap(pf);
    fm_refine(fine, p, 0.58, 4);
  }
  part.swap(p);
}

// ---- vertex separator from an edge bisection: minimum vertex cover of the cut's bipartite graph ------------------------------
// label: 0 / 1 = the two sides, 2 = separator
void vertex_separator(const Graph& g, const std::vector<char>& part, std::vector<char>& label) {
  const int n = g.n;
  label.assign(part.begin(), part.end());
  std::vector<int> lidx(n, -1), ridx(n, -1), L, R;
  for (int v = 0; v < n; ++v) {
    bool b = false;
    for (int k = g.xadj[v]; k < g.xadj[v + 1] && !b; ++k) b = part[g.adj[k]] != part[v];
    if (!b) continue;
    if (part[v] == 0) {
      lidx[v] = (int)L.size();
      L.push_back(v);
    } else {
      ridx[v] = (int)R.size();
      R.push_back(v);
    }
  }
  const int nl = (int)L.size(), nr = (int)R.size();
  // Hopcroft-Karp
  std::vector<int> ml(nl, -1), mr(nr, -1), dist(nl);
  auto bfs = [&]() {
    std::queue<int> q;
    bool found = false;
    for (int i = 0; i < nl; ++i) {
      dist[i] = ml[i] < 0 ? 0 : -1;
      if (ml[i] < 0) q.push(i);
    }
    while (!q.empty()) {
      const int i = q.front();
      q.pop();
      const int v = L[i];
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int j = ridx[g.adj[k]];
        if (j < 0) continue;
        if (mr[j] < 0) found = true;
        else if (dist[mr[j]] < 0) {
          dist[mr[j]] = dist[i] + 1;
          q.push(mr[j]);
        }
      }
    }
    return found;
  };
  std::vector<int> it(nl);
  auto dfs = [&](int root) {
    // iterative augmenting DFS along the BFS layers
    std::vector<int> stack{root};
    std::vector<int> via;  // right vertex used to reach stack[k+1]
    while (!stack.empty()) {
      const int i = stack.back();
      const int v = L[i];
      bool advanced = false;
      for (int& k = it[i]; k < g.xadj[v + 1]; ++k) {
        const int j = ridx[g.adj[k]];
        if (j < 0) continue;
        if (mr[j] < 0) {  // augment along the stack
          int jj = j;
          for (int s = (int)stack.size() - 1; s >= 0; --s) {
            const int ii = stack[s];
            const int prev = ml[ii];
            ml[ii] = jj;
            mr[jj] = ii;
            jj = prev;
          }
          return true;
        }
        if (dist[mr[j]] == dist[i] + 1) {
          stack.push_back(mr[j]);
          ++k;
          advanced = true;
          break;
        }
      }
      if (!advanced) {
        dist[i] = -1;
        stack.pop_back();
      }
    }
    return false;
  };
  while (bfs()) {
    for (int i = 0; i < nl; ++i) it[i] = g.xadj[L[i]];
    for (int i = 0; i < nl; ++i)
      if (ml[i] < 0) dfs(i);
  }
  // Koenig: Z = vertices reachable from unmatched left vertices by alternating paths; cover = (L \ Z) + (R & Z)
  std::vector<char> zl(nl, 0), zr(nr, 0);
  std::queue<int> q;
  for (int i = 0; i < nl; ++i)
    if (ml[i] < 0) {
      zl[i] = 1;
      q.push(i);
    }
  while (!q.empty()) {
    const int i = q.front();
    q.pop();
    const int v = L[i];
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
      const int j = ridx[g.adj[k]];
      if (j < 0 || zr[j] || ml[i] == j) continue;
      zr[j] = 1;
      if (mr[j] >= 0 && !zl[mr[j]]) {
        zl[mr[j]] = 1;
        q.push(mr[j]);
      }
    }
  }
  for (int i = 0; i < nl; ++i)
    if (!zl[i]) label[L[i]] = 2;
  for (int j = 0; j < nr; ++j)
    if (zr[j]) label[R[j]] = 2;
}

// greedy refinement of a vertex separator: move a separator vertex into a side when that pulls less weight into the
// separator than it removes (its neighbours on the OTHER side must join the separator), keeping the sides balanced
void refine_separator(const Graph& g, std::vector<char>& label, double max_frac) {
  const int n = g.n;
  int64_t w[3] = {0, 0, 0};
  for (int v = 0; v < n; ++v) w[(int)label[v]] += g.vw[v];
  const int64_t max_w = (int64_t)(max_frac * (double)(w[0] + w[1] + w[2])) + 1;
  for (int pass = 0; pass < 8; ++pass) {
    bool any = false;
    for (int v = 0; v < n; ++v) {
      if (label[v] != 2) continue;
      int64_t pull[2] = {0, 0};  // weight of neighbours in side s
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
        if (label[g.adj[k]] < 2) pull[(int)label[g.adj[k]]] += g.vw[g.adj[k]];
      // moving v to side s pulls its neighbours of side 1-s into the separator
      int best = -1;
      int64_t best_gain = 0;
      for (int s = 0; s < 2; ++s) {
        const int64_t gain = g.vw[v] - pull[1 - s];
        if (w[s] + g.vw[v] > max_w) continue;
        if (gain > best_gain || (gain == best_gain && gain > 0 && w[s] < w[best < 0 ? s : best])) {
          best = s;
          best_gain = gain;
        }
      }
      if (best < 0 || best_gain <= 0) continue;
      label[v] = (char)best;
      w[2] -= g.vw[v];
      w[best] += g.vw[v];
      for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int u = g.adj[k];
        if (label[u] == 1 - best) {
          label[u] = 2;
          w[1 - best] -= g.vw[u];
          w[2] += g.vw[u];
        }
      }
      any = true;
    }
    if (!any) break;
  }
}

Graph induced(const Graph& g, const std::vector<int>& verts, std::vector<int>& local) {
  Graph s;
  s.n = (int)verts.size();
  for (int i = 0; i < s.n; ++i) local[verts[i]] = i;
  s.xadj.assign(s.n + 1, 0);
  s.vw.resize(s.n);
  for (int i = 0; i < s.n; ++i) {
    const int v = verts[i];
    s.vw[i] = g.vw[v];
    for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
      if (local[g.adj[k]] >= 0) {
        s.adj.push_back(local[g.adj[k]]);
        s.ew.push_back(g.ew[k]);
      }
    s.xadj[i + 1] = (int)s.adj.size();
  }
  for (int v : verts) local[v] = -1;
  return s;
}

struct NdContext {
  const std::vector<std::vector<int>>* adj;  // the caller's adjacency (for the leaf / separator minimum degree)
  const std::vector<int>* w;
  int leaf;
  NdLeafOrder leaf_order;
  std::vector<int>* out;
  Rng rng{12345};
};

// `ids[i]` = the caller's vertex of vertex i of g
void dissect(NdContext& C, const Graph& g, const std::vector<int>& ids) {
  if (g.n <= C.leaf) {
    C.leaf_order(*C.adj, *C.w, ids, *C.out);
    return;
  }
  std::vector<char> part, label;
  bisect(g, C.rng, part);
  vertex_separator(g, part, label);
  refine_separator(g, label, 0.62);
  std::vector<int> side[3];
  for (int v = 0; v < g.n; ++v) side[(int)label[v]].push_back(v);
  if (side[0].empty() || side[1].empty()) {  // no usable split (a clique-like blob): minimum degree on the whole set
    C.leaf_order(*C.adj, *C.w, ids, *C.out);
    return;
  }
  for (int s = 0; s < 2; ++s) {
    std::vector<int> sub_ids(side[s].size());
    for (size_t i = 0; i < side[s].size(); ++i) sub_ids[i] = ids[side[s][i]];
    std::vector<int> local(g.n, -1);
    const Graph sub = induced(g, side[s], local);
    dissect(C, sub, sub_ids);
  }
  std::vector<int> sep_ids(side[2].size());
  for (size_t i = 0; i < side[2].size(); ++i) sep_ids[i] = ids[side[2][i]];
  C.leaf_order(*C.adj, *C.w, sep_ids, *C.out);  // the separator last, minimum degree inside it
}

}  // namespace

void multilevel_nested_dissection(const std::vector<std::vector<int>>& adj, const std::vector<int>& w,
                                  const std::vector<int>& verts, int leaf, NdLeafOrder leaf_order, std::vector<int>& out) {
  // the induced graph on `verts`, unit edge weights
  std::vector<int> local(adj.size(), -1);
  for (size_t i = 0; i < verts.size(); ++i) local[verts[i]] = (int)i;
  Graph g;
  g.n = (int)verts.size();
  g.xadj.assign(g.n + 1, 0);
  g.vw.resize(g.n);
  for (int i = 0; i < g.n; ++i) {
    const int v = verts[i];
    g.vw[i] = std::max(1, w[v]);
    for (int u : adj[v])
      if (local[u] >= 0 && u != v) {
        g.adj.push_back(local[u]);
        g.ew.push_back(1);
      }
    g.xadj[i + 1] = (int)g.adj.size();
  }
  NdContext C;
  C.adj = &adj;
  C.w = &w;
  C.leaf = leaf;
  C.leaf_order = leaf_order;
  C.out = &out;
  dissect(C, g, verts);
}

}  // namespace gsx
