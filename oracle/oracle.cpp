// oracle/oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("the oracle") of the reference's hot path:
//   linearize -> (damped) GaussianFactorGraph -> eliminateMultifrontal with
//   EliminateCholesky per Bayes-tree clique -> GaussianBayesTree::optimize ->
//   LM policy.
// It follows the reference's algorithm structure literally (elimination tree,
// junction tree with its merge rule, per-clique Scatter + augmented Hessian +
// choleskyPartial, symbolic work redone at every solve) so that it can serve as
// the checker for the HIP path and as the `cpu_baseline` ("port", 1 thread) of
// bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load this library; the product (gtsam_petercdev_amd/, libgsx.so)
// never does.
//
// Parity status: PINNED by the reference's own known-answer tests, see
// tests/test_oracle_golden.py (choleskyPartial 7x7, createGaussianFactorGraph /
// createCorrectDelta, smoother junction tree, Pose2 LM cases, dubrovnik-3-7-pre
// final error 0.0199833).  The reference itself (GTSAM) cannot be built under
// this round's rules (needs cmake-generated config.h / dllexport.h); its vendored
// CCOLAMD C source can, see oracle/Makefile (oracle/_ref/libccolamd_ref.so).
//
// All citations are relative to /root/reference/.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../include/gsx.h"
#include "geometry.h"

namespace orc {

using std::vector;
typedef vector<double> Vec;

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------
// Problem description (copied out of the caller's gsx_problem_desc)
// ---------------------------------------------------------------------------
static int state_dim(int type, int dim) {
  switch (type) {
    case GSX_VAR_VECTOR: return dim;
    case GSX_VAR_POSE2: return 3;
    case GSX_VAR_POSE3: return 12;
    case GSX_VAR_CAMERA: return 17;
  }
  return -1;
}

struct Factor {
  int type, rows, noise_kind;
  vector<int> vars;
  Vec meas, noise;
  // noiseModel::Constrained (gtsam/linear/NoiseModel.h:389-500): rows with sigma == 0 are hard constraints; mu is the
  // weight their violation gets in the ERROR functions (default 1000, NoiseModel.cpp:365-368).  Empty: no such row.
  Vec mu;
  bool constrained(int r) const { return !mu.empty() && noise[r] == 0.0; }
};

// A linear factor of the GaussianFactorGraph: either Jacobian [A b] or Hessian.
struct LinFactor {
  bool hessian = false;
  vector<int> vars;   // variable indices
  vector<int> dims;   // tangent dims per variable
  int rows = 0;       // Jacobian: m
  Vec M;              // Jacobian: m x (sum d + 1) col-major; Hessian: (sum d + 1)^2 col-major (upper)
  // Jacobian with a noise model (JacobianFactor::model_): per-row sigmas — 0 = hard constraint — and the constraint
  // weights mu of Constrained::unit() (NoiseModel.cpp:470-476).  Empty: no model (whitened, unit).
  Vec sigmas, mu;
  bool has_constraints() const {
    for (double sg : sigmas)
      if (sg == 0.0) return true;
    return false;
  }
  int cols() const {
    int c = 1;
    for (int d : dims) c += d;
    return c;
  }
};

struct Conditional {       // GaussianConditional [R S d] (gtsam/linear/GaussianConditional.h:243-252)
  vector<int> frontals, parents;  // variable indices
  int nf = 0, ncols = 0;   // frontal scalar dim, total columns incl. rhs
  Vec RSd;                 // nf x ncols col-major
  int parent_clique = -1;
};

struct Problem {
  int n_vars = 0;
  vector<uint64_t> keys;
  vector<int> types, dims, state_off, tan_off;
  int64_t state_size = 0, tan_size = 0, jac_size = 0;
  vector<Factor> factors;
  vector<int64_t> jac_off;

  Vec values;                 // packed state
  vector<int> ordering;       // variable indices in elimination order
  bool has_ordering = false;

  // last linearization
  vector<LinFactor> linear;   // graph order
  bool linearized = false;
  int64_t n_cheirality = 0;

  // last solve
  vector<Conditional> bayes_tree;
  Vec delta;
  bool solved = false;

  // LM state (gtsam/nonlinear/internal/LevenbergMarquardtState.h)
  double lm_lambda = 0, lm_factor = 0, lm_error = 0;
  int lm_iterations = 0, lm_inner = 0;

  // timings (seconds): linearize, damp, eliminate, backsub, linerr, retract, error, symbolic
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // tree stats of the last elimination
  double tree_flops = 0, tree_bytes = 0;
  int64_t tree_cliques = 0, tree_depth = 0, tree_maxf = 0, tree_maxs = 0;
  std::string err;
  // host threads for the two loops the reference runs on TBB: factors in linearize
  // (gtsam/nonlinear/NonlinearFactorGraph.cpp:214-261) and independent subtrees in elimination / back-substitution
  // (gtsam/base/treeTraversal/parallelTraversalTasks.h:35-156); 1 = the serial reference build (TBB off)
  int n_threads = 1;
};

// ---------------------------------------------------------------------------
// Noise whitening — gtsam/linear/NoiseModel.cpp:164-186 (Gaussian),
// :323-341 (Diagonal), :647-674 (Isotropic); Unit is a no-op.
// M is m x ncols col-major (all Jacobian blocks + rhs), whitened in place.
// ---------------------------------------------------------------------------
// mEstimator weight / loss — gtsam/linear/LossFunctions.cpp:179-191 (Huber), :250-267 (Tukey), :217-224 (Cauchy)
static double robust_weight(int loss, double k, double dist) {
  const double a = std::abs(dist);
  if (loss == 1) return (a <= k) ? 1.0 : k / a;
  if (loss == 2) {
    if (a > k) return 0.0;
    const double t = 1.0 - dist * dist / (k * k);
    return t * t;
  }
  return (k * k) / (k * k + dist * dist);
}
static double robust_loss(int loss, double k, double dist) {
  const double a = std::abs(dist);
  if (loss == 1) return (a <= k) ? dist * dist / 2 : k * (a - k / 2);
  if (loss == 2) {
    if (a > k) return k * k / 6.0;
    const double t = 1.0 - dist * dist / (k * k);
    return k * k * (1 - t * t * t) / 6.0;
  }
  return k * k * std::log1p(dist * dist / (k * k)) * 0.5;
}
// whitening with the BASE model only (Robust::unweightedWhiten / noise_->WhitenSystem)
static void whiten_rows_base(const Factor& f, double* M, int m, int ncols);
// Robust::WhitenSystem — NoiseModel.cpp:709-735 with Base::reweight (Block) — LossFunctions.cpp:51-125:
// whiten with the base model, then scale [A b] by sqrt(weight(|b|)); the rhs is the LAST column of M.
static void whiten_rows(const Factor& f, double* M, int m, int ncols) {
  whiten_rows_base(f, M, m, ncols);
  const int loss = f.noise_kind >> 4;
  if (loss) {
    double s = 0;
    for (int r = 0; r < m; ++r) s += M[(size_t)(ncols - 1) * m + r] * M[(size_t)(ncols - 1) * m + r];
    const double w = std::sqrt(robust_weight(loss, f.noise.back(), std::sqrt(s)));
    for (int i = 0; i < m * ncols; ++i) M[i] *= w;
  }
}
static void whiten_rows_base(const Factor& f0, double* M, int m, int ncols) {
  struct { int noise_kind; const Vec& noise; } f{f0.noise_kind & GSX_NOISE_BASE_MASK, f0.noise};
  if (f.noise_kind == GSX_NOISE_UNIT) return;
  if (f.noise_kind == GSX_NOISE_ISOTROPIC) {
    const double inv = 1.0 / f.noise[0];
    for (int i = 0; i < m * ncols; ++i) M[i] *= inv;
  } else if (f.noise_kind == GSX_NOISE_DIAGONAL || f.noise_kind == GSX_NOISE_CONSTRAINED) {
    // (Constrained::Whiten / whiten — NoiseModel.cpp:395-410,447-468: a row with sigma 0 is left as it is)
    for (int c = 0; c < ncols; ++c)
      for (int r = 0; r < m; ++r)
        if (f.noise[r] != 0.0) M[c * m + r] *= (1.0 / f.noise[r]);
  } else {  // GAUSSIAN: R (m x m row-major upper) * M
    Vec col(m);
    for (int c = 0; c < ncols; ++c) {
      for (int r = 0; r < m; ++r) {
        double s = 0;
        for (int k = r; k < m; ++k) s += f.noise[r * m + k] * M[c * m + k];
        col[r] = s;
      }
      for (int r = 0; r < m; ++r) M[c * m + r] = col[r];
    }
  }
}

// ---------------------------------------------------------------------------
// Unwhitened error and Jacobians of one nonlinear factor.
// A (if non-null): m x (sum d) col-major Jacobian blocks in key order; e: m.
// Returns false when the factor is zeroed by cheirality.
// ---------------------------------------------------------------------------
static void local_coords(int type, int dim, const double* x, const double* y, double* out) {
  // traits<T>::Local(x, y) = chart(x^-1 y)
  if (type == GSX_VAR_VECTOR) {
    for (int i = 0; i < dim; ++i) out[i] = y[i] - x[i];
  } else if (type == GSX_VAR_POSE2) {
    // Pose2 ChartAtOrigin::Local = (x, y, theta) — gtsam/geometry/Pose2.cpp:112-122
    Pose2 h = pose2_compose(pose2_inverse(pose2_from(x)), pose2_from(y));
    out[0] = h.x; out[1] = h.y; out[2] = pose2_theta(h);
  } else if (type == GSX_VAR_POSE3) {
    Pose3 h = pose3_compose(pose3_inverse(pose3_from(x)), pose3_from(y));
    pose3_logmap(h, out);  // Pose3 ChartAtOrigin::Local = Logmap — Pose3.cpp:263-277
  } else {  // CAMERA — gtsam/geometry/PinholeCamera.h:206-211
    Pose3 h = pose3_compose(pose3_inverse(pose3_from(x)), pose3_from(y));
    pose3_logmap(h, out);
    out[6] = y[12] - x[12]; out[7] = y[13] - x[13]; out[8] = y[14] - x[14];
  }
}

static bool eval_factor(const Problem& P, const Factor& f, const double* values, double* e, double* A) {
  const int m = f.rows;
  if (f.type == GSX_F_PRIOR) {
    // PriorFactor::evaluateError — gtsam/nonlinear/PriorFactor.h:98-102: e = -Local(x, prior), H = I
    const int v = f.vars[0];
    double l[9];
    local_coords(P.types[v], P.dims[v], values + P.state_off[v], f.meas.data(), l);
    for (int i = 0; i < m; ++i) e[i] = -l[i];
    if (A) {
      for (int i = 0; i < m * m; ++i) A[i] = 0;
      for (int i = 0; i < m; ++i) A[i * m + i] = 1.0;
    }
    return true;
  }
  if (f.type == GSX_F_BETWEEN) {
    // BetweenFactor::evaluateError — gtsam/slam/BetweenFactor.h:111-124;
    // LieGroup::between — gtsam/base/Lie.h:63-69: H1 = -Ad(h^-1), H2 = I.
    const int v1 = f.vars[0], v2 = f.vars[1];
    const int type = P.types[v1];
    const double* x1 = values + P.state_off[v1];
    const double* x2 = values + P.state_off[v2];
    if (type == GSX_VAR_VECTOR) {
      for (int i = 0; i < m; ++i) e[i] = (x2[i] - x1[i]) - f.meas[i];
      if (A) {
        for (int i = 0; i < 2 * m * m; ++i) A[i] = 0;
        for (int i = 0; i < m; ++i) {
          A[i * m + i] = -1.0;
          A[m * m + i * m + i] = 1.0;
        }
      }
    } else if (type == GSX_VAR_POSE2) {
      Pose2 h = pose2_compose(pose2_inverse(pose2_from(x1)), pose2_from(x2));
      Pose2 zh = pose2_compose(pose2_inverse(pose2_from(f.meas.data())), h);
      e[0] = zh.x; e[1] = zh.y; e[2] = pose2_theta(zh);
      if (A) {
        double Ad[9];
        pose2_adjoint(pose2_inverse(h), Ad);
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 3; ++r) {
            A[c * 3 + r] = -Ad[3 * r + c];
            A[9 + c * 3 + r] = (r == c) ? 1.0 : 0.0;
          }
      }
    } else {  // POSE3
      Pose3 h = pose3_compose(pose3_inverse(pose3_from(x1)), pose3_from(x2));
      Pose3 zh = pose3_compose(pose3_inverse(pose3_from(f.meas.data())), h);
      pose3_logmap(zh, e);
      if (A) {
        double Ad[36];
        pose3_adjoint(pose3_inverse(h), Ad);
        for (int c = 0; c < 6; ++c)
          for (int r = 0; r < 6; ++r) {
            A[c * 6 + r] = -Ad[6 * r + c];
            A[36 + c * 6 + r] = (r == c) ? 1.0 : 0.0;
          }
      }
    }
    return true;
  }
  if (f.type == GSX_F_BEARINGRANGE) {
    // BearingRangeFactor (an ExpressionFactor): e = -Local(h(x), z) with the derivatives of h — ExpressionFactor.h:103-115
    const double* pose = values + P.state_off[f.vars[0]];
    const double* pt = values + P.state_off[f.vars[1]];
    double br[2], H1[6], H2[4];
    bearing_range_2d(pose, pt, br, A ? H1 : nullptr, A ? H2 : nullptr);
    const double db = br[0] - f.meas[0];
    e[0] = std::atan2(std::sin(db), std::cos(db));  // Rot2 local coordinates: the wrapped angle difference
    e[1] = br[1] - f.meas[1];
    if (A) {
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 2; ++r) A[c * 2 + r] = H1[3 * r + c];
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 2; ++r) A[6 + c * 2 + r] = H2[2 * r + c];
    }
    return true;
  }
  if (f.type == GSX_F_PROJECTION) {
    // GenericProjectionFactor::evaluateError — gtsam/slam/ProjectionFactor.h:138-166 (no body_P_sensor; default
    // throwCheirality = false: zero Jacobians and the constant error 2 fx)
    const double* pose = values + P.state_off[f.vars[0]];
    const double* pt = values + P.state_off[f.vars[1]];
    const double* K = f.meas.data() + 2;
    double pi[2], H1[12], H2[6];
    const bool ok = pinhole_project_s2(pose, pt, K, pi, A ? H1 : nullptr, A ? H2 : nullptr);
    if (!ok) {
      e[0] = e[1] = 2.0 * K[0];
      if (A)
        for (int i = 0; i < 18; ++i) A[i] = 0;
      return true;  // not a zeroed factor: the constant error stays in the system
    }
    e[0] = pi[0] - f.meas[0];
    e[1] = pi[1] - f.meas[1];
    if (A) {
      for (int c = 0; c < 6; ++c)
        for (int r = 0; r < 2; ++r) A[c * 2 + r] = H1[6 * r + c];
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 2; ++r) A[12 + c * 2 + r] = H2[3 * r + c];
    }
    return true;
  }
  if (f.type == GSX_F_SFM) {
    // GeneralSFMFactor::evaluateError — gtsam/slam/GeneralSFMFactor.h:127-139
    const double* cam = values + P.state_off[f.vars[0]];
    const double* pt = values + P.state_off[f.vars[1]];
    double pi[2], H1[18], H2[6];
    const bool ok = sfm_project(cam, pt, pi, A ? H1 : nullptr, A ? H2 : nullptr);
    if (!ok) {
      e[0] = e[1] = 0;
      if (A)
        for (int i = 0; i < 24; ++i) A[i] = 0;
      return false;
    }
    e[0] = pi[0] - f.meas[0];
    e[1] = pi[1] - f.meas[1];
    if (A) {
      for (int c = 0; c < 9; ++c)
        for (int r = 0; r < 2; ++r) A[c * 2 + r] = H1[9 * r + c];
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 2; ++r) A[18 + c * 2 + r] = H2[3 * r + c];
    }
    return true;
  }
  return true;
}

// NoiseModelFactor::error — gtsam/nonlinear/NonlinearFactor.cpp:138-149.
static double factor_error(const Problem& P, const Factor& f, const double* values) {
  if (f.type == GSX_F_LINEAR) return 0.0;  // handled by the linear path
  double e[16];
  eval_factor(P, f, values, e, nullptr);
  whiten_rows_base(f, e, f.rows, 1);  // squaredMahalanobisDistance of the base model
  double s = 0;
  // (Constrained::squaredMahalanobisDistance — NoiseModel.cpp:438-444: v_i sqrt(mu_i) on the constrained rows)
  for (int i = 0; i < f.rows; ++i) s += e[i] * e[i] * (f.constrained(i) ? f.mu[i] : 1.0);
  const int loss = f.noise_kind >> 4;
  if (loss) return robust_loss(loss, f.noise.back(), std::sqrt(s));  // Robust::loss — NoiseModel.h
  return 0.5 * s;
}

// NonlinearFactorGraph::error — gtsam/nonlinear/NonlinearFactorGraph.cpp:170-179 (serial, graph order).
static double graph_error(const Problem& P, const double* values) {
  double total = 0.0;
  for (const Factor& f : P.factors) total += factor_error(P, f, values);
  return total;
}

// NoiseModelFactor::linearize — gtsam/nonlinear/NonlinearFactor.cpp:152-184;
// GeneralSFMFactor::linearize — gtsam/slam/GeneralSFMFactor.h:141-177.
// run body(begin, end, thread) over [0, n) on up to P.n_threads threads (contiguous chunks)
template <class Body>
static void parallel_chunks(const Problem& P, size_t n, Body body) {
  const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)P.n_threads, n / 256 + 1));
  if (nt == 1) {
    body((size_t)0, n, 0);
    return;
  }
  vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { body(n * t / nt, n * (t + 1) / nt, t); });
  for (auto& x : th) x.join();
}

static void linearize(Problem& P) {
  const double t0 = now_s();
  P.linear.clear();
  P.linear.resize(P.factors.size());
  vector<int64_t> cheir((size_t)std::max(P.n_threads, 1), 0);
  parallel_chunks(P, P.factors.size(), [&](size_t begin, size_t end, int t) {
    for (size_t i = begin; i < end; ++i) {
      const Factor& f = P.factors[i];
      LinFactor& L = P.linear[i];
      L.vars = f.vars;
      for (int v : f.vars) L.dims.push_back(P.dims[v]);
      L.rows = f.rows;
      const int m = f.rows, nc = L.cols();
      L.M.assign((size_t)m * nc, 0.0);
      if (!f.mu.empty()) {  // NoiseModelFactor::linearize returns JacobianFactor(terms, b, constrained->unit()) (:176-180)
        L.sigmas.assign(m, 1.0);
        L.mu = f.mu;
        for (int r = 0; r < m; ++r)
          if (f.constrained(r)) L.sigmas[r] = 0.0;
      }
      if (f.type == GSX_F_LINEAR) {
        L.M = f.meas;  // already [A b]
        whiten_rows(f, L.M.data(), m, nc);
        continue;
      }
      double e[16];
      const bool ok = eval_factor(P, f, P.values.data(), e, L.M.data());
      if (!ok) ++cheir[t];
      for (int r = 0; r < m; ++r) L.M[(size_t)(nc - 1) * m + r] = -e[r];  // b = -e (SFM: z - pi)
      whiten_rows(f, L.M.data(), m, nc);
    }
  });
  P.n_cheirality = 0;
  for (int64_t c : cheir) P.n_cheirality += c;
  P.linearized = true;
  P.t[0] += now_s() - t0;
}

// JacobianFactor::error — gtsam/linear/JacobianFactor.cpp:494-514; GaussianFactorGraph::error :71-78.
static double linear_error(const Problem& P, const vector<LinFactor>& gfg, const double* delta) {
  double total = 0;
  Vec e;
  for (const LinFactor& L : gfg) {
    const int m = L.rows, nc = L.cols();
    e.assign(m, 0.0);
    for (int r = 0; r < m; ++r) e[r] = -L.M[(size_t)(nc - 1) * m + r];
    int col = 0;
    for (size_t k = 0; k < L.vars.size(); ++k) {
      const double* x = delta ? delta + P.tan_off[L.vars[k]] : nullptr;
      for (int c = 0; c < L.dims[k]; ++c, ++col)
        if (x)
          for (int r = 0; r < m; ++r) e[r] += L.M[(size_t)col * m + r] * x[c];
    }
    double s = 0;
    // (with a model: 0.5 * model->squaredMahalanobisDistance(unweighted_error) — JacobianFactor.cpp:508-513)
    for (int r = 0; r < m; ++r) {
      double w = 1.0;
      if (!L.sigmas.empty()) w = L.sigmas[r] == 0.0 ? L.mu[r] : 1.0 / (L.sigmas[r] * L.sigmas[r]);
      s += w * e[r] * e[r];
    }
    total += 0.5 * s;
  }
  return total;
}

// GaussianFactorGraph::hessianDiagonal — GaussianFactorGraph.cpp:279-287, JacobianFactor.cpp:539-564.
static void hessian_diagonal(const Problem& P, const vector<LinFactor>& gfg, double* d) {
  for (int64_t i = 0; i < P.tan_size; ++i) d[i] = 0;
  for (const LinFactor& L : gfg) {
    const int m = L.rows;
    int col = 0;
    for (size_t k = 0; k < L.vars.size(); ++k)
      for (int c = 0; c < L.dims[k]; ++c, ++col) {
        double s = 0;
        // (JacobianFactor::hessianDiagonalAdd whitens the column with the model, :548-556; Constrained::whiten leaves
        //  the rows of sigma 0 as they are)
        for (int r = 0; r < m; ++r) {
          const double w = (L.sigmas.empty() || L.sigmas[r] == 0.0) ? 1.0 : 1.0 / L.sigmas[r];
          s += w * w * L.M[(size_t)col * m + r] * L.M[(size_t)col * m + r];
        }
        d[P.tan_off[L.vars[k]] + c] += s;
      }
  }
}

// Values::retract — gtsam/nonlinear/Values.cpp:53-64,99-101.
static void retract(const Problem& P, const double* values, const double* delta, double* out) {
  for (int v = 0; v < P.n_vars; ++v) {
    const double* x = values + P.state_off[v];
    const double* d = delta + P.tan_off[v];
    double* y = out + P.state_off[v];
    switch (P.types[v]) {
      case GSX_VAR_VECTOR:
        for (int i = 0; i < P.dims[v]; ++i) y[i] = x[i] + d[i];
        break;
      case GSX_VAR_POSE2: {  // x * Pose2(v0,v1,v2) — Pose2.cpp:100-110, Lie.h:131-133
        Pose2 r = pose2_compose(pose2_from(x), pose2_from(d));
        y[0] = r.x; y[1] = r.y; y[2] = pose2_theta(r);
        break;
      }
      case GSX_VAR_POSE3: {  // x * Expmap(xi) — Pose3.cpp:248-250
        pose3_to(pose3_compose(pose3_from(x), pose3_expmap(d)), y);
        break;
      }
      case GSX_VAR_CAMERA: {  // PinholeCamera.h:197-203, Cal3Bundler.h:134-136
        pose3_to(pose3_compose(pose3_from(x), pose3_expmap(d)), y);
        y[12] = x[12] + d[6]; y[13] = x[13] + d[7]; y[14] = x[14] + d[8];
        y[15] = x[15]; y[16] = x[16];
        break;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// gtsam::choleskyPartial — gtsam/base/cholesky.cpp:108-159.
// ABC: n x n col-major, upper triangle significant.
// ---------------------------------------------------------------------------
static bool cholesky_partial(double* ABC, int n, int nf) {
  if (nf == 0) return true;
#define AT(r, c) ABC[(size_t)(c) * n + (r)]
  // Eigen::LLT<Upper> on A: A = R'R (unblocked left-looking; failure iff pivot <= 0)
  for (int j = 0; j < nf; ++j) {
    double x = AT(j, j);
    for (int k = 0; k < j; ++k) x -= AT(k, j) * AT(k, j);
    if (!(x > 0)) return false;
    const double rjj = std::sqrt(x);
    AT(j, j) = rjj;
    // row j of R for the remaining frontal columns AND of S = R^-T B (same recurrence)
    for (int c = j + 1; c < n; ++c) {
      double s = AT(j, c);
      for (int k = 0; k < j; ++k) s -= AT(k, j) * AT(k, c);
      AT(j, c) = s / rjj;
    }
  }
  // C -= S' S (upper)
  for (int c = nf; c < n; ++c)
    for (int r = nf; r <= c; ++r) {
      double s = 0;
      for (int k = 0; k < nf; ++k) s += AT(k, r) * AT(k, c);
      AT(r, c) -= s;
    }
  // conditioning check on the last two pivots — cholesky.cpp:145-158
  const int underconstrainedExponentDifference = 12;
  if (nf >= 2) {
    int exp2, exp1;
    (void)std::frexp(AT(nf - 2, nf - 2), &exp2);
    (void)std::frexp(AT(nf - 1, nf - 1), &exp1);
    return (exp2 - exp1 < underconstrainedExponentDifference);
  } else {
    int exp1;
    (void)std::frexp(AT(0, 0), &exp1);
    return (exp1 > -underconstrainedExponentDifference);
  }
#undef AT
}

// ---------------------------------------------------------------------------
// Symbolic: VariableIndex, EliminationTree, JunctionTree
// ---------------------------------------------------------------------------
struct ENode {  // EliminationTree node
  int var;
  vector<int> factors;    // indices into gfg
  vector<int> children;   // node ids (== elimination position)
};
struct Cluster {  // JunctionTree cluster (ClusterTree::Cluster)
  vector<int> orderedFrontalVars;
  vector<int> factors;
  vector<int> children;   // cluster ids
  int problemSize = 0;
};

struct IndeterminantLinearSystem {
  uint64_t key;
};

// EliminationTree ctor — gtsam/inference/EliminationTree-inst.h:78-156.
static void build_etree(const Problem& P, const vector<LinFactor>& gfg, const vector<int>& order,
                        vector<ENode>& nodes, vector<int>& roots) {
  const size_t m = gfg.size(), n = order.size();
  // VariableIndex::augment — gtsam/inference/VariableIndex-inl.h:27-49
  vector<vector<int>> index(P.n_vars);
  for (size_t i = 0; i < m; ++i)
    for (int v : gfg[i].vars) index[v].push_back((int)i);
  const int none = -1;
  nodes.assign(n, ENode());
  vector<int> parents(n, none), prevCol(m, none);
  for (size_t j = 0; j < n; ++j) {
    ENode& node = nodes[j];
    node.var = order[j];
    for (int i : index[order[j]]) {
      if (prevCol[i] != none) {
        int r = prevCol[i];
        while (parents[r] != none) r = parents[r];
        if (r != (int)j) {
          parents[r] = (int)j;
          node.children.push_back(r);
        }
      } else {
        node.factors.push_back(i);
      }
      prevCol[i] = (int)j;
    }
  }
  roots.clear();
  for (size_t j = 0; j < n; ++j)
    if (parents[j] == none) roots.push_back((int)j);
}

// JunctionTree ctor — gtsam/inference/JunctionTree-inst.h:51-153; symbolic elimination
// gtsam/symbolic/SymbolicFactor-inst.h:36-70; Cluster::mergeChildren ClusterTree-inst.h:58-96.
struct JTBuilder {
  const Problem& P;
  const vector<LinFactor>& gfg;
  const vector<ENode>& nodes;
  vector<Cluster> clusters;
  // per etree node results
  vector<vector<int>> sepVars;   // separator (symbolic factor keys), ascending key == ascending var index
  vector<int> clusterOf;

  JTBuilder(const Problem& p, const vector<LinFactor>& g, const vector<ENode>& n)
      : P(p), gfg(g), nodes(n), sepVars(n.size()), clusterOf(n.size(), -1) {}

  // Post-order processing of etree node j (children have smaller ids than their parent, so
  // visiting j = 0..n-1 is a valid post-order; the reference's DFS does the same per-node work).
  int visit(int j) {
    const ENode& node = nodes[j];
    const int cid = (int)clusters.size();
    clusters.emplace_back();
    clusters[cid].orderedFrontalVars.push_back(node.var);
    clusters[cid].factors = node.factors;
    vector<int> childClusters;
    for (int c : node.children) childClusters.push_back(clusterOf[c]);
    clusters[cid].children = childClusters;
    // symbolic elimination of this node: union of factor keys and child separators
    std::set<int> keyset;
    for (int f : node.factors)
      for (int v : gfg[f].vars) keyset.insert(v);
    for (int c : node.children)
      for (int v : sepVars[c]) keyset.insert(v);
    keyset.erase(node.var);
    sepVars[j].assign(keyset.begin(), keyset.end());
    const size_t nSymbolicFactors = node.factors.size() + node.children.size();
    clusters[cid].problemSize = (int)((sepVars[j].size() + 1) * nSymbolicFactors);
    // merge rule — JunctionTree-inst.h:100-120
    const size_t myNrParents = sepVars[j].size();
    size_t myNrFrontals = 1;
    vector<bool> merge(node.children.size(), false);
    for (size_t i = 0; i < node.children.size(); ++i) {
      if (myNrParents + myNrFrontals == sepVars[node.children[i]].size()) {
        myNrFrontals += clusters[childClusters[i]].orderedFrontalVars.size();
        merge[i] = true;
      }
    }
    // mergeChildren — ClusterTree-inst.h:58-96
    Cluster& me = clusters[cid];
    vector<int> oldChildren = me.children;
    me.children.clear();
    for (size_t i = 0; i < oldChildren.size(); ++i) {
      if (merge[i]) {
        Cluster& ch = clusters[oldChildren[i]];
        me.orderedFrontalVars.insert(me.orderedFrontalVars.end(), ch.orderedFrontalVars.rbegin(),
                                     ch.orderedFrontalVars.rend());
        me.factors.insert(me.factors.end(), ch.factors.begin(), ch.factors.end());
        me.children.insert(me.children.end(), ch.children.begin(), ch.children.end());
        me.problemSize = std::max(me.problemSize, ch.problemSize);
        ch.orderedFrontalVars.clear();  // dead
      } else {
        me.children.push_back(oldChildren[i]);
      }
    }
    std::reverse(me.orderedFrontalVars.begin(), me.orderedFrontalVars.end());
    clusterOf[j] = cid;
    return cid;
  }
};

// ---------------------------------------------------------------------------
// Constrained::QR — gtsam/linear/NoiseModel.cpp:478-620 (check_if_constraint :483-501).
// Ab: m x (n + 1) ROW-major, overwritten with the rows of [R d] (row i starts at column lead[i]); returns the
// precisions of the rows (infinity: still a hard constraint).
// ---------------------------------------------------------------------------
static void constrained_qr(Vec& Ab, int m_in, int n, const Vec& sigmas, vector<int>& lead, Vec& precisions) {
  const double kInf = std::numeric_limits<double>::infinity();
  const int n1 = n + 1;
  int m = m_in;
  const size_t maxRank = (size_t)std::min(m, n);
  Vec invsigmas(m), weights(m);
  for (int i = 0; i < m; ++i) {
    invsigmas[i] = 1.0 / sigmas[i];
    weights[i] = invsigmas[i] * invsigmas[i];
  }
  vector<Vec> Rd;
  vector<int> Rj;
  Vec Rp;
  Vec rd(n1, 0.0);
  for (int j = 0; j < n; ++j) {
    // the constrained row with the largest entry in column j, if that exceeds 1e-9
    int crow = -1;
    double max_element = 1e-9;
    for (int i = 0; i < m; ++i) {
      if (!std::isinf(invsigmas[i])) continue;
      const double a = std::abs(Ab[(size_t)i * n1 + j]);
      if (a > max_element) {
        max_element = a;
        crow = i;
      }
    }
    if (crow >= 0) {
      for (int c = 0; c < n1; ++c) rd[c] = Ab[(size_t)crow * n1 + c];   // the row itself: [R|d] is the row of [A|b]
      Rd.push_back(rd);
      Rj.push_back(j);
      Rp.push_back(kInf);
      if (Rd.size() >= maxRank) break;
      // swap the last valid row in, one row fewer
      m -= 1;
      if (crow != m) {
        for (int c = 0; c < n1; ++c) Ab[(size_t)crow * n1 + c] = Ab[(size_t)m * n1 + c];
        weights[crow] = weights[m];
        invsigmas[crow] = invsigmas[m];
      }
      const double inv = 1.0 / rd[j];
      for (int i = 0; i < m; ++i) {
        const double a = Ab[(size_t)i * n1 + j] * inv;
        Ab[(size_t)i * n1 + j] = a;
        for (int c = j + 1; c < n1; ++c) Ab[(size_t)i * n1 + c] -= a * rd[c];
      }
    } else {
      // Gram-Schmidt with the weighted pseudo-inverse of the column
      double precision = 0;
      Vec pseudo(m);
      for (int i = 0; i < m; ++i) {
        const double ai = Ab[(size_t)i * n1 + j];
        if (std::abs(ai) > 1e-9) {
          pseudo[i] = weights[i] * ai;
          precision += pseudo[i] * ai;
        } else {
          pseudo[i] = 0;
        }
      }
      if (precision > 1e-8) {
        for (int i = 0; i < m; ++i) pseudo[i] /= precision;
        rd[j] = 1.0;
        for (int c = j + 1; c < n1; ++c) {
          double sacc = 0;
          for (int i = 0; i < m; ++i) sacc += pseudo[i] * Ab[(size_t)i * n1 + c];
          rd[c] = sacc;
        }
        Rd.push_back(rd);
        Rj.push_back(j);
        Rp.push_back(precision);
      } else {
        continue;  // no information on this column
      }
      if (Rd.size() >= maxRank) break;
      for (int i = 0; i < m; ++i) {
        const double a = Ab[(size_t)i * n1 + j];
        for (int c = j + 1; c < n1; ++c) Ab[(size_t)i * n1 + c] -= a * rd[c];
      }
    }
  }
  std::fill(Ab.begin(), Ab.end(), 0.0);
  lead = Rj;
  precisions = Rp;
  for (size_t i = 0; i < Rd.size(); ++i)
    for (int c = Rj[i]; c < n1; ++c) Ab[i * n1 + c] = Rd[i][c];
}

// choleskyCareful on the whole matrix — gtsam/base/cholesky.cpp:36-105 (A: n x n col-major, UPPER triangle used / written)
static std::pair<int, bool> cholesky_careful(double* A, int n) {
  int maxrank = 0;
  for (int k = 0; k < n; ++k) {
    double alpha = A[(size_t)k * n + k];
    if (alpha < -1e-1) return {maxrank, false};
    if (alpha < 0.0) alpha = 0.0;
    const double beta = std::sqrt(alpha);
    if (beta > 1e-6) {
      const double betainv = 1.0 / beta;
      A[(size_t)k * n + k] = beta;
      for (int c = k + 1; c < n; ++c) A[(size_t)c * n + k] *= betainv;
      for (int c = k + 1; c < n; ++c)
        for (int r = k + 1; r <= c; ++r) A[(size_t)c * n + r] -= A[(size_t)r * n + k] * A[(size_t)c * n + k];
      maxrank = k + 1;
    } else {
      A[(size_t)k * n + k] = 1e-5;
      for (int c = k + 1; c < n; ++c) A[(size_t)c * n + k] = 0.0;
    }
  }
  return {maxrank, true};
}

static bool has_constraints(const vector<const LinFactor*>& factors) {  // GaussianFactorGraph.cpp:442-451
  for (const LinFactor* f : factors)
    if (!f->hessian && f->has_constraints()) return true;
  return false;
}

// ---------------------------------------------------------------------------
// EliminateCholesky on one clique — gtsam/linear/HessianFactor.cpp:515-535,
// Scatter (gtsam/linear/Scatter.cpp:39-73), HessianFactor merge ctor
// (HessianFactor.cpp:240-253), updateHessian (JacobianFactor.cpp:586-624,
// HessianFactor.cpp:349-373), eliminateCholesky (HessianFactor.cpp:458-486).
// ---------------------------------------------------------------------------
// EliminateQR on one clique with a constrained noise model — gtsam/linear/JacobianFactor.cpp:804-842 (EliminateQR),
// :224-311 (the combined JacobianFactor: rows stacked in factor order, sigmas copied, model Constrained::MixedSigmas),
// :86-112 (JacobianFactor(HessianFactor): choleskyCareful of the whole augmented matrix), :845-897 (splitConditional).
static void eliminate_qr(const Problem& P, const vector<const LinFactor*>& factors, const vector<int>& frontals,
                         Conditional& cond, LinFactor& remaining) {
  vector<int> slots_var = frontals;
  {
    std::set<int> rest;
    for (const LinFactor* f : factors)
      for (int v : f->vars) rest.insert(v);
    for (int v : frontals) rest.erase(v);
    slots_var.insert(slots_var.end(), rest.begin(), rest.end());
  }
  const int nslots = (int)slots_var.size();
  vector<int> off(nslots + 1, 0);
  for (int s = 0; s < nslots; ++s) off[s + 1] = off[s] + P.dims[slots_var[s]];
  const int n = off[nslots], n1 = n + 1;
  std::map<int, int> slot_of;
  for (int s = 0; s < nslots; ++s) slot_of[slots_var[s]] = s;
  // every factor as a Jacobian
  struct Rows {
    int m;
    Vec M;       // m x cols col-major over the factor's own variables + rhs
    Vec sigmas;  // empty: no model
    const LinFactor* f;
  };
  vector<Rows> jac;
  for (const LinFactor* f : factors) {
    Rows R;
    R.f = f;
    if (!f->hessian) {
      R.m = f->rows;
      R.M = f->M;
      R.sigmas = f->sigmas;
    } else {
      const int c = f->cols();
      Vec A = f->M;  // upper triangle holds the information matrix
      const auto res = cholesky_careful(A.data(), c);
      if (!(res.second || res.first == c - 1)) throw IndeterminantLinearSystem{P.keys[f->vars.front()]};
      R.m = res.first;
      R.M.assign((size_t)R.m * c, 0.0);
      for (int col = 0; col < c; ++col)
        for (int r = 0; r < R.m && r <= col; ++r) R.M[(size_t)col * R.m + r] = A[(size_t)col * c + r];
    }
    jac.push_back(std::move(R));
  }
  int m = 0;
  for (const Rows& R : jac) m += R.m;
  Vec Ab((size_t)m * n1, 0.0), sigmas(m, 1.0);  // row-major
  int row0 = 0;
  for (const Rows& R : jac) {
    if (R.m == 0) continue;
    int col = 0;
    const LinFactor* f = R.f;
    for (size_t k = 0; k <= f->vars.size(); ++k) {
      const int d = k < f->vars.size() ? f->dims[k] : 1;
      const int dst = k < f->vars.size() ? off[slot_of.at(f->vars[k])] : n;
      for (int c = 0; c < d; ++c, ++col)
        for (int r = 0; r < R.m; ++r) Ab[(size_t)(row0 + r) * n1 + dst + c] = R.M[(size_t)col * R.m + r];
    }
    if (!R.sigmas.empty())
      for (int r = 0; r < R.m; ++r) sigmas[row0 + r] = R.sigmas[r];
    row0 += R.m;
  }
  vector<int> lead;
  Vec precisions;
  constrained_qr(Ab, m, n, sigmas, lead, precisions);
  const int rank = (int)lead.size();
  int nfs = 0;
  for (int v : frontals) nfs += P.dims[v];
  if (rank < nfs) throw IndeterminantLinearSystem{P.keys[frontals.front()]};  // (model_->dim() < frontalDim, :860-863)
  cond.frontals = frontals;
  cond.parents.assign(slots_var.begin() + frontals.size(), slots_var.end());
  cond.nf = nfs;
  cond.ncols = n1;
  cond.RSd.assign((size_t)nfs * n1, 0.0);
  for (int r = 0; r < nfs; ++r)
    for (int c = 0; c < n1; ++c) cond.RSd[(size_t)c * nfs + r] = Ab[(size_t)r * n1 + c];
  const int rem = std::min(rank - nfs, std::min(n1, m) - nfs);
  remaining.hessian = false;
  remaining.vars = cond.parents;
  remaining.dims.clear();
  for (int v : remaining.vars) remaining.dims.push_back(P.dims[v]);
  remaining.rows = std::max(rem, 0);
  const int c2 = n1 - nfs;
  remaining.M.assign((size_t)remaining.rows * c2, 0.0);
  remaining.sigmas.assign(remaining.rows, 1.0);
  remaining.mu.assign(remaining.rows, 1000.0);  // (MixedSigmas(sigmas) of splitConditional :889-890: the default mu)
  for (int r = 0; r < remaining.rows; ++r) {
    for (int c = 0; c < c2; ++c) remaining.M[(size_t)c * remaining.rows + r] = Ab[(size_t)(nfs + r) * n1 + nfs + c];
    const double pr = precisions[nfs + r];
    remaining.sigmas[r] = std::isinf(pr) ? 0.0 : 1.0 / std::sqrt(pr);
  }
}

static void eliminate_clique(const Problem& P, const vector<const LinFactor*>& factors,
                             const vector<int>& frontals, Conditional& cond, LinFactor& remaining) {
  // EliminatePreferCholesky — gtsam/linear/HessianFactor.cpp:538-551
  if (has_constraints(factors)) {
    eliminate_qr(P, factors, frontals, cond, remaining);
    return;
  }
  // Scatter: frontals first (given order), then the other variables sorted by key
  // (variables are indexed in ascending key order, so sort by index).
  vector<int> slots_var = frontals;
  {
    std::set<int> rest;
    for (const LinFactor* f : factors) {
      if (!f->hessian && f->cols() <= 1) continue;
      for (int v : f->vars) rest.insert(v);
    }
    for (int v : frontals) rest.erase(v);
    slots_var.insert(slots_var.end(), rest.begin(), rest.end());
  }
  const int nslots = (int)slots_var.size();
  vector<int> off(nslots + 2, 0);
  for (int s = 0; s < nslots; ++s) off[s + 1] = off[s] + P.dims[slots_var[s]];
  off[nslots + 1] = off[nslots] + 1;
  const int N = off[nslots + 1];
  std::map<int, int> slot_of;
  for (int s = 0; s < nslots; ++s) slot_of[slots_var[s]] = s;

  Vec info((size_t)N * N, 0.0);
#define INFO(r, c) info[(size_t)(c) * N + (r)]
  for (const LinFactor* f : factors) {
    const int nb = (int)f->vars.size();
    vector<int> fs(nb + 1), foff(nb + 2, 0);
    for (int j = 0; j < nb; ++j) {
      fs[j] = slot_of.at(f->vars[j]);
      foff[j + 1] = foff[j] + f->dims[j];
    }
    fs[nb] = nslots;
    foff[nb + 1] = foff[nb] + 1;
    if (!f->hessian) {
      const int m = f->rows;
      if (m == 0) continue;
      Vec whitened;
      if (!f->sigmas.empty()) {  // HessianFactor(JacobianFactor) whitens with the model (HessianFactor.cpp:210-229)
        whitened = f->M;
        for (int c = 0; c < f->cols(); ++c)
          for (int r = 0; r < m; ++r) whitened[(size_t)c * m + r] /= f->sigmas[r];
      }
      const double* A = whitened.empty() ? f->M.data() : whitened.data();
      for (int j = 0; j <= nb; ++j) {
        const int J = fs[j];
        for (int i = 0; i <= j; ++i) {
          const int I = fs[i];
          // block = A_i' A_j
          for (int cj = foff[j]; cj < foff[j + 1]; ++cj)
            for (int ci = foff[i]; ci < foff[i + 1]; ++ci) {
              if (i == j && ci > cj) continue;  // diagonal block: upper only
              double s = 0;
              for (int r = 0; r < m; ++r) s += A[(size_t)ci * m + r] * A[(size_t)cj * m + r];
              const int ri = off[I] + (ci - foff[i]), rj = off[J] + (cj - foff[j]);
              if (I <= J && !(I == J && i != j))
                INFO(ri, rj) += s;
              else
                INFO(rj, ri) += s;
            }
        }
      }
    } else {
      const int n = f->cols();
      const double* H = f->M.data();
      for (int j = 0; j <= nb; ++j) {
        const int J = fs[j];
        for (int i = 0; i <= j; ++i) {
          const int I = fs[i];
          for (int cj = foff[j]; cj < foff[j + 1]; ++cj)
            for (int ci = foff[i]; ci < foff[i + 1]; ++ci) {
              if (i == j && ci > cj) continue;
              const double s = H[(size_t)cj * n + ci];
              const int ri = off[I] + (ci - foff[i]), rj = off[J] + (cj - foff[j]);
              if (I <= J)
                INFO(ri, rj) += s;
              else
                INFO(rj, ri) += s;
            }
        }
      }
    }
  }
#undef INFO
  int nfs = 0;
  for (size_t k = 0; k < frontals.size(); ++k) nfs += P.dims[frontals[k]];
  if (!cholesky_partial(info.data(), N, nfs)) throw IndeterminantLinearSystem{P.keys[frontals.front()]};

  // split — gtsam/base/SymmetricBlockMatrix.cpp:92-107
  cond.frontals = frontals;
  cond.parents.assign(slots_var.begin() + frontals.size(), slots_var.end());
  cond.nf = nfs;
  cond.ncols = N;
  cond.RSd.assign((size_t)nfs * N, 0.0);
  for (int c = 0; c < N; ++c)
    for (int r = 0; r < nfs && r <= c; ++r) cond.RSd[(size_t)c * nfs + r] = info[(size_t)c * N + r];
  remaining.hessian = true;
  remaining.vars = cond.parents;
  remaining.dims.clear();
  for (int v : remaining.vars) remaining.dims.push_back(P.dims[v]);
  const int n2 = N - nfs;
  remaining.M.assign((size_t)n2 * n2, 0.0);
  for (int c = 0; c < n2; ++c)
    for (int r = 0; r <= c; ++r) remaining.M[(size_t)c * n2 + r] = info[(size_t)(c + nfs) * N + (r + nfs)];
}

// eliminateMultifrontal + GaussianBayesTree::optimize.
// gtsam/inference/EliminateableFactorGraph-inst.h:123-146, ClusterTree-inst.h:219-266,286-318,
// gtsam/linear/linearAlgorithms-inst.h:49-117,142-155.
static void solve_gfg(Problem& P, const vector<LinFactor>& gfg, Vec& delta) {
  double t0 = now_s();
  vector<ENode> nodes;
  vector<int> roots;
  build_etree(P, gfg, P.ordering, nodes, roots);
  JTBuilder jt(P, gfg, nodes);
  for (int j = 0; j < (int)nodes.size(); ++j) jt.visit(j);
  vector<int> rootClusters;
  for (int r : roots) rootClusters.push_back(jt.clusterOf[r]);
  P.t[7] += now_s() - t0;
  t0 = now_s();

  // post-order numbering of the clusters (roots in order, children in order): clique c of the Bayes tree is the c-th
  // cluster eliminated by the serial traversal (ClusterTree-inst.h:286-318), whatever the number of threads
  const int ncl = (int)jt.clusters.size();
  vector<int> post, idx(ncl, -1), parentCluster(ncl, -1), depth(ncl, 1), size(ncl, 1);
  post.reserve(ncl);
  {
    struct Frame {
      int cluster;
      size_t next_child;
    };
    vector<Frame> stack;
    for (int rc : rootClusters) {
      stack.push_back(Frame{rc, 0});
      while (!stack.empty()) {
        Frame& fr = stack.back();
        const Cluster& cl = jt.clusters[fr.cluster];
        if (fr.next_child < cl.children.size()) {
          const int ch = cl.children[fr.next_child++];
          parentCluster[ch] = fr.cluster;
          depth[ch] = depth[fr.cluster] + 1;
          stack.push_back(Frame{ch, 0});
          continue;
        }
        idx[fr.cluster] = (int)post.size();
        post.push_back(fr.cluster);
        if (parentCluster[fr.cluster] >= 0) size[parentCluster[fr.cluster]] += size[fr.cluster];
        stack.pop_back();
      }
    }
  }
  vector<Conditional>& bt = P.bayes_tree;
  bt.clear();
  bt.resize(post.size());
  vector<std::unique_ptr<LinFactor>> remaining(ncl);
  P.tree_flops = P.tree_bytes = 0;
  P.tree_cliques = P.tree_depth = P.tree_maxf = P.tree_maxs = 0;
  std::mutex stat_mutex;
  constexpr int kNoFail = 0x7fffffff;
  std::atomic<int> first_fail{kNoFail};
  vector<uint64_t> fail_key(post.size(), 0);
  // eliminate the cliques post[begin .. end) in order (a whole subtree is a contiguous range of the post-order)
  auto eliminate_range = [&](int begin, int end) {
    double fl = 0, by = 0;
    int64_t maxf = 0, maxs = 0, dep = 0;
    for (int c = begin; c < end; ++c) {
      const int cluster = post[c];
      const Cluster& cl = jt.clusters[cluster];
      // gather factors: own then children's remaining
      vector<const LinFactor*> gathered;
      for (int f : cl.factors) gathered.push_back(&gfg[f]);
      for (int ch : cl.children) gathered.push_back(remaining[ch].get());
      Conditional cond;
      auto rem = std::make_unique<LinFactor>();
      try {
        eliminate_clique(P, gathered, cl.orderedFrontalVars, cond, *rem);
      } catch (const IndeterminantLinearSystem& e) {
        fail_key[c] = e.key;
        int cur = first_fail.load();
        while (c < cur && !first_fail.compare_exchange_weak(cur, c)) {}
        break;  // (the serial traversal stops at its first failure; nothing above this clique can be eliminated)
      }
      for (int ch : cl.children) remaining[ch].reset();
      const double f = cond.nf, s1 = cond.ncols - cond.nf;  // s + 1
      fl += f * f * f / 3 + f * f * s1 + f * s1 * s1;
      by += 8.0 * cond.ncols * cond.ncols;
      maxf = std::max<int64_t>(maxf, cond.nf);
      maxs = std::max<int64_t>(maxs, cond.ncols - cond.nf - 1);
      dep = std::max<int64_t>(dep, depth[cluster]);
      cond.parent_clique = parentCluster[cluster] >= 0 ? idx[parentCluster[cluster]] : -1;
      bt[c] = std::move(cond);
      remaining[cluster] = std::move(rem);
    }
    std::lock_guard<std::mutex> lk(stat_mutex);
    P.tree_flops += fl;
    P.tree_bytes += by;
    P.tree_cliques += end - begin;
    P.tree_maxf = std::max(P.tree_maxf, maxf);
    P.tree_maxs = std::max(P.tree_maxs, maxs);
    P.tree_depth = std::max(P.tree_depth, dep);
  };
  // the split the reference's parallel traversal makes (parallelTraversalTasks.h:35-156: a task per child subtree above a
  // size threshold): clusters whose subtree holds more than a share of the tree form the top, handled in order by this
  // thread; every subtree hanging below it is one task
  vector<std::pair<int, int>> tasks;  // post-order ranges of whole subtrees
  vector<char> is_top(ncl, 0);
  const int nthreads = std::max(1, P.n_threads);
  if (nthreads > 1) {
    const int share = std::max(1, (int)post.size() / (16 * nthreads));
    for (int c = (int)post.size() - 1; c >= 0; --c) {
      const int cluster = post[c];
      const int pc = parentCluster[cluster];
      if (size[cluster] > share && (pc < 0 || is_top[pc])) is_top[cluster] = 1;
      else if (pc < 0 || is_top[pc]) tasks.push_back({idx[cluster] - size[cluster] + 1, idx[cluster] + 1});
    }
    std::sort(tasks.begin(), tasks.end(), [](auto& a, auto& b) { return a.second - a.first > b.second - b.first; });
    std::atomic<size_t> next{0};
    auto worker = [&] {
      for (size_t k = next++; k < tasks.size(); k = next++) eliminate_range(tasks[k].first, tasks[k].second);
    };
    vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
    worker();
    for (auto& x : th) x.join();
    for (int c = 0; c < (int)post.size() && first_fail.load() == kNoFail; ++c)
      if (is_top[post[c]]) eliminate_range(c, c + 1);
  } else {
    eliminate_range(0, (int)post.size());
  }
  if (first_fail.load() != kNoFail) throw IndeterminantLinearSystem{fail_key[first_fail.load()]};
  P.t[2] += now_s() - t0;
  t0 = now_s();

  // back-substitution, parents before children (reverse post-order); subtrees below the top in parallel
  delta.assign(P.tan_size, 0.0);
  std::atomic<int> last_nan{-1};
  vector<uint64_t> nan_key(post.size(), 0);
  auto backsub_range = [&](int begin, int end) {  // cliques end-1 down to begin
    Vec rhs;
    for (int c = end - 1; c >= begin; --c) {
      const Conditional& cd = bt[c];
      const int nf = cd.nf, N = cd.ncols;
      rhs.assign(nf, 0.0);
      for (int r = 0; r < nf; ++r) rhs[r] = cd.RSd[(size_t)(N - 1) * nf + r];
      int col = nf;
      for (int pv : cd.parents)
        for (int k = 0; k < P.dims[pv]; ++k, ++col) {
          const double x = delta[P.tan_off[pv] + k];
          for (int r = 0; r < nf; ++r) rhs[r] -= cd.RSd[(size_t)col * nf + r] * x;
        }
      for (int r = nf - 1; r >= 0; --r) {  // R x = rhs (upper)
        double sacc = rhs[r];
        for (int k = r + 1; k < nf; ++k) sacc -= cd.RSd[(size_t)k * nf + r] * rhs[k];
        rhs[r] = sacc / cd.RSd[(size_t)r * nf + r];
      }
      int row = 0;
      for (int fv : cd.frontals)
        for (int k = 0; k < P.dims[fv]; ++k, ++row) {
          if (std::isnan(rhs[row])) {  // linearAlgorithms-inst.h:100-104
            nan_key[c] = P.keys[fv];
            int cur = last_nan.load();
            while (c > cur && !last_nan.compare_exchange_weak(cur, c)) {}
            return;
          }
          delta[P.tan_off[fv] + k] = rhs[row];
        }
    }
  };
  if (nthreads > 1) {
    for (int c = (int)post.size() - 1; c >= 0 && last_nan.load() < 0; --c)
      if (is_top[post[c]]) backsub_range(c, c + 1);
    if (last_nan.load() < 0) {
      std::atomic<size_t> next{0};
      auto worker = [&] {
        for (size_t k = next++; k < tasks.size(); k = next++) backsub_range(tasks[k].first, tasks[k].second);
      };
      vector<std::thread> th;
      for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
      worker();
      for (auto& x : th) x.join();
    }
  } else {
    backsub_range(0, (int)post.size());
  }
  if (last_nan.load() >= 0) throw IndeterminantLinearSystem{nan_key[last_nan.load()]};
  P.t[3] += now_s() - t0;
}

// LevenbergMarquardtState::buildDampedSystem — LevenbergMarquardtState.h:125-156.
static void build_damped(const Problem& P, const vector<LinFactor>& linear, double lambda, bool diagonal,
                         const double* sqrtHessianDiagonal, vector<LinFactor>& damped) {
  damped = linear;  // "gets copied"
  if (!(lambda > 0)) return;  // sigma = inf: the priors whiten to zero rows
  const double sigma = 1.0 / std::sqrt(lambda);
  const double inv = 1.0 / sigma;
  damped.reserve(linear.size() + P.n_vars);
  for (int v = 0; v < P.n_vars; ++v) {
    LinFactor L;
    const int d = P.dims[v];
    L.vars = {v};
    L.dims = {d};
    L.rows = d;
    L.M.assign((size_t)d * (d + 1), 0.0);
    for (int i = 0; i < d; ++i) {
      const double a = diagonal ? sqrtHessianDiagonal[P.tan_off[v] + i] : 1.0;
      L.M[(size_t)i * d + i] = inv * a;  // whitened by Isotropic(sigma)
    }
    damped.push_back(std::move(L));
  }
}

struct LMTrace {
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

// LevenbergMarquardtOptimizer::tryLambda — LevenbergMarquardtOptimizer.cpp:121-270.
static bool try_lambda(Problem& P, const gsx_lm_params& p, const double* sqrtHD, LMTrace& tr,
                       gsx_lm_result* res) {
  double t0 = now_s();
  vector<LinFactor> damped;
  build_damped(P, P.linear, P.lm_lambda, p.diagonal_damping != 0, sqrtHD, damped);
  P.t[1] += now_s() - t0;
  double modelFidelity = 0.0;
  bool step_is_successful = false, stopSearchingLambda = false;
  double newError = std::numeric_limits<double>::infinity();
  double costChange = 0.0;
  Vec newValues, delta;
  bool systemSolvedSuccessfully;
  try {
    solve_gfg(P, damped, delta);
    systemSolvedSuccessfully = true;
  } catch (const IndeterminantLinearSystem&) {
    systemSolvedSuccessfully = false;
    if (res) res->n_solve_failures++;
  }
  const double lambda_tried = P.lm_lambda;
  if (systemSolvedSuccessfully) {
    t0 = now_s();
    const double oldLinearizedError = linear_error(P, P.linear, nullptr);
    const double newlinearizedError = linear_error(P, P.linear, delta.data());
    P.t[4] += now_s() - t0;
    const double linearizedCostChange = oldLinearizedError - newlinearizedError;
    if (linearizedCostChange >= 0) {
      t0 = now_s();
      newValues.resize(P.state_size);
      retract(P, P.values.data(), delta.data(), newValues.data());
      P.t[5] += now_s() - t0;
      t0 = now_s();
      newError = graph_error(P, newValues.data());
      P.t[6] += now_s() - t0;
      costChange = P.lm_error - newError;
      if (linearizedCostChange > std::numeric_limits<double>::epsilon() * oldLinearizedError) {
        modelFidelity = costChange / linearizedCostChange;
        step_is_successful = modelFidelity > p.min_model_fidelity;
      }
      const double minAbsoluteTolerance = p.relative_error_tol * P.lm_error;
      if (std::abs(costChange) < minAbsoluteTolerance) stopSearchingLambda = true;
    }
  }
  if (p.verbosity >= 1)
    std::printf("%4d %12.6g %12.2e %10.2e %6d\n", P.lm_iterations, newError, costChange, P.lm_lambda,
                (int)systemSolvedSuccessfully);
  if (step_is_successful) {
    // decreaseLambda — LevenbergMarquardtState.h:82-94
    double newLambda = P.lm_lambda, newFactor = P.lm_factor;
    if (p.use_fixed_lambda_factor) {
      newLambda /= P.lm_factor;
    } else {
      newLambda *= std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * modelFidelity - 1.0, 3));
      newFactor = 2.0 * P.lm_factor;
    }
    newLambda = std::max(p.lambda_lower_bound, newLambda);
    P.values = newValues;
    P.lm_error = newError;
    P.lm_lambda = newLambda;
    P.lm_factor = newFactor;
    P.lm_iterations += 1;
    P.lm_inner += 1;
    tr.push(newError, lambda_tried, 1);
    return true;
  } else if (!stopSearchingLambda) {
    // increaseLambda — LevenbergMarquardtState.h:70-76
    P.lm_lambda *= P.lm_factor;
    P.lm_inner += 1;
    if (!p.use_fixed_lambda_factor) P.lm_factor *= 2.0;
    tr.push(newError, lambda_tried, systemSolvedSuccessfully ? 0 : -1);
    if (P.lm_lambda >= p.lambda_upper_bound) return true;  // giving up
    return false;
  } else {
    tr.push(newError, lambda_tried, 0);
    return true;
  }
}

// LevenbergMarquardtOptimizer::iterate — LevenbergMarquardtOptimizer.cpp:273-308.
static void lm_iterate(Problem& P, const gsx_lm_params& p, LMTrace& tr, gsx_lm_result* res) {
  linearize(P);
  Vec sqrtHD;
  if (p.diagonal_damping) {
    sqrtHD.resize(P.tan_size);
    hessian_diagonal(P, P.linear, sqrtHD.data());
    for (double& v : sqrtHD) v = std::sqrt(std::min(std::max(v, p.min_diagonal), p.max_diagonal));
  }
  while (!try_lambda(P, p, sqrtHD.data(), tr, res)) {
  }
}

// checkConvergence — gtsam/nonlinear/NonlinearOptimizer.cpp:182-231.
static bool check_convergence(double relTol, double absTol, double errTol, double currentError,
                              double newError) {
  if (newError <= errTol) return true;
  const double absoluteDecrease = currentError - newError;
  const double relativeDecrease = absoluteDecrease / currentError;
  return (relTol && (relativeDecrease <= relTol)) || (absoluteDecrease <= absTol);
}

static void lm_reset(Problem& P, const gsx_lm_params& p) {
  // LevenbergMarquardtOptimizer ctor — LevenbergMarquardtOptimizer.cpp:47-53
  double t0 = now_s();
  P.lm_error = graph_error(P, P.values.data());
  P.t[6] += now_s() - t0;
  P.lm_lambda = p.lambda_initial;
  P.lm_factor = p.lambda_factor;
  P.lm_iterations = 0;
  P.lm_inner = 0;
}

// NonlinearOptimizer::defaultOptimize — gtsam/nonlinear/NonlinearOptimizer.cpp:62-117.
static void lm_optimize(Problem& P, const gsx_lm_params& p, gsx_lm_result* res) {
  lm_reset(P, p);
  LMTrace tr{res};
  if (res) {
    res->initial_error = P.lm_error;
    res->trace_len = 0;
    res->n_solve_failures = 0;
  }
  double currentError = P.lm_error;
  if (!(currentError <= p.error_tol) && !(P.lm_iterations >= p.max_iterations)) {
    double newError = currentError;
    do {
      currentError = newError;
      lm_iterate(P, p, tr, res);
      newError = P.lm_error;
    } while (P.lm_iterations < p.max_iterations &&
             !check_convergence(p.relative_error_tol, p.absolute_error_tol, p.error_tol, currentError,
                                newError) &&
             std::isfinite(currentError));
  }
  if (res) {
    res->final_error = P.lm_error;
    res->final_lambda = P.lm_lambda;
    res->iterations = P.lm_iterations;
    res->inner_iterations = P.lm_inner;
  }
}

// GaussNewtonOptimizer::iterate — gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66.
static void gn_optimize(Problem& P, int maxIter, double relTol, double absTol, double errTol,
                        gsx_lm_result* res) {
  P.lm_error = graph_error(P, P.values.data());
  P.lm_iterations = 0;
  LMTrace tr{res};
  if (res) {
    res->initial_error = P.lm_error;
    res->trace_len = 0;
    res->n_solve_failures = 0;
  }
  double currentError = P.lm_error;
  if (!(currentError <= errTol) && maxIter > 0) {
    double newError = currentError;
    do {
      currentError = newError;
      linearize(P);
      Vec delta;
      solve_gfg(P, P.linear, delta);
      Vec nv(P.state_size);
      retract(P, P.values.data(), delta.data(), nv.data());
      P.values = nv;
      P.lm_error = graph_error(P, P.values.data());
      P.lm_iterations++;
      newError = P.lm_error;
      tr.push(newError, 0.0, 1);
    } while (P.lm_iterations < maxIter && !check_convergence(relTol, absTol, errTol, currentError, newError) &&
             std::isfinite(currentError));
  }
  if (res) {
    res->final_error = P.lm_error;
    res->final_lambda = 0;
    res->iterations = P.lm_iterations;
    res->inner_iterations = P.lm_iterations;
  }
}

// ---------------------------------------------------------------------------
// Dogleg — gtsam/nonlinear/DoglegOptimizer.cpp:84-121, DoglegOptimizerImpl.h:137-252, DoglegOptimizerImpl.cpp:26-86.
// The Bayes tree of the undamped linearization is used the way the reference uses it: as a GaussianFactorGraph of
// its conditionals (GaussianBayesTree.cpp:73-91), every conditional one unit-noise Jacobian row block [R S | d].
// ---------------------------------------------------------------------------
static void bt_row_block(const Problem& P, const Conditional& cd, const double* x, Vec& out) {  // [R S] x
  out.assign(cd.nf, 0.0);
  int col = 0;
  auto add = [&](int v) {
    const double* xv = x + P.tan_off[v];
    for (int c = 0; c < P.dims[v]; ++c, ++col)
      for (int r = 0; r < cd.nf; ++r) out[r] += cd.RSd[(size_t)col * cd.nf + r] * xv[c];
  };
  for (int v : cd.frontals) add(v);
  for (int v : cd.parents) add(v);
}
static double bt_error(const Problem& P, const double* x) {  // GaussianFactorGraph::error of the conditionals
  double total = 0;
  Vec e;
  for (const Conditional& cd : P.bayes_tree) {
    bt_row_block(P, cd, x, e);
    double s = 0;
    for (int r = 0; r < cd.nf; ++r) {
      const double d = e[r] - cd.RSd[(size_t)(cd.ncols - 1) * cd.nf + r];
      s += d * d;
    }
    total += 0.5 * s;
  }
  return total;
}
static void bt_gradient_at_zero(const Problem& P, Vec& g) {  // -sum [R S]' d  (GaussianFactorGraph.cpp:360-378)
  g.assign(P.tan_size, 0.0);
  for (const Conditional& cd : P.bayes_tree) {
    const double* d = &cd.RSd[(size_t)(cd.ncols - 1) * cd.nf];
    int col = 0;
    auto add = [&](int v) {
      double* gv = g.data() + P.tan_off[v];
      for (int c = 0; c < P.dims[v]; ++c, ++col)
        for (int r = 0; r < cd.nf; ++r) gv[c] -= cd.RSd[(size_t)col * cd.nf + r] * d[r];
    };
    for (int v : cd.frontals) add(v);
    for (int v : cd.parents) add(v);
  }
}
static double vdot(const Vec& a, const Vec& b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}
// DoglegOptimizerImpl::ComputeBlend — DoglegOptimizerImpl.cpp:54-86
static void compute_blend(double delta, const Vec& xu, const Vec& xn, Vec& out) {
  const double un = vdot(xu, xn), uu = vdot(xu, xu), nn = vdot(xn, xn);
  const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - delta * delta;
  const double sq = std::sqrt(b * b - 4 * a * c);
  const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
  const double eps = std::numeric_limits<double>::epsilon();
  const double tau = (-eps <= tau1 && tau1 <= 1.0 + eps) ? tau1 : tau2;
  out.resize(xu.size());
  for (size_t i = 0; i < xu.size(); ++i) out[i] = (1. - tau) * xu[i] + tau * xn[i];
}
// DoglegOptimizerImpl::ComputeDoglegPoint — DoglegOptimizerImpl.cpp:26-51
static void compute_dogleg_point(double delta, const Vec& xu, const Vec& xn, Vec& out) {
  const double deltaSq = delta * delta, uu = vdot(xu, xu), nn = vdot(xn, xn);
  if (deltaSq < uu) {
    const double f = std::sqrt(deltaSq / uu);
    out.resize(xu.size());
    for (size_t i = 0; i < xu.size(); ++i) out[i] = f * xu[i];
  } else if (deltaSq < nn) {
    compute_blend(delta, xu, xn, out);
  } else {
    out = xn;
  }
}
// DoglegOptimizer::iterate with ONE_STEP_PER_ITERATION + NonlinearOptimizer::defaultOptimize
static void dogleg_optimize(Problem& P, double deltaInitial, int maxIter, double relTol, double absTol, double errTol,
                            gsx_lm_result* res) {
  P.lm_error = graph_error(P, P.values.data());
  P.lm_iterations = 0;
  double delta = deltaInitial;
  LMTrace tr{res};
  if (res) {
    res->initial_error = P.lm_error;
    res->trace_len = 0;
    res->n_solve_failures = 0;
  }
  double currentError = P.lm_error;
  if (!(currentError <= errTol) && maxIter > 0) {
    double newError = currentError;
    do {
      currentError = newError;
      linearize(P);
      Vec dx_n;
      solve_gfg(P, P.linear, dx_n);  // Bayes tree + Newton point
      Vec dx_u;
      bt_gradient_at_zero(P, dx_u);  // optimizeGradientSearch — GaussianFactorGraph.cpp:381-407
      {
        const double gg = vdot(dx_u, dx_u);
        double rg = 0;
        Vec e;
        for (const Conditional& cd : P.bayes_tree) {
          bt_row_block(P, cd, dx_u.data(), e);
          for (double q : e) rg += q * q;
        }
        const double step = -gg / rg;
        for (double& q : dx_u) q *= step;
      }
      const double f_error = P.lm_error;
      Vec zero(P.tan_size, 0.0), dx_d, nv(P.state_size);
      const double M_error = bt_error(P, zero.data());
      double result_f = f_error;
      bool stay = true;
      while (stay) {  // DoglegOptimizerImpl::Iterate, mode ONE_STEP_PER_ITERATION
        compute_dogleg_point(delta, dx_u, dx_n, dx_d);
        retract(P, P.values.data(), dx_d.data(), nv.data());
        result_f = graph_error(P, nv.data());
        const double new_M = bt_error(P, dx_d.data());
        const double rho = (std::abs(f_error - result_f) < 1e-15 || std::abs(M_error - new_M) < 1e-15)
                               ? 0.5
                               : (f_error - result_f) / (M_error - new_M);
        if (rho >= 0.75) {
          delta = std::max(delta, 3.0 * std::sqrt(vdot(dx_d, dx_d)));
          stay = false;
        } else if (rho >= 0.25) {
          stay = false;
        } else if (rho >= 0.0) {
          if (delta > 1e-5) delta *= 0.5;
          stay = false;
        } else {  // f increased (also NaN): shrink until it does not
          if (delta > 1e-5) {
            delta *= 0.5;
            stay = true;
          } else {
            std::fill(dx_d.begin(), dx_d.end(), 0.0);
            result_f = f_error;
            stay = false;
          }
        }
      }
      retract(P, P.values.data(), dx_d.data(), nv.data());
      P.values = nv;
      P.lm_error = result_f;
      P.lm_iterations++;
      newError = P.lm_error;
      tr.push(newError, delta, 1);  // the trace's "lambda" column carries the trust-region radius
    } while (P.lm_iterations < maxIter && !check_convergence(relTol, absTol, errTol, currentError, newError) &&
             std::isfinite(currentError));
  }
  if (res) {
    res->final_error = P.lm_error;
    res->final_lambda = delta;
    res->iterations = P.lm_iterations;
    res->inner_iterations = P.lm_iterations;
  }
}

static Problem* build_problem(const gsx_problem_desc* d, std::string& err) {
  auto P = std::make_unique<Problem>();
  P->n_vars = d->n_vars;
  P->keys.assign(d->var_keys, d->var_keys + d->n_vars);
  P->types.assign(d->var_types, d->var_types + d->n_vars);
  P->dims.assign(d->var_dims, d->var_dims + d->n_vars);
  P->state_off.resize(d->n_vars);
  P->tan_off.resize(d->n_vars);
  for (int v = 0; v < d->n_vars; ++v) {
    if (v > 0 && !(P->keys[v] > P->keys[v - 1])) {
      err = "var_keys must be strictly ascending";
      return nullptr;
    }
    const int expect[4] = {P->dims[v], 3, 6, 9};
    if (P->types[v] < 0 || P->types[v] > 3 || P->dims[v] != expect[P->types[v]]) {
      err = "bad variable type/dim";
      return nullptr;
    }
    P->state_off[v] = (int)P->state_size;
    P->tan_off[v] = (int)P->tan_size;
    P->state_size += state_dim(P->types[v], P->dims[v]);
    P->tan_size += P->dims[v];
  }
  P->values.assign(P->state_size, 0.0);
  P->factors.resize(d->n_factors);
  P->jac_off.resize(d->n_factors + 1);
  for (int i = 0; i < d->n_factors; ++i) {
    Factor& f = P->factors[i];
    f.type = d->f_type[i];
    f.rows = d->f_rows[i];
    f.noise_kind = d->f_noise_kind[i];
    f.vars.assign(d->f_vars + d->f_key_ptr[i], d->f_vars + d->f_key_ptr[i + 1]);
    f.meas.assign(d->meas + d->f_meas_ptr[i], d->meas + d->f_meas_ptr[i + 1]);
    f.noise.assign(d->noise + d->f_noise_ptr[i], d->noise + d->f_noise_ptr[i + 1]);
    if (f.noise_kind == GSX_NOISE_CONSTRAINED) {  // sigmas, then mu
      if ((int)f.noise.size() != 2 * f.rows) {
        err = "constrained noise model: 2 m parameters";
        return nullptr;
      }
      f.mu.assign(f.noise.begin() + f.rows, f.noise.end());
      f.noise.resize(f.rows);
    } else if (f.noise_kind == GSX_NOISE_DIAGONAL) {
      bool any = false;
      for (double sg : f.noise) any = any || sg == 0.0;
      if (any) f.mu.assign(f.rows, 1000.0);  // Constrained(sigmas) — NoiseModel.cpp:365-368
    }
    int cols = 1;
    for (int v : f.vars) {
      if (v < 0 || v >= d->n_vars) {
        err = "factor variable index out of range";
        return nullptr;
      }
      cols += P->dims[v];
    }
    P->jac_off[i] = P->jac_size;
    P->jac_size += (int64_t)f.rows * cols;
  }
  P->jac_off[d->n_factors] = P->jac_size;
  return P.release();
}

}  // namespace orc

// ===========================================================================
// C API (mirrors include/gsx.h with the orc_ prefix)
// ===========================================================================
using orc::Problem;
extern "C" {

int orc_create(const gsx_problem_desc* d, void** out) {
  std::string err;
  Problem* P = orc::build_problem(d, err);
  if (!P) {
    std::fprintf(stderr, "orc_create: %s\n", err.c_str());
    return GSX_E_INVALID;
  }
  *out = P;
  return GSX_OK;
}
int orc_destroy(void* h) {
  delete (Problem*)h;
  return GSX_OK;
}
int64_t orc_state_size(void* h) { return ((Problem*)h)->state_size; }
int64_t orc_tangent_size(void* h) { return ((Problem*)h)->tan_size; }
int64_t orc_jacobian_size(void* h) { return ((Problem*)h)->jac_size; }

int orc_set_ordering(void* h, const uint64_t* keys, int32_t n) {
  Problem& P = *(Problem*)h;
  if (n != P.n_vars) return GSX_E_BAD_ORDERING;
  std::map<uint64_t, int> idx;
  for (int v = 0; v < P.n_vars; ++v) idx[P.keys[v]] = v;
  std::vector<int> ord(n);
  std::vector<char> seen(n, 0);
  for (int i = 0; i < n; ++i) {
    auto it = idx.find(keys[i]);
    if (it == idx.end() || seen[it->second]) return GSX_E_BAD_ORDERING;
    seen[it->second] = 1;
    ord[i] = it->second;
  }
  P.ordering = ord;
  P.has_ordering = true;
  return GSX_OK;
}
int orc_set_values(void* h, const double* packed, int64_t n) {
  Problem& P = *(Problem*)h;
  if (n != P.state_size) return GSX_E_INVALID;
  P.values.assign(packed, packed + n);
  P.linearized = false;
  return GSX_OK;
}
int orc_get_values(void* h, double* packed, int64_t n) {
  Problem& P = *(Problem*)h;
  if (n != P.state_size) return GSX_E_INVALID;
  std::memcpy(packed, P.values.data(), n * sizeof(double));
  return GSX_OK;
}
int orc_error(void* h, double* out) {
  Problem& P = *(Problem*)h;
  const double t0 = orc::now_s();
  *out = orc::graph_error(P, P.values.data());
  P.t[6] += orc::now_s() - t0;
  return GSX_OK;
}
int orc_linearize(void* h) {
  orc::linearize(*(Problem*)h);
  return GSX_OK;
}
int orc_get_jacobians(void* h, double* out, int64_t n) {
  Problem& P = *(Problem*)h;
  if (!P.linearized) return GSX_E_STATE;
  if (n != P.jac_size) return GSX_E_INVALID;
  for (size_t i = 0; i < P.linear.size(); ++i)
    std::memcpy(out + P.jac_off[i], P.linear[i].M.data(), P.linear[i].M.size() * sizeof(double));
  return GSX_OK;
}
int orc_hessian_diagonal(void* h, double* out, int64_t n) {
  Problem& P = *(Problem*)h;
  if (!P.linearized) return GSX_E_STATE;
  if (n != P.tan_size) return GSX_E_INVALID;
  orc::hessian_diagonal(P, P.linear, out);
  return GSX_OK;
}
int orc_solve(void* h, double lambda, int32_t diagonal_damping, double min_diagonal, double max_diagonal,
              double* delta_out, int64_t n, uint64_t* bad_key) {
  Problem& P = *(Problem*)h;
  if (!P.linearized || !P.has_ordering) return GSX_E_STATE;
  if (delta_out && n != P.tan_size) return GSX_E_INVALID;
  orc::Vec sqrtHD;
  if (diagonal_damping) {
    sqrtHD.resize(P.tan_size);
    orc::hessian_diagonal(P, P.linear, sqrtHD.data());
    for (double& v : sqrtHD) v = std::sqrt(std::min(std::max(v, min_diagonal), max_diagonal));
  }
  double t0 = orc::now_s();
  std::vector<orc::LinFactor> damped;
  orc::build_damped(P, P.linear, lambda, diagonal_damping != 0, sqrtHD.data(), damped);
  P.t[1] += orc::now_s() - t0;
  try {
    orc::solve_gfg(P, damped, P.delta);
  } catch (const orc::IndeterminantLinearSystem& e) {
    if (bad_key) *bad_key = e.key;
    P.solved = false;
    return GSX_E_INDETERMINATE;
  }
  P.solved = true;
  if (delta_out) std::memcpy(delta_out, P.delta.data(), n * sizeof(double));
  return GSX_OK;
}
int orc_linear_error(void* h, double* e0, double* ed) {
  Problem& P = *(Problem*)h;
  if (!P.linearized) return GSX_E_STATE;
  const double t0 = orc::now_s();
  if (e0) *e0 = orc::linear_error(P, P.linear, nullptr);
  if (ed) {
    if (!P.solved) return GSX_E_STATE;
    *ed = orc::linear_error(P, P.linear, P.delta.data());
  }
  P.t[4] += orc::now_s() - t0;
  return GSX_OK;
}
int orc_retract(void* h, const double* delta, int64_t n, int32_t commit, double* trial_error) {
  Problem& P = *(Problem*)h;
  const double* d = delta;
  if (!d) {
    if (!P.solved) return GSX_E_STATE;
    d = P.delta.data();
  } else if (n != P.tan_size) {
    return GSX_E_INVALID;
  }
  double t0 = orc::now_s();
  orc::Vec nv(P.state_size);
  orc::retract(P, P.values.data(), d, nv.data());
  P.t[5] += orc::now_s() - t0;
  if (trial_error) {
    t0 = orc::now_s();
    *trial_error = orc::graph_error(P, nv.data());
    P.t[6] += orc::now_s() - t0;
  }
  if (commit) {
    P.values = nv;
    P.linearized = false;
  }
  return GSX_OK;
}
int orc_lm_optimize(void* h, const gsx_lm_params* p, gsx_lm_result* r) {
  Problem& P = *(Problem*)h;
  if (!P.has_ordering) return GSX_E_STATE;
  orc::lm_optimize(P, *p, r);
  return GSX_OK;
}
int orc_lm_reset(void* h, const gsx_lm_params* p) {
  orc::lm_reset(*(Problem*)h, *p);
  return GSX_OK;
}
int orc_lm_iterate(void* h, const gsx_lm_params* p, double* error, double* lambda) {
  Problem& P = *(Problem*)h;
  if (!P.has_ordering) return GSX_E_STATE;
  orc::LMTrace tr{nullptr};
  orc::lm_iterate(P, *p, tr, nullptr);
  if (error) *error = P.lm_error;
  if (lambda) *lambda = P.lm_lambda;
  return GSX_OK;
}
int orc_gn_optimize(void* h, int32_t max_iterations, double rel, double abs_, double errtol,
                    gsx_lm_result* r) {
  Problem& P = *(Problem*)h;
  if (!P.has_ordering) return GSX_E_STATE;
  try {
    orc::gn_optimize(P, max_iterations, rel, abs_, errtol, r);
  } catch (const orc::IndeterminantLinearSystem&) {
    return GSX_E_INDETERMINATE;
  }
  return GSX_OK;
}
// Marginals::marginalCovariance — gtsam/nonlinear/Marginals.cpp:107-136: the inverse of the variable's marginal
// information; restated densely (the block of H^-1, H = sum A'A of the current linearization) — for test-sized problems.
// joint version: D x D row-major, blocks in the order of `keys` (Marginals::jointMarginalCovariance,
// gtsam/nonlinear/Marginals.cpp:138-189, returns them sorted by key)
int orc_joint_marginal_covariance(void* h, const uint64_t* keys, int32_t n_keys, double* out, int64_t n_out) {
  Problem& P = *(Problem*)h;
  if (!P.linearized) return GSX_E_STATE;
  std::vector<int64_t> idx;  // global tangent indices of the requested scalars
  for (int q = 0; q < n_keys; ++q) {
    int v = -1;
    for (int i = 0; i < P.n_vars; ++i)
      if (P.keys[i] == keys[q]) v = i;
    if (v < 0) return GSX_E_INVALID;
    for (int c = 0; c < P.dims[v]; ++c) idx.push_back(P.tan_off[v] + c);
  }
  const int64_t D = (int64_t)idx.size();
  const int64_t N = P.tan_size;
  if (n_out != D * D || N > 6000) return GSX_E_INVALID;
  orc::Vec H((size_t)N * N, 0.0);
  for (const orc::LinFactor& L : P.linear) {
    const int m = L.rows;
    std::vector<int> cols;  // global tangent index of every Jacobian column
    for (size_t k = 0; k < L.vars.size(); ++k)
      for (int c = 0; c < L.dims[k]; ++c) cols.push_back(P.tan_off[L.vars[k]] + c);
    for (size_t a = 0; a < cols.size(); ++a)
      for (size_t b = 0; b < cols.size(); ++b) {
        double s = 0;
        for (int r = 0; r < m; ++r) s += L.M[a * m + r] * L.M[b * m + r];
        H[(size_t)cols[a] * N + cols[b]] += s;
      }
  }
  // in-place Cholesky H = G G' (lower), then solve for the unit columns of the requested scalars
  for (int64_t j = 0; j < N; ++j) {
    double s = H[j * N + j];
    for (int64_t k = 0; k < j; ++k) s -= H[j * N + k] * H[j * N + k];
    if (!(s > 0)) return GSX_E_INDETERMINATE;
    const double g = std::sqrt(s);
    H[j * N + j] = g;
    for (int64_t i = j + 1; i < N; ++i) {
      double t = H[i * N + j];
      for (int64_t k = 0; k < j; ++k) t -= H[i * N + k] * H[j * N + k];
      H[i * N + j] = t / g;
    }
  }
  for (int64_t c = 0; c < D; ++c) {
    orc::Vec x(N, 0.0);
    x[idx[c]] = 1.0;
    for (int64_t i = 0; i < N; ++i) {  // G z = e
      double t = x[i];
      for (int64_t k = 0; k < i; ++k) t -= H[i * N + k] * x[k];
      x[i] = t / H[i * N + i];
    }
    for (int64_t i = N - 1; i >= 0; --i) {  // G' x = z
      double t = x[i];
      for (int64_t k = i + 1; k < N; ++k) t -= H[k * N + i] * x[k];
      x[i] = t / H[i * N + i];
    }
    for (int64_t a = 0; a < D; ++a) out[a * D + c] = x[idx[a]];
  }
  return GSX_OK;
}
int orc_marginal_covariance(void* h, uint64_t key, double* out, int64_t n_out) {
  return orc_joint_marginal_covariance(h, &key, 1, out, n_out);  // (symmetric: row- and column-major agree)
}
int orc_dogleg_optimize(void* h, double delta_initial, int32_t max_iterations, double rel, double abs_, double errtol,
                        gsx_lm_result* r) {
  Problem& P = *(Problem*)h;
  if (!P.has_ordering) return GSX_E_STATE;
  try {
    orc::dogleg_optimize(P, delta_initial, max_iterations, rel, abs_, errtol, r);
  } catch (const orc::IndeterminantLinearSystem&) {
    return GSX_E_INDETERMINATE;
  }
  return GSX_OK;
}
// DoglegOptimizerImpl::ComputeDoglegPoint on plain vectors (known-answer tests)
int orc_dogleg_point(double delta, const double* xu, const double* xn, int64_t n, double* out) {
  orc::Vec u(xu, xu + n), nn(xn, xn + n), o;
  orc::compute_dogleg_point(delta, u, nn, o);
  std::copy(o.begin(), o.end(), out);
  return GSX_OK;
}
int orc_dogleg_blend(double delta, const double* xu, const double* xn, int64_t n, double* out) {
  orc::Vec u(xu, xu + n), nn(xn, xn + n), o;
  orc::compute_blend(delta, u, nn, o);
  std::copy(o.begin(), o.end(), out);
  return GSX_OK;
}
// Bayes tree of the last solve: cliques in elimination post-order.
int orc_get_tree(void* h, int32_t* n_fronts, int64_t* n_sep_total, int32_t* parent, int32_t* frontal_ptr,
                 int32_t* frontal_vars, int32_t* sep_ptr, int32_t* sep_vars) {
  Problem& P = *(Problem*)h;
  const int n = (int)P.bayes_tree.size();
  if (n_fronts) *n_fronts = n;
  if (n_sep_total) {
    int64_t t = 0;
    for (const orc::Conditional& cd : P.bayes_tree) t += (int64_t)cd.parents.size();
    *n_sep_total = t;
  }
  if (!parent) return GSX_OK;
  int fp = 0, sp = 0;
  for (int c = 0; c < n; ++c) {
    const orc::Conditional& cd = P.bayes_tree[c];
    parent[c] = cd.parent_clique;
    frontal_ptr[c] = fp;
    sep_ptr[c] = sp;
    for (int v : cd.frontals) frontal_vars[fp++] = v;
    for (int v : cd.parents) sep_vars[sp++] = v;
  }
  frontal_ptr[n] = fp;
  sep_ptr[n] = sp;
  return GSX_OK;
}
// [R S d] of clique c (nf x ncols col-major); pass NULL to query sizes.
int orc_get_conditional(void* h, int32_t c, int32_t* nf, int32_t* ncols, double* out) {
  Problem& P = *(Problem*)h;
  if (c < 0 || c >= (int)P.bayes_tree.size()) return GSX_E_INVALID;
  const orc::Conditional& cd = P.bayes_tree[c];
  if (nf) *nf = cd.nf;
  if (ncols) *ncols = cd.ncols;
  if (out) std::memcpy(out, cd.RSd.data(), cd.RSd.size() * sizeof(double));
  return GSX_OK;
}
// Constrained::QR on a dense [A b] (m x (n + 1) ROW-major, overwritten by the rows of [R d]); lead / precisions: min(m, n)
// entries (infinity = hard constraint).  For the known-answer tests of gtsam/linear/tests/testNoiseModel.cpp:225-390.
int orc_constrained_qr(double* Ab, int32_t m, int32_t n, const double* sigmas, int32_t* lead, double* precisions,
                       int32_t* rank) {
  orc::Vec M(Ab, Ab + (size_t)m * (n + 1)), sg(sigmas, sigmas + m), pr;
  std::vector<int> ld;
  orc::constrained_qr(M, m, n, sg, ld, pr);
  std::copy(M.begin(), M.end(), Ab);
  *rank = (int32_t)ld.size();
  for (size_t i = 0; i < ld.size(); ++i) {
    lead[i] = ld[i];
    precisions[i] = pr[i];
  }
  return GSX_OK;
}
int orc_cholesky_partial(double* abc, int32_t n, int32_t nfrontal, int32_t* ok) {
  *ok = orc::cholesky_partial(abc, n, nfrontal) ? 1 : 0;
  return GSX_OK;
}
// t[8]: linearize, damp, eliminate, backsub, linerr, retract, error, symbolic (seconds, accumulated);
// tree[6]: flops, bytes, cliques, depth, max f, max s of the last elimination.
int orc_get_timing(void* h, double* t, double* tree) {
  Problem& P = *(Problem*)h;
  if (t) std::memcpy(t, P.t, sizeof(P.t));
  if (tree) {
    tree[0] = P.tree_flops; tree[1] = P.tree_bytes; tree[2] = (double)P.tree_cliques;
    tree[3] = (double)P.tree_depth; tree[4] = (double)P.tree_maxf; tree[5] = (double)P.tree_maxs;
  }
  return GSX_OK;
}
int orc_reset_timing(void* h) {
  Problem& P = *(Problem*)h;
  for (double& x : P.t) x = 0;
  return GSX_OK;
}
int64_t orc_n_cheirality(void* h) { return ((Problem*)h)->n_cheirality; }
// host threads of linearize / elimination / back-substitution (1 = serial, the reference without TBB)
int orc_set_threads(void* h, int32_t n) {
  if (!h || n < 1) return GSX_E_INVALID;
  ((Problem*)h)->n_threads = n;
  return GSX_OK;
}

}  // extern "C"
