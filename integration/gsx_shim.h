// gsx_shim.h — the GTSAM-side binding of libgsx.so (include/gsx.h).
//
// This file is compiled where GTSAM's headers exist — a GTSAM checkout or install — NOT in this repository (GTSAM needs
// its cmake-generated config.h / dllexport.h, which the build image lacks).  It is written against the
// reference tree's API (paths relative to /root/reference/):
//
//   L1  GsxLevenbergMarquardtOptimizer::solve        overrides NonlinearOptimizer::solve
//       (gtsam/nonlinear/NonlinearOptimizer.h:129-130, called from LevenbergMarquardtOptimizer.cpp:156; upstream
//       precedent for overriding it: tests/testNonlinearOptimizer.cpp:507-551).  The damped linear graph of every LM
//       trial has ONE structure (same keys, dims, ordering); the handle — symbolic analysis, device tables — is built
//       at the first call and every later call uploads only the [A b] numbers: gsx_solve_gfg_h.
//   L2  GsxLevenbergMarquardtOptimizer::optimize     overrides NonlinearOptimizer::optimize (NonlinearOptimizer.h:98):
//       the nonlinear graph is lowered once by dynamic_cast (the table of INTEGRATION.md) and the whole LM run happens
//       on the device (gsx_lm_optimize); factor types outside the table go through the S5 fallback: the shim calls
//       their own linearize() on the CPU at every linearization point and refreshes their blocks in place
//       (gsx_set_block_jacobians; gtsam/nonlinear/NonlinearFactor.h:145-146), driving the trials itself with the
//       library's decision function (gsx_lm_decide).
//
// Usage: replace `LevenbergMarquardtOptimizer` by `gsx::GsxLevenbergMarquardtOptimizer` in
// examples/SFMExample_bal.cpp:79-86, examples/SFMExample_bal_COLAMD_METIS.cpp:83-117, timing/timeSFMBAL.h:64-96,
// tests/testGeneralSFMFactorB.cpp:44-63, examples/Pose2SLAMExample_g2o.cpp, examples/Pose3SLAMExample_g2o.cpp; link -lgsx.
#pragma once

#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/Cal3_S2.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/geometry/Pose2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/linear/GaussianFactorGraph.h>
#include <gtsam/linear/JacobianFactor.h>
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/linear/VectorValues.h>
#include <gtsam/linear/linearExceptions.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/PriorFactor.h>
#include <gtsam/nonlinear/internal/LevenbergMarquardtState.h>
#include <gtsam/sam/BearingRangeFactor.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>
#include <gtsam/slam/ProjectionFactor.h>

#include <map>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "gsx.h"

namespace gsx {

using gtsam::Key;

// ---- a gsx_problem_desc under construction ---------------------------------------------------------------------------
struct Lowered {
  std::vector<uint64_t> keys;                   // ascending (Values / KeySet order)
  std::map<Key, int> index;
  std::vector<int32_t> types, dims, f_type, f_rows, f_key_ptr{0}, f_vars, f_noise_kind;
  std::vector<int64_t> f_meas_ptr{0}, f_noise_ptr{0};
  std::vector<double> meas, noise;
  std::vector<size_t> fallback;                 // graph indices of the factors lowered as GSX_F_LINEAR slots, in slot order
  int first_slot = 0;                           // the slots are the last factors of the description

  void add_factor(int type, int rows, const std::vector<Key>& ks, const std::vector<double>& m, int noise_kind,
                  const std::vector<double>& noise_params) {
    f_type.push_back(type);
    f_rows.push_back(rows);
    for (Key k : ks) f_vars.push_back(index.at(k));
    f_key_ptr.push_back((int32_t)f_vars.size());
    meas.insert(meas.end(), m.begin(), m.end());
    f_meas_ptr.push_back((int64_t)meas.size());
    f_noise_kind.push_back(noise_kind);
    noise.insert(noise.end(), noise_params.begin(), noise_params.end());
    f_noise_ptr.push_back((int64_t)noise.size());
  }
  gsx_problem_desc desc() const {
    return gsx_problem_desc{(int32_t)keys.size(), keys.data(), types.data(), dims.data(), (int32_t)f_type.size(),
                            f_type.data(), f_rows.data(), f_key_ptr.data(), f_vars.data(), f_meas_ptr.data(), meas.data(),
                            f_noise_kind.data(), f_noise_ptr.data(), noise.data()};
  }
};

// ---- state layouts of include/gsx.h ------------------------------------------------------------------------------------
inline void pack(const gtsam::Pose2& p, std::vector<double>& out) { out.insert(out.end(), {p.x(), p.y(), p.theta()}); }
inline void pack(const gtsam::Pose3& p, std::vector<double>& out) {
  const gtsam::Matrix3 R = p.rotation().matrix();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out.push_back(R(r, c));  // row-major
  const gtsam::Point3 t = p.translation();
  out.insert(out.end(), {t.x(), t.y(), t.z()});
}
inline void pack(const gtsam::Vector& v, std::vector<double>& out) { out.insert(out.end(), v.data(), v.data() + v.size()); }
inline void pack(const gtsam::PinholeCamera<gtsam::Cal3Bundler>& c, std::vector<double>& out) {
  pack(c.pose(), out);
  const gtsam::Cal3Bundler& K = c.calibration();
  out.insert(out.end(), {K.fx(), K.k1(), K.k2(), K.px(), K.py()});
}

// noise model -> (kind, parameters); Robust(Huber / Tukey / Cauchy) is or-ed onto the base kind (include/gsx.h)
inline bool lower_noise(const gtsam::SharedNoiseModel& model, int rows, int& kind, std::vector<double>& params) {
  using namespace gtsam::noiseModel;
  params.clear();
  std::shared_ptr<Base> base = model;
  int robust = 0;
  double k = 0;
  if (auto r = std::dynamic_pointer_cast<Robust>(model)) {
    base = r->noise();
    if (auto h = std::dynamic_pointer_cast<mEstimator::Huber>(r->robust())) robust = GSX_NOISE_ROBUST_HUBER, k = h->modelParameter();
    else if (auto t = std::dynamic_pointer_cast<mEstimator::Tukey>(r->robust())) robust = GSX_NOISE_ROBUST_TUKEY, k = t->modelParameter();
    else if (auto c = std::dynamic_pointer_cast<mEstimator::Cauchy>(r->robust())) robust = GSX_NOISE_ROBUST_CAUCHY, k = c->modelParameter();
    else return false;
  }
  if (!base || base->isUnit()) kind = GSX_NOISE_UNIT;
  else if (auto iso = std::dynamic_pointer_cast<Isotropic>(base)) kind = GSX_NOISE_ISOTROPIC, params = {iso->sigma()};
  else if (auto dg = std::dynamic_pointer_cast<Diagonal>(base)) {
    const gtsam::Vector s = dg->sigmas();
    kind = GSX_NOISE_DIAGONAL, params.assign(s.data(), s.data() + s.size());
    if (auto con = std::dynamic_pointer_cast<Constrained>(base)) {
      // zero sigmas = hard-constraint rows (constraint pivots on the device, constraint.hip), with their error weights mu
      if (robust) return false;
      for (int r = 0; r < s.size(); ++r)
        if (s[r] != 0.0 && !std::isfinite(1.0 / s[r])) return false;  // (Constrained::constrained() of a denormal sigma)
      const gtsam::Vector mu = con->mu();
      kind = GSX_NOISE_CONSTRAINED;
      params.insert(params.end(), mu.data(), mu.data() + mu.size());
    }
  } else if (auto g = std::dynamic_pointer_cast<Gaussian>(base)) {
    const gtsam::Matrix R = g->R();  // upper-triangular sqrt information, row-major in the ABI
    kind = GSX_NOISE_GAUSSIAN;
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < rows; ++c) params.push_back(R(r, c));
  } else {
    return false;
  }
  kind |= robust;
  if (robust) params.push_back(k);
  return true;
}

// ---- L2 lowering table: the factor types libgsx has kernels for ------------------------------------------------------------
// returns false for a factor outside the table (it becomes a GSX_F_LINEAR slot)
inline bool lower_factor(const gtsam::NonlinearFactor::shared_ptr& f, Lowered& L) {
  using namespace gtsam;
  int kind;
  std::vector<double> np, m;
  const std::vector<Key> ks(f->keys().begin(), f->keys().end());
  auto nm = std::dynamic_pointer_cast<NoiseModelFactor>(f);
  if (!nm) return false;
  const int rows = (int)nm->dim();
  if (!lower_noise(nm->noiseModel(), rows, kind, np)) return false;
  if (auto sfm = std::dynamic_pointer_cast<GeneralSFMFactor<PinholeCamera<Cal3Bundler>, Point3>>(f)) {
    const Point2 z = sfm->measured();
    L.add_factor(GSX_F_SFM, 2, ks, {z.x(), z.y()}, kind, np);
  } else if (auto b2 = std::dynamic_pointer_cast<BetweenFactor<Pose2>>(f)) {
    pack(b2->measured(), m), L.add_factor(GSX_F_BETWEEN, 3, ks, m, kind, np);
  } else if (auto b3 = std::dynamic_pointer_cast<BetweenFactor<Pose3>>(f)) {
    pack(b3->measured(), m), L.add_factor(GSX_F_BETWEEN, 6, ks, m, kind, np);
  } else if (auto bp = std::dynamic_pointer_cast<BetweenFactor<Point3>>(f)) {
    pack(Vector(bp->measured()), m), L.add_factor(GSX_F_BETWEEN, 3, ks, m, kind, np);
  } else if (auto p2 = std::dynamic_pointer_cast<PriorFactor<Pose2>>(f)) {
    pack(p2->prior(), m), L.add_factor(GSX_F_PRIOR, 3, ks, m, kind, np);
  } else if (auto p3 = std::dynamic_pointer_cast<PriorFactor<Pose3>>(f)) {
    pack(p3->prior(), m), L.add_factor(GSX_F_PRIOR, 6, ks, m, kind, np);
  } else if (auto pp = std::dynamic_pointer_cast<PriorFactor<Point3>>(f)) {
    pack(Vector(pp->prior()), m), L.add_factor(GSX_F_PRIOR, 3, ks, m, kind, np);
  } else if (auto pc = std::dynamic_pointer_cast<PriorFactor<PinholeCamera<Cal3Bundler>>>(f)) {
    pack(pc->prior(), m), L.add_factor(GSX_F_PRIOR, 9, ks, m, kind, np);
  } else if (auto pr = std::dynamic_pointer_cast<GenericProjectionFactor<Pose3, Point3, Cal3_S2>>(f)) {
    if (pr->body_P_sensor()) return false;
    const Cal3_S2& K = *pr->calibration();
    L.add_factor(GSX_F_PROJECTION, 2, ks, {pr->measured().x(), pr->measured().y(), K.fx(), K.fy(), K.skew(), K.px(), K.py()},
                 kind, np);
  } else if (auto br = std::dynamic_pointer_cast<BearingRangeFactor<Pose2, Point2>>(f)) {
    L.add_factor(GSX_F_BEARINGRANGE, 2, ks, {br->measured().bearing().theta(), br->measured().range()}, kind, np);
  } else {
    return false;
  }
  return true;
}

// variable table + packed Values
inline void lower_values(const gtsam::Values& values, Lowered& L, std::vector<double>& packed) {
  using namespace gtsam;
  for (const auto& kv : values) {
    const Key k = kv.key;
    L.index[k] = (int)L.keys.size();
    L.keys.push_back(k);
    if (auto p = dynamic_cast<const GenericValue<Pose2>*>(&kv.value)) L.types.push_back(GSX_VAR_POSE2), L.dims.push_back(3), pack(p->value(), packed);
    else if (auto q = dynamic_cast<const GenericValue<Pose3>*>(&kv.value)) L.types.push_back(GSX_VAR_POSE3), L.dims.push_back(6), pack(q->value(), packed);
    else if (auto c = dynamic_cast<const GenericValue<PinholeCamera<Cal3Bundler>>*>(&kv.value)) L.types.push_back(GSX_VAR_CAMERA), L.dims.push_back(9), pack(c->value(), packed);
    else if (auto x = dynamic_cast<const GenericValue<Point3>*>(&kv.value)) L.types.push_back(GSX_VAR_VECTOR), L.dims.push_back(3), pack(Vector(x->value()), packed);
    else if (auto y = dynamic_cast<const GenericValue<Point2>*>(&kv.value)) L.types.push_back(GSX_VAR_VECTOR), L.dims.push_back(2), pack(Vector(y->value()), packed);
    else if (auto v = dynamic_cast<const GenericValue<Vector>*>(&kv.value)) L.types.push_back(GSX_VAR_VECTOR), L.dims.push_back((int)v->value().size()), pack(v->value(), packed);
    else throw std::runtime_error("gsx shim: a value type libgsx has no state layout for");
  }
}

inline void check(gsx_status st, gsx_handle h, const char* what) {
  if (st != GSX_OK) throw std::runtime_error(std::string(what) + ": " + (h ? gsx_last_error(h) : "gsx error"));
}

// packed state -> Values (inverse of lower_values)
inline gtsam::Values unpack_values(const gtsam::Values& like, const std::vector<double>& packed) {
  using namespace gtsam;
  Values out;
  size_t o = 0;
  auto pose3 = [&](size_t at) {
    Matrix3 R;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R(r, c) = packed[at + 3 * r + c];
    return Pose3(Rot3(R), Point3(packed[at + 9], packed[at + 10], packed[at + 11]));
  };
  for (const auto& kv : like) {
    if (dynamic_cast<const GenericValue<Pose2>*>(&kv.value)) out.insert(kv.key, Pose2(packed[o], packed[o + 1], packed[o + 2])), o += 3;
    else if (dynamic_cast<const GenericValue<Pose3>*>(&kv.value)) out.insert(kv.key, pose3(o)), o += 12;
    else if (dynamic_cast<const GenericValue<PinholeCamera<Cal3Bundler>>*>(&kv.value)) {
      out.insert(kv.key, PinholeCamera<Cal3Bundler>(pose3(o), Cal3Bundler(packed[o + 12], packed[o + 13], packed[o + 14], packed[o + 15], packed[o + 16])));
      o += 17;
    } else if (dynamic_cast<const GenericValue<Point3>*>(&kv.value)) out.insert(kv.key, Point3(packed[o], packed[o + 1], packed[o + 2])), o += 3;
    else if (dynamic_cast<const GenericValue<Point2>*>(&kv.value)) out.insert(kv.key, Point2(packed[o], packed[o + 1])), o += 2;
    else {
      const size_t d = kv.value.dim();
      out.insert(kv.key, Vector(Eigen::Map<const Vector>(&packed[o], d)));
      o += d;
    }
  }
  return out;
}

inline gsx_lm_params lower_params(const gtsam::LevenbergMarquardtParams& p) {
  gsx_lm_params q;
  q.max_iterations = (int32_t)p.maxIterations, q.relative_error_tol = p.relativeErrorTol, q.absolute_error_tol = p.absoluteErrorTol;
  q.error_tol = p.errorTol, q.lambda_initial = p.lambdaInitial, q.lambda_factor = p.lambdaFactor;
  q.lambda_upper_bound = p.lambdaUpperBound, q.lambda_lower_bound = p.lambdaLowerBound, q.min_model_fidelity = p.minModelFidelity;
  q.diagonal_damping = p.diagonalDamping, q.use_fixed_lambda_factor = p.useFixedLambdaFactor;
  q.min_diagonal = p.minDiagonal, q.max_diagonal = p.maxDiagonal, q.verbosity = 0;
  return q;
}

// =====================================================================================================================
class GsxLevenbergMarquardtOptimizer : public gtsam::LevenbergMarquardtOptimizer {
 public:
  using gtsam::LevenbergMarquardtOptimizer::LevenbergMarquardtOptimizer;
  ~GsxLevenbergMarquardtOptimizer() override {
    if (linear_) gsx_destroy(linear_);
  }
  int device = 0;

  // ---- L1: the linear solve of one LM trial, structure cached --------------------------------------------------------
  gtsam::VectorValues solve(const gtsam::GaussianFactorGraph& gfg, const gtsam::NonlinearOptimizerParams& params) const override {
    using namespace gtsam;
    // [A b] blocks in graph order, noise folded in (JacobianFactor::whiten)
    std::vector<double> blocks;
    std::vector<JacobianFactor> white;
    std::vector<std::vector<double>> unit_sigmas;   // per factor: empty, or Constrained::unit()'s sigmas (0 = hard constraint)
    white.reserve(gfg.size());
    for (const auto& f : gfg) {
      if (!f) continue;
      auto jf = std::dynamic_pointer_cast<JacobianFactor>(f);
      unit_sigmas.emplace_back();
      if (jf && jf->get_model() && jf->get_model()->isConstrained()) {
        const Vector s = jf->get_model()->sigmas();
        for (int r = 0; r < s.size(); ++r) unit_sigmas.back().push_back(s[r] == 0.0 ? 0.0 : 1.0);
      }
      // (whiten() leaves a constraint row as it is: Constrained::WhitenInPlace, NoiseModel.cpp:456-468)
      white.push_back(jf ? jf->whiten() : JacobianFactor(*f).whiten());  // (a HessianFactor is rare on this path)
      const Matrix Ab = white.back().augmentedJacobian();                // m x (sum d + 1), column-major
      blocks.insert(blocks.end(), Ab.data(), Ab.data() + Ab.size());
    }
    if (!linear_ || white.size() != linear_factors_ || unit_sigmas != linear_unit_sigmas_)
      build_linear_handle(gfg, white, unit_sigmas, params);
    std::vector<double> delta(linear_dim_);
    uint64_t bad = 0;
    const gsx_status st = gsx_solve_gfg_h(linear_, blocks.data(), (int64_t)blocks.size(), delta.data(), (int64_t)delta.size(), &bad);
    if (st == GSX_E_INDETERMINATE) throw IndeterminantLinearSystemException(bad);  // caught at LevenbergMarquardtOptimizer.cpp:158
    check(st, linear_, "gsx_solve_gfg_h");
    VectorValues x;
    size_t off = 0;
    for (size_t i = 0; i < linear_keys_.size(); ++i) {
      x.insert(linear_keys_[i], Eigen::Map<Vector>(&delta[off], linear_dims_[i]));
      off += linear_dims_[i];
    }
    return x;
  }

  // ---- L2: the whole run on the device -------------------------------------------------------------------------------------
  const gtsam::Values& optimize() override {
    using namespace gtsam;
    Lowered L;
    std::vector<double> packed;
    lower_values(values(), L, packed);
    const NonlinearFactorGraph& g = graph();
    for (size_t i = 0; i < g.size(); ++i)
      if (g[i] && !lower_factor(g[i], L)) L.fallback.push_back(i);
    L.first_slot = (int)L.f_type.size();
    // S5 slots: one GSX_F_LINEAR factor per fallback factor, rows = its dim, unit noise ([A b] arrives whitened)
    for (size_t i : L.fallback) {
      const auto& f = g[i];
      const std::vector<Key> ks(f->keys().begin(), f->keys().end());
      size_t cols = 1;
      for (Key k : ks) cols += L.dims[L.index.at(k)];
      L.add_factor(GSX_F_LINEAR, (int)f->dim(), ks, std::vector<double>(f->dim() * cols, 0.0), GSX_NOISE_UNIT, {});
    }
    const gsx_problem_desc d = L.desc();
    gsx_handle h = nullptr;
    check(gsx_create(&d, device, &h), nullptr, "gsx_create");
    struct Guard {
      gsx_handle h;
      ~Guard() { gsx_destroy(h); }
    } guard{h};
    check(gsx_set_values(h, packed.data(), (int64_t)packed.size()), h, "gsx_set_values");
    // GTSAM's own ordering (params.ordering, or Ordering::Create as LevenbergMarquardtParams.h:112-117 does)
    const Ordering ordering = params().ordering ? *params().ordering : Ordering::Create(params().orderingType, g);
    const std::vector<uint64_t> ord(ordering.begin(), ordering.end());
    check(gsx_set_ordering(h, ord.data(), (int32_t)ord.size()), h, "gsx_set_ordering");
    const gsx_lm_params p = lower_params(params());
    double final_error = 0, final_lambda = 0;
    int iterations = 0, inner = 0;
    if (L.fallback.empty()) {
      gsx_lm_result r{};
      check(gsx_lm_optimize(h, &p, &r), h, "gsx_lm_optimize");
      final_error = r.final_error, final_lambda = r.final_lambda, iterations = r.iterations, inner = r.inner_iterations;
    } else {
      run_with_fallback(h, L, p, packed, final_error, final_lambda, iterations, inner);
    }
    check(gsx_get_values(h, packed.data(), (int64_t)packed.size()), h, "gsx_get_values");
    state_.reset(new internal::LevenbergMarquardtState(unpack_values(values(), packed), final_error, final_lambda,
                                                       params().lambdaFactor, iterations, inner));
    return values();
  }

 private:
  // the trial loop of LevenbergMarquardtOptimizer::iterate / tryLambda with the CPU-linearized slots refreshed per
  // linearization point and their nonlinear error added on the host; decisions by gsx_lm_decide
  void run_with_fallback(gsx_handle h, const Lowered& L, const gsx_lm_params& p, std::vector<double>& packed, double& final_error,
                         double& final_lambda, int& iterations, int& inner) {
    using namespace gtsam;
    const NonlinearFactorGraph& g = graph();
    Values cur = values();
    auto fallback_error = [&](const Values& v) {
      double e = 0;
      for (size_t i : L.fallback) e += g[i]->error(v);
      return e;
    };
    double dev_err = 0;
    check(gsx_error(h, &dev_err), h, "gsx_error");
    gsx_lm_state st{p.lambda_initial, p.lambda_factor, dev_err + fallback_error(cur), 0, 0};
    std::vector<double> blocks, delta((size_t)gsx_tangent_size(h));
    while (st.outer_iterations < p.max_iterations) {
      const double before = st.cost;
      blocks.clear();
      for (size_t i : L.fallback) {  // factor->linearize(values): seam S5
        const JacobianFactor w = JacobianFactor(*g[i]->linearize(cur)).whiten();
        const Matrix Ab = w.augmentedJacobian();
        blocks.insert(blocks.end(), Ab.data(), Ab.data() + Ab.size());
      }
      check(gsx_set_block_jacobians(h, L.first_slot, (int32_t)L.fallback.size(), blocks.data(), (int64_t)blocks.size()), h,
            "gsx_set_block_jacobians");
      check(gsx_linearize(h), h, "gsx_linearize");
      gsx_lm_decision dec{};
      do {
        uint64_t bad = 0;
        const gsx_status s = gsx_solve(h, st.lambda, p.diagonal_damping, p.min_diagonal, p.max_diagonal, delta.data(),
                                       (int64_t)delta.size(), &bad);
        double lin0 = 0, lind = 0, trial = 0;
        Values trial_values;
        if (s == GSX_OK) {
          check(gsx_linear_error(h, &lin0, &lind), h, "gsx_linear_error");
          check(gsx_retract(h, nullptr, 0, 0, &trial), h, "gsx_retract");   // native factors at the trial point
          VectorValues dx;
          size_t off = 0;
          for (size_t k = 0; k < L.keys.size(); ++k) dx.insert(L.keys[k], Eigen::Map<Vector>(&delta[off], L.dims[k])), off += L.dims[k];
          trial_values = cur.retract(dx);
          trial += fallback_error(trial_values);                             // + the fallback factors, on the host
        } else if (s != GSX_E_INDETERMINATE) {
          check(s, h, "gsx_solve");
        }
        check(gsx_lm_decide(&p, &st, s == GSX_OK, lin0, lind, trial, &dec), h, "gsx_lm_decide");
        if (dec.verdict == GSX_LM_TAKE) {
          check(gsx_retract(h, nullptr, 0, 1, nullptr), h, "gsx_retract(commit)");
          cur = trial_values;
        }
      } while (dec.verdict == GSX_LM_RETRY);
      const double decrease = before - st.cost;
      if (dec.verdict == GSX_LM_GIVE_UP || st.cost <= p.error_tol ||
          (p.relative_error_tol && decrease / before <= p.relative_error_tol) || decrease <= p.absolute_error_tol)
        break;  // NonlinearOptimizer.cpp:182-231
    }
    final_error = st.cost, final_lambda = st.lambda, iterations = st.outer_iterations, inner = st.inner_iterations;
    (void)packed;
  }

  void build_linear_handle(const gtsam::GaussianFactorGraph& gfg, const std::vector<gtsam::JacobianFactor>& white,
                           const std::vector<std::vector<double>>& unit_sigmas,
                           const gtsam::NonlinearOptimizerParams& params) const {
    using namespace gtsam;
    if (linear_) gsx_destroy(linear_);
    linear_ = nullptr;
    Lowered L;
    for (Key k : gfg.keys()) L.index[k] = (int)L.keys.size(), L.keys.push_back(k);
    L.types.assign(L.keys.size(), GSX_VAR_VECTOR);
    L.dims.assign(L.keys.size(), 0);
    for (size_t i = 0; i < white.size(); ++i) {
      const JacobianFactor& w = white[i];
      const Matrix Ab = w.augmentedJacobian();
      for (auto it = w.begin(); it != w.end(); ++it) L.dims[L.index.at(*it)] = (int)w.getDim(it);
      // a factor with hard-constraint rows: a diagonal model of sigmas 0 / 1 (the device eliminates the zero rows exactly)
      L.add_factor(GSX_F_LINEAR, (int)Ab.rows(), std::vector<Key>(w.begin(), w.end()),
                   std::vector<double>(Ab.data(), Ab.data() + Ab.size()),
                   unit_sigmas[i].empty() ? GSX_NOISE_UNIT : GSX_NOISE_DIAGONAL, unit_sigmas[i]);
    }
    linear_unit_sigmas_ = unit_sigmas;
    const gsx_problem_desc d = L.desc();
    check(gsx_create(&d, device, &linear_), nullptr, "gsx_create");
    const Ordering ordering = params.ordering ? *params.ordering : Ordering::Create(params.orderingType, gfg);
    const std::vector<uint64_t> ord(ordering.begin(), ordering.end());
    check(gsx_set_ordering(linear_, ord.data(), (int32_t)ord.size()), linear_, "gsx_set_ordering");
    linear_keys_.assign(L.keys.begin(), L.keys.end());
    linear_dims_ = L.dims;
    linear_dim_ = std::accumulate(L.dims.begin(), L.dims.end(), (size_t)0);
    linear_factors_ = white.size();
  }

  mutable gsx_handle linear_ = nullptr;          // the L1 handle: built at the first solve(), kept for the run
  mutable std::vector<Key> linear_keys_;
  mutable std::vector<int32_t> linear_dims_;
  mutable std::vector<std::vector<double>> linear_unit_sigmas_;
  mutable size_t linear_dim_ = 0, linear_factors_ = 0;
};

}  // namespace gsx
