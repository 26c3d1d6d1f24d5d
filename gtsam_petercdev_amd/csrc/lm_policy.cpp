// lm_policy.cpp — the Levenberg-Marquardt trust policy as a pure host function (product; no device code).
//
// One damped trial produces three numbers on the device: the quadratic model's cost at zero and at the
// step (both on the UNDAMPED linearized graph) and the true cost at the retracted point.  gsx_lm_decide
// turns them, together with the controller state (damping weight, its growth multiplier, the cost at the
// current point), into one of four verdicts and the next controller state.  It must take the same
// decisions as LevenbergMarquardtOptimizer::tryLambda (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:
// 121-270) with LevenbergMarquardtState::increaseLambda / decreaseLambda
// (gtsam/nonlinear/internal/LevenbergMarquardtState.h:70-94); the rules, in this file's own words:
//
//   predicted = model(0) - model(step), actual = cost(now) - cost(trial)
//   * a trial whose linear system could not be factored, or whose model predicts an increase, is a
//     rejection without looking at the trial cost;
//   * the gain ratio actual / predicted is only formed when predicted exceeds machine-epsilon times
//     model(0); the step is taken iff the ratio is above the acceptance threshold;
//   * |actual| below relative_error_tol x cost(now) ends the search at this linearization point whatever
//     the ratio says (nothing more to gain here);
//   * taken: damping shrinks — divided by the multiplier (fixed schedule) or scaled by
//     max(1/3, 1 - (2 ratio - 1)^3) with the multiplier doubled (adaptive schedule) — floored at the lower bound;
//   * rejected: damping grows by the multiplier (which doubles under the adaptive schedule); past the upper
//     bound the optimizer gives up.
#include <algorithm>
#include <cmath>
#include <limits>

#include "../../include/gsx.h"

extern "C" gsx_status gsx_lm_decide(const gsx_lm_params* p, gsx_lm_state* s, int32_t solved, double model_at_zero,
                                    double model_at_step, double trial_cost, gsx_lm_decision* out) {
  if (!p || !s || !out) return GSX_E_INVALID;
  gsx_lm_decision d;
  d.verdict = GSX_LM_RETRY;
  d.gain_ratio = 0.0;
  d.cost_change = 0.0;
  d.trial_cost = std::numeric_limits<double>::infinity();
  d.lambda_tried = s->lambda;
  d.solved = solved ? 1 : 0;

  bool take = false, settle = false;
  const double predicted = model_at_zero - model_at_step;
  if (solved && predicted >= 0.0) {
    d.trial_cost = trial_cost;
    d.cost_change = s->cost - trial_cost;
    if (predicted > std::numeric_limits<double>::epsilon() * model_at_zero) {
      d.gain_ratio = d.cost_change / predicted;
      take = d.gain_ratio > p->min_model_fidelity;
    }
    settle = std::fabs(d.cost_change) < p->relative_error_tol * s->cost;
  }

  if (take) {
    double next = s->lambda;
    if (p->use_fixed_lambda_factor) {
      next /= s->factor;
    } else {
      const double t = 2.0 * d.gain_ratio - 1.0;
      next *= std::max(1.0 / 3.0, 1.0 - std::pow(t, 3));
      s->factor *= 2.0;
    }
    s->lambda = std::max(p->lambda_lower_bound, next);
    s->cost = trial_cost;
    s->outer_iterations += 1;
    s->inner_iterations += 1;
    d.verdict = GSX_LM_TAKE;
  } else if (settle) {
    d.verdict = GSX_LM_SETTLE;   // controller state untouched
  } else {
    s->lambda *= s->factor;
    if (!p->use_fixed_lambda_factor) s->factor *= 2.0;
    s->inner_iterations += 1;
    d.verdict = (s->lambda >= p->lambda_upper_bound) ? GSX_LM_GIVE_UP : GSX_LM_RETRY;
  }
  *out = d;
  return GSX_OK;
}
