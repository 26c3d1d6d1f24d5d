// problem.cpp — validation and lowering of gsx_problem_desc into host tables.
#include <cmath>
#include <algorithm>

#include "gsx_internal.h"

namespace gsx {

static int state_dim(int type, int dim) {
  switch (type) {
    case GSX_VAR_VECTOR: return dim;
    case GSX_VAR_POSE2: return 3;
    case GSX_VAR_POSE3: return 12;
    case GSX_VAR_CAMERA: return 17;
  }
  return -1;
}

gsx_status lower_problem(const gsx_problem_desc* d, HostProblem& P, std::string& err) {
  if (!d || d->n_vars < 0 || d->n_factors < 0) {
    err = "null or negative-sized description";
    return GSX_E_INVALID;
  }
  P.n_vars = d->n_vars;
  P.n_factors = d->n_factors;
  P.con_factor.clear();
  P.con_row.clear();
  P.con_mu.clear();
  P.keys.assign(d->var_keys, d->var_keys + d->n_vars);
  P.types.assign(d->var_types, d->var_types + d->n_vars);
  P.dims.assign(d->var_dims, d->var_dims + d->n_vars);
  P.state_off.resize(P.n_vars + 1);
  P.tan_off.resize(P.n_vars + 1);
  P.state_size = P.tan_size = 0;
  for (int v = 0; v < P.n_vars; ++v) {
    if (v > 0 && !(P.keys[v] > P.keys[v - 1])) {
      err = "var_keys must be strictly ascending";
      return GSX_E_INVALID;
    }
    const int t = P.types[v];
    const int expect[4] = {P.dims[v], 3, 6, 9};
    if (t < 0 || t > 3 || P.dims[v] != expect[t] || P.dims[v] <= 0 || (t == GSX_VAR_VECTOR && P.dims[v] > 64)) {
      err = "bad variable type/dim at variable " + std::to_string(v);
      return GSX_E_INVALID;
    }
    P.state_off[v] = (int)P.state_size;
    P.tan_off[v] = (int)P.tan_size;
    P.state_size += state_dim(t, P.dims[v]);
    P.tan_size += P.dims[v];
  }
  P.state_off[P.n_vars] = (int)P.state_size;
  P.tan_off[P.n_vars] = (int)P.tan_size;
  const int nf = P.n_factors;
  P.f_type.assign(d->f_type, d->f_type + nf);
  P.f_rows.assign(d->f_rows, d->f_rows + nf);
  P.f_key_ptr.assign(d->f_key_ptr, d->f_key_ptr + nf + 1);
  P.f_vars.assign(d->f_vars, d->f_vars + (nf ? d->f_key_ptr[nf] : 0));
  P.f_meas_ptr.assign(d->f_meas_ptr, d->f_meas_ptr + nf + 1);
  P.meas.assign(d->meas, d->meas + (nf ? d->f_meas_ptr[nf] : 0));
  P.f_noise_kind.assign(d->f_noise_kind, d->f_noise_kind + nf);
  P.f_noise_ptr.assign(d->f_noise_ptr, d->f_noise_ptr + nf + 1);
  P.noise.assign(d->noise, d->noise + (nf ? d->f_noise_ptr[nf] : 0));
  P.f_jac_off.resize(nf + 1);
  P.f_cols.resize(nf);
  P.jac_size = 0;
  for (int f = 0; f < nf; ++f) {
    const int nk = P.f_key_ptr[f + 1] - P.f_key_ptr[f];
    const int m = P.f_rows[f];
    int cols = 1;
    for (int k = P.f_key_ptr[f]; k < P.f_key_ptr[f + 1]; ++k) {
      const int v = P.f_vars[k];
      if (v < 0 || v >= P.n_vars) {
        err = "factor " + std::to_string(f) + " refers to a variable out of range";
        return GSX_E_INVALID;
      }
      for (int k2 = P.f_key_ptr[f]; k2 < k; ++k2)
        if (P.f_vars[k2] == v) {
          err = "factor " + std::to_string(f) + " lists a variable twice";
          return GSX_E_INVALID;
        }
      cols += P.dims[v];
    }
    const int64_t nmeas = P.f_meas_ptr[f + 1] - P.f_meas_ptr[f];
    const int64_t nnoise = P.f_noise_ptr[f + 1] - P.f_noise_ptr[f];
    bool ok = m > 0 && m <= 64;
    auto tv = [&](int slot) { return P.types[P.f_vars[P.f_key_ptr[f] + slot]]; };
    auto dv = [&](int slot) { return P.dims[P.f_vars[P.f_key_ptr[f] + slot]]; };
    switch (P.f_type[f]) {
      case GSX_F_LINEAR:
        ok = ok && nk >= 1 && nk <= 8 && nmeas == (int64_t)m * cols;
        break;
      case GSX_F_PRIOR:
        ok = ok && nk == 1 && m == dv(0) && nmeas == state_dim(tv(0), dv(0)) && m <= 9;
        break;
      case GSX_F_BETWEEN:
        ok = ok && nk == 2 && tv(0) == tv(1) && dv(0) == dv(1) && tv(0) != GSX_VAR_CAMERA && m == dv(0) &&
             nmeas == state_dim(tv(0), dv(0)) && m <= 9;
        break;
      case GSX_F_SFM:
        ok = ok && nk == 2 && tv(0) == GSX_VAR_CAMERA && tv(1) == GSX_VAR_VECTOR && dv(1) == 3 && m == 2 && nmeas == 2;
        break;
      case GSX_F_BEARINGRANGE:
        ok = ok && nk == 2 && tv(0) == GSX_VAR_POSE2 && tv(1) == GSX_VAR_VECTOR && dv(1) == 2 && m == 2 && nmeas == 2;
        break;
      case GSX_F_PROJECTION:
        ok = ok && nk == 2 && tv(0) == GSX_VAR_POSE3 && tv(1) == GSX_VAR_VECTOR && dv(1) == 3 && m == 2 && nmeas == 7;
        break;
      default:
        ok = false;
    }
    {
      const int kind = P.f_noise_kind[f], loss = kind >> 4;
      const int64_t extra = loss ? 1 : 0;  // a robust model appends its k / c
      ok = ok && kind >= 0 && loss <= 3;
      if (loss && P.f_type[f] == GSX_F_LINEAR) ok = false;  // a given linear factor is already whitened
      switch (kind & GSX_NOISE_BASE_MASK) {
        case GSX_NOISE_UNIT: ok = ok && nnoise == 0 + extra; break;
        case GSX_NOISE_ISOTROPIC: ok = ok && nnoise == 1 + extra; break;
        case GSX_NOISE_DIAGONAL: ok = ok && nnoise == m + extra; break;
        case GSX_NOISE_GAUSSIAN: ok = ok && nnoise == (int64_t)m * m + extra; break;
        case GSX_NOISE_CONSTRAINED: ok = ok && nnoise == 2 * (int64_t)m && !loss; break;
        default: ok = false;
      }
      if (loss && ok) ok = P.noise[P.f_noise_ptr[f + 1] - 1] > 0;
      const int base = kind & GSX_NOISE_BASE_MASK;
      if (ok && (base == GSX_NOISE_ISOTROPIC || base == GSX_NOISE_DIAGONAL || base == GSX_NOISE_CONSTRAINED)) {
        const int64_t ns = base == GSX_NOISE_ISOTROPIC ? 1 : m;
        for (int64_t k = 0; k < ns; ++k) {
          const double sg = P.noise[P.f_noise_ptr[f] + k];
          // a sigma of exactly 0 in a diagonal model is a hard-constraint row (noiseModel::Constrained, NoiseModel.h:389-500)
          const bool constraint = sg == 0.0 && base != GSX_NOISE_ISOTROPIC && !loss;
          if (!constraint && (!(sg > 0) || !std::isfinite(sg))) {
            err = "factor " + std::to_string(f) + ": sigma " + std::to_string(sg) +
                  " (a hard constraint is a sigma of exactly 0 in a GSX_NOISE_DIAGONAL / GSX_NOISE_CONSTRAINED model)";
            return GSX_E_INVALID;
          }
          if (constraint) {
            const double mu = base == GSX_NOISE_CONSTRAINED ? P.noise[P.f_noise_ptr[f] + m + k] : 1000.0;
            if (!(mu > 0) || !std::isfinite(mu)) {
              err = "factor " + std::to_string(f) + ": constraint weight mu " + std::to_string(mu);
              return GSX_E_INVALID;
            }
            P.con_factor.push_back(f);
            P.con_row.push_back((int)k);
            P.con_mu.push_back(mu);
          }
        }
      }
    }
    if (!ok) {
      err = "malformed factor " + std::to_string(f);
      return GSX_E_INVALID;
    }
    P.f_cols[f] = cols;
    P.f_jac_off[f] = P.jac_size;
    P.jac_size += (int64_t)m * cols;
  }
  P.f_jac_off[nf] = P.jac_size;
  // Hard-constraint rows on the device: a DIAGONAL model whose constraint rows carry sigma = 1 / sqrt(mu).  Every kernel
  // then treats them as ordinary rows — the graph error gets mu e^2 / 2 for a violated constraint
  // (Constrained::squaredMahalanobisDistance, NoiseModel.cpp:438-444), the linearized error likewise
  // (JacobianFactor::error with the unit() model, JacobianFactor.cpp:508-513), and J'J a penalty term that is constant on
  // the feasible set — and the factorization eliminates the rows EXACTLY by constraint pivots (solver.hip: constraint_*).
  if (!P.con_factor.empty() || std::find(P.f_noise_kind.begin(), P.f_noise_kind.end(), (int)GSX_NOISE_CONSTRAINED) !=
                                   P.f_noise_kind.end()) {
    std::vector<double> noise;
    std::vector<int64_t> ptr(nf + 1, 0);
    size_t ci = 0;
    for (int f = 0; f < nf; ++f) {
      const int base = P.f_noise_kind[f] & GSX_NOISE_BASE_MASK, m = P.f_rows[f];
      const int64_t b = P.f_noise_ptr[f], e = P.f_noise_ptr[f + 1];
      ptr[f] = (int64_t)noise.size();
      if (base == GSX_NOISE_DIAGONAL || base == GSX_NOISE_CONSTRAINED) {
        for (int k = 0; k < m; ++k) {
          double sg = P.noise[b + k];
          if (ci < P.con_factor.size() && P.con_factor[ci] == f && P.con_row[ci] == k) sg = 1.0 / std::sqrt(P.con_mu[ci++]);
          noise.push_back(sg);
        }
        if (base == GSX_NOISE_DIAGONAL)
          for (int64_t k = b + m; k < e; ++k) noise.push_back(P.noise[k]);   // (a robust model's parameter)
        else
          P.f_noise_kind[f] = GSX_NOISE_DIAGONAL;
      } else {
        noise.insert(noise.end(), P.noise.begin() + b, P.noise.begin() + e);
      }
    }
    ptr[nf] = (int64_t)noise.size();
    P.noise.swap(noise);
    P.f_noise_ptr.swap(ptr);
  }
  return GSX_OK;
}

}  // namespace gsx
