// kernels.hip — hand-written HIP kernels of the hot path for gfx950 (MI355X, wave64).
//
//   linearize_*        per-factor residual/Jacobian -> [A b] blocks           (HBM-bound)
//   error / linear_error / retract                                            (HBM-bound)
//   assemble_h         Jacobians -> block-sparse Hessian panels (J'J, J'b), per-variable
//                      deterministic gather through LDS                        (HBM-bound)
//   front_small        fused assemble + partial Cholesky of one Bayes-tree clique in LDS
//                      (extend-add of child Schur complements, choleskyPartial semantics of
//                      gtsam/base/cholesky.cpp:108-159), >= 97 % of all cliques   (HBM/LDS-bound)
//   big_*              blocked right-looking partial Cholesky of large cliques in HBM
//                      (trsm + syrk tiles, FP64 MFMA in the syrk)               (MFMA-bound)
//   backsolve          x_F = L11^-T (d - L21^T x_S) per clique, top-down        (HBM-bound)
//
// Fronts are stored column-major, lower triangle significant: L = R' of the reference's
// upper-triangular [R S d] (gtsam/linear/GaussianConditional.h:243-252), the rhs d is the last ROW.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstdlib>

#include "device_geometry.h"
#include "gsx_internal.h"
#include "kernels.h"

namespace gsx {
using namespace gsxd;

#define T kTile

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// total in thread 0 of the block (fixed summation order => deterministic)
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i64 readlane_i64(i64 v, int src_lane) {  // src_lane must be wave-uniform
  int lo = (int)(v & 0xFFFFFFFFll), hi = (int)(v >> 32);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return ((i64)hi << 32) | (unsigned)lo;
}

// e = q d + r for a SMALL quotient (q < 1024, e < 2^23): an integer division by a run-time divisor costs ~40
// instructions on gfx950, and these index splits sit in the inner loops of issue-bound kernels.  rd = 1.0f / d.
__device__ __forceinline__ void divmod_small(int e, int d, float rd, int& q, int& r) {
  q = (int)(((float)e + 0.5f) * rd);
  r = e - q * d;
}

__device__ __forceinline__ double block_sum(double v) {
  __shared__ double ws[16];
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) ws[wave] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += ws[w];
  return s;
}

// mEstimator weight / loss of a whitened error norm — gtsam/linear/LossFunctions.cpp:179-191 (Huber), :250-267 (Tukey),
// :217-224 (Cauchy); `loss` = GSX_NOISE_ROBUST_* >> 4, k = the model's parameter.
__device__ __forceinline__ double robust_weight(int loss, double k, double dist) {
  const double a = fabs(dist);
  if (loss == 1) return (a <= k) ? 1.0 : k / a;
  if (loss == 2) {
    if (a > k) return 0.0;
    const double t = 1.0 - dist * dist / (k * k);
    return t * t;
  }
  return (k * k) / (k * k + dist * dist);
}
__device__ __forceinline__ double robust_loss(int loss, double k, double dist) {
  const double a = fabs(dist);
  if (loss == 1) return (a <= k) ? dist * dist / 2 : k * (a - k / 2);
  if (loss == 2) {
    if (a > k) return k * k / 6.0;
    const double t = 1.0 - dist * dist / (k * k);
    return k * k * (1 - t * t * t) / 6.0;
  }
  return k * k * log1p(dist * dist / (k * k)) * 0.5;
}
// number of parameters of the base noise model (the robust parameter follows them)
__device__ __forceinline__ int noise_base_params(int base, int m) {
  return base == GSX_NOISE_UNIT ? 0 : (base == GSX_NOISE_ISOTROPIC ? 1 : (base == GSX_NOISE_DIAGONAL ? m : m * m));
}

template <int M, int NC>
__device__ __forceinline__ void whiten_store(double (&J)[M * NC], int kind_full, const double* np, double* out) {
  const int kind = kind_full & GSX_NOISE_BASE_MASK, loss = kind_full >> 4;
  if (kind == GSX_NOISE_ISOTROPIC) {
    const double inv = 1.0 / np[0];
#pragma unroll
    for (int i = 0; i < M * NC; ++i) J[i] *= inv;
  } else if (kind == GSX_NOISE_DIAGONAL) {
    double inv[M];
#pragma unroll
    for (int r = 0; r < M; ++r) inv[r] = 1.0 / np[r];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < M; ++r) J[c * M + r] *= inv[r];
  } else if (kind == GSX_NOISE_GAUSSIAN) {
    double R[M * M];
#pragma unroll
    for (int i = 0; i < M * M; ++i) R[i] = np[i];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double col[M];
#pragma unroll
      for (int r = 0; r < M; ++r) {
        double s = 0;
#pragma unroll
        for (int k = r; k < M; ++k) s += R[r * M + k] * J[c * M + k];
        col[r] = s;
      }
#pragma unroll
      for (int r = 0; r < M; ++r) J[c * M + r] = col[r];
    }
  }
  if (loss) {  // Robust::WhitenSystem: Block reweighting by sqrt(weight(|b|)) — NoiseModel.cpp:714-722
    double s = 0;
#pragma unroll
    for (int r = 0; r < M; ++r) s += J[(NC - 1) * M + r] * J[(NC - 1) * M + r];
    const double w = sqrt(robust_weight(loss, np[noise_base_params(kind, M)], sqrt(s)));
#pragma unroll
    for (int i = 0; i < M * NC; ++i) J[i] *= w;
  }
  // a lane owns its factor's block: 16-byte stores halve the memory instructions when the block allows it
  if constexpr ((M * NC) % 2 == 0) {
    if ((reinterpret_cast<size_t>(out) & 15) == 0) {
      double2* out2 = reinterpret_cast<double2*>(out);
#pragma unroll
      for (int i = 0; i < M * NC / 2; ++i) out2[i] = double2{J[2 * i], J[2 * i + 1]};
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < M * NC; ++i) out[i] = J[i];
}

template <int M>
__device__ __forceinline__ double whitened_half_sqnorm(double (&e)[M], int kind_full, const double* np) {
  const int kind = kind_full & GSX_NOISE_BASE_MASK, loss = kind_full >> 4;
  double s = 0;
  if (kind == GSX_NOISE_UNIT) {
#pragma unroll
    for (int r = 0; r < M; ++r) s += e[r] * e[r];
  } else if (kind == GSX_NOISE_ISOTROPIC) {
    const double inv = 1.0 / np[0];
#pragma unroll
    for (int r = 0; r < M; ++r) s += (e[r] * inv) * (e[r] * inv);
  } else if (kind == GSX_NOISE_DIAGONAL) {
#pragma unroll
    for (int r = 0; r < M; ++r) {
      const double x = e[r] * (1.0 / np[r]);
      s += x * x;
    }
  } else {
#pragma unroll
    for (int r = 0; r < M; ++r) {
      double x = 0;
#pragma unroll
      for (int k = r; k < M; ++k) x += np[r * M + k] * e[k];
      s += x * x;
    }
  }
  // NoiseModelFactor::error = loss(sqrt(squaredMahalanobisDistance)) — NonlinearFactor.cpp:138-149, NoiseModel.h Robust::loss
  if (loss) return robust_loss(loss, np[noise_base_params(kind, M)], sqrt(s));
  return 0.5 * s;
}

// runtime-dimension whitening of an m x nc column-major block living in global memory
__device__ inline void whiten_inplace(double* J, int m, int nc, int kind_full, const double* np) {
  const int kind = kind_full & GSX_NOISE_BASE_MASK, loss = kind_full >> 4;
  if (kind == GSX_NOISE_UNIT) {
  } else if (kind == GSX_NOISE_ISOTROPIC) {
    const double inv = 1.0 / np[0];
    for (int i = 0; i < m * nc; ++i) J[i] *= inv;
  } else if (kind == GSX_NOISE_DIAGONAL) {
    for (int c = 0; c < nc; ++c)
      for (int r = 0; r < m; ++r) J[c * m + r] *= (1.0 / np[r]);
  } else {
    for (int c = 0; c < nc; ++c)
      for (int r = 0; r < m; ++r) {  // ascending r: row r only needs rows k >= r (still unwhitened)
        double s = 0;
        for (int k = r; k < m; ++k) s += np[r * m + k] * J[c * m + k];
        J[c * m + r] = s;
      }
  }
  if (loss) {
    double s = 0;
    for (int r = 0; r < m; ++r) s += J[(nc - 1) * m + r] * J[(nc - 1) * m + r];
    const double w = sqrt(robust_weight(loss, np[noise_base_params(kind, m)], sqrt(s)));
    for (int i = 0; i < m * nc; ++i) J[i] *= w;
  }
}

// traits<T>::Local(x, y) = chart(x^-1 y) for every variable type
__device__ inline void local_coords(int type, int dim, const double* x, const double* y, double* out) {
  if (type == GSX_VAR_VECTOR) {
    for (int i = 0; i < dim; ++i) out[i] = y[i] - x[i];
  } else if (type == GSX_VAR_POSE2) {
    const P2 h = compose(inverse(load_pose2(x)), load_pose2(y));
    out[0] = h.x; out[1] = h.y; out[2] = theta(h);
  } else {
    pose3_logmap(between(load_pose3(x), load_pose3(y)), out);
    if (type == GSX_VAR_CAMERA) {
      out[6] = y[12] - x[12]; out[7] = y[13] - x[13]; out[8] = y[14] - x[14];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// linearize: one thread per factor (register-resident [A b], whitening fused)
// ---------------------------------------------------------------------------------------------
// GeneralSFMFactor::linearize — gtsam/slam/GeneralSFMFactor.h:141-177
// (defined below) priors and other rare factors ride in the first blocks of the launch of a main factor family
// (launch_linearize): a launch of their own is pure latency — 22 us for the 1 723 camera priors of BAL-1723
__device__ __forceinline__ void linearize_generic_body(const DevProblem& P, const int* list, int n, const double* values,
                                                       double* jac, int bid);
__global__ void __launch_bounds__(256) linearize_sfm_kernel(DevProblem P, const int* list, int n, const double* values,
                                                            double* jac, DevStatus* status, const int* glist, int gn) {
  const int gb = (gn + 255) >> 8;
  if ((int)blockIdx.x < gb) {
    linearize_generic_body(P, glist, gn, values, jac, blockIdx.x);
    return;
  }
  const int i = ((int)blockIdx.x - gb) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = list[i];
  const FactorRec fr = P.frec[f];
  const double* cam = values + fr.s0;
  const double* pt = values + fr.s1;
  const double* z = P.meas + fr.meas_off;
  double camr[17], ptr3[3], pi[2], H1[18], H2[6], J[26];
#pragma unroll
  for (int k = 0; k < 17; ++k) camr[k] = cam[k];
  ptr3[0] = pt[0]; ptr3[1] = pt[1]; ptr3[2] = pt[2];
  const bool ok = sfm_project(camr, ptr3, pi, H1, H2);
  if (ok) {
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      J[2 * c] = H1[c];
      J[2 * c + 1] = H1[9 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      J[18 + 2 * c] = H2[c];
      J[18 + 2 * c + 1] = H2[3 + c];
    }
    J[24] = z[0] - pi[0];
    J[25] = z[1] - pi[1];
  } else {
#pragma unroll
    for (int k = 0; k < 26; ++k) J[k] = 0;
    atomicAdd(&status->n_cheirality, 1);
  }
  whiten_store<2, 13>(J, (fr.type_kind >> 8) & 0xffff, P.noise + fr.noise_off, jac + fr.jac_off);
}

// BearingRangeFactor<Pose2,Point2> — gtsam/sam/BearingRangeFactor.h (ExpressionFactor: e = -Local(h(x), z))
__global__ void __launch_bounds__(256) linearize_bearingrange_kernel(DevProblem P, const int* list, int n,
                                                                     const double* values, double* jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = list[i];
  const int kp = P.f_key_ptr[f];
  const double* pose = values + P.var_state_off[P.f_vars[kp]];
  const double* pt = values + P.var_state_off[P.f_vars[kp + 1]];
  const double* z = P.meas + P.f_meas_off[f];
  double ps[3] = {pose[0], pose[1], pose[2]}, p2[2] = {pt[0], pt[1]}, br[2], H1[6], H2[4], J[12];
  bearing_range_2d(ps, p2, br, H1, H2);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    J[2 * c] = H1[c];
    J[2 * c + 1] = H1[3 + c];
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    J[6 + 2 * c] = H2[c];
    J[6 + 2 * c + 1] = H2[2 + c];
  }
  J[10] = -wrap_angle(br[0] - z[0]);
  J[11] = -(br[1] - z[1]);
  whiten_store<2, 6>(J, P.f_noise_kind[f], P.noise + P.f_noise_off[f], jac + P.f_jac_off[f]);
}

// GenericProjectionFactor<Pose3,Point3,Cal3_S2>::evaluateError — gtsam/slam/ProjectionFactor.h:138-166
__global__ void __launch_bounds__(256) linearize_projection_kernel(DevProblem P, const int* list, int n, const double* values,
                                                                   double* jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = list[i];
  const int kp = P.f_key_ptr[f];
  const double* pose = values + P.var_state_off[P.f_vars[kp]];
  const double* pt = values + P.var_state_off[P.f_vars[kp + 1]];
  const double* z = P.meas + P.f_meas_off[f];  // u v fx fy s u0 v0
  double pr[12], p3[3], K[5], pi[2], H1[12], H2[6], J[20];
#pragma unroll
  for (int k = 0; k < 12; ++k) pr[k] = pose[k];
  p3[0] = pt[0]; p3[1] = pt[1]; p3[2] = pt[2];
#pragma unroll
  for (int k = 0; k < 5; ++k) K[k] = z[2 + k];
  if (pinhole_project_s2(pr, p3, K, pi, H1, H2)) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      J[2 * c] = H1[c];
      J[2 * c + 1] = H1[6 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      J[12 + 2 * c] = H2[c];
      J[12 + 2 * c + 1] = H2[3 + c];
    }
    J[18] = z[0] - pi[0];  // b = -(h(x) - z)
    J[19] = z[1] - pi[1];
  } else {  // point behind the camera: zero Jacobians, error (2 fx, 2 fx)
#pragma unroll
    for (int k = 0; k < 18; ++k) J[k] = 0;
    J[18] = J[19] = -2.0 * K[0];
  }
  whiten_store<2, 10>(J, P.f_noise_kind[f], P.noise + P.f_noise_off[f], jac + P.f_jac_off[f]);
}

// BetweenFactor<Pose2> via NoiseModelFactor::linearize — gtsam/nonlinear/NonlinearFactor.cpp:152-184
__global__ void __launch_bounds__(256) linearize_between_pose2_kernel(DevProblem P, const int* list, int n,
                                                                      const double* values, double* jac, const int* glist,
                                                                      int gn) {
  const int gb = (gn + 255) >> 8;
  if ((int)blockIdx.x < gb) {
    linearize_generic_body(P, glist, gn, values, jac, blockIdx.x);
    return;
  }
  const int i = ((int)blockIdx.x - gb) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = list[i];
  const FactorRec fr = P.frec[f];
  const P2 x1 = load_pose2(values + fr.s0);
  const P2 x2 = load_pose2(values + fr.s1);
  const P2 h = compose(inverse(x1), x2);
  const P2 zh = compose(inverse(load_pose2(P.meas + fr.meas_off)), h);
  const P2 hi = inverse(h);
  // Ad(h^-1) = [[c,-s,y],[s,c,-x],[0,0,1]]; A1 = -Ad, A2 = I, b = -e
  double J[21];
  J[0] = -hi.c; J[1] = -hi.s; J[2] = 0;
  J[3] = hi.s; J[4] = -hi.c; J[5] = 0;
  J[6] = -hi.y; J[7] = hi.x; J[8] = -1;
  J[9] = 1; J[10] = 0; J[11] = 0;
  J[12] = 0; J[13] = 1; J[14] = 0;
  J[15] = 0; J[16] = 0; J[17] = 1;
  J[18] = -zh.x; J[19] = -zh.y; J[20] = -theta(zh);
  whiten_store<3, 7>(J, (fr.type_kind >> 8) & 0xffff, P.noise + fr.noise_off, jac + fr.jac_off);
}

__global__ void __launch_bounds__(256) linearize_between_pose3_kernel(DevProblem P, const int* list, int n,
                                                                      const double* values, double* jac, const int* glist,
                                                                      int gn) {
  const int gb = (gn + 255) >> 8;
  if ((int)blockIdx.x < gb) {
    linearize_generic_body(P, glist, gn, values, jac, blockIdx.x);
    return;
  }
  const int i = ((int)blockIdx.x - gb) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = list[i];
  const FactorRec fr = P.frec[f];
  const P3 x1 = load_pose3(values + fr.s0);
  const P3 x2 = load_pose3(values + fr.s1);
  const P3 h = between(x1, x2);
  const P3 zh = compose(inverse(load_pose3(P.meas + fr.meas_off)), h);
  double e[6], Ad[36], J[78];
  pose3_logmap(zh, e);
  pose3_adjoint(inverse(h), Ad);
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      J[c * 6 + r] = -Ad[6 * r + c];
      J[36 + c * 6 + r] = (r == c) ? 1.0 : 0.0;
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) J[72 + r] = -e[r];
  whiten_store<6, 13>(J, (fr.type_kind >> 8) & 0xffff, P.noise + fr.noise_off, jac + fr.jac_off);
}

// priors of every type and BetweenFactor on vector spaces: rare, runtime dims, built in place
__device__ __forceinline__ void linearize_generic_body(const DevProblem& P, const int* list, int n, const double* values,
                                                       double* jac, int bid) {
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = list[i];
  const int kp = P.f_key_ptr[f];
  const int m = P.f_rows[f];
  double* J = jac + P.f_jac_off[f];
  const double* z = P.meas + P.f_meas_off[f];
  if (P.f_type[f] == GSX_F_PRIOR) {
    // PriorFactor — gtsam/nonlinear/PriorFactor.h:98-102: e = -Local(x, prior), H = I, b = -e
    const int v = P.f_vars[kp];
    double l[9];
    local_coords(P.var_type[v], P.var_dim[v], values + P.var_state_off[v], z, l);
    for (int c = 0; c < m; ++c)
      for (int r = 0; r < m; ++r) J[c * m + r] = (r == c) ? 1.0 : 0.0;
    for (int r = 0; r < m; ++r) J[m * m + r] = l[r];
    whiten_inplace(J, m, m + 1, P.f_noise_kind[f], P.noise + P.f_noise_off[f]);
  } else {  // BETWEEN on VECTOR: e = (x2 - x1) - z, A1 = -I, A2 = I
    const double* x1 = values + P.var_state_off[P.f_vars[kp]];
    const double* x2 = values + P.var_state_off[P.f_vars[kp + 1]];
    for (int c = 0; c < m; ++c)
      for (int r = 0; r < m; ++r) {
        J[c * m + r] = (r == c) ? -1.0 : 0.0;
        J[m * m + c * m + r] = (r == c) ? 1.0 : 0.0;
      }
    for (int r = 0; r < m; ++r) J[2 * m * m + r] = -((x2[r] - x1[r]) - z[r]);
    whiten_inplace(J, m, 2 * m + 1, P.f_noise_kind[f], P.noise + P.f_noise_off[f]);
  }
}
__global__ void linearize_generic_kernel(DevProblem P, const int* list, int n, const double* values, double* jac) {
  linearize_generic_body(P, list, n, values, jac, blockIdx.x);
}

void launch_linearize(const DevProblem& P, const int* const type_lists[6], const int type_counts[6],
                      const double* values, double* jac, DevStatus* status, hipStream_t st) {
  auto grid = [](int n) { return dim3((n + 255) / 256); };
  // the generic family (priors, vector-space factors: few) rides in the first blocks of the first main family's launch
  const int* glist = type_lists[3];
  int gn = type_counts[3];
  auto take = [&](int n) {  // blocks of a main launch: its own + the generic ones, handed out once
    const int g = gn;
    gn = 0;
    return std::make_pair(dim3((n + 255) / 256 + (g + 255) / 256), g);
  };
  if (type_counts[0]) {
    const auto gg = take(type_counts[0]);
    linearize_sfm_kernel<<<gg.first, 256, 0, st>>>(P, type_lists[0], type_counts[0], values, jac, status, glist, gg.second);
  }
  if (type_counts[1]) {
    const auto gg = take(type_counts[1]);
    linearize_between_pose2_kernel<<<gg.first, 256, 0, st>>>(P, type_lists[1], type_counts[1], values, jac, glist, gg.second);
  }
  if (type_counts[2]) {
    const auto gg = take(type_counts[2]);
    linearize_between_pose3_kernel<<<gg.first, 256, 0, st>>>(P, type_lists[2], type_counts[2], values, jac, glist, gg.second);
  }
  if (gn) linearize_generic_kernel<<<grid(gn), 256, 0, st>>>(P, glist, gn, values, jac);
  if (type_counts[4])
    linearize_projection_kernel<<<grid(type_counts[4]), 256, 0, st>>>(P, type_lists[4], type_counts[4], values, jac);
  if (type_counts[5])
    linearize_bearingrange_kernel<<<grid(type_counts[5]), 256, 0, st>>>(P, type_lists[5], type_counts[5], values, jac);
}

// ---------------------------------------------------------------------------------------------
// graph error (NoiseModelFactor::error, gtsam/nonlinear/NonlinearFactor.cpp:138-149)
// ---------------------------------------------------------------------------------------------
// FT / FV >= 0: the factor type / the type of its first variable are known at compile time (the per-type kernels below:
// the one-kernel-for-all form needed 247 registers, two waves a SIMD); -1: read from the record.
template <int FT = -1, int FV = -1>
__device__ inline double factor_error(const DevProblem& P, int f, const double* values) {
  const FactorRec fr = P.frec[f];
  const int type = FT >= 0 ? FT : (fr.type_kind & 0xff);
  if (type == GSX_F_LINEAR) return 0.0;
  const int kind = (fr.type_kind >> 8) & 0xffff;
  const double* np = P.noise + fr.noise_off;
  const double* z = P.meas + fr.meas_off;
  if (type == GSX_F_SFM) {
    const double* cam = values + fr.s0;
    const double* pt = values + fr.s1;
    double camr[17], ptr3[3], pi[2], e[2];
#pragma unroll
    for (int k = 0; k < 17; ++k) camr[k] = cam[k];
    ptr3[0] = pt[0]; ptr3[1] = pt[1]; ptr3[2] = pt[2];
    if (!sfm_project(camr, ptr3, pi, nullptr, nullptr)) return 0.0;
    e[0] = pi[0] - z[0];
    e[1] = pi[1] - z[1];
    return whitened_half_sqnorm<2>(e, kind, np);
  }
  if (type == GSX_F_BEARINGRANGE) {
    const double* pose = values + fr.s0;
    const double* pt = values + fr.s1;
    double ps[3] = {pose[0], pose[1], pose[2]}, p2[2] = {pt[0], pt[1]}, br[2], e[2];
    bearing_range_2d(ps, p2, br, nullptr, nullptr);
    e[0] = wrap_angle(br[0] - z[0]);
    e[1] = br[1] - z[1];
    return whitened_half_sqnorm<2>(e, kind, np);
  }
  if (type == GSX_F_PROJECTION) {
    const double* pose = values + fr.s0;
    const double* pt = values + fr.s1;
    double pr[12], p3[3], K[5], pi[2], e[2];
#pragma unroll
    for (int k = 0; k < 12; ++k) pr[k] = pose[k];
    p3[0] = pt[0]; p3[1] = pt[1]; p3[2] = pt[2];
#pragma unroll
    for (int k = 0; k < 5; ++k) K[k] = z[2 + k];
    if (pinhole_project_s2(pr, p3, K, pi, nullptr, nullptr)) {
      e[0] = pi[0] - z[0];
      e[1] = pi[1] - z[1];
    } else {
      e[0] = e[1] = 2.0 * K[0];
    }
    return whitened_half_sqnorm<2>(e, kind, np);
  }
  const int vt = FV >= 0 ? FV : ((fr.type_kind >> 24) & 0xff);
  if (type == GSX_F_BETWEEN && vt == GSX_VAR_POSE2) {
    const P2 h = compose(inverse(load_pose2(values + fr.s0)), load_pose2(values + fr.s1));
    const P2 zh = compose(inverse(load_pose2(z)), h);
    double e[3] = {zh.x, zh.y, theta(zh)};
    return whitened_half_sqnorm<3>(e, kind, np);
  }
  if (type == GSX_F_BETWEEN && vt == GSX_VAR_POSE3) {
    const P3 h = between(load_pose3(values + fr.s0), load_pose3(values + fr.s1));
    double e[6];
    pose3_logmap(compose(inverse(load_pose3(z)), h), e);
    return whitened_half_sqnorm<6>(e, kind, np);
  }
  // generic: priors, vector between
  const int m = fr.rows_dim & 0xff;
  double e[9];
  if (type == GSX_F_PRIOR) {
    local_coords(vt, (fr.rows_dim >> 8) & 0xff, values + fr.s0, z, e);
  } else {
    const double* x1 = values + fr.s0;
    const double* x2 = values + fr.s1;
    for (int r = 0; r < m; ++r) e[r] = (x2[r] - x1[r]) - z[r];
  }
  const int base = kind & GSX_NOISE_BASE_MASK, loss = kind >> 4;
  double s = 0;
  for (int r = 0; r < m; ++r) {
    double x;
    if (base == GSX_NOISE_UNIT) x = e[r];
    else if (base == GSX_NOISE_ISOTROPIC) x = e[r] * (1.0 / np[0]);
    else if (base == GSX_NOISE_DIAGONAL) x = e[r] * (1.0 / np[r]);
    else {
      x = 0;
      for (int k = r; k < m; ++k) x += np[r * m + k] * e[k];
    }
    s += x * x;
  }
  if (loss) return robust_loss(loss, np[noise_base_params(base, m)], sqrt(s));
  return 0.5 * s;
}

__global__ void __launch_bounds__(256) reduce_final_kernel(const double* partials, int n, int stride, double* scalars,
                                                           int slot) {
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[(size_t)i * stride];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) scalars[slot] = s;
}

// two interleaved partial sums (stride 2), a block each, into two consecutive scalar slots: the same additions in the
// same order as two launches of the kernel above
__global__ void __launch_bounds__(256) reduce_final_pair_kernel(const double* partials, int n, double* scalars, int slot0) {
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[(size_t)i * 2 + blockIdx.x];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) scalars[slot0 + blockIdx.x] = s;
}

// the factors of one type list (the lists of launch_linearize), the type a compile-time constant
template <int FT, int FV>
__global__ void __launch_bounds__(256) error_list_kernel(DevProblem P, const int* list, int n, const double* values,
                                                         double* partials) {
  double acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc += factor_error<FT, FV>(P, list[i], values);
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

void launch_error(const DevProblem& P, const int* const type_lists[6], const int type_counts[6], const double* values,
                  double* partials, int cap, double* scalars, int slot, hipStream_t st) {
  // a launch per non-empty list, its blocks a share of the partial sums proportional to its factors; the sums are added in
  // list order, block order: deterministic
  long long total = 0;
  for (int k = 0; k < 6; ++k) total += type_counts[k];
  int off = 0;
  for (int k = 0; k < 6; ++k) {
    const int n = type_counts[k];
    if (!n) continue;
    int nb = (n + 255) / 256;
    const int share = (int)std::max<long long>(1, (long long)(cap - 6) * n / std::max<long long>(total, 1));
    nb = std::min(nb, share);
    double* out = partials + off;
    switch (k) {
      case 0: error_list_kernel<GSX_F_SFM, GSX_VAR_CAMERA><<<nb, 256, 0, st>>>(P, type_lists[k], n, values, out); break;
      case 1: error_list_kernel<GSX_F_BETWEEN, GSX_VAR_POSE2><<<nb, 256, 0, st>>>(P, type_lists[k], n, values, out); break;
      case 2: error_list_kernel<GSX_F_BETWEEN, GSX_VAR_POSE3><<<nb, 256, 0, st>>>(P, type_lists[k], n, values, out); break;
      case 4: error_list_kernel<GSX_F_PROJECTION, GSX_VAR_POSE3><<<nb, 256, 0, st>>>(P, type_lists[k], n, values, out); break;
      case 5: error_list_kernel<GSX_F_BEARINGRANGE, GSX_VAR_POSE2><<<nb, 256, 0, st>>>(P, type_lists[k], n, values, out); break;
      default: error_list_kernel<-1, -1><<<nb, 256, 0, st>>>(P, type_lists[k], n, values, out); break;
    }
    off += nb;
  }
  reduce_final_kernel<<<1, 256, 0, st>>>(partials, off, 1, scalars, slot);
}

// ---------------------------------------------------------------------------------------------
// linear error 0.5 sum |A delta - b|^2 at 0 and at delta (JacobianFactor::error,
// gtsam/linear/JacobianFactor.cpp:494-514)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) linear_error_kernel(DevProblem P, const double* jac, const double* delta,
                                                           double* partials) {
  double acc0 = 0, accd = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_active; i += gridDim.x * blockDim.x) {
    const int f = P.f_active ? P.f_active[i] : i;
    const int m = P.f_rows[f], nc = P.f_cols[f];
    const double* J = jac + P.f_jac_off[f];
    if (m == 2 && (P.f_jac_off[f] & 1) == 0) {
      // two-row factors (every SFM observation): a column is one 16-byte load — half the memory instructions of a
      // kernel whose lanes each walk their own 208-byte block
      const double2* J2 = reinterpret_cast<const double2*>(J);
      const double2 b = J2[nc - 1];
      double e0 = -b.x, e1 = -b.y;
      int col = 0;
      for (int k = P.f_key_ptr[f]; k < P.f_key_ptr[f + 1]; ++k) {
        const int v = P.f_vars[k];
        const double* x = delta + P.var_tan_off[v];
        const int d = P.var_dim[v];
        for (int c = 0; c < d; ++c, ++col) {
          const double2 j = J2[col];
          e0 += j.x * x[c];
          e1 += j.y * x[c];
        }
      }
      acc0 += b.x * b.x;
      accd += e0 * e0;
      acc0 += b.y * b.y;
      accd += e1 * e1;
      continue;
    }
    for (int r = 0; r < m; ++r) {
      const double b = J[(nc - 1) * m + r];
      double e = -b;
      int col = 0;
      for (int k = P.f_key_ptr[f]; k < P.f_key_ptr[f + 1]; ++k) {
        const int v = P.f_vars[k];
        const double* x = delta + P.var_tan_off[v];
        const int d = P.var_dim[v];
        for (int c = 0; c < d; ++c, ++col) e += J[col * m + r] * x[c];
      }
      acc0 += b * b;
      accd += e * e;
    }
  }
  const double s0 = block_sum(0.5 * acc0);
  const double sd = block_sum(0.5 * accd);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = s0;
    partials[2 * blockIdx.x + 1] = sd;
  }
}

// The same sums with the Jacobian blocks read ONCE and coalesced: a wave takes 64 consecutive factors, whose [A b]
// blocks are one contiguous range of `jac`, copies the range into its LDS slice (eight loads a lane in flight) and
// every lane then walks its own block out of LDS.  (A lane walking its 208-byte block in global memory touches a new
// cache line per load: 1.9 x the algorithmic bytes fetched from HBM, r01_bal1723_pmc_traffic.json.)
// `slice` = doubles per wave, the largest range of any 64 consecutive factors (host: upload_problem).
__global__ void __launch_bounds__(128) linear_error_staged_kernel(DevProblem P, const double* jac, const double* delta,
                                                                  double* partials, int slice) {
  extern __shared__ double stage[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* W = stage + (size_t)wave * slice;
  const int ngroups = (P.n_factors + 63) >> 6;
  double acc0 = 0, accd = 0;
  for (int g = blockIdx.x * 2 + wave; g < ngroups; g += gridDim.x * 2) {
    const int i0 = g * 64;
    const bool live = i0 + lane < P.n_factors;
    const int f = live ? i0 + lane : P.n_factors - 1;
    const int m = P.f_rows[f], nc = P.f_cols[f];
    const i64 off = P.f_jac_off[f];
    const i64 base = readlane_i64(off, 0);
    const int last = min(63, P.n_factors - 1 - i0);
    const int len = (int)(readlane_i64(off + (i64)m * nc, last) - base);
    const double* src = jac + base;
    for (int e0 = lane; e0 < len; e0 += 64 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (e0 + 64 * u < len) ? src[e0 + 64 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + 64 * u < len) W[e0 + 64 * u] = v[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (live) {
      const double* J = W + (off - base);
      for (int r = 0; r < m; ++r) {
        const double b = J[(nc - 1) * m + r];
        double e = -b;
        int col = 0;
        for (int k = P.f_key_ptr[f]; k < P.f_key_ptr[f + 1]; ++k) {
          const int v = P.f_vars[k];
          const double* x = delta + P.var_tan_off[v];
          const int d = P.var_dim[v];
          for (int c = 0; c < d; ++c, ++col) e += J[col * m + r] * x[c];
        }
        acc0 += b * b;
        accd += e * e;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  const double s0 = block_sum(0.5 * acc0);
  const double sd = block_sum(0.5 * accd);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = s0;
    partials[2 * blockIdx.x + 1] = sd;
  }
}

void launch_linear_error(const DevProblem& P, const double* jac, const double* delta, double* partials, int cap,
                         double* scalars, int slice, hipStream_t st) {
  int nb;
  if (slice > 0 && !P.f_active) {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)linear_error_staged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr = true;
    }
    nb = ((P.n_factors + 63) / 64 + 1) / 2;
    nb = nb < 1 ? 1 : (nb > cap / 2 ? cap / 2 : nb);
    linear_error_staged_kernel<<<nb, 128, (size_t)2 * slice * sizeof(double), st>>>(P, jac, delta, partials, slice);
  } else {
    nb = (P.n_active + 255) / 256;
    nb = nb < 1 ? 1 : (nb > cap / 2 ? cap / 2 : nb);
    linear_error_kernel<<<nb, 256, 0, st>>>(P, jac, delta, partials);
  }
  static_assert(SC_LIND == SC_LIN0 + 1, "the two linearized errors are reduced by one launch, a block each");
  reduce_final_pair_kernel<<<2, 256, 0, st>>>(partials, nb, scalars, SC_LIN0);
}

// The two linearized errors of an LM trial WITHOUT a pass over the Jacobians (GaussianFactorGraph::error at 0 and at
// delta, gtsam/linear/GaussianFactorGraph.cpp:71-78, LevenbergMarquardtOptimizer.cpp:178-184):
//   e(0) = 1/2 sum |b_i|^2                      — the right-hand sides alone, once per linearization (lin0_kernel);
//   e(delta) = e(0) - 1/2 g'delta - 1/2 lambda delta'D delta   for the delta that solves (H + lambda D) delta = g
// (1/2 |A delta - b|^2 = 1/2 delta'H delta - g'delta + 1/2 b'b and delta'H delta = g'delta - lambda delta'D delta), with
// g = J'b read from the rhs rows of the H panels — a few hundred KB instead of every [A b] block.  The model decrease
// e(0) - e(delta) comes out without the cancellation of the direct evaluation.
__global__ void __launch_bounds__(256) lin0_kernel(DevProblem P, const double* jac, double* partials) {
  double acc = 0;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P.n_active; k += gridDim.x * blockDim.x) {
    const int f = P.f_active ? P.f_active[k] : k;
    const FactorRec fr = P.frec[f];   // (one 32-byte record instead of three table lookups before the first load of b)
    const int m = fr.rows_dim & 0xff;
    const double* b = jac + fr.jac_off + (i64)m * (int)(((unsigned)fr.rows_dim >> 16) - 1u);
    double s = 0;
    for (int r = 0; r < m; ++r) s += b[r] * b[r];
    acc += 0.5 * s;
  }
  const double t = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256) model_error_kernel(DevProblem P, DevSymbolic S, const int* vars, int nvars,
                                                          const double* H, const double* delta, const double* damp,
                                                          const double* scalars, double* partials) {
  const double lambda = scalars[SC_LAMBDA];
  double acc = 0;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nvars; k += gridDim.x * blockDim.x) {
    const int v = vars[k];
    const int d = P.var_dim[v], rows = S.h_rows[v], to = P.var_tan_off[v];
    const double* p = H + S.h_off[v] + (rows - 1);   // the rhs row of the panel = (J'b)_v
    for (int j = 0; j < d; ++j) {
      const double x = delta[to + j];
      acc += x * (p[(i64)j * rows] + lambda * damp[to + j] * x);
    }
  }
  const double t = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// scalars[SC_LIN0] = e(0) (this rank's share), scalars[SC_LIND] = e(0) - 1/2 sum partials
__global__ void __launch_bounds__(256) model_final_kernel(const double* partials, int n, double* scalars) {
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) {
    const double e0 = scalars[SC_LIN0_LOCAL];
    scalars[SC_LIN0] = e0;
    scalars[SC_LIND] = e0 - 0.5 * s;
  }
}
void launch_lin0(const DevProblem& P, const double* jac, double* partials, int cap, double* scalars, hipStream_t st) {
  int nb = (P.n_active + 255) / 256;
  nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
  lin0_kernel<<<nb, 256, 0, st>>>(P, jac, partials);
  reduce_final_kernel<<<1, 256, 0, st>>>(partials, nb, 1, scalars, SC_LIN0_LOCAL);
}
void launch_model_error(const DevProblem& P, const DevSymbolic& S, const int* vars, int nvars, const double* H,
                        const double* delta, const double* damp, double* partials, int cap, double* scalars,
                        hipStream_t st) {
  int nb = (nvars + 255) / 256;
  nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
  model_error_kernel<<<nb, 256, 0, st>>>(P, S, vars, nvars, H, delta, damp, scalars, partials);
  model_final_kernel<<<1, 256, 0, st>>>(partials, nb, scalars);
}

// ---------------------------------------------------------------------------------------------
// retract (Values::retract, gtsam/nonlinear/Values.cpp:53-64): one thread per variable
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) retract_kernel(DevProblem P, const double* values, const double* delta,
                                                      double* out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P.n_vars) return;
  const double* x = values + P.var_state_off[v];
  const double* d = delta + P.var_tan_off[v];
  double* y = out + P.var_state_off[v];
  const int type = P.var_type[v];
  if (type == GSX_VAR_VECTOR) {
    for (int i = 0; i < P.var_dim[v]; ++i) y[i] = x[i] + d[i];
  } else if (type == GSX_VAR_POSE2) {
    const P2 r = compose(load_pose2(x), load_pose2(d));
    y[0] = r.x; y[1] = r.y; y[2] = theta(r);
  } else {
    double xi[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xi[i] = d[i];
    store_pose3(compose(load_pose3(x), pose3_expmap(xi)), y);
    if (type == GSX_VAR_CAMERA) {
      y[12] = x[12] + d[6]; y[13] = x[13] + d[7]; y[14] = x[14] + d[8];
      y[15] = x[15]; y[16] = x[16];
    }
  }
}
void launch_retract(const DevProblem& P, const double* values, const double* delta, double* out, hipStream_t st) {
  if (P.n_vars) retract_kernel<<<(P.n_vars + 255) / 256, 256, 0, st>>>(P, values, delta, out);
}

// ---------------------------------------------------------------------------------------------
// Hessian panels: for variable A (rows: A itself, its later-eliminated neighbours, rhs)
//   panel[dst+i, j] = sum over terms  sum_r J[r, colB+i] * J[r, colA+j]
// One workgroup per variable, `copies` waves each owning a private LDS copy of the panel and a
// strided share of the term list; copies are summed in wave order => deterministic.
// (JacobianFactor::updateHessian, gtsam/linear/JacobianFactor.cpp:586-624, restructured as a gather.)
// ---------------------------------------------------------------------------------------------
__global__ void assemble_h_kernel(DevProblem P, DevSymbolic S, const int* vars, const double* jac, double* H) {
  extern __shared__ double lds[];
  const int v = vars[blockIdx.x];
  const int dA = P.var_dim[v], rows = S.h_rows[v], psize = rows * dA;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  double* panel = lds + (size_t)wave * psize;
  for (int e = lane; e < psize; e += 64) panel[e] = 0;
  // Each wave takes a CONTIGUOUS chunk of the term list (the diagonal and rhs terms of one factor are neighbours and
  // share the factor's Jacobian cache lines).  The loop is bound by DEPENDENT MEMORY ROUND TRIPS: the 32-byte
  // records of up to 64 terms are fetched with one coalesced load (a record per lane) and broadcast with
  // v_readlane, and the Jacobian loads of kTG terms — rows 0 and 1 of every term, i.e. everything for the 2-row
  // SFM factors — are issued back to back before the first FMA.  (The compiler serialised the per-row loads of
  // the earlier two-term scalar-record form: six round trips per pair of terms.)  Terms are accumulated into the
  // panel in list order: deterministic.
  constexpr int kTG = 3;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const i64 tb0 = S.term_ptr[v], tn = S.term_ptr[v + 1] - tb0;
  const i64 chunk = ((tn + nw - 1) / nw + 1) & ~i64(1);
  const i64 t1 = min(tb0 + tn, tb0 + (uw + 1) * chunk);
  for (i64 base = tb0 + uw * chunk; base < t1; base += 64) {
    const int nblk = (int)min((i64)64, t1 - base);
    TermRec rec{0, 0, 0, 0, 0, 0, 0};
    if (lane < nblk) rec = S.terms[base + lane];
    for (int g = 0; g < nblk; g += kTG) {
      const double* tj[kTG];
      int tm[kTG], tA[kTG], tB[kTG], tdB[kTG], tdst[kTG];
      float trd[kTG];
      bool tpair[kTG];
      int mmax = 0, nemax = 0;
#pragma unroll
      for (int i = 0; i < kTG; ++i) {
        const int src = min(g + i, 63);  // wave-uniform
        const bool in = g + i < nblk;
        const i64 joff = readlane_i64(rec.jac, src);
        tj[i] = jac + joff;
        tm[i] = in ? __builtin_amdgcn_readlane(rec.m, src) : 0;
        tA[i] = __builtin_amdgcn_readlane(rec.colA, src);
        tB[i] = __builtin_amdgcn_readlane(rec.colB, src);
        tdB[i] = in ? __builtin_amdgcn_readlane(rec.dB, src) : 0;
        tdst[i] = __builtin_amdgcn_readlane(rec.dst, src);
        trd[i] = __builtin_amdgcn_rcpf((float)max(tdB[i], 1));
        tpair[i] = tm[i] == 2 && (joff & 1) == 0;
        mmax = max(mmax, tm[i]);
        nemax = max(nemax, tdB[i] * dA);
      }
      for (int e = lane; e < nemax; e += 64) {
        double acc[kTG], a0[kTG], b0[kTG], a1[kTG], b1[kTG];
        unsigned oa[kTG], ob[kTG];  // 32-bit lane offsets from the term's (wave-uniform, scalar) Jacobian base
        bool on[kTG];
        int ii[kTG], jj[kTG];
#pragma unroll
        for (int i = 0; i < kTG; ++i) {
          on[i] = e < tdB[i] * dA;
          divmod_small(e, tdB[i], trd[i], jj[i], ii[i]);
          oa[i] = (unsigned)((tB[i] + ii[i]) * tm[i]);
          ob[i] = (unsigned)((tA[i] + jj[i]) * tm[i]);
        }
#pragma unroll
        for (int i = 0; i < kTG; ++i) {
          if (tpair[i]) {  // wave-uniform: two-row factor at an even offset — rows 0 and 1 of a column in one 16-byte load
            double2 av = {0.0, 0.0}, bv = {0.0, 0.0};
            if (on[i]) {
              av = *reinterpret_cast<const double2*>(tj[i] + oa[i]);
              bv = *reinterpret_cast<const double2*>(tj[i] + ob[i]);
            }
            a0[i] = av.x; a1[i] = av.y; b0[i] = bv.x; b1[i] = bv.y;
          } else {
            const bool r0 = on[i] && tm[i] > 0, r1 = on[i] && tm[i] > 1;
            a0[i] = r0 ? tj[i][oa[i]] : 0.0;
            b0[i] = r0 ? tj[i][ob[i]] : 0.0;
            a1[i] = r1 ? tj[i][oa[i] + 1u] : 0.0;
            b1[i] = r1 ? tj[i][ob[i] + 1u] : 0.0;
          }
        }
#pragma unroll
        for (int i = 0; i < kTG; ++i) acc[i] = a0[i] * b0[i] + a1[i] * b1[i];
        for (int r = 2; r < mmax; ++r) {  // (16-byte pair loads here too were measured: the extra registers cost more)
          double ar[kTG], br[kTG];
#pragma unroll
          for (int i = 0; i < kTG; ++i) {
            const bool rr = on[i] && r < tm[i];
            ar[i] = rr ? tj[i][oa[i] + (unsigned)r] : 0.0;
            br[i] = rr ? tj[i][ob[i] + (unsigned)r] : 0.0;
          }
#pragma unroll
          for (int i = 0; i < kTG; ++i) acc[i] += ar[i] * br[i];
        }
#pragma unroll
        for (int i = 0; i < kTG; ++i)
          if (on[i]) panel[tdst[i] + ii[i] + jj[i] * rows] += acc[i];
      }
    }
  }
  __syncthreads();
  double* out = H + S.h_off[v];
  for (int e = threadIdx.x; e < psize; e += blockDim.x) {
    double s = 0;
    for (int w = 0; w < nw; ++w) s += lds[(size_t)w * psize + e];
    out[e] = s;
  }
}
// Variables whose factors are all NARROW — at most 16 columns with the rhs, at most 8 rows: every factor of a pose graph
// (6 x 13, 3 x 7), priors, pose-landmark projections (2 x 10) — and that have at most 64 terms.  A wave per variable, four
// variables a workgroup.  Per factor ONE matrix-core product G = [A b]'[A b] (16 x 16 tile, v_mfma_f64_16x16x4 per four
// rows; the operand X[k][i] = entry (row k, column i) of the block is both the A and the B operand), loaded with one or two
// loads a lane — the generic kernel above reads every entry of the block once per panel entry it contributes to, a
// dozen dependent round trips per variable where this one has two: the term records, then the blocks of up to kTF factors
// together.  The panel entries of the factor's terms are picked out of G and added to the LDS panel term by term in list
// order: deterministic.  (JacobianFactor::updateHessian, gtsam/linear/JacobianFactor.cpp:586-624.)
constexpr int kTF = 8;   // factors whose blocks are in flight together
__global__ void __launch_bounds__(256) assemble_h_tile_kernel(DevProblem P, DevSymbolic S, const int* vars, int count,
                                                              int panel_stride, const double* jac, double* H) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (slot >= count) return;
  const int v = vars[slot];
  const int dA = P.var_dim[v], rows = S.h_rows[v], psize = rows * dA;
  double* panel = lds + (size_t)wave * panel_stride;
  for (int e = lane; e < psize; e += 64) panel[e] = 0;
  const i64 tb0 = S.term_ptr[v];
  const int nt = (int)(S.term_ptr[v + 1] - tb0);   // <= 64 (host)
  TermRec rec{0, 0, 0, 0, 0, 0, 0};
  if (lane < nt) rec = S.terms[tb0 + lane];
  const int li = lane & 15, lk = lane >> 4;
  int t = 0;
  while (t < nt) {
    // the next (up to kTF) factors: a factor's terms are consecutive and share its Jacobian offset; the rhs term is its last
    int fbeg[kTF + 1];
    double x0[kTF], x1[kTF];
    int nf = 0, tt = t;
#pragma unroll
    for (int q = 0; q < kTF; ++q) {
      fbeg[q] = tt;
      if (tt < nt) {
        const i64 jo = readlane_i64(rec.jac, tt);
        int te = tt + 1;
        while (te < nt && readlane_i64(rec.jac, te) == jo) ++te;
        const int m = __builtin_amdgcn_readlane(rec.m, tt);
        const int ncols = __builtin_amdgcn_readlane(rec.colB, te - 1) + 1;
        const double* J = jac + jo;
        const bool in = li < ncols;
        x0[q] = (in && lk < m) ? J[li * m + lk] : 0.0;
        x1[q] = (in && lk + 4 < m) ? J[li * m + lk + 4] : 0.0;
        tt = te;
        nf = q + 1;
      } else {
        x0[q] = x1[q] = 0.0;
      }
    }
    fbeg[kTF] = tt;
#pragma unroll
    for (int q = 0; q < kTF; ++q) {
      if (q >= nf) break;
      v4d g = {0.0, 0.0, 0.0, 0.0};
      g = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], x0[q], g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], x1[q], g, 0, 0, 0);
      const int te = fbeg[q + 1];
      for (int k = fbeg[q]; k < te; ++k) {
        const int colA = __builtin_amdgcn_readlane(rec.colA, k), colB = __builtin_amdgcn_readlane(rec.colB, k);
        const int dB = __builtin_amdgcn_readlane(rec.dB, k), dst = __builtin_amdgcn_readlane(rec.dst, k);
        const int j = li - colA;
        if (j >= 0 && j < dA) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int i = lk + 4 * r4 - colB;   // G[lk + 4 r4][li] = sum_k X[k][colB + i] X[k][colA + j]
            if (i >= 0 && i < dB) panel[dst + i + j * rows] += g[r4];
          }
        }
      }
    }
    t = tt;
  }
  double* out = H + S.h_off[v];
  for (int e = lane; e < psize; e += 64) out[e] = panel[e];
}

// panels too large for LDS: one wave accumulates straight into the (zeroed) global panel
__global__ void assemble_h_global_kernel(DevProblem P, DevSymbolic S, const int* vars, const double* jac, double* H) {
  const int v = vars[blockIdx.x];
  const int dA = P.var_dim[v], rows = S.h_rows[v], psize = rows * dA;
  const int lane = threadIdx.x;
  double* panel = H + S.h_off[v];
  for (int e = lane; e < psize; e += 64) panel[e] = 0;
  __syncthreads();
  const i64 t1 = S.term_ptr[v + 1];
  for (i64 t = S.term_ptr[v]; t < t1; ++t) {
    const TermRec tr = S.terms[t];
    const double* Jf = jac + tr.jac;
    const int m = tr.m, colA = tr.colA, colB = tr.colB, dB = tr.dB, dst = tr.dst;
    const float rdB = 1.0f / (float)dB;
    for (int e = lane; e < dB * dA; e += 64) {
      int i, j;
      divmod_small(e, dB, rdB, j, i);
      double acc = 0;
      for (int r = 0; r < m; ++r) acc += Jf[(colB + i) * m + r] * Jf[(colA + j) * m + r];
      panel[dst + i + j * rows] += acc;  // an entry is always owned by the same lane
    }
  }
}

// Variables that are the LAST-eliminated one of all their factors (the cameras of a bundle adjustment under a Schur
// ordering) have a panel of just their own d x d block and the rhs row: panel = [J_A | b]' [J_A] summed over the factors —
// a (d+1) x K by K x d product with K = all the factors' rows, i.e. matrix-core work.  One workgroup of 4 waves per
// variable; a wave walks a contiguous quarter of the factor list, one v_mfma_f64_16x16x4 per four factor rows (two 2-row
// SFM factors per instruction; other row counts one factor at a time), the operand X[k][i] = column i of [J_A | b] at
// factor row k being BOTH the A and the B operand of the instruction (A[i][k] and B[k][j] share the lane layout).  Loads
// of kU steps are issued before their products.  The four partial 16 x 16 tiles are added in wave order: deterministic.
// (Reference: JacobianFactor::updateHessian, gtsam/linear/JacobianFactor.cpp:586-624.)
__device__ __forceinline__ void assemble_h_diag_body(const DevProblem& P, const DevSymbolic& S, const int* vars,
                                                     const double* jac, double* H, int bid) {
  __shared__ double red[4][256];
  const int v = vars[bid];
  const int d = P.var_dim[v], rows = d + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int uw = __builtin_amdgcn_readfirstlane(wave);
  const i64 t0 = S.term_ptr[v];
  const int nf = (int)((S.term_ptr[v + 1] - t0) >> 1);  // two terms (diagonal, rhs) per factor
  const int chunk = (((nf + 3) >> 2) + 1) & ~1;         // even: 2-row factors pair up inside a wave's share
  const int f0 = uw * chunk, f1 = min(nf, f0 + chunk);
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  constexpr int kU = 4;
  for (int base = f0; base < f1; base += 64) {
    const int nblk = min(64, f1 - base);
    // a factor per lane: where its [A b] starts, its rows, the column of this variable and of b
    i64 r_jac = 0;
    int r_m = 0, r_ca = 0, r_cb = 0;
    if (lane < nblk) {
      const TermRec a = S.terms[t0 + 2 * (i64)(base + lane)];
      const TermRec b = S.terms[t0 + 2 * (i64)(base + lane) + 1];
      r_jac = a.jac;
      r_m = a.m;
      r_ca = a.colA;
      r_cb = b.colB;
    }
    int q = 0;
    while (q < nblk) {
      double x[kU];
      int used = 0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        x[u] = 0.0;
        if (q < nblk) {  // wave-uniform
          const int m0 = __builtin_amdgcn_readlane(r_m, q);
          const bool pair = m0 == 2 && q + 1 < nblk && __builtin_amdgcn_readlane(r_m, min(q + 1, 63)) == 2;
          if (m0 <= 4) {
            // rows lk (and, paired, rows of the next factor in k slots 2, 3)
            // (v_readlane takes a wave-uniform lane: fetch both factors' records, then choose per lane)
            const int q1 = min(q + 1, 63);
            const i64 offA = readlane_i64(r_jac, q), offB = readlane_i64(r_jac, q1);
            const int caA = __builtin_amdgcn_readlane(r_ca, q), caB = __builtin_amdgcn_readlane(r_ca, q1);
            const int cbA = __builtin_amdgcn_readlane(r_cb, q), cbB = __builtin_amdgcn_readlane(r_cb, q1);
            const bool second = pair && lk >= 2;
            const int r = second ? lk - 2 : lk;
            const i64 off = second ? offB : offA;
            const int col = li < d ? (second ? caB : caA) + li : (second ? cbB : cbA);
            if (li <= d && r < m0) x[u] = jac[off + (i64)col * m0 + r];
            q += pair ? 2 : 1;
            used = u + 1;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (u < used) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], x[u], acc, 0, 0, 0);
      if (q < nblk && __builtin_amdgcn_readlane(r_m, q) > 4) {
        // a taller factor (a 9-row camera prior): four of its rows per instruction
        const i64 off = readlane_i64(r_jac, q);
        const int m = __builtin_amdgcn_readlane(r_m, q);
        const int col = li < d ? __builtin_amdgcn_readlane(r_ca, q) + li : __builtin_amdgcn_readlane(r_cb, q);
        for (int r0 = 0; r0 < m; r0 += 4) {
          const int r = r0 + lk;
          const double xv = (li <= d && r < m) ? jac[off + (i64)col * m + r] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, xv, acc, 0, 0, 0);
        }
        ++q;
      }
    }
  }
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) red[wave][(lk + 4 * qq) * 16 + li] = acc[qq];
  __syncthreads();
  double* out = H + S.h_off[v];
  for (int e = threadIdx.x; e < 256; e += blockDim.x) {
    const int i = e >> 4, j = e & 15;  // entry (i, j) of X'X: i = row of the panel (d = the rhs row), j = column
    if (i <= d && j < d) out[i + j * rows] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
  }
}
__global__ void __launch_bounds__(256) assemble_h_diag_kernel(DevProblem P, DevSymbolic S, const int* vars, const double* jac,
                                                              double* H) {
  assemble_h_diag_body(P, S, vars, jac, H, blockIdx.x);
}

// "Star" variables: every factor of the variable is a binary factor with a LATER-eliminated partner and all factors
// have the same shape (the landmarks of a bundle adjustment: (camera, point) factors of 2 rows).  The panel is then
// one off-diagonal block per factor, each written exactly once, plus the variable's own block and the rhs row summed
// over the factors.  One wave per variable, four variables per workgroup, no LDS: ALL output entries of up to 64
// factors are spread flat over the lanes (entry -> factor, row, column by two integer divisions; the factor's record
// comes from the lane that loaded it, ds_bpermute), so the lanes stay busy where the term-by-term kernel keeps 27 / 9 /
// 3 of 64 active; the d*d + d entries of the own block and the rhs are accumulated by one lane each over the factors in
// list order: deterministic.  2-row factors load a column as one 16-byte pair.
__global__ void __launch_bounds__(256) assemble_h_star_kernel(DevProblem P, DevSymbolic S, const int* vars, int count,
                                                              const double* jac, double* H) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= count) return;
  const int v = vars[w];
  const int d = P.var_dim[v], rows = S.h_rows[v];
  const i64 t0 = S.term_ptr[v];
  const int nf = (int)((S.term_ptr[v + 1] - t0) / 3);  // (diagonal, partner block, rhs) per factor
  double* panel = H + S.h_off[v];
  const TermRec a0 = S.terms[t0], b0 = S.terms[t0 + 1], c0 = S.terms[t0 + 2];  // the common shape
  const int m = a0.m, colA = a0.colA, colB = b0.colB, dB = b0.dB, colR = c0.colB;
  const int nper = dB * d, nown = d * d + d;
  // this lane's entry of the own block / rhs row: columns (ci, cj) of the factor's [A b], destination
  int ci = 0, cj = 0, odst = 0;
  if (lane < d * d) {
    const int i = lane % d, j = lane / d;
    ci = colA + i; cj = colA + j; odst = i + j * rows;
  } else if (lane < nown) {
    const int j = lane - d * d;
    ci = colR; cj = colA + j; odst = (rows - 1) + j * rows;
  }
  double own = 0.0;
  const float rnper = 1.0f / (float)nper, rdB = 1.0f / (float)dB;
  for (int fb = 0; fb < nf; fb += 64) {
    const int nfb = min(64, nf - fb);
    i64 r_jac = 0;
    int r_dst = 0;
    if (lane < nfb) {
      r_jac = S.terms[t0 + 3 * (i64)(fb + lane)].jac;
      r_dst = S.terms[t0 + 3 * (i64)(fb + lane) + 1].dst;
    }
    // partner blocks: entry e -> factor q, row i of the partner's dB, column j of this variable's d
    const int total = nfb * nper;
    for (int e0 = 0; e0 < total; e0 += 64) {
      const int e = e0 + lane;
      const bool on = e < total;
      int q, rem, i, j;
      divmod_small(on ? e : 0, nper, rnper, q, rem);
      divmod_small(rem, dB, rdB, j, i);
      const int lo = __shfl((int)(r_jac & 0xffffffff), q), hi = __shfl((int)(r_jac >> 32), q);
      const i64 joff = ((i64)hi << 32) | (unsigned)lo;
      const int dst = __shfl(r_dst, q);
      if (on) {
        const double* Jf = jac + joff;
        double acc;
        if (m == 2 && (joff & 1) == 0) {
          const double2 x = reinterpret_cast<const double2*>(Jf)[colB + i];
          const double2 y = reinterpret_cast<const double2*>(Jf)[colA + j];
          acc = x.x * y.x + x.y * y.y;
        } else {
          acc = 0.0;
          for (int r = 0; r < m; ++r) acc += Jf[(colB + i) * m + r] * Jf[(colA + j) * m + r];
        }
        panel[dst + i + j * rows] = acc;
      }
    }
    // own block and rhs row: one lane per entry, over the factors of the chunk in list order
    for (int q = 0; q < nfb; ++q) {
      const i64 joff = readlane_i64(r_jac, q);
      if (lane < nown) {
        const double* Jf = jac + joff;
        if (m == 2 && (joff & 1) == 0) {
          const double2 x = reinterpret_cast<const double2*>(Jf)[ci];
          const double2 y = reinterpret_cast<const double2*>(Jf)[cj];
          own += x.x * y.x + x.y * y.y;
        } else {
          for (int r = 0; r < m; ++r) own += Jf[ci * m + r] * Jf[cj * m + r];
        }
      }
    }
  }
  if (lane < nown) panel[odst] = own;
}

// The same work for a BUNDLE of up to kStarVars consecutive star variables of one shape with at most 64 factors between
// them, one wave per bundle.  A wave of assemble_h_star_kernel walks four dependent load levels (variable tables ->
// term records -> Jacobians -> stores) for ONE variable and is bound by that chain times the number of waves; here the
// four levels are walked once per bundle, with all the bundle's records fetched by one load instruction per level and
// all its partner-block entries flat over the lanes.  Entry for entry the arithmetic is that of the one-variable
// kernel (which gsx_relinearize_partial keeps using on its filtered lists): same bits.
constexpr int kStarVars = 5;  // kStarVars * (d*d + d) own entries must fit a wave (d = 3: 60 lanes)
__device__ __forceinline__ void assemble_h_star_bundle_body(const DevProblem& P, const DevSymbolic& S, const int* vars,
                                                            const int2* bundles, int count, const double* jac, double* H,
                                                            int bid) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(bid * 4 + (threadIdx.x >> 6));
  if (w >= count) return;
  const int2 bd = bundles[w];  // first variable (position in vars), number of variables
  const int nv = bd.y;
  // level 1: the variables' tables, a variable per lane
  int s_v = 0, s_rows = 0, s_nf = 0;
  i64 s_t0 = 0, s_hoff = 0;
  if (lane < nv) {
    s_v = vars[bd.x + lane];
    s_rows = S.h_rows[s_v];
    s_t0 = S.term_ptr[s_v];
    s_nf = (int)((S.term_ptr[s_v + 1] - s_t0) / 3);
    s_hoff = S.h_off[s_v];
  }
  // first factor of each variable in the bundle's factor numbering (exclusive prefix sum over <= kStarVars lanes)
  int s_first = 0, nft = 0;
#pragma unroll
  for (int k = 0; k < kStarVars; ++k) {
    const int c = __builtin_amdgcn_readlane(s_nf, k);
    if (lane == k) s_first = nft;
    nft += (k < nv) ? c : 0;
  }
  const int d = P.var_dim[__builtin_amdgcn_readfirstlane(s_v)];
  // level 2: the factors' records, a factor per lane (nft <= 64), and the common shape from the first factor
  int f_slot = 0;
#pragma unroll
  for (int k = 1; k < kStarVars; ++k)
    if (k < nv && lane >= __builtin_amdgcn_readlane(s_first, k)) f_slot = k;
  i64 r_jac = 0;
  int r_dst = 0;
  {
    const i64 t0s = ((i64)__shfl((int)(s_t0 >> 32), f_slot) << 32) | (unsigned)__shfl((int)(s_t0 & 0xffffffff), f_slot);
    const int fl = lane - __shfl(s_first, f_slot);
    if (lane < nft) {
      r_jac = S.terms[t0s + 3 * (i64)fl].jac;
      r_dst = S.terms[t0s + 3 * (i64)fl + 1].dst;
    }
  }
  const i64 t00 = readlane_i64(s_t0, 0);
  const TermRec a0 = S.terms[t00], b0 = S.terms[t00 + 1], c0t = S.terms[t00 + 2];
  const int m = a0.m, colA = a0.colA, colB = b0.colB, dB = b0.dB, colR = c0t.colB;
  const int nper = dB * d, nown = d * d + d;
  const int f_rows = __shfl(s_rows, f_slot);
  const i64 f_hoff = ((i64)__shfl((int)(s_hoff >> 32), f_slot) << 32) | (unsigned)__shfl((int)(s_hoff & 0xffffffff), f_slot);
  // level 3: partner blocks, all entries of all factors flat over the lanes
  const float rnper = 1.0f / (float)nper, rdB = 1.0f / (float)dB;
  const int total = nft * nper;
  for (int e0 = 0; e0 < total; e0 += 64) {
    const int e = e0 + lane;
    const bool on = e < total;
    int q, rem, i, j;
    divmod_small(on ? e : 0, nper, rnper, q, rem);
    divmod_small(rem, dB, rdB, j, i);
    const i64 joff = ((i64)__shfl((int)(r_jac >> 32), q) << 32) | (unsigned)__shfl((int)(r_jac & 0xffffffff), q);
    const int dst = __shfl(r_dst, q);
    const int rows = __shfl(f_rows, q);
    const i64 hoff = ((i64)__shfl((int)(f_hoff >> 32), q) << 32) | (unsigned)__shfl((int)(f_hoff & 0xffffffff), q);
    if (on) {
      const double* Jf = jac + joff;
      double acc;
      if (m == 2 && (joff & 1) == 0) {
        const double2 x = reinterpret_cast<const double2*>(Jf)[colB + i];
        const double2 y = reinterpret_cast<const double2*>(Jf)[colA + j];
        acc = x.x * y.x + x.y * y.y;
      } else {
        acc = 0.0;
        for (int r = 0; r < m; ++r) acc += Jf[(colB + i) * m + r] * Jf[(colA + j) * m + r];
      }
      H[hoff + dst + i + j * rows] = acc;
    }
  }
  // own block and rhs row: a lane per (variable, entry), over that variable's factors in list order
  {
    int slot, ent;
    divmod_small(lane, nown, 1.0f / (float)nown, slot, ent);
    const bool on = slot < nv;
    const int sl = on ? slot : 0;
    const int rows = __shfl(s_rows, sl), first = __shfl(s_first, sl), nf = __shfl(s_nf, sl);
    const i64 hoff = ((i64)__shfl((int)(s_hoff >> 32), sl) << 32) | (unsigned)__shfl((int)(s_hoff & 0xffffffff), sl);
    int ci, cj, odst;
    if (ent < d * d) {
      const int i = ent % d, j = ent / d;
      ci = colA + i; cj = colA + j; odst = i + j * rows;
    } else {
      const int j = ent - d * d;
      ci = colR; cj = colA + j; odst = (rows - 1) + j * rows;
    }
    int nfmax = 0;
#pragma unroll
    for (int k = 0; k < kStarVars; ++k) nfmax = max(nfmax, k < nv ? __builtin_amdgcn_readlane(s_nf, k) : 0);
    double own = 0.0;
    for (int it = 0; it < nfmax; ++it) {
      const int q = min(first + it, 63);
      const i64 joff = ((i64)__shfl((int)(r_jac >> 32), q) << 32) | (unsigned)__shfl((int)(r_jac & 0xffffffff), q);
      if (on && it < nf) {
        const double* Jf = jac + joff;
        if (m == 2 && (joff & 1) == 0) {
          const double2 x = reinterpret_cast<const double2*>(Jf)[ci];
          const double2 y = reinterpret_cast<const double2*>(Jf)[cj];
          own += x.x * y.x + x.y * y.y;
        } else {
          for (int r = 0; r < m; ++r) own += Jf[ci * m + r] * Jf[cj * m + r];
        }
      }
    }
    if (on) H[hoff + odst] = own;
  }
}
__global__ void __launch_bounds__(256) assemble_h_star_bundle_kernel(DevProblem P, DevSymbolic S, const int* vars,
                                                                    const int2* bundles, int count, const double* jac,
                                                                    double* H) {
  assemble_h_star_bundle_body(P, S, vars, bundles, count, jac, H, blockIdx.x);
}
// Both in ONE launch: the diagonal-panel variables (the cameras: a workgroup each, far fewer than the GPU holds) go
// first and run beside the star bundles (the landmarks) instead of before them — the two kernels write different panels
// and each is bound by its own chain of dependent loads, not by a resource the other needs.
__global__ void __launch_bounds__(256) assemble_h_diag_star_kernel(DevProblem P, DevSymbolic S, const int* dvars, int dcount,
                                                                  const int* svars, const int2* bundles, int scount,
                                                                  const double* jac, double* H) {
  if ((int)blockIdx.x < dcount) assemble_h_diag_body(P, S, dvars, jac, H, blockIdx.x);
  else assemble_h_star_bundle_body(P, S, svars, bundles, scount, jac, H, (int)blockIdx.x - dcount);
}
void launch_assemble_h_star_bundles(const DevProblem& P, const DevSymbolic& S, const int* vars, const int2* bundles,
                                    int count, const double* jac, double* H, hipStream_t st) {
  if (count) assemble_h_star_bundle_kernel<<<(count + 3) / 4, 256, 0, st>>>(P, S, vars, bundles, count, jac, H);
}
void launch_assemble_h_diag_star(const DevProblem& P, const DevSymbolic& S, const int* dvars, int dcount, const int* svars,
                                 const int2* bundles, int scount, const double* jac, double* H, hipStream_t st) {
  assemble_h_diag_star_kernel<<<dcount + (scount + 3) / 4, 256, 0, st>>>(P, S, dvars, dcount, svars, bundles, scount, jac, H);
}

static int g_max_lds = -1;
int max_dynamic_lds() {
  if (g_max_lds < 0) {
    int dev = 0, v = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 65536;
    g_max_lds = v;
  }
  return g_max_lds;
}

void launch_assemble_h_group(const DevProblem& P, const DevSymbolic& S, const int* vars, int count, int threads,
                             int lds_bytes, bool global, const double* jac, double* H, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)assemble_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr = true;
  }
  if (!count) return;
  if (threads == 0) {  // (mode of the group: variables whose panel is their own block + rhs — matrix-core kernel)
    assemble_h_diag_kernel<<<count, 256, 0, st>>>(P, S, vars, jac, H);
    return;
  }
  if (threads == -1) {  // (mode: star variables, a wave each)
    assemble_h_star_kernel<<<(count + 3) / 4, 256, 0, st>>>(P, S, vars, count, jac, H);
    return;
  }
  if (threads == -2) {  // (mode: variables of narrow factors, a wave each; lds_bytes = four panels)
    assemble_h_tile_kernel<<<(count + 3) / 4, 256, lds_bytes, st>>>(P, S, vars, count, lds_bytes / 32, jac, H);
    return;
  }
  if (global) assemble_h_global_kernel<<<count, 64, 0, st>>>(P, S, vars, jac, H);
  else assemble_h_kernel<<<count, threads, lds_bytes, st>>>(P, S, vars, jac, H);
}

__global__ void hessian_diag_kernel(DevProblem P, DevSymbolic S, const double* H, double* diag) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P.n_vars) return;
  const int d = P.var_dim[v], rows = S.h_rows[v];
  const double* p = H + S.h_off[v];
  for (int k = 0; k < d; ++k) diag[P.var_tan_off[v] + k] = p[k + k * rows];
}
void launch_hessian_diag(const DevProblem& P, const DevSymbolic& S, const double* H, double* diag, hipStream_t st) {
  if (P.n_vars) hessian_diag_kernel<<<(P.n_vars + 255) / 256, 256, 0, st>>>(P, S, H, diag);
}

// damping weights D: 1 (lambda I) or clamp(diag H) (diagonalDamping) —
// gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:293-299, internal/LevenbergMarquardtState.h:125-156
__global__ void make_damping_kernel(int n, const double* hdiag, int diagonal, double mind, double maxd, double* damp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  damp[i] = diagonal ? fmin(fmax(hdiag[i], mind), maxd) : 1.0;
}
void launch_make_damping(int n, const double* hdiag, int diagonal, double mind, double maxd, double* damp,
                         hipStream_t st) {
  if (n) make_damping_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, hdiag, diagonal, mind, maxd, damp);
}

// ---- marginal covariance of one variable ---------------------------------------------------------------------
// H = L L' (variable rows only).  (H^-1)_vv = sum_k (L^-1)_{k,v}' (L^-1)_{k,v}, and the columns (L^-1)_{:,v} are the
// forward substitution L y = e_v, which only touches the cliques on the path from v's clique to the root: at a clique
// y_F = L11^-1 w_F, w_S -= L21 y_F, Sigma += y_F' y_F, and w_S moves to the parent's rows through the child's row map.
// One workgroup; w lives in LDS (two buffers of max_n x dA).  (The reference gets the same block by eliminating the
// Bayes tree down to a marginal factor and inverting its information: gtsam/nonlinear/Marginals.cpp:107-136.)
// With Y != nullptr the columns y = (L^-1)_{:,v} themselves are kept (Y[g + c * n_tan], zero off the path: the caller
// clears Y): the joint covariance of several variables is then Y_a' Y_b (joint_cross_kernel).
__global__ void __launch_bounds__(256) marginal_path_kernel(DevSymbolic S, const int* path, int npath, int loc, int dA,
                                                            int max_n, const double* arena, double* out, double* Y,
                                                            i64 n_tan) {
  extern __shared__ double mw[];
  double* w = mw;                        // (n - 1) x dA, column-major, ld = max_n
  double* wn = mw + (size_t)max_n * dA;  // the parent's
  __shared__ double yj[16];
  __shared__ double sig[16 * 16];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int e = tid; e < dA * dA; e += nt) sig[e] = 0.0;
  for (int e = tid; e < max_n * dA; e += nt) w[e] = 0.0;
  __syncthreads();
  if (tid < dA) w[(loc + tid) + tid * max_n] = 1.0;
  __syncthreads();
  for (int pi = 0; pi < npath; ++pi) {
    const int f = path[pi];
    const int n = S.fr_N[f], F = S.fr_F[f];
    const double* A = arena + S.fr_off[f];
    const bool big = (S.fr_lean[f] & 2) != 0;
    const double* Lp = big ? A + big_panel_offset(n) : A;
    for (int j = 0; j < F; ++j) {
      // L(r, j): inside j's 32-row diagonal tile of a big front the value sits in the front itself, below it in the L panel
      const int tile_end = big ? min((j / T + 1) * T, F) : 0;
      if (tid < dA) yj[tid] = w[j + tid * max_n] / A[j + (i64)j * n];
      __syncthreads();
      if (tid < dA) w[j + tid * max_n] = yj[tid];
      for (int e = tid; e < (n - 2 - j) * dA; e += nt) {
        const int r = j + 1 + e % (n - 2 - j), c = e / (n - 2 - j);
        const double l = (big && r >= tile_end) ? Lp[r + (i64)j * n] : A[r + (i64)j * n];
        w[r + c * max_n] -= l * yj[c];
      }
      __syncthreads();
    }
    // Sigma += y_F' y_F
    for (int e = tid; e < dA * dA; e += nt) {
      const int a = e % dA, b = e / dA;
      double acc = 0;
      for (int r = 0; r < F; ++r) acc += w[r + a * max_n] * w[r + b * max_n];
      sig[e] += acc;
    }
    if (Y) {
      const int* gi = S.gidx + S.gidx_ptr[f];
      for (int e = tid; e < F * dA; e += nt) {
        const int r = e % F, c = e / F;
        Y[gi[r] + (i64)c * n_tan] = w[r + c * max_n];
      }
    }
    if (pi + 1 < npath) {
      const int np = S.fr_N[path[pi + 1]];
      for (int e = tid; e < max_n * dA; e += nt) wn[e] = 0.0;
      __syncthreads();
      const int* cm = S.cmap + S.cmap_ptr[f];
      const int s1 = n - F;  // separator rows + rhs row; the rhs row (last) is not part of the system
      for (int e = tid; e < (s1 - 1) * dA; e += nt) {
        const int i = e % (s1 - 1), c = e / (s1 - 1);
        wn[cm[i] + c * max_n] = w[(F + i) + c * max_n];
      }
      (void)np;
      __syncthreads();
      double* t = w;
      w = wn;
      wn = t;
    }
    __syncthreads();
  }
  for (int e = tid; e < dA * dA; e += nt) out[e] = sig[e];
}
// out (dA x dB, column-major) = Ya' Yb over the n_tan rows: one workgroup per entry
__global__ void __launch_bounds__(256) joint_cross_kernel(const double* Ya, const double* Yb, i64 n_tan, int dA, double* out) {
  const int a = blockIdx.x % dA, b = blockIdx.x / dA;
  const double* pa = Ya + (i64)a * n_tan;
  const double* pb = Yb + (i64)b * n_tan;
  double acc = 0;
  for (i64 i = threadIdx.x; i < n_tan; i += blockDim.x) acc += pa[i] * pb[i];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
void launch_joint_cross(const double* Ya, const double* Yb, int64_t n_tan, int dA, int dB, double* out, hipStream_t st) {
  joint_cross_kernel<<<dA * dB, 256, 0, st>>>(Ya, Yb, n_tan, dA, out);
}
void launch_marginal_path(const DevSymbolic& S, const int* path, int npath, int loc, int dA, int max_n, const double* arena,
                          double* out, double* Y, int64_t n_tan, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)marginal_path_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr = true;
  }
  marginal_path_kernel<<<1, 256, (size_t)2 * max_n * dA * sizeof(double), st>>>(S, path, npath, loc, dA, max_n, arena, out,
                                                                                Y, n_tan);
}

// ---- Dogleg helpers -------------------------------------------------------------------------------------------
__global__ void gradient_kernel(DevProblem P, DevSymbolic S, const double* H, double* g) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P.n_vars) return;
  const int d = P.var_dim[v], rows = S.h_rows[v];
  const double* p = H + S.h_off[v];
  for (int k = 0; k < d; ++k) g[P.var_tan_off[v] + k] = p[(rows - 1) + k * rows];  // rhs row of the panel = (A'b)_v
}
void launch_gradient(const DevProblem& P, const DevSymbolic& S, const double* H, double* g, hipStream_t st) {
  gradient_kernel<<<(P.n_vars + 255) / 256, 256, 0, st>>>(P, S, H, g);
}
__global__ void __launch_bounds__(256) ax_sqnorm_kernel(DevProblem P, const double* jac, const double* xv, double* partials) {
  double acc = 0;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < P.n_factors; f += gridDim.x * blockDim.x) {
    const int m = P.f_rows[f];
    const double* J = jac + P.f_jac_off[f];
    for (int r = 0; r < m; ++r) {
      double e = 0;
      int col = 0;
      for (int k = P.f_key_ptr[f]; k < P.f_key_ptr[f + 1]; ++k) {
        const int v = P.f_vars[k];
        const double* x = xv + P.var_tan_off[v];
        const int d = P.var_dim[v];
        for (int c = 0; c < d; ++c, ++col) e += J[col * m + r] * x[c];
      }
      acc += e * e;
    }
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
void launch_ax_sqnorm(const DevProblem& P, const double* jac, const double* x, double* partials, int cap, double* scalars,
                      int slot, hipStream_t st) {
  int nb = (P.n_factors + 255) / 256;
  nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
  ax_sqnorm_kernel<<<nb, 256, 0, st>>>(P, jac, x, partials);
  reduce_final_kernel<<<1, 256, 0, st>>>(partials, nb, 1, scalars, slot);
}
__global__ void __launch_bounds__(256) vec_dot_kernel(const double* a, const double* b, i64 n, double* partials) {
  double acc = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) acc += a[i] * b[i];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
void launch_vec_dot(const double* a, const double* b, int64_t n, double* partials, int cap, double* scalars, int slot,
                    hipStream_t st) {
  int nb = (int)((n + 1023) / 1024);
  nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
  vec_dot_kernel<<<nb, 256, 0, st>>>(a, b, n, partials);
  reduce_final_kernel<<<1, 256, 0, st>>>(partials, nb, 1, scalars, slot);
}
__global__ void vec_axpby_kernel(double* out, double alpha, const double* a, double beta, const double* b, i64 n) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] + beta * b[i];
}
void launch_vec_axpby(double* out, double alpha, const double* a, double beta, const double* b, int64_t n, hipStream_t st) {
  if (n > 0) vec_axpby_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(out, alpha, a, beta, b, n);
}

// ---- sharded problems: what crosses the exchange callback besides the cap fronts ----------------------------------
__global__ void shard_pack_kernel(const double* scalars, const DevStatus* status, unsigned dirty, double* x) {
  const int k = threadIdx.x;
  if (k >= kXScalars) return;
  double v = 0.0;
  if (k < SC_COUNT) v = ((dirty >> k) & 1u) ? scalars[k] : 0.0;
  else if (k == 12) v = (dirty & kXFact) ? (double)status->n_fail : 0.0;
  else if (k == 13) v = (dirty & kXFact) ? (double)status->n_nonfinite : 0.0;
  else if (k == 14) v = (dirty & kXLin) ? (double)status->n_cheirality : 0.0;
  x[k] = v;
}
__global__ void shard_unpack_kernel(const double* x, unsigned dirty, double* scalars, DevStatus* status) {
  const int k = threadIdx.x;
  if (k >= kXScalars) return;
  const double cap = 2147483647.0;
  if (k < SC_COUNT) {
    if ((dirty >> k) & 1u) scalars[k] = x[k];
  } else if (k == 12) {
    if (dirty & kXFact) status->n_fail = (int)fmin(x[k], cap);
  } else if (k == 13) {
    if (dirty & kXFact) status->n_nonfinite = (int)fmin(x[k], cap);
  } else if (k == 14) {
    if (dirty & kXLin) status->n_cheirality = (int)fmin(x[k], cap);
  }
}
void launch_shard_pack(const double* scalars, const DevStatus* status, unsigned dirty, double* x, hipStream_t st) {
  static_assert(SC_COUNT == 12 && kXScalars >= 15, "exchange vector layout");
  shard_pack_kernel<<<1, 64, 0, st>>>(scalars, status, dirty, x);
}
void launch_shard_unpack(const double* x, unsigned dirty, double* scalars, DevStatus* status, hipStream_t st) {
  shard_unpack_kernel<<<1, 64, 0, st>>>(x, dirty, scalars, status);
}
__global__ void mask_copy_kernel(const double* in, const unsigned char* mask, i64 n, double* out) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mask[i] ? in[i] : 0.0;
}
void launch_mask_copy(const double* in, const unsigned char* mask, int64_t n, double* out, hipStream_t st) {
  if (n > 0) mask_copy_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, mask, n, out);
}

__device__ inline int state_len(int type, int dim) {
  return type == GSX_VAR_POSE2 ? 3 : (type == GSX_VAR_POSE3 ? 12 : (type == GSX_VAR_CAMERA ? 17 : dim));
}
__global__ void scatter_states_kernel(DevProblem P, const int* vars, const int* src_off, int n, const double* src,
                                      double* values) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = vars[i];
  const int len = state_len(P.var_type[v], P.var_dim[v]);
  const double* s = src + src_off[i];
  double* d = values + P.var_state_off[v];
  for (int k = 0; k < len; ++k) d[k] = s[k];
}
void launch_scatter_states(const DevProblem& P, const int* vars, const int* src_off, int n, const double* src,
                           double* values, hipStream_t st) {
  if (n > 0) scatter_states_kernel<<<(n + 255) / 256, 256, 0, st>>>(P, vars, src_off, n, src, values);
}

__global__ void set_scalar_kernel(double* scalars, int slot, double v) { scalars[slot] = v; }
void launch_set_scalar(double* scalars, int slot, double v, hipStream_t st) {
  set_scalar_kernel<<<1, 1, 0, st>>>(scalars, slot, v);
}
// start of a factorization: lambda + the status words of the factorization / back-substitution in ONE tiny launch
// (three memset / memcpy nodes cost ~5 us each on the stream; the cheirality count of the last linearize is kept)
__global__ void begin_factorization_kernel(double* scalars, double lambda, DevStatus* status, int* tree_cursors) {
  if (tree_cursors)
    for (int k = 0; k < kTreeCursors; ++k) tree_cursors[k] = 0;
  scalars[SC_LAMBDA] = lambda;
  status->n_fail = 0;
  status->first_front = 0x7fffffff;
  status->n_nonfinite = 0;
  status->n_backsub = 0;
}
void launch_begin_factorization(double* scalars, double lambda, DevStatus* status, int* tree_cursors, hipStream_t st) {
  begin_factorization_kernel<<<1, 1, 0, st>>>(scalars, lambda, status, tree_cursors);
}

constexpr int kVarStage = 32, kChildStage = 16;
// LDS scratch of one workgroup of the LDS-front kernels, declared once per kernel and handed to the bodies (two bodies
// inlined into one kernel would otherwise each bring their own copy: the medium tier's kernel needs every kilobyte)
struct FrontScratch {
  VarRec vrec[kVarStage];
  ChildRec crec[kChildStage];
  double Eb[16 * 17];   // (L_pp^-1)[i][c] of the current 16 columns at i * 17 + c
  int vpre[kVarStage + 1];
  int failed;
};

// A launch that asks for more LDS than a CU has must never reach the GPU (the hardware faults instead of refusing):
// dynamic + the kernel's static LDS against the 160 KB of a gfx950 CU.  A refused launch leaves its fronts unfactored,
// which the status words / the parity tests show; the symbolic analysis sizes every class so that this never fires.
static bool lds_fits(const void* kernel, size_t dynamic_bytes, const char* what) {
  hipFuncAttributes a;
  size_t fixed = 8192;
  if (hipFuncGetAttributes(&a, kernel) == hipSuccess) fixed = a.sharedSizeBytes;
  if (dynamic_bytes + fixed <= (size_t)160 * 1024) return true;
  fprintf(stderr, "gsx: %s needs %zu + %zu bytes of LDS: launch refused\n", what, dynamic_bytes, fixed);
  return false;
}

// ---------------------------------------------------------------------------------------------
// in-LDS partial Cholesky (choleskyPartial, gtsam/base/cholesky.cpp:108-159), lower form:
// columns 0..F-1 become [L11; L21] (incl. the rhs row), the trailing block C -= L21 L21'.
// All threads of the block take part.  Returns (in every thread) 0 ok / 1 failed.
// ---------------------------------------------------------------------------------------------
#ifdef GSX_STAMP
__device__ unsigned long long g_stamp[8];
#define STAMP_BEGIN unsigned long long st0__ = __builtin_amdgcn_s_memtime();
#define STAMP_ADD(slot)                                                      \
  {                                                                          \
    unsigned long long t__ = __builtin_amdgcn_s_memtime();                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_stamp[slot] += t__ - st0__;   \
    st0__ = t__;                                                             \
  }
#else
#define STAMP_BEGIN
#define STAMP_ADD(slot)
#endif
template <int PB>
__device__ inline int lds_partial_cholesky_t(double* L, int n, int F, bool gap_test) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  int fail = 0;
  // Blocked right-looking: panels of PB columns are factored with rank-1 updates confined to the
  // panel (n x PB entries per pivot instead of n x n), then ONE rank-PB sweep updates the rest of the
  // matrix — 1/PB of the LDS traffic and of the full-matrix passes of the unblocked form.
  // Inside a panel thread r keeps ROW r of the panel (PB values) in registers; per pivot the only shared
  // data is column j restricted to the PB x PB diagonal block, double-buffered in LDS => ONE barrier per
  // pivot (requires blockDim >= n, which the size classes guarantee).
  __shared__ double dcol[2][PB];
  const int row = tid;
  for (int j0 = 0; j0 < F; j0 += PB) {
    const int pb = min(PB, F - j0), jend = j0 + pb;
    STAMP_BEGIN
    const bool active = row >= j0 && row < n;
    const int dk = row - j0;  // position inside the diagonal block (valid when 0 <= dk < pb)
    double a[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) a[k] = (active && k < pb) ? L[row + (j0 + k) * n] : 0.0;
    if (dk >= 0 && dk < pb) dcol[0][dk] = a[0];
#pragma unroll
    for (int jj = 0; jj < PB; ++jj) {
      if (jj < pb) {
        __syncthreads();
        const double* cur = dcol[jj & 1];
        const double p = cur[jj];
        if (!(p > 0)) fail = 1;  // Eigen::LLT reports NumericalIssue on a non-positive pivot
        const double inv = (p > 0) ? rsqrt(p) : 1.0;
        if (active && dk >= jj) {
          const double l = (dk == jj) ? p * inv : a[jj] * inv;
          a[jj] = l;
#pragma unroll
          for (int k = jj + 1; k < PB; ++k)
            if (k < pb && dk >= k) a[k] -= l * (cur[k] * inv);
          if (jj + 1 < pb && dk > jj && dk < pb) dcol[(jj + 1) & 1][dk] = a[jj + 1 < PB ? jj + 1 : 0];
        }
      }
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < PB; ++k)
        if (k < pb && dk >= k) L[row + (j0 + k) * n] = a[k];
    }
    __syncthreads();
    STAMP_ADD(0)
    // rank-pb trailing update C -= P P' on the FP64 matrix cores: one wave per 16x16 tile of the lower
    // triangle, v_mfma_f64_16x16x4 (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
    // C/D row = (lane>>4) + 4 reg, col = lane&15), two k-steps for the 8-column panel.  Per lane that is
    // 4 LDS reads of operands + 8 of the C tile for 32 FMA-equivalents (the VALU form needed 18 reads for
    // 16 FMAs and was bound by the ~100-cycle LDS round trip).
    {
      const int m = n - jend, nt16 = (m + 15) >> 4, ntiles = nt16 * (nt16 + 1) / 2;
      const int li = lane & 15, lk = lane >> 4;
      for (int t = wave; t < ntiles; t += nw) {
        int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        const int tj = t - ti * (ti + 1) / 2;
        const int i0 = jend + ti * 16, c0 = jend + tj * 16;
        const int col = c0 + li;
        v4d acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = i0 + lk + 4 * q;
          acc[q] = (rr < n && col < n) ? L[rr + col * n] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < PB; kk += 4) {
          if (kk < pb) {  // wave-uniform
            const int k = kk + lk;
            const bool kin = k < pb;
            const double av = (kin && i0 + li < n) ? -L[(i0 + li) + (j0 + k) * n] : 0.0;
            const double bv = (kin && c0 + li < n) ? L[(c0 + li) + (j0 + k) * n] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = i0 + lk + 4 * q;
          if (rr < n && col < n) L[rr + col * n] = acc[q];
        }
      }
    }
    __syncthreads();
    STAMP_ADD(1)
  }
  // conditioning test on the last two pivots — cholesky.cpp:145-158 (the fronts of a tree are tested per REFERENCE
  // clique by cond_check_kernel instead: a relaxed front holds several)
  if (!gap_test) return fail;
  if (F >= 2) {
    int e2, e1;
    (void)frexp(L[(F - 2) + (F - 2) * n], &e2);
    (void)frexp(L[(F - 1) + (F - 1) * n], &e1);
    if (!(e2 - e1 < 12)) fail = 1;
  } else if (F == 1) {
    int e1;
    (void)frexp(L[0], &e1);
    if (!(e1 > -12)) fail = 1;
  }
  return fail;
}
// ---------------------------------------------------------------------------------------------
// The same partial Cholesky with the sequential part in REGISTERS (the scheme of bigfront.hip's big_diag): the
// row-per-thread panel above costs ~1300 cycles a pivot (an LDS write -> barrier -> read round, an rsqrt and a chain of
// dependent LDS reads), 16 us for the 30 frontal scalars of a typical pose-graph front — most of a front's life.
// Here, per panel of 16 columns:
//   D. wave 0 holds the diagonal 16 x 16 tile in the matrix-core accumulator layout (lane (li, lk): entries (row li,
//      column 4 q + lk)) and factors it with no LDS traffic and no barrier: per pivot one v_readlane, an rsqrt, and ONE
//      rank-1 matrix-core update of the tile; an identity tile carried through the same column operations comes out as
//      L_pp^-T (~280 cycles a pivot, measured).  Columns / rows beyond F are identity padding.
//   X. every 16-row tile below: X = A_ip L_pp^-T, four matrix-core instructions (the tile itself is the row-side
//      operand), in place.
//   U. rank-16 update of the trailing matrix from the panel's columns, one wave per 16 x 16 tile.
// Three workgroup barriers a panel (free for the one-wave fronts).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double tile_rsqrt(double d) {  // 1 / sqrt(d) from the hardware seed + one cubic correction
  const double y0 = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
__device__ __forceinline__ double tile_readlane(double v, int src_lane) {  // src_lane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ inline int lds_partial_cholesky_mfma(double* Lm, int n, int F, bool gap_test, FrontScratch& sc) {
  double* Eb = sc.Eb;
  int& failed = sc.failed;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (tid == 0) failed = 0;
  for (int jb = 0; jb < F; jb += 16) {
    const int w = min(16, F - jb), rb = jb + w;
    if (wave == 0) {
      v4d pt, E;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = jb + li, c = jb + 4 * q + lk;
        pt[q] = (r < F && c < F) ? (r >= c ? Lm[r + c * n] : 0.0) : (r == c ? 1.0 : 0.0);
        E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
      }
      // (straight-line: a branch between a matrix-core instruction and the v_readlane of its result hides the
      //  dependency from the compiler's hazard padding)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, lkj = j & 3;
        __builtin_amdgcn_sched_barrier(0);
        int lj = li;
        asm volatile("" : "+v"(lj));
        const double dj = tile_readlane(pt[q], lkj * 16 + j);
        const double sj = tile_rsqrt(dj);  // (a non-positive pivot makes L_jj a NaN: caught below)
        const bool colj = lk == lkj;
        const double xj = pt[q] * sj;
        const double xm = (colj && lj >= j) ? xj : 0.0;
        const double ej = colj ? E[q] * sj : 0.0;
        pt[q] = colj ? xm : pt[q];
        E[q] = colj ? ej : E[q];
        if (j < 15) {
          const double xu = (lj > j) ? xm : 0.0;
          pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
        }
      }
      if (lk == (li & 3)) {  // entry (li, li) sits in the lane with lk = li & 3, register li >> 2
        const int qd = li >> 2;
        const double l = (qd == 0) ? pt[0] : ((qd == 1) ? pt[1] : ((qd == 2) ? pt[2] : pt[3]));
        if (jb + li < F && !(l > 0)) failed = 1;  // non-positive or non-finite pivot (Eigen::LLT NumericalIssue)
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = jb + li, c = jb + 4 * q + lk;
        if (r < F && c < F && r >= c) Lm[r + c * n] = pt[q];
        Eb[(4 * q + lk) * 17 + li] = E[q];  // E[row li][col 4q+lk] = (L^-1)[4q+lk][li]
      }
    }
    __syncthreads();
    const int m = n - rb, T16 = (m + 15) >> 4;
    for (int t = wave; t < T16; t += nw) {
      const int r = rb + 16 * t + li;
      v4d nv = {0.0, 0.0, 0.0, 0.0};
      double pv[4], av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pv[s] = (r < n && 4 * s + lk < w) ? Lm[r + (jb + 4 * s + lk) * n] : 0.0;
        av[s] = Eb[li * 17 + 4 * s + lk];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) nv = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], pv[s], nv, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (r < n && 4 * q + lk < w) Lm[r + (jb + 4 * q + lk) * n] = nv[q];
    }
    __syncthreads();
    {
      const int ntiles = T16 * (T16 + 1) / 2;
      for (int t = wave; t < ntiles; t += nw) {
        int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        const int tj = t - ti * (ti + 1) / 2;
        const int i0 = rb + ti * 16, c0 = rb + tj * 16;
        const int col = c0 + li;
        v4d acc;
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = i0 + lk + 4 * q;
          acc[q] = (rr < n && col < n) ? Lm[rr + col * n] : 0.0;
          const int k = 4 * q + lk;
          av[q] = (k < w && i0 + li < n) ? -Lm[(i0 + li) + (jb + k) * n] : 0.0;
          bv[q] = (k < w && c0 + li < n) ? Lm[(c0 + li) + (jb + k) * n] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = i0 + lk + 4 * q;
          if (rr < n && col < n) Lm[rr + col * n] = acc[q];
        }
      }
    }
    __syncthreads();
  }
  int fail = failed;
  // conditioning test on the last two pivots — cholesky.cpp:145-158 (the fronts of a tree are tested per REFERENCE
  // clique by cond_check_kernel instead: a relaxed front holds several)
  if (!gap_test) return fail;
  if (F >= 2) {
    int e2, e1;
    (void)frexp(Lm[(F - 2) + (F - 2) * n], &e2);
    (void)frexp(Lm[(F - 1) + (F - 1) * n], &e1);
    if (!(e2 - e1 < 12)) fail = 1;
  } else if (F == 1) {
    int e1;
    (void)frexp(Lm[0], &e1);
    if (!(e1 > -12)) fail = 1;
  }
  return fail;
}
// panel width by frontal size: narrow panels keep the per-pivot register work small for the many cliques
// with a handful of frontal scalars, wide panels halve the number of trailing sweeps of the larger ones
__device__ inline int lds_partial_cholesky(double* L, int n, int F, bool gap_test, FrontScratch& sc) {
#ifdef GSX_OLD_LDS_CHOLESKY
  return (F <= 24) ? lds_partial_cholesky_t<8>(L, n, F, gap_test) : lds_partial_cholesky_t<16>(L, n, F, gap_test);
#else
  return lds_partial_cholesky_mfma(L, n, F, gap_test, sc);
#endif
}

__device__ inline void report_failure(DevStatus* status, int front) {
  atomicAdd(&status->n_fail, 1);
  atomicMin(&status->first_front, front);
}

// ---------------------------------------------------------------------------------------------
// front_small: one workgroup assembles and eliminates one clique entirely inside LDS.
//   1. H panels of the frontal variables (+ lambda D on their diagonal)      [HessianFactor merge ctor]
//   2. extend-add of the children's Schur complements, in child order        [updateHessian of child factors]
//   3. partial Cholesky                                                      [choleskyPartial]
//   4. L panel -> arena (kept for back-substitution); Schur complement -> arena (pulled by a small
//      parent, gathered by a big one).
// ---------------------------------------------------------------------------------------------
#ifdef GSX_STAMP
__device__ unsigned long long g_fs_stamp[12];
#define FS_BEGIN unsigned long long fs0__ = wall_clock64();
#define FS_ADD(slot)                                                                  \
  {                                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                       \
    unsigned long long t__ = wall_clock64();                                          \
    if (threadIdx.x == 0) atomicAdd(&g_fs_stamp[slot], t__ - fs0__);                   \
    fs0__ = t__;                                                                      \
  }
#else
#define FS_BEGIN
#define FS_ADD(slot)
#endif
// COH: the children's Schur complements are read, and this front's is written, with agent-scope accesses that go past
// the (per-XCD, mutually non-coherent) L2s — what front_tree_kernel needs to hand a front from one workgroup to another
// inside a launch without cache-wide write-back / invalidate fences.
template <bool COH>
__device__ __forceinline__ void front_small_body(const DevProblem& P, const DevSymbolic& S, const int f, const double* H,
                                                 const double* damp, const double lambda, double* arena,
                                                 DevStatus* status, double* L, FrontScratch& sc) {
  VarRec* vrec = sc.vrec;
  int* vpre = sc.vpre;
  ChildRec* crec = sc.crec;
  FS_BEGIN
  const FrontRec fr = S.front_recs[f];
  const int n = fr.n, F = fr.F;
  const i64 off = fr.off;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  // A front this small is a chain of dependent memory round trips (more than a microsecond each), not bytes or flops:
  // the records of its variables and children are staged in LDS by one round trip, and every loop below puts all the
  // loads of a lane in flight before the first LDS store that needs one.
  int* cml = (int*)(L + (size_t)n * n);  // staging of two children's row maps (2 x n ints)
  const int nfv = fr.nfv, nchild = fr.nchild;
  if (tid < min(nfv, kVarStage)) vrec[tid] = S.fvar_recs[fr.fvar_ptr + tid];
  if (tid < min(nchild, kChildStage)) crec[tid] = S.child_recs[fr.child_ptr + tid];
  for (int e = tid; e < n * n; e += nt) L[e] = 0;
  FS_ADD(0)
  for (int k0 = 0; k0 < nfv; k0 += kVarStage) {
    const int nb = min(kVarStage, nfv - k0);
    if (k0 > 0) {
      __syncthreads();
      if (tid < nb) vrec[tid] = S.fvar_recs[fr.fvar_ptr + k0 + tid];
    }
    __syncthreads();
    if (tid <= nb) {
      int s = 0;
      for (int k = 0; k < tid; ++k) s += vrec[k].rows * vrec[k].dA;
      vpre[tid] = s;
    }
    __syncthreads();
    // H panels of the staged variables, one flat index space over (variable, entry)
    const int total = vpre[nb];
    constexpr int U = 8;
    for (int e0 = tid; e0 < total; e0 += nt * U) {
      double x[U];
      int dst[U];
      int k = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + nt * u;
        x[u] = 0.0;
        dst[u] = -1;
        if (e < total) {
          while (e >= vpre[k + 1]) ++k;
          const VarRec vr = vrec[k];
          const int el = e - vpre[k];
          int r, j;
          divmod_small(el, vr.rows, 1.0f / (float)vr.rows, j, r);
          x[u] = H[vr.h_off + el];
          dst[u] = S.hmap[vr.hmap_off + r] + (vr.loc + j) * n;
          if (r == j) x[u] += lambda * damp[vr.toff + j];  // rows 0..dA-1 of the panel are the variable itself
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (dst[u] >= 0) L[dst[u]] = x[u];
    }
  }
  // extend-add of the children, in child order.  A child's row map is staged in LDS (no dependent global loads in the
  // entry loop) while the previous child is being added; its Schur complement is one flat index space over the lower
  // triangle, sixteen loads a lane in flight.
  FS_ADD(1)
  auto child = [&](int k) -> ChildRec { return k < kChildStage ? crec[k] : S.child_recs[fr.child_ptr + k]; };
  if (nchild > 0) {
    const ChildRec c0 = child(0);
    for (int r = tid; r < c0.s1; r += nt) cml[r] = S.cmap[c0.cmap_off + r];
  }
  __syncthreads();
  for (int ci = 0; ci < nchild; ++ci) {
    const ChildRec cr = child(ci);
    const int* cm = cml + (ci & 1) * n;
    // the next child's row map: requested now, stored after this child's entries
    int nxt[3] = {0, 0, 0};
    ChildRec cn = cr;
    if (ci + 1 < nchild) {
      cn = child(ci + 1);
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (tid + nt * u < cn.s1) nxt[u] = S.cmap[cn.cmap_off + tid + nt * u];
    }
    const double* src0 = arena + cr.src0;
    const int s1 = cr.s1, tot = s1 * (s1 + 1) / 2;
    const float b = (float)(2 * s1 + 1);
    constexpr int U = 16;
    for (int e0 = tid; e0 < tot; e0 += nt * U) {
      double v[U];
      int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + nt * u;
        v[u] = 0.0;
        dst[u] = -1;
        if (e < tot) {
          // column c of the lower triangle starts at entry c s1 - c (c - 1) / 2
          int c = (int)((b - sqrtf(b * b - 8.0f * (float)e)) * 0.5f);
          c = max(0, min(c, s1 - 1));
          if (c + 1 < s1 && (c + 1) * s1 - (c + 1) * c / 2 <= e) ++c;
          if (c * s1 - c * (c - 1) / 2 > e) --c;
          const int r = c + (e - (c * s1 - c * (c - 1) / 2));
          if constexpr (COH) v[u] = __hip_atomic_load(&src0[(i64)c * cr.nc + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else v[u] = src0[(i64)c * cr.nc + r];
          dst[u] = cm[r] + cm[c] * n;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (dst[u] >= 0) L[dst[u]] += v[u];
    }
    if (ci + 1 < nchild) {
      int* cmn = cml + ((ci + 1) & 1) * n;
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (tid + nt * u < cn.s1) cmn[tid + nt * u] = nxt[u];
    }
    __syncthreads();
  }
  FS_ADD(2)
  const int fail = lds_partial_cholesky(L, n, F, false, sc);
  FS_ADD(3)
  if (fail && tid == 0) report_failure(status, f);
  // L panel
  double* A = arena + off;
  for (int c = wave; c < F; c += nw)
    for (int r = c + lane; r < n; r += 64) A[r + (i64)c * n] = L[r + c * n];
  // Schur complement -> own arena square (a small parent pulls it, a big parent gathers it)
  const int s1 = n - F;
  for (int col = wave; col < s1; col += nw)
    for (int r = col + lane; r < s1; r += 64) {
      if constexpr (COH)
        __hip_atomic_store(&A[(F + r) + (i64)(F + col) * n], L[(F + r) + (F + col) * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else A[(F + r) + (i64)(F + col) * n] = L[(F + r) + (F + col) * n];
    }
  FS_ADD(4)
#ifdef GSX_STAMP
  if (tid == 0) { atomicAdd(&g_fs_stamp[5], (unsigned long long)n); atomicAdd(&g_fs_stamp[6], (unsigned long long)F); atomicAdd(&g_fs_stamp[7], (unsigned long long)nchild); atomicAdd(&g_fs_stamp[8], (unsigned long long)nfv); atomicAdd(&g_fs_stamp[9], 1ull); }
#endif
}
__global__ void __launch_bounds__(512) front_small_kernel(DevProblem P, DevSymbolic S, const int* ids, const double* H, const double* damp,
                                   const double* scalars, double* arena, DevStatus* status) {
  extern __shared__ double L[];
  __shared__ FrontScratch sc;
  front_small_body<false>(P, S, ids[blockIdx.x], H, damp, scalars[SC_LAMBDA], arena, status, L, sc);
}

// ---------------------------------------------------------------------------------------------
// MEDIUM fronts (gsx_internal.h): n rows do not fit LDS as a square, the n x F frontal panel does.  One workgroup:
//   1. panel [A11; A21; g'] in LDS from the H panels (+ lambda D) — H only ever touches the frontal columns;
//   2. children in child order: entries that land in a frontal column go to the LDS panel, the others are added to the
//      trailing block where it lives, in the arena (zeroed first; plain read-modify-write of this workgroup's own data);
//   3. the panel is factored in LDS: per 16 columns D (one wave, registers), X (the tiles below), U — restricted to the
//      panel's own columns;
//   4. the trailing block takes C -= L21 L21' tile by tile on the matrix cores, operands from the LDS panel, the tile read
//      from and written back to the arena once;
//   5. the panel goes out as the front's L columns.
// Storage = an LDS front's (n x n, column-major, L in the first F columns, the Schur complement behind them).
// choleskyPartial, gtsam/base/cholesky.cpp:108-159; HessianFactor merge ctor + updateHessian, HessianFactor.cpp:240-373.
// ---------------------------------------------------------------------------------------------
__device__ inline int lds_panel_cholesky_mfma(double* Pm, int n, int F, FrontScratch& sc) {
  double* Ebm = sc.Eb;
  int& failedm = sc.failed;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (tid == 0) failedm = 0;
  for (int jb = 0; jb < F; jb += 16) {
    const int w = min(16, F - jb), rb = jb + w;
    if (wave == 0) {
      v4d pt, E;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = jb + li, c = jb + 4 * q + lk;
        pt[q] = (r < F && c < F) ? (r >= c ? Pm[r + c * n] : 0.0) : (r == c ? 1.0 : 0.0);
        E[q] = (li == 4 * q + lk) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {   // (straight-line: see lds_partial_cholesky_mfma)
        const int q = j >> 2, lkj = j & 3;
        __builtin_amdgcn_sched_barrier(0);
        int lj = li;
        asm volatile("" : "+v"(lj));
        const double dj = tile_readlane(pt[q], lkj * 16 + j);
        const double sj = tile_rsqrt(dj);
        const bool colj = lk == lkj;
        const double xj = pt[q] * sj;
        const double xm = (colj && lj >= j) ? xj : 0.0;
        const double ej = colj ? E[q] * sj : 0.0;
        pt[q] = colj ? xm : pt[q];
        E[q] = colj ? ej : E[q];
        if (j < 15) {
          const double xu = (lj > j) ? xm : 0.0;
          pt = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, xu, pt, 0, 0, 0);
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(-xu, ej, E, 0, 0, 0);
        }
      }
      if (lk == (li & 3)) {
        const int qd = li >> 2;
        const double l = (qd == 0) ? pt[0] : ((qd == 1) ? pt[1] : ((qd == 2) ? pt[2] : pt[3]));
        if (jb + li < F && !(l > 0)) failedm = 1;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = jb + li, c = jb + 4 * q + lk;
        if (r < F && c < F && r >= c) Pm[r + c * n] = pt[q];
        Ebm[(4 * q + lk) * 17 + li] = E[q];
      }
    }
    __syncthreads();
    // X: every 16-row tile below the diagonal tile, X = A_ip L_pp^-T
    const int m = n - rb, T16 = (m + 15) >> 4;
    for (int t = wave; t < T16; t += nw) {
      const int r = rb + 16 * t + li;
      v4d nv = {0.0, 0.0, 0.0, 0.0};
      double pv[4], av[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        pv[s4] = (r < n && 4 * s4 + lk < w) ? Pm[r + (jb + 4 * s4 + lk) * n] : 0.0;
        av[s4] = Ebm[li * 17 + 4 * s4 + lk];
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) nv = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], pv[s4], nv, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (r < n && 4 * q + lk < w) Pm[r + (jb + 4 * q + lk) * n] = nv[q];
    }
    __syncthreads();
    // U: rank-16 update of the panel's remaining columns rb..F-1 (all rows below): tile (row tile ti, column tile tj),
    // column tiles only while they start inside the panel
    {
      const int C16 = (F - rb + 15) >> 4;   // column tiles left in the panel
      const int ntiles = C16 > 0 ? C16 * T16 - C16 * (C16 - 1) / 2 : 0;   // tj < C16, ti >= tj
      for (int t = wave; t < ntiles; t += nw) {
        int tj = 0, rem = t;
        while (rem >= T16 - tj) {
          rem -= T16 - tj;
          ++tj;
        }
        const int ti = tj + rem;
        const int i0 = rb + ti * 16, c0 = rb + tj * 16;
        const int col = c0 + li;
        v4d acc;
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = i0 + lk + 4 * q;
          acc[q] = (rr < n && col < F) ? Pm[rr + col * n] : 0.0;
          const int k = 4 * q + lk;
          av[q] = (k < w && i0 + li < n) ? -Pm[(i0 + li) + (jb + k) * n] : 0.0;
          bv[q] = (k < w && c0 + li < n) ? Pm[(c0 + li) + (jb + k) * n] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = i0 + lk + 4 * q;
          if (rr < n && col < F) Pm[rr + col * n] = acc[q];
        }
      }
    }
    __syncthreads();
  }
  return failedm;
}

template <bool COH>
__device__ __forceinline__ void front_medium_body(const DevProblem& P, const DevSymbolic& S, const int f, const double* H,
                                                  const double* damp, const double lambda, double* arena,
                                                  DevStatus* status, double* Pm, FrontScratch& sc) {
  VarRec* mvrec = sc.vrec;
  int* mvpre = sc.vpre;
  ChildRec* mcrec = sc.crec;
  FS_BEGIN
  const FrontRec fr = S.front_recs[f];
  const int n = fr.n, F = fr.F;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  const int li = lane & 15, lk = lane >> 4;
  double* A = arena + fr.off;
  int* cml = (int*)(Pm + (size_t)n * F);  // two children's row maps (2 x n ints)
  const int nfv = fr.nfv, nchild = fr.nchild;
  const int s1 = n - F;
  if (tid < min(nfv, kVarStage)) mvrec[tid] = S.fvar_recs[fr.fvar_ptr + tid];
  if (tid < min(nchild, kChildStage)) mcrec[tid] = S.child_recs[fr.child_ptr + tid];
  for (int e = tid; e < n * F; e += nt) Pm[e] = 0;
  // the trailing block's lower triangle in the arena: zero (the children add into it)
  for (int col = wave; col < s1; col += nw)
    for (int r = col + lane; r < s1; r += 64) A[(F + r) + (i64)(F + col) * n] = 0.0;
  FS_ADD(0)
  for (int k0 = 0; k0 < nfv; k0 += kVarStage) {
    const int nb = min(kVarStage, nfv - k0);
    if (k0 > 0) {
      __syncthreads();
      if (tid < nb) mvrec[tid] = S.fvar_recs[fr.fvar_ptr + k0 + tid];
    }
    __syncthreads();
    if (tid <= nb) {
      int sacc = 0;
      for (int k = 0; k < tid; ++k) sacc += mvrec[k].rows * mvrec[k].dA;
      mvpre[tid] = sacc;
    }
    __syncthreads();
    const int total = mvpre[nb];
    constexpr int U = 8;
    for (int e0 = tid; e0 < total; e0 += nt * U) {
      double x[U];
      int dst[U];
      int k = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + nt * u;
        x[u] = 0.0;
        dst[u] = -1;
        if (e < total) {
          while (e >= mvpre[k + 1]) ++k;
          const VarRec vr = mvrec[k];
          const int el = e - mvpre[k];
          int r, j;
          divmod_small(el, vr.rows, 1.0f / (float)vr.rows, j, r);
          x[u] = H[vr.h_off + el];
          dst[u] = S.hmap[vr.hmap_off + r] + (vr.loc + j) * n;   // (columns of a frontal variable: inside the panel)
          if (r == j) x[u] += lambda * damp[vr.toff + j];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (dst[u] >= 0) Pm[dst[u]] = x[u];
    }
  }
  FS_ADD(1)
  // children, in child order (a barrier — which also drains the arena stores — between one child and the next)
  auto child = [&](int k) -> ChildRec { return k < kChildStage ? mcrec[k] : S.child_recs[fr.child_ptr + k]; };
  if (nchild > 0) {
    const ChildRec c0 = child(0);
    for (int r = tid; r < c0.s1; r += nt) cml[r] = S.cmap[c0.cmap_off + r];
  }
  __syncthreads();
  for (int ci = 0; ci < nchild; ++ci) {
    const ChildRec cr = child(ci);
    const int* cm = cml + (ci & 1) * n;
    if (ci + 1 < nchild) {   // the next child's row map into the other buffer (nobody reads that one now)
      const ChildRec cn = child(ci + 1);
      int* cmn = cml + ((ci + 1) & 1) * n;
      for (int r = tid; r < cn.s1; r += nt) cmn[r] = S.cmap[cn.cmap_off + r];
    }
    const double* src0 = arena + cr.src0;
    const int cs1 = cr.s1, tot = cs1 * (cs1 + 1) / 2;
    const float b = (float)(2 * cs1 + 1);
    constexpr int U = 8;
    for (int e0 = tid; e0 < tot; e0 += nt * U) {
      double v[U];
      int dr[U], dc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + nt * u;
        v[u] = 0.0;
        dr[u] = -1;
        dc[u] = 0;
        if (e < tot) {
          int c = (int)((b - sqrtf(b * b - 8.0f * (float)e)) * 0.5f);
          c = max(0, min(c, cs1 - 1));
          if (c + 1 < cs1 && (c + 1) * cs1 - (c + 1) * c / 2 <= e) ++c;
          if (c * cs1 - c * (c - 1) / 2 > e) --c;
          const int r = c + (e - (c * cs1 - c * (c - 1) / 2));
          if constexpr (COH) v[u] = __hip_atomic_load(&src0[(i64)c * cr.nc + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else v[u] = src0[(i64)c * cr.nc + r];
          dr[u] = cm[r];
          dc[u] = cm[c];
        }
      }
      // the trailing entries: read-modify-write in the arena, all loads of a lane before its stores
      double old[U];
#pragma unroll
      for (int u = 0; u < U; ++u) old[u] = (dr[u] >= 0 && dc[u] >= F) ? A[dr[u] + (i64)dc[u] * n] : 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (dr[u] < 0) continue;
        if (dc[u] < F) Pm[dr[u] + dc[u] * n] += v[u];
        else A[dr[u] + (i64)dc[u] * n] = old[u] + v[u];
      }
    }
    __syncthreads();
  }
  FS_ADD(2)
  const int fail = lds_panel_cholesky_mfma(Pm, n, F, sc);
  FS_ADD(3)
  if (fail && tid == 0) report_failure(status, f);
  // the L columns
  for (int c = wave; c < F; c += nw)
    for (int r = c + lane; r < n; r += 64) A[r + (i64)c * n] = Pm[r + c * n];
  FS_ADD(4)
  // Schur complement: C -= L21 L21', one wave per 16 x 16 tile of the trailing block's lower triangle
  {
    const int T16 = (s1 + 15) >> 4, ntiles = T16 * (T16 + 1) / 2;
    for (int t = wave; t < ntiles; t += nw) {
      int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      while (ti * (ti + 1) / 2 > t) --ti;
      const int tj = t - ti * (ti + 1) / 2;
      const int i0 = F + ti * 16, c0 = F + tj * 16;
      const int col = c0 + li;
      v4d acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = i0 + lk + 4 * q;
        acc[q] = (rr < n && col < n && rr >= col) ? A[rr + (i64)col * n] : 0.0;
      }
      for (int k0 = 0; k0 < F; k0 += 4) {
        const int k = k0 + lk;
        const double av = (k < F && i0 + li < n) ? -Pm[(i0 + li) + k * n] : 0.0;
        const double bv = (k < F && c0 + li < n) ? Pm[(c0 + li) + k * n] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = i0 + lk + 4 * q;
        if (rr < n && col < n && rr >= col) {
          if constexpr (COH) __hip_atomic_store(&A[rr + (i64)col * n], acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else A[rr + (i64)col * n] = acc[q];
        }
      }
    }
  }
  FS_ADD(11)
#ifdef GSX_STAMP
  if (tid == 0) { atomicAdd(&g_fs_stamp[5], (unsigned long long)n); atomicAdd(&g_fs_stamp[6], (unsigned long long)F); atomicAdd(&g_fs_stamp[7], (unsigned long long)nchild); atomicAdd(&g_fs_stamp[8], (unsigned long long)nfv); atomicAdd(&g_fs_stamp[9], 1ull); }
#endif
}
// medium fronts of one launch group, a workgroup each (the level schedule and gsx_relinearize_partial)
__global__ void __launch_bounds__(512) front_medium_kernel(DevProblem P, DevSymbolic S, const int* ids, const double* H,
                                                           const double* damp, const double* scalars, double* arena,
                                                           DevStatus* status) {
  extern __shared__ double L[];
  __shared__ FrontScratch sc;
  front_medium_body<false>(P, S, ids[blockIdx.x], H, damp, scalars[SC_LAMBDA], arena, status, L, sc);
}
void launch_front_medium(const DevProblem& P, const DevSymbolic& S, const int* ids, int count, int max_panel, int max_n,
                         const double* H, const double* damp, const double* scalars, double* arena, DevStatus* status,
                         hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)front_medium_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr = true;
  }
  if (count && lds_fits((const void*)front_medium_kernel, ((size_t)max_panel + max_n) * sizeof(double), "front_medium"))
    front_medium_kernel<<<count, 512, ((size_t)max_panel + max_n) * sizeof(double), st>>>(P, S, ids, H, damp, scalars, arena,
                                                                                          status);
}

// ---------------------------------------------------------------------------------------------
// front_tree: the LDS-class fronts of one tier (Symbolic::tree_*), whole subtrees without a kernel boundary or a level
// barrier.  A workgroup claims a start front (no unfinished child in the tier), eliminates it exactly as front_small does,
// and then climbs: it tells the parent that one more child has arrived (one device-scope atomic), and when it was the
// LAST to arrive it eliminates the parent itself — whose children's Schur complements are then all in the arena, written
// by this or by other workgroups: the Schur complements travel with agent-scope stores and loads (past the per-XCD L2s),
// ordered by the workgroup's own wait for its stores before the atomic — a release / acquire fence pair instead writes
// back and invalidates a whole L2 per front, measured at 5x the level-by-level time.  Nobody ever waits: a
// workgroup that is not the last arrival claims the next start front, and leaves when there is none.  The extend-add
// stays a pull in child order, so the numbers do not depend on who arrives when (bitwise reproducible).
// Post-order elimination of gtsam/inference/ClusterTree-inst.h:219-318.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 4) front_tree_kernel(DevProblem P, DevSymbolic S, TreeArgs T, const double* H,
                                                         const double* damp, const double* scalars, double* arena,
                                                         DevStatus* status) {
  extern __shared__ double L[];
  __shared__ FrontScratch sc;
  __shared__ int s_next;
  const double lambda = scalars[SC_LAMBDA];
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(T.cursor, 1);
    __syncthreads();
    const int k = s_next;
    __syncthreads();
    if (k >= T.nstart) return;
    int f = T.start[k];
    while (f >= 0) {
      front_small_body<true>(P, S, f, H, damp, lambda, arena, status, L, sc);
#ifdef GSX_STAMP
      unsigned long long hs0 = wall_clock64();
#endif
      __syncthreads();  // (drains every store of the workgroup: s_waitcnt vmcnt(0) before the barrier)
      if (threadIdx.x == 0) {
        int nxt = -1;
        const int p = T.up[f];
        if (p >= 0 && atomicSub(&T.pending[p], 1) == 1) {
          T.pending[p] = T.npend[p];  // (every child has arrived: nobody touches the counter again in this launch)
          nxt = p;
        }
        s_next = nxt;
      }
      __syncthreads();
      f = s_next;
      __syncthreads();
#ifdef GSX_STAMP
      if (threadIdx.x == 0) atomicAdd(&g_fs_stamp[10], wall_clock64() - hs0);
#endif
    }
  }
}
#ifdef GSX_STAMP
static void fs_stamp_zero() {
  unsigned long long z[12] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(g_fs_stamp), z, sizeof(z));
}
static void fs_stamp_print(const char* what, int count, int threads, int max_n, hipStream_t st) {
  unsigned long long h[12];
  hipStreamSynchronize(st);
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fs_stamp), sizeof(h));
  const double c = h[9] ? (double)h[9] : 1.0;
  printf("[%s] count %6d thr %3d max_n %3d | fronts %llu mean n=%.0f F=%.0f ch=%.1f nfv=%.1f | mean us: head %.1f  H %.1f  children %.1f  chol %.1f  store %.1f  handoff %.1f  schur(medium) %.1f\n",
         what, count, threads, max_n, h[9], h[5] / c, h[6] / c, h[7] / c, h[8] / c, h[0] / c / 100, h[1] / c / 100, h[2] / c / 100,
         h[3] / c / 100, h[4] / c / 100, h[10] / c / 100, h[11] / c / 100);
}
#endif
// the tier of the medium fronts: they, and the LDS fronts above them in their subtrees
__global__ void __launch_bounds__(512) front_tree_med_kernel(DevProblem P, DevSymbolic S, TreeArgs T, const double* H,
                                                             const double* damp, const double* scalars, double* arena,
                                                             DevStatus* status) {
  extern __shared__ double L[];
  __shared__ FrontScratch sc;
  __shared__ int s_nextm;
  const double lambda = scalars[SC_LAMBDA];
  for (;;) {
    if (threadIdx.x == 0) s_nextm = atomicAdd(T.cursor, 1);
    __syncthreads();
    const int k = s_nextm;
    __syncthreads();
    if (k >= T.nstart) return;
    int f = T.start[k];
    while (f >= 0) {
      if (S.front_recs[f].n > kSmallMaxN) front_medium_body<true>(P, S, f, H, damp, lambda, arena, status, L, sc);
      else front_small_body<true>(P, S, f, H, damp, lambda, arena, status, L, sc);
      __syncthreads();  // (drains every store of the workgroup)
      if (threadIdx.x == 0) {
        int nxt = -1;
        const int p = T.up[f];
        if (p >= 0 && atomicSub(&T.pending[p], 1) == 1) {
          T.pending[p] = T.npend[p];
          nxt = p;
        }
        s_nextm = nxt;
      }
      __syncthreads();
      f = s_nextm;
      __syncthreads();
    }
  }
}
void launch_front_tree_med(const DevProblem& P, const DevSymbolic& S, const TreeArgs& T, size_t lds_bytes, const double* H,
                           const double* damp, const double* scalars, double* arena, DevStatus* status, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)front_tree_med_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr = true;
  }
  if (T.nstart <= 0 || !lds_fits((const void*)front_tree_med_kernel, lds_bytes, "front_tree_med")) return;
#ifdef GSX_STAMP
  fs_stamp_zero();
#endif
  front_tree_med_kernel<<<std::min(T.nstart, 256), 512, lds_bytes, st>>>(P, S, T, H, damp, scalars, arena, status);
#ifdef GSX_STAMP
  fs_stamp_print("tree-med", std::min(T.nstart, 256), 512, (int)(lds_bytes / 1024), st);
#endif
}

void launch_front_tree(const DevProblem& P, const DevSymbolic& S, const TreeArgs& T, int max_n, int threads, const double* H,
                       const double* damp, const double* scalars, double* arena, DevStatus* status, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)front_tree_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr = true;
  }
  if (T.nstart <= 0) return;
  const size_t lds = ((size_t)max_n * max_n + max_n) * sizeof(double);
  if (!lds_fits((const void*)front_tree_kernel, lds, "front_tree")) return;
  // as many workgroups as can be resident (LDS, 2048 threads a CU); the rest would only find the start list empty
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>({(size_t)(156 * 1024) / (lds + 2048), (size_t)2048 / threads, 16}));
  const int grid = std::min(T.nstart, 256 * per_cu);
#ifdef GSX_STAMP
  fs_stamp_zero();
#endif
  front_tree_kernel<<<grid, threads, lds, st>>>(P, S, T, H, damp, scalars, arena, status);
#ifdef GSX_STAMP
  fs_stamp_print("tree", grid, threads, max_n, st);
#endif
}

void launch_front_small(const DevProblem& P, const DevSymbolic& S, const int* ids, int count, int max_n, int threads,
                        const double* H, const double* damp, const double* scalars, double* arena, DevStatus* status,
                        hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)front_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr = true;
  }
#ifdef GSX_STAMP
  fs_stamp_zero();
#endif
  if (count && lds_fits((const void*)front_small_kernel, ((size_t)max_n * max_n + max_n) * sizeof(double), "front_small"))
    front_small_kernel<<<count, threads, ((size_t)max_n * max_n + max_n) * sizeof(double), st>>>(P, S, ids, H, damp, scalars,
                                                                                         arena, status);
#ifdef GSX_STAMP
  if (count) fs_stamp_print("fs", count, threads, max_n, st);
#endif
}

// ---------------------------------------------------------------------------------------------
// front_leaf: cliques WITHOUT children and with few frontal scalars (every BAL landmark, most
// pose-graph leaves).  Only the n x F panel [A; B; g'] lives in LDS: it is filled from the H panels
// (+ lambda D), factored in place ([L11; L21; d']), written out for back-substitution, and the Schur
// complement -L21 L21' is formed on the fly as an outer product straight into the arena — no n x n
// matrix in LDS, no zeroing of it, no trailing-update sweeps (the reference allocates and sweeps the full
// (F+S+1)^2 augmented Hessian per landmark: HessianFactor.cpp:240-253, cholesky.cpp:108-143).
// ---------------------------------------------------------------------------------------------
// PACK: four cliques a workgroup, a WAVE each (the launch groups whose cliques run with 64 threads: every BAL landmark) —
// an experiment (see launch_front_leaf): 78 000 one-wave workgroups in 150 us keep a tenth of the chip's wave slots
// occupied (SQ counters: 3 600 cycles a wave), but starting fewer, larger workgroups does not change that.  The
// arithmetic of a clique is the same instruction sequence either way.
template <bool PACK>
__device__ __forceinline__ void leaf_sync() {
  if (PACK) {  // one wave: LDS traffic of its lanes is ordered once the counter has drained
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
template <bool PACK>
__global__ void front_leaf_kernel(DevProblem P, DevSymbolic S, const LeafRec* recs, int count, int panel_stride,
                                  const double* H, const double* damp, const double* scalars, double* arena,
                                  DevStatus* status) {
  extern __shared__ double leaf_lds[];  // n x F, column-major, ld = n (PACK: four of them, panel_stride apart)
  const int sub = PACK ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // (wave-uniform: the record stays scalar)
  const int slot = PACK ? (int)blockIdx.x * 4 + sub : (int)blockIdx.x;
  if (slot >= count) return;   // (PACK: a whole wave; no workgroup barrier below)
  double* Pn = leaf_lds + (size_t)sub * panel_stride;
  const LeafRec rec = recs[slot];
  const int f = rec.front;
  const int n = rec.n, F = rec.F;
  const int tid = PACK ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nt = PACK ? 64 : (int)blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  const double lambda = scalars[SC_LAMBDA];
  for (int e = tid; e < n * F; e += nt) Pn[e] = 0;
  leaf_sync<PACK>();
  for (int k = 0; k < rec.nfv; ++k) {
    // the first frontal variable's panel is described by the record itself; further ones (merged leaf cliques of
    // pose graphs) by their VarRec
    int dA = rec.dA, rows = rec.rows, c0 = rec.loc, toff = rec.toff;
    const double* hp = H + rec.h_off;
    const int* hm = S.hmap + rec.hmap_off;
    if (k > 0) {
      const VarRec vr = S.var_recs[S.fvars[rec.fvar_ptr + k]];
      dA = vr.dA; rows = vr.rows; c0 = vr.loc; toff = vr.toff;
      hp = H + vr.h_off;
      hm = S.hmap + vr.hmap_off;
    }
    const float rrows = 1.0f / (float)rows;
    const int total = rows * dA;
    // (four loads a lane in flight before the first LDS store: a load -> store loop pays a memory round trip per pass)
    for (int e0 = tid; e0 < total; e0 += nt * 4) {
      double x[4];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + nt * u;
        x[u] = 0.0;
        dst[u] = -1;
        if (e < total) {
          int r, j;
          divmod_small(e, rows, rrows, j, r);
          x[u] = hp[e];
          if (r == j) x[u] += lambda * damp[toff + j];
          dst[u] = hm[r] + (c0 + j) * n;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dst[u] >= 0) Pn[dst[u]] = x[u];
    }
  }
  leaf_sync<PACK>();
  int fail = 0;
  for (int j = 0; j < F; ++j) {
    const double p = Pn[j + j * n];
    if (!(p > 0)) fail = 1;
    const double s = (p > 0) ? sqrt(p) : 1.0;
    const double inv = 1.0 / s;
    leaf_sync<PACK>();
    for (int r = j + tid; r < n; r += nt) Pn[r + j * n] = (r == j) ? s : Pn[r + j * n] * inv;
    leaf_sync<PACK>();
    for (int c = j + 1 + wave; c < F; c += nw) {
      const double lc = Pn[c + j * n];
      for (int r = c + lane; r < n; r += 64) Pn[r + c * n] -= Pn[r + j * n] * lc;
    }
    leaf_sync<PACK>();
  }
  // (choleskyPartial's conditioning test runs per reference clique after the factorization: cond_check_kernel)
  if (fail && tid == 0) report_failure(status, f);
  double* A = arena + rec.off;
  for (int c = wave; c < F; c += nw)
    for (int r = c + lane; r < n; r += 64) A[r + (i64)c * n] = Pn[r + c * n];
  // a lean leaf stops here: its big parent's gather forms the Schur complement blocks from this panel
  if (rec.lean) return;
  // Schur complement -L21 L21' as an outer product, thread-per-row: row r's F values stay in registers
  // (F <= kLeafMaxF), the threads of a row split the columns, two columns in flight per iteration.
  const int s1 = n - F;
  {
    const int G = max(1, nt / s1);
    // (a launch with fewer threads than trailing rows — the host never chooses one — still covers every row)
    for (int rr = tid; rr < s1 * G; rr += nt) {
      const int r = rr % s1, g = rr / s1;
      double rv[kLeafMaxF];
#pragma unroll
      for (int k = 0; k < kLeafMaxF; ++k) rv[k] = (k < F) ? Pn[F + r + k * n] : 0.0;
      double* out = A + (i64)F * n + F + r;  // entry (F + r, F + col) at out[col * n]
      int col = g;
      for (; col + G <= r; col += 2 * G) {
        double acc0 = 0, acc1 = 0;
#pragma unroll
        for (int k = 0; k < kLeafMaxF; ++k)
          if (k < F) {
            acc0 += rv[k] * Pn[F + col + k * n];
            acc1 += rv[k] * Pn[F + col + G + k * n];
          }
        out[(i64)col * n] = -acc0;
        out[(i64)(col + G) * n] = -acc1;
      }
      if (col <= r) {
        double acc0 = 0;
#pragma unroll
        for (int k = 0; k < kLeafMaxF; ++k)
          if (k < F) acc0 += rv[k] * Pn[F + col + k * n];
        out[(i64)col * n] = -acc0;
      }
    }
  }
}

// choleskyPartial's conditioning test (gtsam/base/cholesky.cpp:145-158) for every clique of the REFERENCE tree: a thread
// per clique reads the last two diagonal entries of L of its frontal block wherever the (possibly relaxed) front that
// holds them stored it.  Exponent gap >= 12 between them, or an exponent <= -12 of a lone pivot, is the reference's
// "underconstrained" verdict; a failure is reported against the front, like a non-positive pivot.
__global__ void __launch_bounds__(256) cond_check_kernel(int n, const i64* last, const i64* prev, const int* front,
                                                         const double* arena, DevStatus* status) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int e1, e2;
  (void)frexp(arena[last[k]], &e1);
  bool bad;
  if (prev[k] >= 0) {
    (void)frexp(arena[prev[k]], &e2);
    bad = !(e2 - e1 < 12);
  } else {
    bad = !(e1 > -12);
  }
  if (bad) report_failure(status, front[k]);
}
void launch_cond_check(int n, const i64* last, const i64* prev, const int* front, const double* arena, DevStatus* status,
                       hipStream_t st) {
  if (n > 0) cond_check_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, last, prev, front, arena, status);
}

void launch_front_leaf(const DevProblem& P, const DevSymbolic& S, const LeafRec* recs, int count, int max_panel, int threads,
                       const double* H, const double* damp, const double* scalars, double* arena, DevStatus* status,
                       hipStream_t st) {
  if (!count) return;
  // one-wave cliques four to a workgroup while four panels stay a small share of a CU's LDS
  // MEASURED AND SWITCHED OFF (GSX_LEAF_PACK=1 with GSX_LEAF_SPLIT=1024 turns it on): on BAL-1723 the main-queue launch
  // goes 148 -> 138 us, the low-priority one 178 -> 346 us, the LM iteration 1.797 -> 1.793 ms (noise) — the leaf launches
  // are not bound by how fast workgroups start (a thousand waves in flight either way; tools/r03_leaf.sh)
  static const bool pack_on = std::getenv("GSX_LEAF_PACK") != nullptr;
  if (pack_on && threads == 64 && (size_t)max_panel * sizeof(double) * 4 <= 32 * 1024)
    front_leaf_kernel<true><<<(count + 3) / 4, 256, (size_t)max_panel * sizeof(double) * 4, st>>>(
        P, S, recs, count, max_panel, H, damp, scalars, arena, status);
  else
    front_leaf_kernel<false><<<count, threads, (size_t)max_panel * sizeof(double), st>>>(P, S, recs, count, 0, H, damp,
                                                                                          scalars, arena, status);
}

// ---------------------------------------------------------------------------------------------
// big_gather: deterministic extend-add into big parents.  A task = one destination block (a pair of
// parent variables) with the list, in child order, of the matching blocks of the children's Schur
// complements (built once by the symbolic analysis).  Lists are cut into segments of <= 96 sources,
// one wave each (two block entries per lane, four loads in flight).  A single-segment task adds its
// sum straight into the pre-initialised front; a multi-segment task parks partial sums in scratch
// slots that the combine pass adds in slot order.  No atomics: bitwise reproducible.
// (HessianFactor::updateHessian of the children's remaining factors,
//  gtsam/linear/HessianFactor.cpp:349-373, turned into a gather.)
// ---------------------------------------------------------------------------------------------
// Operands of one group of kLG consecutive sources in product form (F <= 4).  The records sit one per lane (rec_*);
// the group's bases are broadcast with v_readlane so that the 16 loads per lane go out back to back.
constexpr int kLG = 4;  // sources per group
struct LeanGroup {
  double x[kLG], y[kLG];
};
// SAME: a diagonal destination block — both operands are the same rows of the panel (d2 = 0, dB = dA): one load a source
// instead of two (a third of a landmark's blocks are diagonal ones, and the kernel is bound by the texture-address path:
// profiles/r02_bal1723_gather_pmc.json)
template <bool SAME>
__device__ __forceinline__ void lean_group_load(LeanGroup& g, const double* arena, i64 rec_off, int rec_d2, int rec_ldF,
                                                int first, int cnt, int li, int lk, bool rowB, bool colA) {
#pragma unroll
  for (int i = 0; i < kLG; ++i) {
    const int src = first + i;  // wave-uniform
    const i64 off = readlane_i64(rec_off, src & 63);
    const int ldF = __builtin_amdgcn_readlane(rec_ldF, src & 63);
    const double* pb = arena + off;
    const unsigned o = (unsigned)li + (unsigned)lk * (unsigned)(ldF & 0xFFFFFF);
    const bool in = i < cnt && lk < (ldF >> 24);
    g.x[i] = (rowB && in) ? pb[o] : 0.0;
    if (SAME) {
      g.y[i] = g.x[i];
    } else {
      const int d2 = __builtin_amdgcn_readlane(rec_d2, src & 63);
      g.y[i] = (colA && in) ? pb[(i64)d2 + o] : 0.0;
    }
  }
}
__device__ __forceinline__ void lean_group_mfma(const LeanGroup& g, v4d& acc) {
#pragma unroll
  for (int i = 0; i < kLG; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-g.x[i], g.y[i], acc, 0, 0, 0);
}

__global__ void __launch_bounds__(256) big_gather_seg_kernel(GatherArgs G, int seg0, int nseg, double* arena) {
  const int lane = threadIdx.x & 63;
  // Workgroups go round-robin over the 8 XCDs, each with its own L2.  The segments are sorted by destination front and
  // a child's panel is read by every destination block it touches — all in its parent: XCD x takes the x-th contiguous
  // eighth of the segments, so that a panel is fetched into ONE L2 instead of all eight (measured before: 5.6 x the
  // algorithmic bytes fetched from HBM).
  const int per_xcd = gridDim.x >> 3;
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int sw = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
  if (sw >= nseg) return;
  const GatherSeg sg = G.segs[seg0 + sw];  // one 32-byte record through the scalar path
  const int dB = sg.dims & 255, dA = (sg.dims >> 8) & 255, diag = sg.dims >> 16;
  const int ld = sg.ld, slot = sg.slot, n = sg.n;
  double* dst = arena + sg.dst;
  if (dB <= 16 && dA <= 16) {
    // The destination block is one 16 x 16 FP64 matrix-core tile (entry (lk + 4q, li) in acc[q]).  Stored
    // sources are added entry by entry; a lean leaf's source is the rank-F product -W_b W_a' of two row blocks of
    // its L panel, one v_mfma_f64_16x16x4 per four panel columns — the Schur complement of a landmark is never
    // written to or read from HBM.  Sources are taken in list (child) order: deterministic.
    const int li = lane & 15, lk = lane >> 4;
    const bool rowB = li < dB, colA = li < dA;
    bool act[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) act[q] = (lk + 4 * q < dB) && colA && !(diag && lk + 4 * q < li);
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    // the destination entries are read now, while the sources are on their way
    double dv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dv[q] = (slot < 0 && act[q]) ? dst[(lk + 4 * q) + (i64)li * ld] : 0.0;
    // all (<= 64) source records of the segment with ONE coalesced load, lane l <- record l
    GatherSrc rec{0, 0, 0};
    if (lane < n) rec = G.srcs[sg.src + lane];
    const bool fast_lane = (rec.ldF >> 24) > 0 && (rec.ldF >> 24) <= 4;
    const unsigned long long fast = __ballot(fast_lane);
    const unsigned long long valid = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    int s = 0;
    // groups of kLG product-form sources (BAL landmarks): 2 kLG loads per lane back to back, then kLG matrix-core
    // instructions.  (Most destination blocks have only a handful of sources: the kernel is bound by the chain of
    // dependent memory round trips per wave times the number of waves in flight, so registers are kept low for
    // occupancy rather than spent on deeper per-wave pipelining.)
    while (s < n) {
      const unsigned long long want = (valid >> s) & ((1ull << kLG) - 1ull);
      if (((fast >> s) & want) == want) {
        LeanGroup ga;
        if (diag && dB == dA) lean_group_load<true>(ga, arena, rec.off, rec.d2, rec.ldF, s, n - s, li, lk, rowB, colA);
        else lean_group_load<false>(ga, arena, rec.off, rec.d2, rec.ldF, s, n - s, li, lk, rowB, colA);
        lean_group_mfma(ga, acc);
        s += kLG;
        continue;
      }
      // generic source (stored Schur-complement block, or a product with F > 4)
      const i64 o = readlane_i64(rec.off, s);
      const int d2 = __builtin_amdgcn_readlane(rec.d2, s);
      const int ldF = __builtin_amdgcn_readlane(rec.ldF, s);
      const i64 l = ldF & 0xFFFFFF;
      const int Fc = ldF >> 24;
      if (Fc == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (act[q]) acc[q] += arena[o + (lk + 4 * q) + li * l];
      } else {
        for (int kk = 0; kk < Fc; kk += 4) {
          const int k = kk + lk;
          const double x = (k < Fc && rowB) ? arena[o + li + k * l] : 0.0;
          const double y = (k < Fc && colA) ? arena[o + d2 + li + k * l] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-x, y, acc, 0, 0, 0);
        }
      }
      ++s;
    }
    if (slot < 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (act[q]) dst[(lk + 4 * q) + (i64)li * ld] = dv[q] + acc[q];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (act[q]) G.scratch[(i64)slot * 256 + q * 64 + lane] = acc[q];
    }
    return;
  }
  // blocks wider than a matrix-core tile (variables of more than 16 dimensions): stored sources only, never split
  const int ne = dB * dA;
  for (int eb = 0; eb < ne; eb += 128) {
    const int ea = eb + lane, ebb = eb + 64 + lane;
    const int ia = ea % dB, ja = ea / dB, ib = ebb % dB, jb = ebb / dB;
    const bool act_a = (ea < ne) && !(diag && ia < ja), act_b = (ebb < ne) && !(diag && ib < jb);
    double acc_a = 0, acc_b = 0;
    for (int s = 0; s < n; ++s) {
      const GatherSrc r = G.srcs[sg.src + s];
      const i64 o0 = r.off, l0 = r.ldF & 0xFFFFFF;
      if (act_a) acc_a += arena[o0 + ia + ja * l0];
      if (act_b) acc_b += arena[o0 + ib + jb * l0];
    }
    if (act_a) dst[ia + (i64)ja * ld] += acc_a;
    if (act_b) dst[ib + (i64)jb * ld] += acc_b;
  }
}

__global__ void __launch_bounds__(256) big_gather_combine_kernel(GatherArgs G, int m0, int nm, double* arena) {
  const int lane = threadIdx.x & 63;
  const int mw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (mw >= nm) return;
  const int t = G.gm_task[m0 + mw], slot0 = G.gm_slot[m0 + mw], ns = G.gm_nslots[m0 + mw];
  const int dims = G.gt_dims[t], dB = dims & 255, dA = (dims >> 8) & 255, diag = dims >> 16;
  const int ld = G.gt_ld[t];
  double* dst = arena + G.gt_dst[t];
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = lk + 4 * q;
    if (r >= dB || li >= dA || (diag && r < li)) continue;
    double acc = 0;
    for (int k = 0; k < ns; ++k) acc += G.scratch[(i64)(slot0 + k) * 256 + q * 64 + lane];
    dst[r + (i64)li * ld] += acc;
  }
}

void launch_big_gather(const GatherArgs& G, int seg0, int nseg, int m0, int nm, double* arena, hipStream_t st) {
  if (nseg > 0) big_gather_seg_kernel<<<((nseg + 31) / 32) * 8, 256, 0, st>>>(G, seg0, nseg, arena);  // a multiple of 8 workgroups
  if (nm > 0) big_gather_combine_kernel<<<(nm + 3) / 4, 256, 0, st>>>(G, m0, nm, arena);
}

// ---------------------------------------------------------------------------------------------
// big fronts (n > kSmallMaxN): blocked right-looking partial Cholesky in HBM, tile T = 32
// ---------------------------------------------------------------------------------------------
// Only what the factorization reads before writing is cleared: of column c the rows from the top of its diagonal 32-tile
// down (the lower triangle in tile granularity — the tiles above the diagonal are never touched by any kernel).
__global__ void big_zero_kernel(const BigDesc* descs, double* arena) {
  const BigDesc d = descs[blockIdx.y];
  double* A = arena + d.off;
  for (int c = blockIdx.x; c < d.N; c += gridDim.x) {
    double* col = A + (i64)c * d.N;
    for (int r = (c & ~(T - 1)) + threadIdx.x; r < d.N; r += blockDim.x) col[r] = 0;
  }
}
__global__ void big_add_h_kernel(DevProblem P, DevSymbolic S, const BigDesc* descs, const double* H, const double* damp,
                                 const double* scalars, double* arena) {
  const BigDesc d = descs[blockIdx.y];
  const int f = d.front;
  if ((int)blockIdx.x >= S.fr_nfv[f]) return;
  const int v = S.fvars[S.fr_fvar_ptr[f] + blockIdx.x];
  const int dA = P.var_dim[v], rows = S.h_rows[v], c0 = S.h_loc[v], n = d.N;
  const double* hp = H + S.h_off[v];
  const int* hm = S.hmap + S.hmap_ptr[v];
  const int toff = P.var_tan_off[v];
  const double lambda = scalars[SC_LAMBDA];
  double* A = arena + d.off;
  const float rrows = 1.0f / (float)rows;
  for (int e = threadIdx.x; e < rows * dA; e += blockDim.x) {
    int r, j;
    divmod_small(e, rows, rrows, j, r);
    double x = hp[e];
    if (r == j) x += lambda * damp[toff + j];
    A[hm[r] + (i64)(c0 + j) * n] = x;
  }
}
void launch_big_init(const DevProblem& P, const DevSymbolic& S, const BigDesc* descs, int count, int max_n, int max_nfv,
                     const double* H, const double* damp, const double* scalars, double* arena, hipStream_t st) {
  if (!count) return;
  // (columns per workgroup so that the whole launch is a few thousand workgroups)
  int bx = std::max(1, std::min(max_n, 8192 / std::max(count, 1)));
  big_zero_kernel<<<dim3(bx, count), 256, 0, st>>>(descs, arena);
  big_add_h_kernel<<<dim3(max_nfv, count), 128, 0, st>>>(P, S, descs, H, damp, scalars, arena);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access
// (s_waitcnt vmcnt(0)) — which would put the latency of in-flight prefetches and of
// just-issued stores on the critical path of the latency-bound kernels below.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (The blocked fronts' factorization kernels live in bigfront.hip.)
// gsx_cholesky_partial's large case: move the L panel back under the diagonal tiles of the matrix
__global__ void big_copy_panel_kernel(const BigDesc* descs, double* arena) {
  const BigDesc d = descs[0];
  double* A = arena + d.off;
  const double* X = arena + d.xoff;
  const i64 total = (i64)d.N * d.F;
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x) {
    const int r = (int)(e % d.N), c = (int)(e / d.N);
    if (r >= min((c / T + 1) * T, d.F)) A[r + (i64)c * d.N] = X[r + (i64)c * d.N];
  }
}

// ---------------------------------------------------------------------------------------------
// back-substitution of one clique per workgroup (OptimizeClique, gtsam/linear/linearAlgorithms-inst.h:49-117):
//   L11' x_F = d - L21' x_S, blocked by 32 columns from the last panel to the first.
// ---------------------------------------------------------------------------------------------
constexpr int TB = 32;  // panel width of the back-substitution (independent of the factorization tile)
__global__ void backsolve_kernel(DevSymbolic S, const int* ids, const double* arena, double* delta, DevStatus* status) {
  extern __shared__ double xs[];  // n-1 solution entries of this front (frontal + separator)
  __shared__ double tile[TB][TB + 1];
  __shared__ double y[TB];
  __shared__ double dv[TB];
  const int f = ids[blockIdx.x];
  if (wildfire_skip(S, f, threadIdx.x == 0)) return;  // a clique no change reaches
  const int n = S.fr_N[f], F = S.fr_F[f];
  const double* A = arena + S.fr_off[f];
  const bool big = (S.fr_lean[f] & 2) != 0;  // blocked layout (n > kSmallMaxN, and every cap front of a sharded problem)
  // big fronts keep the rows of L below each diagonal tile in their L-panel area right after the n x n front
  const double* Lp = big ? A + big_panel_offset(n) : A;
  const int* gi = S.gidx + S.gidx_ptr[f];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  for (int r = F + tid; r < n - 1; r += nt) xs[r] = delta[gi[r]];
  __syncthreads();
  // big fronts carry (L^-1)' of every 32x32 diagonal tile in that tile's strictly upper triangle and 1 / L_cc in
  // the L-panel area (big_diag, bigfront.hip; kTile == TB): the tile solve is then a 32-term dot product per lane instead
  // of a 32-step serial substitution with divisions.
  const bool has_inv = big && (T == TB);
  const int nblk = (F + TB - 1) / TB;
  constexpr int kTR = 16;  // tile entries per thread at the smallest block size (64 threads)
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int c0 = kb * TB, w = min(TB, F - c0);
    // the diagonal tile (and the reciprocal diagonal) travels in registers while the dot products below stream the
    // panel: nothing but LDS sits between the two barriers of a panel
    double treg[kTR], dreg = 0.0;
    const float rw = __builtin_amdgcn_rcpf((float)w);
#pragma unroll
    for (int q = 0; q < kTR; ++q) {
      const int e = tid + q * nt;
      treg[q] = 0.0;
      if (e < w * w) {
        int r, c;
        divmod_small(e, w, rw, c, r);
        if (has_inv || r >= c) treg[q] = A[(c0 + r) + (i64)(c0 + c) * n];
      }
    }
    if (has_inv && tid < w) dreg = Lp[(c0 + tid) + (i64)(c0 + tid) * n];
    for (int c = 2 * wave; c < w; c += 2 * nw) {  // two columns per wave: independent loads and reductions
      const double* col0 = Lp + (i64)(c0 + c) * n;
      const bool two = c + 1 < w;
      const double* col1 = two ? col0 + n : col0;
      double acc0 = 0, acc1 = 0;
      for (int r = c0 + w + lane; r < n - 1; r += 64) {
        const double x = xs[r];
        acc0 += col0[r] * x;
        acc1 += col1[r] * x;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        acc0 += __shfl_down(acc0, o, 64);
        acc1 += __shfl_down(acc1, o, 64);
      }
      if (lane == 0) {
        y[c] = col0[n - 1] - acc0;
        if (two) y[c + 1] = col1[n - 1] - acc1;
      }
    }
#pragma unroll
    for (int q = 0; q < kTR; ++q) {
      const int e = tid + q * nt;
      if (e < w * w) {
        int r, c;
        divmod_small(e, w, rw, c, r);
        tile[r][c] = treg[q];
      }
    }
    if (tid < w) dv[tid] = dreg;
    __syncthreads();
    if (wave == 0) {
      double yr;
      if (has_inv) {
        // x = (L^-1)' y:  x[c] = y[c] / L[c][c] + sum_{r > c} (L^-1)[r][c] y[r],  (L^-1)[r][c] = tile[c][r]
        yr = 0;
        if (lane < w) {
          yr = y[lane] * dv[lane];
          for (int r = lane + 1; r < w; ++r) yr += tile[lane][r] * y[r];
        }
      } else {
        // solve tile' x = y backwards; lane r holds y_r
        yr = (lane < w) ? y[lane] : 0.0;
        for (int c = w - 1; c >= 0; --c) {
          const double xc = __shfl(yr, c, 64) / tile[c][c];
          if (lane == c) yr = xc;
          else if (lane < c) yr -= tile[c][lane] * xc;
        }
      }
      if (lane < w) xs[c0 + lane] = yr;
    }
    lds_barrier();
  }
  // the frontal part of the solution goes out once, coalesced, off the chain of panels
  int bad = 0;
  for (int r = tid; r < F; r += nt) {
    const double x = xs[r];
    delta[gi[r]] = x;
    if (!isfinite(x)) bad = 1;
  }
  if (bad) atomicAdd(&status->n_nonfinite, 1);
}
// (A thread-per-column, right-looking variant for big fronts — separator part first, panel rows prefetched into
//  registers two panels ahead — was built and measured: 36-65 us per level against 25-39 us for the kernel above.)
// Leaf cliques (F <= kLeafMaxF: BAL landmarks, pose-graph leaves): one WAVE per clique, four per workgroup, no
// barriers.  A wave is a chain of dependent memory round trips and nothing else, so everything that needs only the
// clique's record — the row indices, the rows of L21 for the first 128 separator rows, the rhs row, the F x F
// triangle — is requested together; the parents' solution is the second round trip, the store the third.
// FM = the launch's largest F rounded up (register arrays are sized by it).
template <int FM>
__global__ void __launch_bounds__(256) backsolve_leaf_kernel(DevSymbolic S, const LeafRec* recs, int count,
                                                             const double* arena, double* delta, DevStatus* status) {
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (k >= count) return;
  const LeafRec rec = recs[k];
  if (wildfire_skip(S, rec.front, lane == 0)) return;  // a clique no change reaches
  const int n = rec.n, F = rec.F;
  const double* A = arena + rec.off;
  const int* gi = S.gidx + rec.gidx_ptr;
  constexpr int U = 2;
  int g[U];
  double a[U][FM];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int r = F + lane + 64 * u;
    const bool ok = r < n - 1;
    g[u] = ok ? gi[r] : -1;
#pragma unroll
    for (int c = 0; c < FM; ++c) a[u][c] = (ok && c < F) ? A[r + (i64)c * n] : 0.0;
  }
  const int gf = lane < F ? gi[lane] : -1;
  // lane e holds entry (e % F, e / F) of the triangle and lane c the rhs of column c (F * F <= 64 when F <= 8; wider
  // leaves read the triangle from memory in the substitution)
  const int tr = lane % max(F, 1), tc = lane / max(F, 1);
  const double tri = (FM <= 8 && tc < F) ? A[tr + (i64)tc * n] : 0.0;
  const double dl = lane < F ? A[(n - 1) + (i64)lane * n] : 0.0;
  // y_c = d_c - sum_{r >= F} L[r][c] x_r
  double yc[FM];
#pragma unroll
  for (int c = 0; c < FM; ++c) yc[c] = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double x = g[u] >= 0 ? delta[g[u]] : 0.0;
#pragma unroll
    for (int c = 0; c < FM; ++c) yc[c] = fma(a[u][c], x, yc[c]);
  }
  for (int r = F + lane + 64 * U; r < n - 1; r += 64) {  // (separators of more than 128 rows: rare)
    const double x = delta[gi[r]];
#pragma unroll
    for (int c = 0; c < FM; ++c)
      if (c < F) yc[c] += A[r + (i64)c * n] * x;
  }
#pragma unroll
  for (int c = 0; c < FM; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) yc[c] += __shfl_xor(yc[c], o, 64);
    yc[c] = tile_readlane(dl, c) - yc[c];
  }
  // L11' x = y, backwards (every lane computes the same values)
  double xv[FM];
#pragma unroll
  for (int c = FM - 1; c >= 0; --c) {
    xv[c] = 0.0;
    if (c < F) {
      double acc = yc[c];
#pragma unroll
      for (int r = c + 1; r < FM; ++r)
        if (r < F) acc -= (FM <= 8 ? tile_readlane(tri, r + c * F) : A[r + (i64)c * n]) * xv[r];
      xv[c] = acc / (FM <= 8 ? tile_readlane(tri, c + c * F) : A[c + (i64)c * n]);
    }
  }
  double mine = 0.0;
#pragma unroll
  for (int c = 0; c < FM; ++c) mine = (lane == c) ? xv[c] : mine;
  if (lane < F) {
    delta[gf] = mine;
    if (!isfinite(mine)) atomicAdd(&status->n_nonfinite, 1);
  }
}

void launch_backsolve_leaf(const DevSymbolic& S, const LeafRec* recs, int count, int max_F, const double* arena,
                           double* delta, DevStatus* status, hipStream_t st) {
  if (!count) return;
  const int grid = (count + 3) / 4;
  if (max_F <= 4) backsolve_leaf_kernel<4><<<grid, 256, 0, st>>>(S, recs, count, arena, delta, status);
  else if (max_F <= 8) backsolve_leaf_kernel<8><<<grid, 256, 0, st>>>(S, recs, count, arena, delta, status);
  else backsolve_leaf_kernel<kLeafMaxF><<<grid, 256, 0, st>>>(S, recs, count, arena, delta, status);
}

// ---------------------------------------------------------------------------------------------
// ISAM2's partial ("wildfire") back-substitution: the pass after a level's kernels (kernels.h: WildfireArgs;
// gtsam/nonlinear/ISAM2Clique.cpp valuesChanged :175-183, restoreFromOriginals :193-201, optimizeWildfireNode :237-259).
// The dirty rule itself sits at the top of the back-substitution kernels (wildfire_skip).
// ---------------------------------------------------------------------------------------------
// WAVE: a wave per clique (a thread walking hundreds of frontal scalars behind dependent loads takes as long as the
// level's back-substitution); !WAVE: a thread per clique, for the leaf cliques (F <= 16, up to 10^5 a level).  The count
// goes through LDS: one global atomic a workgroup.
template <bool WAVE>
__global__ void __launch_bounds__(256) wildfire_post_kernel(DevSymbolic S, const int* ids, int count, WildfireArgs W,
                                                            double* delta) {
  __shared__ int solved;
  if (threadIdx.x == 0) solved = 0;
  __syncthreads();
  const int lane = WAVE ? (threadIdx.x & 63) : 0, step = WAVE ? 64 : 1;
  const int k = WAVE ? blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x * 256 + threadIdx.x;
  if (k < count && W.dirty[ids[k]]) {
    const int f = ids[k];
    const int F = S.fr_F[f], nfv = S.fr_nfv[f];
    const int* gi = S.gidx + S.gidx_ptr[f];
    if (lane == 0) atomicAdd(&solved, nfv);
    bool keep = W.replaced[f] != 0;
    if (!keep) {
      double m = 0.0;
      for (int r = lane; r < F; r += step) {
        const int g = gi[r];
        m = fmax(m, fabs(W.old_delta[g] - delta[g]));
      }
      keep = WAVE ? (bool)__any(m >= W.threshold) : (m >= W.threshold);
    }
    if (keep) {
      for (int q = lane; q < nfv; q += step) W.changed[S.fvars[S.fr_fvar_ptr[f] + q]] = 1;
    } else {
      for (int r = lane; r < F; r += step) {
        const int g = gi[r];
        delta[g] = W.old_delta[g];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && solved) atomicAdd(&W.status->n_backsub, solved);
}
void launch_wildfire_post(const DevSymbolic& S, const int* ids, int count, bool wave_per_clique, const WildfireArgs& W,
                          double* delta, hipStream_t st) {
  if (count <= 0) return;
  if (wave_per_clique) wildfire_post_kernel<true><<<(count + 3) / 4, 256, 0, st>>>(S, ids, count, W, delta);
  else wildfire_post_kernel<false><<<(count + 255) / 256, 256, 0, st>>>(S, ids, count, W, delta);
}
__global__ void __launch_bounds__(256) mark_fronts_kernel(const int* ids, int count, unsigned char* flags) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < count) flags[ids[k]] = 1;
}
void launch_mark_fronts(const int* ids, int count, unsigned char* flags, hipStream_t st) {
  if (count > 0) mark_fronts_kernel<<<(count + 255) / 256, 256, 0, st>>>(ids, count, flags);
}

void launch_backsolve(const DevSymbolic& S, const int* ids, int count, int threads, int max_n, const double* arena,
                      double* delta, DevStatus* status, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)backsolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr = true;
  }
  if (count)
    backsolve_kernel<<<count, threads, (size_t)max_n * sizeof(double), st>>>(S, ids, arena, delta, status);
}

// ---------------------------------------------------------------------------------------------
// dense unit entry (gsx_cholesky_partial): small matrices through the LDS kernel, large ones
// through the blocked path.
// ---------------------------------------------------------------------------------------------
__global__ void dense_small_kernel(double* a, int n, int nf, DevStatus* status) {
  extern __shared__ double L[];
  __shared__ FrontScratch sc;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) L[e] = a[e];
  __syncthreads();
  const int fail = lds_partial_cholesky(L, n, nf, true, sc);   // the dense entry: one clique, tested here
  if (fail && threadIdx.x == 0) report_failure(status, 0);
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int r = e % n, c = e / n;
    if (r >= c) a[e] = L[e];
  }
}
void launch_dense_partial(double* a, int n, int nf, DevStatus* status, hipStream_t st) {
  if (nf == 0) return;
  if (n <= kSmallMaxN) {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)dense_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
      attr = true;
    }
    dense_small_kernel<<<1, 512, (size_t)n * n * sizeof(double), st>>>(a, n, nf, status);
#ifdef GSX_STAMP
    {
      unsigned long long h[8];
      hipStreamSynchronize(st);
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
      printf("[stamp] n=%d F=%d panel-factor %llu cycles, trailing %llu cycles\n", n, nf, h[0], h[1]);
      unsigned long long z[8] = {0};
      hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z));
    }
#endif
    return;
  }
  // the L panel is produced next to the matrix and copied back under the diagonal tiles at the end
  double* work = nullptr;
  const i64 xoff = big_panel_offset(n);
  hipMalloc(&work, (size_t)(xoff + (i64)n * nf) * sizeof(double));
  hipMemcpyAsync(work, a, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, st);
  double* const user = a;
  a = work;
  BigDesc h{0, xoff, n, nf, 0, kStandaloneFront};
  BigDesc* d = nullptr;
  hipMalloc(&d, sizeof(BigDesc));
  hipMemcpyAsync(d, &h, sizeof(BigDesc), hipMemcpyHostToDevice, st);
  BigPlan plan;
  plan_big_group(&h, 1, plan);
  for (int r = 0; r < plan.rounds(); ++r) {
    launch_big_diag(d, 1, plan, r, a, status, st);
    launch_big_rows(d, 1, plan, r, a, st);
    launch_big_schur(d, 1, plan, r, a, st);
  }
  big_copy_panel_kernel<<<64, 256, 0, st>>>(d, a);
  hipMemcpyAsync(user, work, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, st);
  hipStreamSynchronize(st);
#ifdef GSX_STAMP
  {
    char what[64];
    snprintf(what, sizeof(what), "n=%d F=%d", n, nf);
    big_stamp_dump(what);
  }
#endif
  hipFree(d);
  hipFree(work);
}

}  // namespace gsx
