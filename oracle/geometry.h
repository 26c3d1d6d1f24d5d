// oracle/geometry.h — TEST INFRASTRUCTURE ONLY (see oracle/oracle.cpp header).
//
// CPU restatement of the Lie-group / camera math on the hot path.  Every
// function cites the reference file:line it follows (paths relative to
// /root/reference/).  Compile-time switches assumed (cmake/HandleGeneralOptions.cmake:32-50):
// GTSAM_POSE3_EXPMAP=ON, GTSAM_ROT3_EXPMAP=ON, GTSAM_USE_QUATERNIONS=OFF,
// GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR=OFF, GTSAM_SLOW_BUT_CORRECT_EXPMAP=OFF,
// GTSAM_THROW_CHEIRALITY_EXCEPTION=ON.
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

// 3x3 matrices are row-major double[9]; vectors double[3].
inline void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  std::memcpy(C, T, sizeof(T));
}
inline void mat3_tmul_vec(const double* R, const double* v, double* out) {  // R' v
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
  out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}
inline void mat3_mul_vec(const double* R, const double* v, double* out) {  // R v
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
  out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}
inline void mat3_transpose(const double* R, double* Rt) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = R[3 * j + i];
  std::memcpy(Rt, T, sizeof(T));
}
inline void skew(const double* w, double* W) {  // gtsam/base/Matrix.h skewSymmetric
  W[0] = 0; W[1] = -w[2]; W[2] = w[1];
  W[3] = w[2]; W[4] = 0; W[5] = -w[0];
  W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}
inline void cross3(const double* a, const double* b, double* c) {
  double t[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  c[0] = t[0]; c[1] = t[1]; c[2] = t[2];
}

// so3::ExpmapFunctor / DexpFunctor coefficients — gtsam/geometry/SO3.cpp:61-112.
struct ExpCoef {
  double A, B, C;
  bool nearZero;
};
inline ExpCoef exp_coef(const double* w, bool nearZeroApprox) {
  ExpCoef e;
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double theta = std::sqrt(theta2);
  e.nearZero = nearZeroApprox || (theta2 <= 2.220446049250313e-16);  // SO3.cpp:62-63
  if (!e.nearZero) {
    const double sin_theta = std::sin(theta);
    e.A = sin_theta / theta;
    const double s2 = std::sin(theta / 2.0);
    const double one_minus_cos = 2.0 * s2 * s2;  // SO3.cpp:67-69
    e.B = one_minus_cos / theta2;
    e.C = (1 - e.A) / theta2;  // SO3.cpp:101
  } else {
    e.A = 1.0 - theta2 * (1.0 / 6.0);   // SO3.cpp:73
    e.B = 0.5 - theta2 * (1.0 / 24.0);  // SO3.cpp:74
    e.C = (1.0 / 6.0) - theta2 * (1.0 / 120.0);  // SO3.cpp:108
  }
  return e;
}
// R = I + A W + B W W — SO3.cpp:96.
inline void so3_expmap_coef(const double* w, const ExpCoef& e, double* R) {
  double W[9], WW[9];
  skew(w, W);
  mat3_mul(W, W, WW);
  for (int i = 0; i < 9; ++i) R[i] = e.A * W[i] + e.B * WW[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}
// Rot3::Rodrigues / Rot3::Expmap -> SO3::Expmap (exact-zero threshold only).
inline void so3_expmap(const double* w, double* R) { so3_expmap_coef(w, exp_coef(w, false), R); }

// SO3::Logmap — gtsam/geometry/SO3.cpp:299-375.
inline void so3_logmap(const double* R, double* omega) {
  const double R11 = R[0], R12 = R[1], R13 = R[2];
  const double R21 = R[3], R22 = R[4], R23 = R[5];
  const double R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (tr + 1.0 < 1e-3) {
    if (R33 > R22 && R33 > R11) {
      const double W = R21 - R12, Q1 = 2.0 + 2.0 * R33, Q2 = R31 + R13, Q3 = R23 + R32;
      const double r = std::sqrt(Q1), one_over_r = 1 / r;
      const double norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
      const double sgn_w = W < 0 ? -1.0 : 1.0;
      const double mag = M_PI - (2 * sgn_w * W) / norm;
      const double scale = 0.5 * one_over_r * mag;
      omega[0] = sgn_w * scale * Q2; omega[1] = sgn_w * scale * Q3; omega[2] = sgn_w * scale * Q1;
    } else if (R22 > R11) {
      const double W = R13 - R31, Q1 = 2.0 + 2.0 * R22, Q2 = R23 + R32, Q3 = R12 + R21;
      const double r = std::sqrt(Q1), one_over_r = 1 / r;
      const double norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
      const double sgn_w = W < 0 ? -1.0 : 1.0;
      const double mag = M_PI - (2 * sgn_w * W) / norm;
      const double scale = 0.5 * one_over_r * mag;
      omega[0] = sgn_w * scale * Q3; omega[1] = sgn_w * scale * Q1; omega[2] = sgn_w * scale * Q2;
    } else {
      const double W = R32 - R23, Q1 = 2.0 + 2.0 * R11, Q2 = R12 + R21, Q3 = R31 + R13;
      const double r = std::sqrt(Q1), one_over_r = 1 / r;
      const double norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
      const double sgn_w = W < 0 ? -1.0 : 1.0;
      const double mag = M_PI - (2 * sgn_w * W) / norm;
      const double scale = 0.5 * one_over_r * mag;
      omega[0] = sgn_w * scale * Q1; omega[1] = sgn_w * scale * Q2; omega[2] = sgn_w * scale * Q3;
    }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-6) {
      const double theta = std::acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * std::sin(theta));
    } else {
      magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    }
    omega[0] = magnitude * (R32 - R23);
    omega[1] = magnitude * (R13 - R31);
    omega[2] = magnitude * (R21 - R12);
  }
}

// ---- Pose3: state = R[9] row-major, t[3] ---------------------------------
struct Pose3 {
  double R[9];
  double t[3];
};
inline Pose3 pose3_from(const double* s) {
  Pose3 p;
  std::memcpy(p.R, s, 9 * sizeof(double));
  std::memcpy(p.t, s + 9, 3 * sizeof(double));
  return p;
}
inline void pose3_to(const Pose3& p, double* s) {
  std::memcpy(s, p.R, 9 * sizeof(double));
  std::memcpy(s + 9, p.t, 3 * sizeof(double));
}
// Pose3::operator* — gtsam/geometry/Pose3.h: (R1 R2, t1 + R1 t2)
inline Pose3 pose3_compose(const Pose3& a, const Pose3& b) {
  Pose3 c;
  mat3_mul(a.R, b.R, c.R);
  double rt[3];
  mat3_mul_vec(a.R, b.t, rt);
  for (int i = 0; i < 3; ++i) c.t[i] = a.t[i] + rt[i];
  return c;
}
// Pose3::inverse — gtsam/geometry/Pose3.cpp:61-65: (R', R' * (-t))
inline Pose3 pose3_inverse(const Pose3& a) {
  Pose3 c;
  mat3_transpose(a.R, c.R);
  double nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
  mat3_mul_vec(c.R, nt, c.t);
  return c;
}
// Pose3::AdjointMap — gtsam/geometry/Pose3.cpp:69-75: [[R,0],[[t]x R, R]]; 6x6 row-major.
inline void pose3_adjoint(const Pose3& p, double* Ad) {
  double T[9], TR[9];
  skew(p.t, T);
  mat3_mul(T, p.R, TR);
  for (int i = 0; i < 36; ++i) Ad[i] = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Ad[6 * i + j] = p.R[3 * i + j];
      Ad[6 * (i + 3) + j] = TR[3 * i + j];
      Ad[6 * (i + 3) + j + 3] = p.R[3 * i + j];
    }
}
// Pose3::Expmap — gtsam/geometry/Pose3.cpp:184-222 (nearZero iff w.w <= 1e-5);
// t = applyLeftJacobian(v) = v + B (w x v) + C (w x (w x v)) — SO3.cpp:163-174.
inline Pose3 pose3_expmap(const double* xi) {
  const double* w = xi;
  const double* v = xi + 3;
  const bool nearZero = (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) <= 1e-5;
  const ExpCoef e = exp_coef(w, nearZero);
  Pose3 p;
  so3_expmap_coef(w, e, p.R);
  double wv[3], wwv[3];
  cross3(w, v, wv);
  cross3(w, wv, wwv);
  for (int i = 0; i < 3; ++i) p.t[i] = v[i] + e.B * wv[i] + e.C * wwv[i];
  return p;
}
// Pose3::Logmap — gtsam/geometry/Pose3.cpp:225-245.
inline void pose3_logmap(const Pose3& p, double* xi) {
  double w[3];
  so3_logmap(p.R, w);
  const double* T = p.t;
  const double t = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (t < 1e-10) {
    xi[3] = T[0]; xi[4] = T[1]; xi[5] = T[2];
  } else {
    double wn[3] = {w[0] / t, w[1] / t, w[2] / t}, W[9], WT[3], WWT[3];
    skew(wn, W);
    const double Tan = std::tan(0.5 * t);
    mat3_mul_vec(W, T, WT);
    mat3_mul_vec(W, WT, WWT);
    for (int i = 0; i < 3; ++i)
      xi[3 + i] = T[i] - (0.5 * t) * WT[i] + (1 - t / (2. * Tan)) * WWT[i];
  }
}

// ---- Pose2: state = (x, y, theta) -----------------------------------------
struct Pose2 {
  double x, y, c, s;
};
// Rot2::normalize — gtsam/geometry/Rot2.cpp:56-64
inline void rot2_normalize(double& c, double& s) {
  double scale = c * c + s * s;
  if (std::fabs(scale - 1.0) > 1e-10) {
    scale = 1 / std::sqrt(scale);
    c *= scale;
    s *= scale;
  }
}
inline Pose2 pose2_from(const double* st) {  // Pose2(x,y,theta): Rot2::fromAngle
  Pose2 p;
  p.x = st[0]; p.y = st[1];
  p.c = std::cos(st[2]); p.s = std::sin(st[2]);
  return p;
}
inline double pose2_theta(const Pose2& p) { return std::atan2(p.s, p.c); }  // Rot2::theta
// Pose2::operator* — gtsam/geometry/Pose2.h: (r1*r2, t1 + r1*t2), Rot2::operator* Rot2.h:120-122
inline Pose2 pose2_compose(const Pose2& a, const Pose2& b) {
  Pose2 r;
  r.c = a.c * b.c - a.s * b.s;
  r.s = a.s * b.c + a.c * b.s;
  rot2_normalize(r.c, r.s);
  r.x = a.x + a.c * b.x - a.s * b.y;
  r.y = a.y + a.s * b.x + a.c * b.y;
  return r;
}
// Pose2::inverse — gtsam/geometry/Pose2.cpp:202-204
inline Pose2 pose2_inverse(const Pose2& a) {
  Pose2 r;
  r.c = a.c; r.s = -a.s;
  const double tx = -a.x, ty = -a.y;  // unrotate: R' * p
  r.x = a.c * tx + a.s * ty;
  r.y = -a.s * tx + a.c * ty;
  return r;
}
// Pose2::AdjointMap — gtsam/geometry/Pose2.cpp:127-135; 3x3 row-major
inline void pose2_adjoint(const Pose2& p, double* Ad) {
  Ad[0] = p.c; Ad[1] = -p.s; Ad[2] = p.y;
  Ad[3] = p.s; Ad[4] = p.c; Ad[5] = -p.x;
  Ad[6] = 0; Ad[7] = 0; Ad[8] = 1;
}

// ---- BAL projection: PinholeCamera<Cal3Bundler>::project2 ------------------
// gtsam/geometry/PinholeCamera.h:228-240, PinholePose.h:90-109,
// CalibratedCamera.cpp:27-46,88-135, Pose3.cpp:380-397, Cal3Bundler.cpp:64-90.
// cam = R[9] t[3] f k1 k2 u0 v0.  Returns false on cheirality (z <= 0).
// H1: 2x9 row-major, H2: 2x3 row-major, pi: 2.
inline bool sfm_project(const double* cam, const double* pt, double* pi, double* H1, double* H2) {
  const double* R = cam;
  const double* t = cam + 9;
  const double f = cam[12], k1 = cam[13], k2 = cam[14], u0 = cam[15], v0 = cam[16];
  double d3[3] = {pt[0] - t[0], pt[1] - t[1], pt[2] - t[2]}, q[3];
  mat3_tmul_vec(R, d3, q);  // Pose3::transformTo
  if (q[2] <= 0) return false;  // CalibratedCamera.cpp:121-122
  const double d = 1.0 / q[2];
  const double u = q[0] * d, v = q[1] * d;
  const double uv = u * v, uu = u * u, vv = v * v;
  // Dpose — CalibratedCamera.cpp:27-33
  const double Dpose[12] = {uv, -1 - uu, v, -d, 0, d * u, 1 + vv, -uv, -u, 0, -d, d * v};
  // Dpoint — CalibratedCamera.cpp:36-46: d * [R'(0,:) - u R'(2,:); R'(1,:) - v R'(2,:)]
  double Dpoint[6];
  for (int j = 0; j < 3; ++j) {
    Dpoint[j] = d * (R[3 * j + 0] - u * R[3 * j + 2]);
    Dpoint[3 + j] = d * (R[3 * j + 1] - v * R[3 * j + 2]);
  }
  // Cal3Bundler::uncalibrate — Cal3Bundler.cpp:64-90
  const double r = uu + vv;
  const double g = 1. + (k1 + k2 * r) * r;
  const double gu = g * u, gv = g * v;
  pi[0] = u0 + f * gu;
  pi[1] = v0 + f * gv;
  if (H1 && H2) {
    const double rx = r * u, ry = r * v;
    const double Dcal[6] = {gu, f * rx, f * r * rx, gv, f * ry, f * r * ry};
    const double a = 2. * (k1 + 2. * k2 * r);  // Cal3Bundler.cpp:82-85
    const double axx = a * u * u, axy = a * u * v, ayy = a * v * v;
    const double Dp[4] = {f * (g + axx), f * axy, f * axy, f * (g + ayy)};
    for (int i = 0; i < 2; ++i) {
      for (int j = 0; j < 6; ++j) H1[9 * i + j] = Dp[2 * i] * Dpose[j] + Dp[2 * i + 1] * Dpose[6 + j];
      for (int j = 0; j < 3; ++j) H1[9 * i + 6 + j] = Dcal[3 * i + j];
      for (int j = 0; j < 3; ++j) H2[3 * i + j] = Dp[2 * i] * Dpoint[j] + Dp[2 * i + 1] * Dpoint[3 + j];
    }
  }
  return true;
}

// GenericProjectionFactor's camera: PinholeCamera<Cal3_S2>(pose, K).project(point) — gtsam/geometry/PinholePose.h:90-109,
// CalibratedCamera.cpp:27-46,116-135, Cal3_S2.cpp:54-62.  pose: R9 t3; K = (fx, fy, s, u0, v0); false = cheirality.
inline bool pinhole_project_s2(const double* pose, const double* pt, const double* K, double* pi, double* H1, double* H2) {
  const double dx = pt[0] - pose[9], dy = pt[1] - pose[10], dz = pt[2] - pose[11];
  const double qx = pose[0] * dx + pose[3] * dy + pose[6] * dz;
  const double qy = pose[1] * dx + pose[4] * dy + pose[7] * dz;
  const double qz = pose[2] * dx + pose[5] * dy + pose[8] * dz;
  if (qz <= 0) return false;
  const double d = 1.0 / qz;
  const double u = qx * d, v = qy * d;
  pi[0] = K[0] * u + K[2] * v + K[3];
  pi[1] = K[1] * v + K[4];
  if (H1) {
    const double uv = u * v, uu = u * u, vv = v * v;
    const double Dpose[12] = {uv, -1 - uu, v, -d, 0, d * u, 1 + vv, -uv, -u, 0, -d, d * v};
    double Dpoint[6];
    for (int j = 0; j < 3; ++j) {
      Dpoint[j] = d * (pose[3 * j + 0] - u * pose[3 * j + 2]);
      Dpoint[3 + j] = d * (pose[3 * j + 1] - v * pose[3 * j + 2]);
    }
    for (int j = 0; j < 6; ++j) {
      H1[j] = K[0] * Dpose[j] + K[2] * Dpose[6 + j];
      H1[6 + j] = K[1] * Dpose[6 + j];
    }
    for (int j = 0; j < 3; ++j) {
      H2[j] = K[0] * Dpoint[j] + K[2] * Dpoint[3 + j];
      H2[3 + j] = K[1] * Dpoint[3 + j];
    }
  }
  return true;
}

// BearingRange<Pose2, Point2>::Measure — Pose2::bearing / range (gtsam/geometry/Pose2.cpp:246-285), Rot2::relativeBearing
// (Rot2.cpp:119-130).  pose = (x, y, theta); br = (bearing angle, range); H1 2x3 / H2 2x2 row-major.
inline void bearing_range_2d(const double* pose, const double* pt, double* br, double* H1, double* H2) {
  const double c = std::cos(pose[2]), s = std::sin(pose[2]);
  const double dx = pt[0] - pose[0], dy = pt[1] - pose[1];
  const double qx = c * dx + s * dy, qy = -s * dx + c * dy;
  const double d2 = qx * qx + qy * qy, n = std::sqrt(d2);
  const bool far = std::abs(n) > 1e-5;
  br[0] = far ? std::atan2(qy, qx) : 0.0;
  br[1] = n;
  if (H1) {
    const double bx = far ? -qy / d2 : 0.0, by = far ? qx / d2 : 0.0;
    H1[0] = -bx; H1[1] = -by; H1[2] = bx * qy - by * qx;
    H2[0] = bx * c - by * s; H2[1] = bx * s + by * c;
    const double rx = dx / n, ry = dy / n;
    H1[3] = -rx * c - ry * s; H1[4] = rx * s - ry * c; H1[5] = 0.0;
    H2[2] = rx; H2[3] = ry;
  }
}

}  // namespace orc
