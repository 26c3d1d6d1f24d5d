// device_geometry.h — FP64 __device__ restatement of the Lie-group / camera math of the hot path
// (product code; gfx950 only).  Reference semantics (paths relative to /root/reference/):
//   SO3 Expmap/Logmap        gtsam/geometry/SO3.cpp:61-112,299-375
//   Pose3 compose/inverse/AdjointMap/Expmap/Logmap  gtsam/geometry/Pose3.cpp:61-75,184-245
//   Pose2 compose/inverse/AdjointMap/chart          gtsam/geometry/Pose2.cpp:100-135,202-204, Rot2.cpp:56-64
//   BAL projection chain     gtsam/geometry/CalibratedCamera.cpp:27-46,116-135, Cal3Bundler.cpp:64-90
// with the reference's default compile-time switches (full Pose3/Rot3 expmap, no quaternions,
// BetweenFactor without the Local Jacobian, cheirality -> zeroed factor).
#pragma once
#include <hip/hip_runtime.h>

namespace gsxd {

struct M3 {
  double a[9];  // row-major
};
struct V3 {
  double x, y, z;
};
struct P3 {
  M3 R;
  V3 t;
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
  return C;
}
__device__ __forceinline__ M3 transpose(const M3& A) {
  M3 T;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T.a[3 * i + j] = A.a[3 * j + i];
  return T;
}
__device__ __forceinline__ V3 mulv(const M3& A, V3 v) {
  return V3{A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
            A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z};
}
__device__ __forceinline__ V3 tmulv(const M3& A, V3 v) {  // A' v
  return V3{A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z, A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z,
            A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z};
}
__device__ __forceinline__ M3 skew(V3 w) {
  M3 W;
  W.a[0] = 0; W.a[1] = -w.z; W.a[2] = w.y;
  W.a[3] = w.z; W.a[4] = 0; W.a[5] = -w.x;
  W.a[6] = -w.y; W.a[7] = w.x; W.a[8] = 0;
  return W;
}

__device__ __forceinline__ P3 load_pose3(const double* s) {
  P3 p;
#pragma unroll
  for (int i = 0; i < 9; ++i) p.R.a[i] = s[i];
  p.t = V3{s[9], s[10], s[11]};
  return p;
}
__device__ __forceinline__ void store_pose3(const P3& p, double* s) {
#pragma unroll
  for (int i = 0; i < 9; ++i) s[i] = p.R.a[i];
  s[9] = p.t.x; s[10] = p.t.y; s[11] = p.t.z;
}
__device__ __forceinline__ P3 compose(const P3& a, const P3& b) { return P3{mul(a.R, b.R), a.t + mulv(a.R, b.t)}; }
__device__ __forceinline__ P3 inverse(const P3& a) {
  P3 c;
  c.R = transpose(a.R);
  c.t = mulv(c.R, V3{-a.t.x, -a.t.y, -a.t.z});
  return c;
}
__device__ __forceinline__ P3 between(const P3& a, const P3& b) { return compose(inverse(a), b); }

// Pose3::Expmap with the w.w <= 1e-5 Taylor branch (Pose3.cpp:189, SO3.cpp:61-112).
__device__ inline P3 pose3_expmap(const double* xi) {
  const V3 w{xi[0], xi[1], xi[2]}, v{xi[3], xi[4], xi[5]};
  const double theta2 = dot(w, w);
  double A, B, C;
  if (theta2 <= 1e-5) {
    A = 1.0 - theta2 * (1.0 / 6.0);
    B = 0.5 - theta2 * (1.0 / 24.0);
    C = (1.0 / 6.0) - theta2 * (1.0 / 120.0);
  } else {
    const double theta = sqrt(theta2);
    A = sin(theta) / theta;
    const double s2 = sin(theta / 2.0);
    B = 2.0 * s2 * s2 / theta2;
    C = (1 - A) / theta2;
  }
  const M3 W = skew(w), WW = mul(W, W);
  P3 p;
#pragma unroll
  for (int i = 0; i < 9; ++i) p.R.a[i] = A * W.a[i] + B * WW.a[i];
  p.R.a[0] += 1.0; p.R.a[4] += 1.0; p.R.a[8] += 1.0;
  const V3 wv = cross(w, v), wwv = cross(w, wv);
  p.t = v + B * wv + C * wwv;
  return p;
}

// SO3::Logmap (SO3.cpp:299-375)
__device__ inline V3 so3_logmap(const M3& Rm) {
  const double R11 = Rm.a[0], R12 = Rm.a[1], R13 = Rm.a[2];
  const double R21 = Rm.a[3], R22 = Rm.a[4], R23 = Rm.a[5];
  const double R31 = Rm.a[6], R32 = Rm.a[7], R33 = Rm.a[8];
  const double tr = R11 + R22 + R33;
  const double kPi = 3.14159265358979323846;
  if (tr + 1.0 < 1e-3) {
    double W, Q1, Q2, Q3;
    int perm;
    if (R33 > R22 && R33 > R11) {
      W = R21 - R12; Q1 = 2.0 + 2.0 * R33; Q2 = R31 + R13; Q3 = R23 + R32; perm = 0;
    } else if (R22 > R11) {
      W = R13 - R31; Q1 = 2.0 + 2.0 * R22; Q2 = R23 + R32; Q3 = R12 + R21; perm = 1;
    } else {
      W = R32 - R23; Q1 = 2.0 + 2.0 * R11; Q2 = R12 + R21; Q3 = R31 + R13; perm = 2;
    }
    const double r = sqrt(Q1), one_over_r = 1 / r;
    const double norm = sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
    const double sgn_w = W < 0 ? -1.0 : 1.0;
    const double mag = kPi - (2 * sgn_w * W) / norm;
    const double scale = sgn_w * 0.5 * one_over_r * mag;
    if (perm == 0) return V3{scale * Q2, scale * Q3, scale * Q1};
    if (perm == 1) return V3{scale * Q3, scale * Q1, scale * Q2};
    return V3{scale * Q1, scale * Q2, scale * Q3};
  }
  double magnitude;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-6) {
    const double theta = acos((tr - 1.0) / 2.0);
    magnitude = theta / (2.0 * sin(theta));
  } else {
    magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
  }
  return V3{magnitude * (R32 - R23), magnitude * (R13 - R31), magnitude * (R21 - R12)};
}

// Pose3::Logmap (Pose3.cpp:225-245)
__device__ inline void pose3_logmap(const P3& p, double* xi) {
  const V3 w = so3_logmap(p.R);
  const V3 T = p.t;
  const double t = sqrt(dot(w, w));
  xi[0] = w.x; xi[1] = w.y; xi[2] = w.z;
  if (t < 1e-10) {
    xi[3] = T.x; xi[4] = T.y; xi[5] = T.z;
  } else {
    const M3 W = skew((1.0 / t) * w);
    const double Tan = tan(0.5 * t);
    const V3 WT = mulv(W, T);
    const V3 u = T - (0.5 * t) * WT + (1 - t / (2. * Tan)) * mulv(W, WT);
    xi[3] = u.x; xi[4] = u.y; xi[5] = u.z;
  }
}

// 6x6 AdjointMap entry (row r, col c) of pose p: [[R,0],[[t]x R, R]] (Pose3.cpp:69-75)
__device__ inline void pose3_adjoint(const P3& p, double* Ad /*36 row-major*/) {
  const M3 TR = mul(skew(p.t), p.R);
#pragma unroll
  for (int i = 0; i < 36; ++i) Ad[i] = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Ad[6 * i + j] = p.R.a[3 * i + j];
      Ad[6 * (i + 3) + j] = TR.a[3 * i + j];
      Ad[6 * (i + 3) + j + 3] = p.R.a[3 * i + j];
    }
}

// ---- Pose2 -----------------------------------------------------------------------------------
struct P2 {
  double x, y, c, s;
};
__device__ __forceinline__ P2 load_pose2(const double* st) { return P2{st[0], st[1], cos(st[2]), sin(st[2])}; }
__device__ __forceinline__ void rot2_normalize(double& c, double& s) {
  double scale = c * c + s * s;
  if (fabs(scale - 1.0) > 1e-10) {
    scale = 1 / sqrt(scale);
    c *= scale;
    s *= scale;
  }
}
__device__ __forceinline__ P2 compose(const P2& a, const P2& b) {
  P2 r;
  r.c = a.c * b.c - a.s * b.s;
  r.s = a.s * b.c + a.c * b.s;
  rot2_normalize(r.c, r.s);
  r.x = a.x + a.c * b.x - a.s * b.y;
  r.y = a.y + a.s * b.x + a.c * b.y;
  return r;
}
__device__ __forceinline__ P2 inverse(const P2& a) {
  const double tx = -a.x, ty = -a.y;
  return P2{a.c * tx + a.s * ty, -a.s * tx + a.c * ty, a.c, -a.s};
}
__device__ __forceinline__ double theta(const P2& p) { return atan2(p.s, p.c); }

// ---- BAL projection (returns false on cheirality) ---------------------------------------------
// cam: R9 t3 f k1 k2 u0 v0.  H1 2x9 / H2 2x3 row-major when non-null.
__device__ inline bool sfm_project(const double* cam, const double* pt, double* pi, double* H1, double* H2) {
  const double f = cam[12], k1 = cam[13], k2 = cam[14], u0 = cam[15], v0 = cam[16];
  const double dx = pt[0] - cam[9], dy = pt[1] - cam[10], dz = pt[2] - cam[11];
  const double qx = cam[0] * dx + cam[3] * dy + cam[6] * dz;
  const double qy = cam[1] * dx + cam[4] * dy + cam[7] * dz;
  const double qz = cam[2] * dx + cam[5] * dy + cam[8] * dz;
  if (qz <= 0) return false;
  const double d = 1.0 / qz;
  const double u = qx * d, v = qy * d;
  const double uv = u * v, uu = u * u, vv = v * v;
  const double r = uu + vv;
  const double g = 1. + (k1 + k2 * r) * r;
  const double gu = g * u, gv = g * v;
  pi[0] = u0 + f * gu;
  pi[1] = v0 + f * gv;
  if (H1) {
    const double Dpose[12] = {uv, -1 - uu, v, -d, 0, d * u, 1 + vv, -uv, -u, 0, -d, d * v};
    double Dpoint[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Dpoint[j] = d * (cam[3 * j + 0] - u * cam[3 * j + 2]);
      Dpoint[3 + j] = d * (cam[3 * j + 1] - v * cam[3 * j + 2]);
    }
    const double rx = r * u, ry = r * v;
    const double Dcal[6] = {gu, f * rx, f * r * rx, gv, f * ry, f * r * ry};
    const double a = 2. * (k1 + 2. * k2 * r);
    const double axx = a * u * u, axy = a * u * v, ayy = a * v * v;
    const double Dp[4] = {f * (g + axx), f * axy, f * axy, f * (g + ayy)};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) H1[9 * i + j] = Dp[2 * i] * Dpose[j] + Dp[2 * i + 1] * Dpose[6 + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) H1[9 * i + 6 + j] = Dcal[3 * i + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) H2[3 * i + j] = Dp[2 * i] * Dpoint[j] + Dp[2 * i + 1] * Dpoint[3 + j];
    }
  }
  return true;
}

// ---- pinhole projection with a fixed Cal3_S2 (GenericProjectionFactor; returns false on cheirality) -------------
// pose: R9 t3; K = (fx, fy, s, u0, v0).  H1 2x6 / H2 2x3 row-major when non-null.
// PinholeBase::project2 + Dpose / Dpoint (gtsam/geometry/CalibratedCamera.cpp:27-46,116-135), Cal3_S2::uncalibrate
// (gtsam/geometry/Cal3_S2.cpp:54-62): (u, v) = (fx x + s y + u0, fy y + v0), Dp = [[fx, s], [0, fy]].
__device__ inline bool pinhole_project_s2(const double* pose, const double* pt, const double* K, double* pi, double* H1,
                                          double* H2) {
  const double dx = pt[0] - pose[9], dy = pt[1] - pose[10], dz = pt[2] - pose[11];
  const double qx = pose[0] * dx + pose[3] * dy + pose[6] * dz;
  const double qy = pose[1] * dx + pose[4] * dy + pose[7] * dz;
  const double qz = pose[2] * dx + pose[5] * dy + pose[8] * dz;
  if (qz <= 0) return false;
  const double d = 1.0 / qz;
  const double u = qx * d, v = qy * d;
  const double fx = K[0], fy = K[1], sk = K[2];
  pi[0] = fx * u + sk * v + K[3];
  pi[1] = fy * v + K[4];
  if (H1) {
    const double uv = u * v, uu = u * u, vv = v * v;
    const double Dpose[12] = {uv, -1 - uu, v, -d, 0, d * u, 1 + vv, -uv, -u, 0, -d, d * v};
    double Dpoint[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Dpoint[j] = d * (pose[3 * j + 0] - u * pose[3 * j + 2]);
      Dpoint[3 + j] = d * (pose[3 * j + 1] - v * pose[3 * j + 2]);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      H1[j] = fx * Dpose[j] + sk * Dpose[6 + j];
      H1[6 + j] = fy * Dpose[6 + j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      H2[j] = fx * Dpoint[j] + sk * Dpoint[3 + j];
      H2[3 + j] = fy * Dpoint[3 + j];
    }
  }
  return true;
}

// ---- BearingRange<Pose2, Point2>::Measure with Jacobians (Pose2::bearing / range, gtsam/geometry/Pose2.cpp:246-285;
// Rot2::relativeBearing, Rot2.cpp:119-130).  pose = (x, y, theta); out: bearing angle, range; H1 2x3 / H2 2x2 row-major.
__device__ inline void bearing_range_2d(const double* pose, const double* pt, double* br, double* H1, double* H2) {
  const double c = cos(pose[2]), s = sin(pose[2]);
  const double dx = pt[0] - pose[0], dy = pt[1] - pose[1];
  const double qx = c * dx + s * dy, qy = -s * dx + c * dy;  // transformTo
  const double d2 = qx * qx + qy * qy, n = sqrt(d2);
  const bool far = fabs(n) > 1e-5;
  br[0] = far ? atan2(qy, qx) : 0.0;
  br[1] = n;
  if (H1) {
    const double bx = far ? -qy / d2 : 0.0, by = far ? qx / d2 : 0.0;  // D bearing / D q
    // D q / D pose = [-1 0 qy; 0 -1 -qx], D q / D point = R'
    H1[0] = -bx; H1[1] = -by; H1[2] = bx * qy - by * qx;
    H2[0] = bx * c - by * s; H2[1] = bx * s + by * c;
    // range: d = point - t (world), D r / D d = d' / r, D d / D pose = [-c s 0; -s -c 0]
    const double rx = dx / n, ry = dy / n;
    H1[3] = -rx * c - ry * s; H1[4] = rx * s - ry * c; H1[5] = 0.0;
    H2[2] = rx; H2[3] = ry;
  }
}
__device__ inline double wrap_angle(double a) { return atan2(sin(a), cos(a)); }

}  // namespace gsxd
