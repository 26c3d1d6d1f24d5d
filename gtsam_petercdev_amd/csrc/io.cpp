// io.cpp — native readers of the on-disk formats either side of the hot path: g2o pose graphs
// and BAL bundle-adjustment files, lowered straight to a gsx_problem_desc + packed initial Values.  Host only.
//
// Behaviour follows the reference's parsers (paths relative to /root/reference/):
//   g2o 2-D  gtsam/slam/dataset.cpp:216-296,505-633  VERTEX_SE2|VERTEX2 id x y theta;
//            EDGE_SE2|EDGE2|EDGE|ODOMETRY i j x y theta I11 I12 I13 I22 I23 I33; a vertex that only appears in edges is
//            created by chaining the odometry (:541-546)
//   g2o 3-D  gtsam/slam/dataset.cpp:756-863           VERTEX_SE3:QUAT id x y z qx qy qz qw; EDGE_SE3:QUAT i j x y z
//            qx qy qz qw + 21 upper-triangular information entries in (t, R) order, permuted to GTSAM's (R, t) (:850-856)
//   noise    noiseModel::Gaussian::Information(I): R = LLT(I).matrixU() (gtsam/linear/NoiseModel.cpp:98-112)
//   anchor   the prior the reference's examples add on the first pose, appended last
//            (examples/Pose2SLAMExample_g2o.cpp:65-67, examples/Pose3SLAMExample_g2o.cpp:42-48)
//   BAL      gtsam/sfm/SfmData.cpp:189-245 (file order, values read as `float`), openGL2gtsam :79-85, measurement (u, -v),
//            tracks in file order; graph of examples/SFMExample_bal.cpp:55-68 (unit pixel noise, optional priors
//            Isotropic(9, 0.1) on camera 0 and Isotropic(3, 0.1) on point 0 appended last); keys: cameras 0.., points
//            Symbol('p', j)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "gsx_internal.h"

struct gsx_dataset {
  std::vector<uint64_t> var_keys;
  std::vector<int32_t> var_types, var_dims, f_type, f_rows, f_key_ptr, f_vars, f_noise_kind;
  std::vector<int64_t> f_meas_ptr, f_noise_ptr;
  std::vector<double> meas, noise, values;
  std::string err;
};

namespace {

void quat_to_R(double w, double x, double y, double z, double* R) {  // row-major, Eigen's normalised convention
  const double n = std::sqrt(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// upper Cholesky factor R (row-major d x d, R'R = I) of a symmetric positive-definite information matrix
bool chol_upper(const double* I, int d, double* R) {
  std::vector<double> L(d * d, 0.0);  // lower, row-major
  for (int j = 0; j < d; ++j) {
    double s = I[j * d + j];
    for (int k = 0; k < j; ++k) s -= L[j * d + k] * L[j * d + k];
    if (!(s > 0)) return false;
    const double ljj = std::sqrt(s);
    L[j * d + j] = ljj;
    for (int i = j + 1; i < d; ++i) {
      double t = I[i * d + j];
      for (int k = 0; k < j; ++k) t -= L[i * d + k] * L[j * d + k];
      L[i * d + j] = t / ljj;
    }
  }
  for (int r = 0; r < d; ++r)
    for (int c = 0; c < d; ++c) R[r * d + c] = (c >= r) ? L[c * d + r] : 0.0;
  return true;
}

void so3_expmap(const double* w, double* R) {  // gtsam/geometry/SO3.cpp:61-96, row-major
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double a, b;
  if (th2 <= 2.220446049250313e-16) {
    a = 1.0 - th2 / 6.0;
    b = 0.5 - th2 / 24.0;
  } else {
    const double th = std::sqrt(th2), s2 = std::sin(th / 2.0);
    a = std::sin(th) / th;
    b = 2.0 * s2 * s2 / th2;
  }
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double WW[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) WW[i * 3 + j] = W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j] + W[i * 3 + 2] * W[6 + j];
  for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * W[i] + b * WW[i];
}

// Rodrigues vector of a rotation matrix (row-major) — Rot3::Logmap; through the unit quaternion, which stays accurate
// near 0 and near pi
void so3_logmap(const double* R, double* w) {
  double q[4];  // (w, x, y, z)
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(1.0 + R[i * 4] - R[j * 4] - R[k * 4]) * 2;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) / s;
    q[1 + i] = 0.25 * s;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) / s;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) / s;
  }
  if (q[0] < 0)
    for (double& x : q) x = -x;
  const double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double scale = n < 1e-12 ? 2.0 : 2.0 * std::atan2(n, q[0]) / n;
  w[0] = scale * q[1]; w[1] = scale * q[2]; w[2] = scale * q[3];
}

// Rot3::Ypr(y, p, r) = Rz(y) Ry(p) Rx(r) (gtsam/geometry/Rot3.h), row-major
void ypr_to_R(double y, double p, double r, double* R) {
  const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}
// packed Pose3 states (R row-major, then t)
std::vector<double> pose3_compose(const std::vector<double>& a, const std::vector<double>& b) {
  std::vector<double> c(12);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) c[r * 3 + k] = a[r * 3] * b[k] + a[r * 3 + 1] * b[3 + k] + a[r * 3 + 2] * b[6 + k];
    c[9 + r] = a[9 + r] + a[r * 3] * b[9] + a[r * 3 + 1] * b[10] + a[r * 3 + 2] * b[11];
  }
  return c;
}
std::vector<double> pose3_inverse(const std::vector<double>& a) {
  std::vector<double> c(12);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) c[r * 3 + k] = a[k * 3 + r];
    c[9 + r] = -(a[r] * a[9] + a[3 + r] * a[10] + a[6 + r] * a[11]);
  }
  return c;
}

// createNoiseModel (gtsam/slam/dataset.cpp:216-296) for the six numbers of a 2-D edge; appends kind and parameters.
// smart: Gaussian::Information / Covariance with smart = true fall through Diagonal -> Isotropic -> Unit when the matrix
// allows (gtsam/linear/NoiseModel.cpp:98-131, 284-315, 625-634).
bool emit_noise_2d(const double v[6], int format, bool smart, int kernel, gsx_dataset* D, std::string* err);

struct Edge {
  long long i, j;
  std::vector<double> z;  // measurement in packed state layout
  std::vector<double> R;  // d x d upper factor
};

bool emit_noise_2d(const double v[6], int format, bool smart, int kernel, gsx_dataset* D, std::string* err) {
  if (format == GSX_NOISE_FORMAT_AUTO) {
    if (v[0] != 0.0 && v[1] == 0.0 && v[2] != 0.0 && v[3] != 0.0 && v[4] == 0.0 && v[5] == 0.0) format = GSX_NOISE_FORMAT_GRAPH;
    else if (v[0] != 0.0 && v[1] == 0.0 && v[2] == 0.0 && v[3] != 0.0 && v[4] == 0.0 && v[5] != 0.0) format = GSX_NOISE_FORMAT_COV;
    else {
      *err = "load2D: unrecognized covariance matrix format in dataset file";
      return false;
    }
  }
  double M[9];
  if (format == GSX_NOISE_FORMAT_G2O || format == GSX_NOISE_FORMAT_COV) {
    if (v[0] == 0.0 || v[3] == 0.0 || v[5] == 0.0) {
      *err = "load2D: not G2O matrix order";
      return false;
    }
    const double m[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
    std::copy(m, m + 9, M);
  } else {
    if (v[0] == 0.0 || v[2] == 0.0 || v[3] == 0.0) {
      *err = "load2D: not TORO matrix order";
      return false;
    }
    const double m[9] = {v[0], v[1], v[4], v[1], v[2], v[5], v[4], v[5], v[3]};
    std::copy(m, m + 9, M);
  }
  const bool is_cov = format == GSX_NOISE_FORMAT_GRAPH || format == GSX_NOISE_FORMAT_COV;
  const bool diagonal = std::abs(M[1]) <= 1e-9 && std::abs(M[2]) <= 1e-9 && std::abs(M[5]) <= 1e-9;  // checkIfDiagonal
  int kind;
  std::vector<double> params;
  if (smart && diagonal) {
    double var[3] = {M[0], M[4], M[8]};
    if (!is_cov)
      for (double& q : var) q = 1.0 / q;  // Precisions -> Variances
    if (var[0] == var[1] && var[0] == var[2]) {
      if (std::abs(var[0] - 1.0) < 1e-9) {
        kind = GSX_NOISE_UNIT;
      } else {
        kind = GSX_NOISE_ISOTROPIC;
        params = {std::sqrt(var[0])};
      }
    } else {
      kind = GSX_NOISE_DIAGONAL;
      params = {std::sqrt(var[0]), std::sqrt(var[1]), std::sqrt(var[2])};
    }
  } else {
    double I[9];
    if (is_cov) {  // Information(covariance.inverse(), false)
      const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                         M[2] * (M[3] * M[7] - M[4] * M[6]);
      if (det == 0.0) {
        *err = "load2D: singular covariance";
        return false;
      }
      const double a[9] = {M[4] * M[8] - M[5] * M[7], M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                           M[5] * M[6] - M[3] * M[8], M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                           M[3] * M[7] - M[4] * M[6], M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
      for (int k = 0; k < 9; ++k) I[k] = a[k] / det;
    } else {
      std::copy(M, M + 9, I);
    }
    kind = GSX_NOISE_GAUSSIAN;
    params.resize(9);
    if (!chol_upper(I, 3, params.data())) {
      *err = "load2D: information matrix is not positive definite";
      return false;
    }
  }
  if (kernel == 1) {  // KernelFunctionTypeHUBER: mEstimator::Huber::Create(1.345)
    kind |= GSX_NOISE_ROBUST_HUBER;
    params.push_back(1.345);
  } else if (kernel == 2) {  // KernelFunctionTypeTUKEY: mEstimator::Tukey::Create(4.6851)
    kind |= GSX_NOISE_ROBUST_TUKEY;
    params.push_back(4.6851);
  }
  D->f_noise_kind.push_back(kind);
  D->noise.insert(D->noise.end(), params.begin(), params.end());
  return true;
}

}  // namespace

extern "C" {

static gsx_status read_g2o_impl(const char* path, int32_t is3d, gsx_dataset** out) {
  if (!path || !out) return GSX_E_INVALID;
  std::ifstream in(path);
  if (!in) return GSX_E_INVALID;
  const int d = is3d ? 6 : 3, sd = is3d ? 12 : 3;
  std::map<long long, std::vector<double>> states;
  std::vector<Edge> edges;
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    if (!(ss >> tag)) continue;
    if (!is3d && (tag == "VERTEX_SE2" || tag == "VERTEX2")) {
      long long id;
      double x, y, th;
      if (!(ss >> id >> x >> y >> th)) return GSX_E_INVALID;
      states[id] = {x, y, th};
    } else if (!is3d && (tag == "EDGE_SE2" || tag == "EDGE2" || tag == "EDGE" || tag == "ODOMETRY")) {
      Edge e;
      double x, y, th, v[6];
      if (!(ss >> e.i >> e.j >> x >> y >> th)) return GSX_E_INVALID;
      for (double& q : v)
        if (!(ss >> q)) return GSX_E_INVALID;
      e.z = {x, y, th};
      const double I[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
      e.R.resize(9);
      if (!chol_upper(I, 3, e.R.data())) return GSX_E_INVALID;
      edges.push_back(e);
    } else if (is3d && tag == "VERTEX_SE3:QUAT") {
      long long id;
      double x, y, z, qx, qy, qz, qw;
      if (!(ss >> id >> x >> y >> z >> qx >> qy >> qz >> qw)) return GSX_E_INVALID;
      std::vector<double> s(12);
      quat_to_R(qw, qx, qy, qz, s.data());
      s[9] = x; s[10] = y; s[11] = z;
      states[id] = s;
    } else if (is3d && tag == "VERTEX3") {
      // TORO: id x y z roll pitch yaw, R = Rot3::Ypr(yaw, pitch, roll) (dataset.cpp:741-764)
      long long id;
      double x, y, z, roll, pitch, yaw;
      if (!(ss >> id >> x >> y >> z >> roll >> pitch >> yaw)) return GSX_E_INVALID;
      std::vector<double> s(12);
      ypr_to_R(yaw, pitch, roll, s.data());
      s[9] = x; s[10] = y; s[11] = z;
      states[id] = s;
    } else if (is3d && tag == "EDGE3") {
      // TORO: i j x y z roll pitch yaw + 21 upper-triangular information entries, taken as they are (dataset.cpp:829-840)
      Edge e;
      double x, y, z, roll, pitch, yaw, up[21];
      if (!(ss >> e.i >> e.j >> x >> y >> z >> roll >> pitch >> yaw)) return GSX_E_INVALID;
      for (double& q : up)
        if (!(ss >> q)) return GSX_E_INVALID;
      e.z.resize(12);
      ypr_to_R(yaw, pitch, roll, e.z.data());
      e.z[9] = x; e.z[10] = y; e.z[11] = z;
      double m[36];
      int k = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) m[r * 6 + c] = m[c * 6 + r] = up[k++];
      e.R.resize(36);
      if (!chol_upper(m, 6, e.R.data())) return GSX_E_INVALID;
      edges.push_back(e);
    } else if (is3d && tag == "EDGE_SE3:QUAT") {
      Edge e;
      double x, y, z, qx, qy, qz, qw, up[21];
      if (!(ss >> e.i >> e.j >> x >> y >> z >> qx >> qy >> qz >> qw)) return GSX_E_INVALID;
      for (double& q : up)
        if (!(ss >> q)) return GSX_E_INVALID;
      e.z.resize(12);
      quat_to_R(qw, qx, qy, qz, e.z.data());
      e.z[9] = x; e.z[10] = y; e.z[11] = z;
      double m[36], mg[36];
      int k = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) m[r * 6 + c] = m[c * 6 + r] = up[k++];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          mg[r * 6 + c] = m[(3 + r) * 6 + (3 + c)];        // R block first
          mg[(3 + r) * 6 + (3 + c)] = m[r * 6 + c];          // then t
          mg[r * 6 + (3 + c)] = m[(3 + r) * 6 + c];
          mg[(3 + r) * 6 + c] = m[r * 6 + (3 + c)];
        }
      e.R.resize(36);
      if (!chol_upper(mg, 6, e.R.data())) return GSX_E_INVALID;
      edges.push_back(e);
    }
  }
  if (!is3d)
    for (const Edge& e : edges) {  // chain the odometry for vertices without a VERTEX line
      if (!states.count(e.i)) states[e.i] = {0.0, 0.0, 0.0};
      if (!states.count(e.j)) {
        const std::vector<double>& a = states[e.i];
        const double c = std::cos(a[2]), s = std::sin(a[2]);
        states[e.j] = {a[0] + c * e.z[0] - s * e.z[1], a[1] + s * e.z[0] + c * e.z[1], a[2] + e.z[2]};
      }
    }
  if (is3d) {
    // The C++ reader leaves poses without a VERTEX line out of the Values (dataset.cpp:929-931); files of edges only
    // (examples/Data/sphere2500.txt) are read by the reference's MATLAB loader, which starts at the origin and chains
    // the successive odometry edges (matlab/+gtsam/load3D.m:21-53) — done here for the poses that are missing.
    bool missing = false;
    for (const Edge& e : edges) missing = missing || !states.count(e.i) || !states.count(e.j);
    if (missing) {
      if (!states.count(0)) states[0] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
      for (const Edge& e : edges) {
        if (e.j == e.i + 1 && states.count(e.i) && !states.count(e.j)) states[e.j] = pose3_compose(states[e.i], e.z);
        else if (e.i == e.j + 1 && states.count(e.j) && !states.count(e.i))
          states[e.i] = pose3_compose(states[e.j], pose3_inverse(e.z));
      }
    }
  }
  if (states.empty()) return GSX_E_INVALID;
  gsx_dataset* D = new gsx_dataset();
  std::map<long long, int> index;
  for (const auto& kv : states) {
    index[kv.first] = (int)D->var_keys.size();
    D->var_keys.push_back((uint64_t)kv.first);
    D->var_types.push_back(is3d ? GSX_VAR_POSE3 : GSX_VAR_POSE2);
    D->var_dims.push_back(d);
    D->values.insert(D->values.end(), kv.second.begin(), kv.second.end());
  }
  D->f_key_ptr.push_back(0);
  D->f_meas_ptr.push_back(0);
  D->f_noise_ptr.push_back(0);
  for (const Edge& e : edges) {
    if (!index.count(e.i) || !index.count(e.j)) {
      delete D;
      return GSX_E_INVALID;
    }
    D->f_type.push_back(GSX_F_BETWEEN);
    D->f_rows.push_back(d);
    D->f_vars.push_back(index[e.i]);
    D->f_vars.push_back(index[e.j]);
    D->f_key_ptr.push_back((int32_t)D->f_vars.size());
    D->meas.insert(D->meas.end(), e.z.begin(), e.z.end());
    D->f_meas_ptr.push_back((int64_t)D->meas.size());
    D->f_noise_kind.push_back(GSX_NOISE_GAUSSIAN);
    D->noise.insert(D->noise.end(), e.R.begin(), e.R.end());
    D->f_noise_ptr.push_back((int64_t)D->noise.size());
  }
  // anchoring prior on the first pose, as the reference's example programs add it
  D->f_type.push_back(GSX_F_PRIOR);
  D->f_rows.push_back(d);
  D->f_vars.push_back(0);
  D->f_key_ptr.push_back((int32_t)D->f_vars.size());
  D->meas.insert(D->meas.end(), D->values.begin(), D->values.begin() + sd);
  D->f_meas_ptr.push_back((int64_t)D->meas.size());
  D->f_noise_kind.push_back(GSX_NOISE_DIAGONAL);
  if (is3d)
    for (int k = 0; k < 6; ++k) D->noise.push_back(std::sqrt(k < 3 ? 1e-6 : 1e-4));
  else
    for (double var : {1e-6, 1e-6, 1e-8}) D->noise.push_back(std::sqrt(var));
  D->f_noise_ptr.push_back((int64_t)D->noise.size());
  *out = D;
  return GSX_OK;
}

// load2D — gtsam/slam/dataset.cpp:505-570
static gsx_status load2d_impl(const char* path, const double* model_sigmas, int64_t max_index, int32_t smart, int32_t noise_format,
                      int32_t kernel, gsx_dataset** out) {
  if (!path || !out || noise_format < GSX_NOISE_FORMAT_G2O || noise_format > GSX_NOISE_FORMAT_AUTO || kernel < 0 ||
      kernel > 2 || max_index < 0)
    return GSX_E_INVALID;
  std::ifstream in(path);
  if (!in) return GSX_E_INVALID;
  const uint64_t lbase = (uint64_t)'l' << 56;  // L(j), gtsam/inference/Symbol.h
  const unsigned long long mi = (unsigned long long)max_index;
  struct Rec {
    int kind;  // 0 between, 1 bearing-range
    unsigned long long i, j;
    double z[3], v[6];
  };
  std::map<uint64_t, std::vector<double>> states;  // key -> packed state (3: pose, 2: landmark)
  std::vector<Rec> recs;
  std::string line;
  // first pass: vertices; second (here: recorded in file order): factors — dataset.cpp:512-527
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    if (!(ss >> tag)) continue;
    if (tag == "VERTEX2" || tag == "VERTEX_SE2" || tag == "VERTEX") {
      unsigned long long id;
      double x, y, th;
      if (!(ss >> id >> x >> y >> th)) return GSX_E_INVALID;
      if (!mi || id <= mi) states[id] = {x, y, th};
    } else if (tag == "VERTEX_XY") {
      unsigned long long id;
      double x, y;
      if (!(ss >> id >> x >> y)) return GSX_E_INVALID;
      if (!mi || id <= mi) states[lbase | id] = {x, y};
    } else if (tag == "EDGE2" || tag == "EDGE" || tag == "EDGE_SE2" || tag == "ODOMETRY") {
      Rec r{};
      r.kind = 0;
      if (!(ss >> r.i >> r.j >> r.z[0] >> r.z[1] >> r.z[2])) return GSX_E_INVALID;
      for (double& q : r.v) ss >> q;   // (as the reference: a short line leaves zeros, which the format check rejects)
      if (mi && (r.i > mi || r.j > mi)) continue;
      recs.push_back(r);
    } else if (tag == "BR" || tag == "LANDMARK") {
      Rec r{};
      r.kind = 1;
      if (!(ss >> r.i >> r.j)) return GSX_E_INVALID;
      double bearing, range, bstd, rstd;
      if (tag == "BR") {
        if (!(ss >> bearing >> range >> bstd >> rstd)) return GSX_E_INVALID;
      } else {
        // a landmark sighting (x, y) + covariance (v1 v2 v3), converted to bearing-range (dataset.cpp:463-481)
        double lx, ly, v1, v2, v3;
        if (!(ss >> lx >> ly >> v1 >> v2 >> v3)) return GSX_E_INVALID;
        bearing = std::atan2(ly, lx);
        range = std::sqrt(lx * lx + ly * ly);
        if (std::abs(v1 - v3) < 1e-4) {
          bstd = std::sqrt(v1 / 10.0);
          rstd = std::sqrt(v1);
        } else {
          bstd = rstd = 1;
        }
      }
      if (mi && r.i > mi) continue;
      r.z[0] = bearing; r.z[1] = range; r.v[0] = bstd; r.v[1] = rstd;
      recs.push_back(r);
    }
  }
  // variables referenced but not declared: odometry / sighting from the first pose that sees them (dataset.cpp:540-563)
  for (const Rec& r : recs) {
    if (!states.count(r.i)) states[r.i] = {0.0, 0.0, 0.0};
    const std::vector<double> a = states[r.i];
    const double c = std::cos(a[2]), s = std::sin(a[2]);
    if (r.kind == 0) {
      if (!states.count(r.j)) states[r.j] = {a[0] + c * r.z[0] - s * r.z[1], a[1] + s * r.z[0] + c * r.z[1], a[2] + r.z[2]};
    } else if (!states.count(lbase | r.j)) {
      const double lx = r.z[1] * std::cos(r.z[0]), ly = r.z[1] * std::sin(r.z[0]);
      states[lbase | r.j] = {a[0] + c * lx - s * ly, a[1] + s * lx + c * ly};
    }
  }
  if (states.empty()) return GSX_E_INVALID;
  gsx_dataset* D = new gsx_dataset();
  std::map<uint64_t, int> index;
  for (const auto& kv : states) {
    index[kv.first] = (int)D->var_keys.size();
    D->var_keys.push_back(kv.first);
    const bool pose = kv.second.size() == 3;
    D->var_types.push_back(pose ? GSX_VAR_POSE2 : GSX_VAR_VECTOR);
    D->var_dims.push_back(pose ? 3 : 2);
    D->values.insert(D->values.end(), kv.second.begin(), kv.second.end());
  }
  D->f_key_ptr.push_back(0);
  D->f_meas_ptr.push_back(0);
  D->f_noise_ptr.push_back(0);
  for (const Rec& r : recs) {
    const bool between = r.kind == 0;
    D->f_type.push_back(between ? GSX_F_BETWEEN : GSX_F_BEARINGRANGE);
    D->f_rows.push_back(between ? 3 : 2);
    D->f_vars.push_back(index[r.i]);
    D->f_vars.push_back(index[between ? (uint64_t)r.j : (lbase | r.j)]);
    D->f_key_ptr.push_back((int32_t)D->f_vars.size());
    D->meas.insert(D->meas.end(), r.z, r.z + (between ? 3 : 2));
    D->f_meas_ptr.push_back((int64_t)D->meas.size());
    if (!between) {
      D->f_noise_kind.push_back(GSX_NOISE_DIAGONAL);  // Diagonal::Sigmas(bearing_std, range_std), not smart
      D->noise.push_back(r.v[0]);
      D->noise.push_back(r.v[1]);
    } else if (model_sigmas) {  // "If this is not null, will use instead of parsed model" (dataset.cpp:348-349)
      D->f_noise_kind.push_back(GSX_NOISE_DIAGONAL);
      D->noise.insert(D->noise.end(), model_sigmas, model_sigmas + 3);
    } else if (!emit_noise_2d(r.v, noise_format, smart != 0, kernel, D, &D->err)) {
      std::fprintf(stderr, "gsx_load2d: %s\n", D->err.c_str());
      delete D;
      return GSX_E_INVALID;
    }
    D->f_noise_ptr.push_back((int64_t)D->noise.size());
  }
  *out = D;
  return GSX_OK;
}

static gsx_status read_bal_impl(const char* path, int32_t add_priors, gsx_dataset** out) {
  if (!path || !out) return GSX_E_INVALID;
  std::ifstream in(path);
  if (!in) return GSX_E_INVALID;
  long long nc, np, nobs;
  if (!(in >> nc >> np >> nobs) || nc <= 0 || np <= 0 || nobs < 0) return GSX_E_INVALID;
  {
    // counts a file of this size cannot hold are refused before anything is sized from them (an observation line is at
    // least 8 characters, a camera 18, a point 6)
    const std::streampos here = in.tellg();
    in.seekg(0, std::ios::end);
    const long long bytes = (long long)in.tellg();
    in.seekg(here);
    if (nobs > bytes / 8 || nc > bytes / 18 || np > bytes / 6 || nc > (1LL << 31) - 2 || np > (1LL << 31) - 2) return GSX_E_INVALID;
  }
  std::vector<long long> ci(nobs), pj(nobs);
  std::vector<double> u(nobs), v(nobs);
  for (long long k = 0; k < nobs; ++k) {
    double a, b;
    if (!(in >> ci[k] >> pj[k] >> a >> b)) return GSX_E_INVALID;
    if (ci[k] < 0 || ci[k] >= nc || pj[k] < 0 || pj[k] >= np) return GSX_E_INVALID;
    u[k] = (double)(float)a;   // "float u, v;" in the reference's reader
    v[k] = -(double)(float)b;  // BAL's image y axis points the other way
  }
  gsx_dataset* D = new gsx_dataset();
  const uint64_t pbase = (uint64_t)'p' << 56;
  for (long long i = 0; i < nc; ++i) {
    double c[9];
    for (double& q : c) {
      double t;
      if (!(in >> t)) {
        delete D;
        return GSX_E_INVALID;
      }
      q = (double)(float)t;
    }
    double R[9];
    so3_expmap(c, R);
    // openGL2gtsam: wRc = R' diag(1,-1,-1), t_w = -R' t
    double s[17];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) s[r * 3 + k] = R[k * 3 + r] * (k == 0 ? 1.0 : -1.0);
    for (int r = 0; r < 3; ++r) s[9 + r] = -(R[0 * 3 + r] * c[3] + R[1 * 3 + r] * c[4] + R[2 * 3 + r] * c[5]);
    s[12] = c[6]; s[13] = c[7]; s[14] = c[8]; s[15] = 0.0; s[16] = 0.0;
    D->var_keys.push_back((uint64_t)i);
    D->var_types.push_back(GSX_VAR_CAMERA);
    D->var_dims.push_back(9);
    D->values.insert(D->values.end(), s, s + 17);
  }
  for (long long j = 0; j < np; ++j) {
    double p[3];
    for (double& q : p) {
      double t;
      if (!(in >> t)) {
        delete D;
        return GSX_E_INVALID;
      }
      q = (double)(float)t;
    }
    D->var_keys.push_back(pbase + (uint64_t)j);
    D->var_types.push_back(GSX_VAR_VECTOR);
    D->var_dims.push_back(3);
    D->values.insert(D->values.end(), p, p + 3);
  }
  // factors in track order: observations stably sorted by point (tracks[j].measurements in file order)
  std::vector<long long> order(nobs);
  for (long long k = 0; k < nobs; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](long long a, long long b) { return pj[a] < pj[b]; });
  D->f_key_ptr.push_back(0);
  D->f_meas_ptr.push_back(0);
  D->f_noise_ptr.push_back(0);
  for (long long k : order) {
    D->f_type.push_back(GSX_F_SFM);
    D->f_rows.push_back(2);
    D->f_vars.push_back((int32_t)ci[k]);
    D->f_vars.push_back((int32_t)(nc + pj[k]));
    D->f_key_ptr.push_back((int32_t)D->f_vars.size());
    D->meas.push_back(u[k]);
    D->meas.push_back(v[k]);
    D->f_meas_ptr.push_back((int64_t)D->meas.size());
    D->f_noise_kind.push_back(GSX_NOISE_UNIT);
    D->f_noise_ptr.push_back((int64_t)D->noise.size());
  }
  if (add_priors) {
    for (int which = 0; which < 2; ++which) {
      const int var = which == 0 ? 0 : (int)nc, dim = which == 0 ? 9 : 3, sdim = which == 0 ? 17 : 3;
      const size_t off = which == 0 ? 0 : (size_t)nc * 17;
      D->f_type.push_back(GSX_F_PRIOR);
      D->f_rows.push_back(dim);
      D->f_vars.push_back(var);
      D->f_key_ptr.push_back((int32_t)D->f_vars.size());
      D->meas.insert(D->meas.end(), D->values.begin() + off, D->values.begin() + off + sdim);
      D->f_meas_ptr.push_back((int64_t)D->meas.size());
      D->f_noise_kind.push_back(GSX_NOISE_ISOTROPIC);
      D->noise.push_back(0.1);
      D->f_noise_ptr.push_back((int64_t)D->noise.size());
    }
  }
  *out = D;
  return GSX_OK;
}

gsx_status gsx_dataset_get(const gsx_dataset* D, gsx_problem_desc* desc, const double** values, int64_t* n_values) {
  if (!D || !desc) return GSX_E_INVALID;
  static const double zero = 0.0;
  desc->n_vars = (int32_t)D->var_keys.size();
  desc->var_keys = D->var_keys.data();
  desc->var_types = D->var_types.data();
  desc->var_dims = D->var_dims.data();
  desc->n_factors = (int32_t)D->f_type.size();
  desc->f_type = D->f_type.data();
  desc->f_rows = D->f_rows.data();
  desc->f_key_ptr = D->f_key_ptr.data();
  desc->f_vars = D->f_vars.data();
  desc->f_meas_ptr = D->f_meas_ptr.data();
  desc->meas = D->meas.empty() ? &zero : D->meas.data();
  desc->f_noise_kind = D->f_noise_kind.data();
  desc->f_noise_ptr = D->f_noise_ptr.data();
  desc->noise = D->noise.empty() ? &zero : D->noise.data();
  if (values) *values = D->values.data();
  if (n_values) *n_values = (int64_t)D->values.size();
  return GSX_OK;
}

void gsx_dataset_free(gsx_dataset* D) { delete D; }

// writeG2o — gtsam/slam/dataset.cpp:636-735: VERTEX_SE2 / VERTEX_SE3:QUAT for the Pose2 / Pose3 variables, EDGE_SE2 /
// EDGE_SE3:QUAT for the between factors with the upper triangle of the information matrix (3-D: permuted back to the
// file's (t, R) order).  Priors and other factor types are not part of the format.  17 significant digits: the file
// reads back to the same doubles.
// save2D — gtsam/slam/dataset.cpp:587-617: VERTEX2 lines for the Pose2 values, and for every BetweenFactor<Pose2> an
// EDGE2 line with the keys swapped and the measurement inverted, all with the information R'R of the ONE model handed in,
// in TORO order (the reference does not use the factors' own models either).
static gsx_status save2d_impl(const gsx_problem_desc* d, const double* values, int64_t n_values, const double* model_sigmas,
                      const char* path) {
  if (!d || !values || !path || !model_sigmas) return GSX_E_INVALID;
  std::vector<int64_t> soff(d->n_vars + 1, 0);
  for (int v = 0; v < d->n_vars; ++v) {
    const int t = d->var_types[v];
    soff[v + 1] = soff[v] + (t == GSX_VAR_POSE2 ? 3 : (t == GSX_VAR_POSE3 ? 12 : (t == GSX_VAR_CAMERA ? 17 : d->var_dims[v])));
  }
  if (soff[d->n_vars] != n_values) return GSX_E_INVALID;
  FILE* fh = std::fopen(path, "w");
  if (!fh) return GSX_E_INVALID;
  for (int v = 0; v < d->n_vars; ++v)
    if (d->var_types[v] == GSX_VAR_POSE2) {
      const double* s = values + soff[v];
      std::fprintf(fh, "VERTEX2 %llu %.17g %.17g %.17g\n", (unsigned long long)d->var_keys[v], s[0], s[1],
                   std::atan2(std::sin(s[2]), std::cos(s[2])));
    }
  const double i0 = 1.0 / (model_sigmas[0] * model_sigmas[0]), i1 = 1.0 / (model_sigmas[1] * model_sigmas[1]),
               i2 = 1.0 / (model_sigmas[2] * model_sigmas[2]);
  for (int f = 0; f < d->n_factors; ++f) {
    if (d->f_type[f] != GSX_F_BETWEEN || d->f_rows[f] != 3) continue;
    const int a = d->f_vars[d->f_key_ptr[f]], b = d->f_vars[d->f_key_ptr[f] + 1];
    if (d->var_types[a] != GSX_VAR_POSE2) continue;
    const double* z = d->meas + d->f_meas_ptr[f];
    const double c = std::cos(z[2]), s = std::sin(z[2]);
    // Pose2::inverse: (-R' t, -theta)
    std::fprintf(fh, "EDGE2 %llu %llu %.17g %.17g %.17g %.17g 0 %.17g %.17g 0 0\n", (unsigned long long)d->var_keys[b],
                 (unsigned long long)d->var_keys[a], -(c * z[0] + s * z[1]), -(-s * z[0] + c * z[1]),
                 std::atan2(-s, c), i0, i1, i2);
  }
  std::fclose(fh);
  return GSX_OK;
}

// writeBAL / writeBALfromValues — gtsam/sfm/SfmData.cpp:249-377: the cameras and points of `values` and the
// observations of the GeneralSFMFactors, grouped by point in file order of the factors; pose back to the OpenGL
// convention (gtsam2openGL, :88-99), rotation as its Rodrigues vector, measurement (u, -v).  17 significant digits
// (the reference writes its stream's default 6).
static gsx_status write_bal_impl(const gsx_problem_desc* d, const double* values, int64_t n_values, const char* path) {
  if (!d || !values || !path) return GSX_E_INVALID;
  std::vector<int64_t> soff(d->n_vars + 1, 0);
  std::vector<int> cam_id(d->n_vars, -1), pt_id(d->n_vars, -1);
  int nc = 0, np = 0;
  for (int v = 0; v < d->n_vars; ++v) {
    const int t = d->var_types[v];
    soff[v + 1] = soff[v] + (t == GSX_VAR_POSE2 ? 3 : (t == GSX_VAR_POSE3 ? 12 : (t == GSX_VAR_CAMERA ? 17 : d->var_dims[v])));
    if (t == GSX_VAR_CAMERA) cam_id[v] = nc++;
    else if (t == GSX_VAR_VECTOR && d->var_dims[v] == 3) pt_id[v] = np++;
  }
  if (soff[d->n_vars] != n_values || nc == 0 || np == 0) return GSX_E_INVALID;
  std::vector<std::vector<int>> track(np);  // point -> SFM factors, in factor order
  long long nobs = 0;
  for (int f = 0; f < d->n_factors; ++f) {
    if (d->f_type[f] != GSX_F_SFM) continue;
    const int p = pt_id[d->f_vars[d->f_key_ptr[f] + 1]];
    if (p < 0 || cam_id[d->f_vars[d->f_key_ptr[f]]] < 0) return GSX_E_INVALID;
    track[p].push_back(f);
    ++nobs;
  }
  FILE* fh = std::fopen(path, "w");
  if (!fh) return GSX_E_INVALID;
  std::fprintf(fh, "%d %d %lld\n\n", nc, np, nobs);
  for (int p = 0; p < np; ++p)
    for (int f : track[p]) {
      const double* z = d->meas + d->f_meas_ptr[f];
      std::fprintf(fh, "%d %d %.17g %.17g\n", cam_id[d->f_vars[d->f_key_ptr[f]]], p, z[0], -z[1]);
    }
  std::fprintf(fh, "\n");
  for (int v = 0; v < d->n_vars; ++v) {
    if (cam_id[v] < 0) continue;
    const double* s = values + soff[v];
    // cRw_openGL = R90 * wRc' with R90 = diag(1, -1, -1); t_openGL = cRw_openGL * (-t)
    double R[9], t[3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = (r == 0 ? 1.0 : -1.0) * s[c * 3 + r];
    for (int r = 0; r < 3; ++r) t[r] = -(R[r * 3] * s[9] + R[r * 3 + 1] * s[10] + R[r * 3 + 2] * s[11]);
    double w[3];
    so3_logmap(R, w);
    std::fprintf(fh, "%.17g\n%.17g\n%.17g\n%.17g\n%.17g\n%.17g\n%.17g\n%.17g\n%.17g\n\n", w[0], w[1], w[2], t[0], t[1], t[2],
                 s[12], s[13], s[14]);
  }
  for (int v = 0; v < d->n_vars; ++v) {
    if (pt_id[v] < 0) continue;
    const double* s = values + soff[v];
    std::fprintf(fh, "%.17g\n%.17g\n%.17g\n\n", s[0], s[1], s[2]);
  }
  std::fclose(fh);
  return GSX_OK;
}

static gsx_status write_g2o_impl(const gsx_problem_desc* d, const double* values, int64_t n_values, const char* path) {
  if (!d || !values || !path) return GSX_E_INVALID;
  std::vector<int64_t> soff(d->n_vars + 1, 0);
  for (int v = 0; v < d->n_vars; ++v) {
    const int t = d->var_types[v];
    const int sd = t == GSX_VAR_POSE2 ? 3 : (t == GSX_VAR_POSE3 ? 12 : (t == GSX_VAR_CAMERA ? 17 : d->var_dims[v]));
    soff[v + 1] = soff[v] + sd;
  }
  if (soff[d->n_vars] != n_values) return GSX_E_INVALID;
  FILE* fh = std::fopen(path, "w");
  if (!fh) return GSX_E_INVALID;
  auto quat = [](const double* R, double* q) {  // q = (w, x, y, z); R row-major
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      const double s = std::sqrt(tr + 1.0) * 2;
      q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
      return;
    }
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(1.0 + R[i * 4] - R[j * 4] - R[k * 4]) * 2;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) / s;
    q[1 + i] = 0.25 * s;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) / s;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) / s;
  };
  for (int v = 0; v < d->n_vars; ++v) {
    const double* s = values + soff[v];
    if (d->var_types[v] == GSX_VAR_POSE2) {
      std::fprintf(fh, "VERTEX_SE2 %llu %.17g %.17g %.17g\n", (unsigned long long)d->var_keys[v], s[0], s[1], s[2]);
    } else if (d->var_types[v] == GSX_VAR_POSE3) {
      double q[4];
      quat(s, q);
      std::fprintf(fh, "VERTEX_SE3:QUAT %llu %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n",
                   (unsigned long long)d->var_keys[v], s[9], s[10], s[11], q[1], q[2], q[3], q[0]);
    }
  }
  for (int f = 0; f < d->n_factors; ++f) {
    if (d->f_type[f] != GSX_F_BETWEEN) continue;
    const int dd = d->f_rows[f];
    if (dd != 3 && dd != 6) continue;
    const int a = d->f_vars[d->f_key_ptr[f]], b = d->f_vars[d->f_key_ptr[f] + 1];
    const double* z = d->meas + d->f_meas_ptr[f];
    const double* nz = d->noise + d->f_noise_ptr[f];
    double info[36] = {0};
    for (int r = 0; r < dd; ++r)
      for (int c = 0; c < dd; ++c) {
        double x = 0;
        if (d->f_noise_kind[f] == GSX_NOISE_GAUSSIAN) {
          for (int k = 0; k < dd; ++k) x += nz[k * dd + r] * nz[k * dd + c];  // R'R
        } else if (r == c) {
          x = d->f_noise_kind[f] == GSX_NOISE_DIAGONAL ? 1.0 / (nz[r] * nz[r])
              : (d->f_noise_kind[f] == GSX_NOISE_ISOTROPIC ? 1.0 / (nz[0] * nz[0]) : 1.0);
        }
        info[r * dd + c] = x;
      }
    if (dd == 3) {
      std::fprintf(fh, "EDGE_SE2 %llu %llu %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n",
                   (unsigned long long)d->var_keys[a], (unsigned long long)d->var_keys[b], z[0], z[1], z[2], info[0],
                   info[1], info[2], info[4], info[5], info[8]);
    } else {
      double q[4], m[36];
      quat(z, q);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {  // GTSAM's (R, t) order back to the file's (t, R)
          m[r * 6 + c] = info[(3 + r) * 6 + (3 + c)];
          m[(3 + r) * 6 + (3 + c)] = info[r * 6 + c];
          m[r * 6 + (3 + c)] = info[(3 + r) * 6 + c];
          m[(3 + r) * 6 + c] = info[r * 6 + (3 + c)];
        }
      std::fprintf(fh, "EDGE_SE3:QUAT %llu %llu %.17g %.17g %.17g %.17g %.17g %.17g %.17g",
                   (unsigned long long)d->var_keys[a], (unsigned long long)d->var_keys[b], z[9], z[10], z[11], q[1], q[2],
                   q[3], q[0]);
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) std::fprintf(fh, " %.17g", m[r * 6 + c]);
      std::fprintf(fh, "\n");
    }
  }
  std::fclose(fh);
  return GSX_OK;
}


// No exception crosses the C boundary (include/gsx.h): a malformed or hostile file (absurd counts in a header, a
// truncated record) ends as GSX_E_INVALID / GSX_E_NOMEM, never as std::terminate in the caller's process.
#define GSX_NO_THROW(call)                  \
  try {                                     \
    return call;                            \
  } catch (const std::bad_alloc&) {         \
    return GSX_E_NOMEM;                     \
  } catch (...) {                           \
    return GSX_E_INVALID;                   \
  }
gsx_status gsx_read_g2o(const char* path, int32_t is3d, gsx_dataset** out) {
  GSX_NO_THROW(read_g2o_impl(path, is3d, out))
}

gsx_status gsx_load2d(const char* path, const double* model_sigmas, int64_t max_index, int32_t smart, int32_t noise_format, int32_t kernel, gsx_dataset** out) {
  GSX_NO_THROW(load2d_impl(path, model_sigmas, max_index, smart, noise_format, kernel, out))
}

gsx_status gsx_read_bal(const char* path, int32_t add_priors, gsx_dataset** out) {
  GSX_NO_THROW(read_bal_impl(path, add_priors, out))
}

gsx_status gsx_save2d(const gsx_problem_desc* d, const double* values, int64_t n_values, const double* model_sigmas, const char* path) {
  GSX_NO_THROW(save2d_impl(d, values, n_values, model_sigmas, path))
}

gsx_status gsx_write_bal(const gsx_problem_desc* d, const double* values, int64_t n_values, const char* path) {
  GSX_NO_THROW(write_bal_impl(d, values, n_values, path))
}

gsx_status gsx_write_g2o(const gsx_problem_desc* d, const double* values, int64_t n_values, const char* path) {
  GSX_NO_THROW(write_g2o_impl(d, values, n_values, path))
}

}  // extern "C"
