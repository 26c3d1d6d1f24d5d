"""Host-side mirror of the reference's interface for the hot path.

Same names, argument meaning and error behaviour as the reference classes so
that the parity tests read like the reference's own tests
(tests/testNonlinearOptimizer.cpp, tests/testGeneralSFMFactorB.cpp, ...):

  NonlinearFactorGraph / Values / Ordering          gtsam/nonlinear, gtsam/inference
  PriorFactor / BetweenFactor / GeneralSFMFactor    gtsam/nonlinear/PriorFactor.h,
                                                    gtsam/slam/BetweenFactor.h,
                                                    gtsam/slam/GeneralSFMFactor.h
  noiseModel.{Unit,Isotropic,Diagonal,Gaussian}     gtsam/linear/NoiseModel.cpp
  LevenbergMarquardtParams / ...Optimizer           gtsam/nonlinear/LevenbergMarquardt*.h
  GaussNewtonOptimizer                              gtsam/nonlinear/GaussNewtonOptimizer.cpp
  GaussianFactorGraph / JacobianFactor              gtsam/linear

These classes only *describe* a problem (they lower it to the flat arrays of
include/gsx.h); every number on the hot path is computed behind the C-ABI.
The backend is pluggable only so that tests can push the identical description
through the CPU oracle; the default is the HIP library and it fails loudly when
that is missing.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import _abi as A

Key = int


# ---- Symbol (gtsam/inference/Symbol.cpp:29-47) --------------------------------
def symbol(c: str, j: int) -> Key:
    return (ord(c) << 56) | int(j)


def X(j):
    return symbol("x", j)


def L(j):
    return symbol("l", j)


def P(j):
    return symbol("p", j)


def C(j):
    return symbol("c", j)


# ---- geometry value types (host-side construction of Values only) -----------------
class Pose2:
    type_code = A.VAR_POSE2
    dim = 3

    def __init__(self, x=0.0, y=0.0, theta=0.0):
        self.x_, self.y_, self.theta_ = float(x), float(y), float(theta)

    def state(self):
        return np.array([self.x_, self.y_, self.theta_])

    @staticmethod
    def from_state(s):
        return Pose2(s[0], s[1], s[2])

    def x(self):
        return self.x_

    def y(self):
        return self.y_

    def theta(self):
        return self.theta_

    def equals(self, o, tol=1e-9):
        dth = math.atan2(math.sin(self.theta_ - o.theta_), math.cos(self.theta_ - o.theta_))
        return abs(self.x_ - o.x_) < tol and abs(self.y_ - o.y_) < tol and abs(dth) < tol

    def __repr__(self):
        return f"Pose2({self.x_}, {self.y_}, {self.theta_})"


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


class Rot3:
    def __init__(self, R=None):
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=float).reshape(3, 3)

    @staticmethod
    def Rodrigues(wx, wy=None, wz=None):
        """Rot3::Rodrigues -> SO3::Expmap (gtsam/geometry/SO3.cpp:61-96)."""
        w = np.array([wx, wy, wz], dtype=float) if wy is not None else np.asarray(wx, dtype=float)
        th2 = float(w @ w)
        W = _skew(w)
        if th2 <= np.finfo(float).eps:
            a, b = 1.0 - th2 / 6.0, 0.5 - th2 / 24.0
        else:
            th = math.sqrt(th2)
            a = math.sin(th) / th
            s2 = math.sin(th / 2.0)
            b = 2.0 * s2 * s2 / th2
        return Rot3(np.eye(3) + a * W + b * (W @ W))

    Expmap = Rodrigues

    @staticmethod
    def Quaternion(w, x, y, z):
        """Rot3::Quaternion(w,x,y,z) (Eigen quaternion -> matrix)."""
        n = math.sqrt(w * w + x * x + y * y + z * z)
        w, x, y, z = w / n, x / n, y / n, z / n
        return Rot3([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    @staticmethod
    def RzRyRx(x, y, z):
        """Rot3::RzRyRx(x,y,z) = Rz(z) Ry(y) Rx(x) (gtsam/geometry/Rot3M.cpp)."""
        cx, sx, cy, sy, cz, sz = math.cos(x), math.sin(x), math.cos(y), math.sin(y), math.cos(z), math.sin(z)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        return Rot3(Rz @ Ry @ Rx)

    Ypr = staticmethod(lambda y, p, r: Rot3.RzRyRx(r, p, y))

    def matrix(self):
        return self.R

    def inverse(self):
        return Rot3(self.R.T)

    def compose(self, o):
        return Rot3(self.R @ o.R)


class Pose3:
    type_code = A.VAR_POSE3
    dim = 6

    def __init__(self, R: Optional[Rot3] = None, t=None):
        self.R_ = R if R is not None else Rot3()
        self.t_ = np.zeros(3) if t is None else np.asarray(t, dtype=float).reshape(3)

    def state(self):
        return np.concatenate([self.R_.R.reshape(9), self.t_])

    @staticmethod
    def from_state(s):
        return Pose3(Rot3(np.asarray(s[:9]).reshape(3, 3)), s[9:12])

    def rotation(self):
        return self.R_

    def translation(self):
        return self.t_

    def compose(self, o):
        return Pose3(Rot3(self.R_.R @ o.R_.R), self.t_ + self.R_.R @ o.t_)

    def inverse(self):
        return Pose3(Rot3(self.R_.R.T), -self.R_.R.T @ self.t_)

    def between(self, o):
        return self.inverse().compose(o)

    def equals(self, o, tol=1e-9):
        return np.allclose(self.R_.R, o.R_.R, atol=tol) and np.allclose(self.t_, o.t_, atol=tol)


class Cal3Bundler:
    def __init__(self, f=1.0, k1=0.0, k2=0.0, u0=0.0, v0=0.0):
        self.f, self.k1, self.k2, self.u0, self.v0 = map(float, (f, k1, k2, u0, v0))

    def vector(self):
        return np.array([self.f, self.k1, self.k2, self.u0, self.v0])


class PinholeCameraCal3Bundler:
    """SfmCamera = PinholeCamera<Cal3Bundler> (gtsam/geometry/PinholeCamera.h)."""
    type_code = A.VAR_CAMERA
    dim = 9

    def __init__(self, pose: Pose3, K: Cal3Bundler):
        self.pose_, self.K_ = pose, K

    def state(self):
        return np.concatenate([self.pose_.state(), self.K_.vector()])

    @staticmethod
    def from_state(s):
        return PinholeCameraCal3Bundler(Pose3.from_state(s[:12]), Cal3Bundler(*s[12:17]))

    def pose(self):
        return self.pose_

    def calibration(self):
        return self.K_


class _Vector:
    type_code = A.VAR_VECTOR

    def __init__(self, v):
        self.v = np.atleast_1d(np.asarray(v, dtype=float))
        self.dim = int(self.v.size)

    def state(self):
        return self.v


def Point2(x, y):
    return np.array([x, y], dtype=float)


def Point3(x, y, z):
    return np.array([x, y, z], dtype=float)


def _wrap_value(v):
    return v if hasattr(v, "type_code") else _Vector(v)


def _unwrap(type_code, s):
    if type_code == A.VAR_POSE2:
        return Pose2.from_state(s)
    if type_code == A.VAR_POSE3:
        return Pose3.from_state(s)
    if type_code == A.VAR_CAMERA:
        return PinholeCameraCal3Bundler.from_state(s)
    return np.array(s)


# ---- noise models (gtsam/linear/NoiseModel.cpp) -------------------------------------
class _Noise:
    def __init__(self, kind, dim, params=()):
        self.kind, self.dim_ = kind, int(dim)
        self.params = np.asarray(params, dtype=float).reshape(-1)

    def dim(self):
        return self.dim_

    def isUnit(self):
        return self.kind == A.NOISE_UNIT


class noiseModel:
    class Unit:
        @staticmethod
        def Create(dim):
            return _Noise(A.NOISE_UNIT, dim)

    class Isotropic:
        @staticmethod
        def Sigma(dim, sigma, smart=True):
            # NoiseModel.cpp:625-634: sigma == 1 -> Unit
            if smart and abs(sigma - 1.0) < 1e-9:
                return noiseModel.Unit.Create(dim)
            return _Noise(A.NOISE_ISOTROPIC, dim, [sigma])

        @staticmethod
        def Variance(dim, variance, smart=True):
            return noiseModel.Isotropic.Sigma(dim, math.sqrt(variance), smart)

        @staticmethod
        def Precision(dim, precision, smart=True):
            return noiseModel.Isotropic.Sigma(dim, 1.0 / math.sqrt(precision), smart)

    class Diagonal:
        @staticmethod
        def Sigmas(sigmas, smart=True):
            s = np.asarray(sigmas, dtype=float).reshape(-1)
            # NoiseModel.cpp:292-309: a (near-)zero sigma makes the model Constrained; all sigmas equal -> Isotropic
            if smart and s.size > 0 and np.any(s < 1e-8):
                return noiseModel.Constrained.MixedSigmas(s)
            if smart and s.size > 0 and np.all(np.abs(s - s[0]) < 1e-9):
                return noiseModel.Isotropic.Sigma(s.size, float(s[0]), True)
            return _Noise(A.NOISE_DIAGONAL, s.size, s)

        @staticmethod
        def Variances(variances, smart=True):
            return noiseModel.Diagonal.Sigmas(np.sqrt(np.asarray(variances, dtype=float)), smart)

        @staticmethod
        def Precisions(precisions, smart=True):
            return noiseModel.Diagonal.Sigmas(1.0 / np.sqrt(np.asarray(precisions, dtype=float)), smart)

    class Constrained:
        """noiseModel::Constrained (gtsam/linear/NoiseModel.h:389-500): rows with sigma == 0 are hard constraints (eliminated
        by constraint pivots instead of Cholesky, NoiseModel.cpp:503-620); mu weighs their violation in the error functions."""

        @staticmethod
        def MixedSigmas(*args):
            if len(args) == 1:
                sigmas = np.asarray(args[0], dtype=float).reshape(-1)
                mu = np.full(sigmas.size, 1000.0)
            else:
                sigmas = np.asarray(args[1], dtype=float).reshape(-1)
                mu = np.broadcast_to(np.asarray(args[0], dtype=float), sigmas.shape).astype(float)
            return _Noise(A.NOISE_CONSTRAINED, sigmas.size, np.concatenate([sigmas, mu]))

        @staticmethod
        def All(dim, mu=1000.0):
            return noiseModel.Constrained.MixedSigmas(mu, np.zeros(int(dim)))

    class Gaussian:
        @staticmethod
        def SqrtInformation(R, smart=True):
            R = np.asarray(R, dtype=float)
            m = R.shape[0]
            if smart and np.count_nonzero(R - np.diag(np.diagonal(R))) == 0:
                return noiseModel.Diagonal.Sigmas(1.0 / np.diagonal(R), True)  # NoiseModel.cpp:84-96
            return _Noise(A.NOISE_GAUSSIAN, m, np.triu(R).reshape(-1))

        @staticmethod
        def Information(M, smart=True):
            M = np.asarray(M, dtype=float)
            if smart and np.count_nonzero(M - np.diag(np.diagonal(M))) == 0:
                return noiseModel.Diagonal.Precisions(np.diagonal(M), True)  # NoiseModel.cpp:98-112
            Lc = np.linalg.cholesky(M)  # LLT(information).matrixU() = L'
            return _Noise(A.NOISE_GAUSSIAN, M.shape[0], Lc.T.reshape(-1))

        @staticmethod
        def Covariance(S, smart=True):
            return noiseModel.Gaussian.Information(np.linalg.inv(np.asarray(S, dtype=float)), smart)

    class mEstimator:
        """gtsam/linear/LossFunctions.h: robust M-estimators (Block reweighting)."""

        class _M:
            def __init__(self, code, k):
                self.code, self.k = code, float(k)

        class Huber:
            @staticmethod
            def Create(k=1.345):
                return noiseModel.mEstimator._M(A.NOISE_ROBUST_HUBER, k)

        class Tukey:
            @staticmethod
            def Create(c=4.6851):
                return noiseModel.mEstimator._M(A.NOISE_ROBUST_TUKEY, c)

        class Cauchy:
            @staticmethod
            def Create(k=0.1):
                return noiseModel.mEstimator._M(A.NOISE_ROBUST_CAUCHY, k)

    class Robust:
        @staticmethod
        def Create(robust, noise):
            """noiseModel::Robust::Create(mEstimator, baseNoise) — gtsam/linear/NoiseModel.cpp:740-743."""
            return _Noise(noise.kind | robust.code, noise.dim(), np.concatenate([noise.params, [robust.k]]))


# ---- factors ---------------------------------------------------------------------------
class _Factor:
    def __init__(self, ftype, keys, rows, meas, noise: Optional[_Noise]):
        self.ftype, self.keys_, self.rows = ftype, [int(k) for k in keys], int(rows)
        self.meas = np.asarray(meas, dtype=float).reshape(-1)
        self.noise = noise if noise is not None else noiseModel.Unit.Create(rows)
        if self.noise.dim() != self.rows:
            raise ValueError("noise model dimension does not match factor dimension")

    def keys(self):
        return list(self.keys_)


def PriorFactor(key, prior, model=None):
    """PriorFactor<T>(key, prior, model) — gtsam/nonlinear/PriorFactor.h."""
    v = _wrap_value(prior)
    f = _Factor(A.F_PRIOR, [key], v.dim, v.state(), model)
    f.value_type = v.type_code
    return f


def BetweenFactor(key1, key2, measured, model=None):
    """BetweenFactor<T>(key1, key2, measured, model) — gtsam/slam/BetweenFactor.h."""
    v = _wrap_value(measured)
    if v.type_code == A.VAR_CAMERA:
        raise ValueError("BetweenFactor<camera> is not on the supported path")
    f = _Factor(A.F_BETWEEN, [key1, key2], v.dim, v.state(), model)
    f.value_type = v.type_code
    return f


BetweenFactorPose2 = BetweenFactorPose3 = BetweenFactorPoint2 = BetweenFactorPoint3 = BetweenFactor
PriorFactorPose2 = PriorFactorPose3 = PriorFactorPoint2 = PriorFactorPoint3 = PriorFactor


def GeneralSFMFactor(measured, model, cameraKey, landmarkKey):
    """GeneralSFMFactor<SfmCamera,Point3>(measured, model, cameraKey, landmarkKey)."""
    return _Factor(A.F_SFM, [cameraKey, landmarkKey], 2, measured, model)


GeneralSFMFactorCal3Bundler = GeneralSFMFactor


class Cal3_S2:
    """gtsam/geometry/Cal3_S2.h: (fx, fy, s, u0, v0), or (fov degrees, w, h) — Cal3_S2.cpp:28-41."""

    def __init__(self, *args):
        if len(args) == 3:
            fov, w, h = args
            a = fov * math.pi / 360.0  # fov/2 in radians
            f = w / (2.0 * math.tan(a))
            self.v = np.array([f, f, 0.0, w / 2.0, h / 2.0])
        elif len(args) == 5:
            self.v = np.asarray(args, dtype=float)
        else:
            self.v = np.array([1.0, 1.0, 0.0, 0.0, 0.0])

    def fx(self):
        return float(self.v[0])

    def vector(self):
        return self.v.copy()


def GenericProjectionFactor(measured, model, poseKey, pointKey, K: "Cal3_S2"):
    """GenericProjectionFactor<Pose3, Point3, Cal3_S2>(measured, model, poseKey, pointKey, K) —
    gtsam/slam/ProjectionFactor.h (no body_P_sensor, default cheirality flags)."""
    meas = np.concatenate([np.asarray(measured, dtype=float).reshape(2), K.vector()])
    return _Factor(A.F_PROJECTION, [poseKey, pointKey], 2, meas, model)


GenericProjectionFactorCal3_S2 = GenericProjectionFactor


def BearingRangeFactor(poseKey, pointKey, bearing, range_, model):
    """BearingRangeFactor<Pose2, Point2>(poseKey, pointKey, Rot2 bearing, double range, model) — gtsam/sam/
    BearingRangeFactor.h; `bearing` is the angle in radians (Rot2::fromAngle)."""
    return _Factor(A.F_BEARINGRANGE, [poseKey, pointKey], 2, [float(bearing), float(range_)], model)


BearingRangeFactor2D = BearingRangeFactor


def JacobianFactor(*args):
    """JacobianFactor(key1, A1, [key2, A2, ...], b[, model]) — gtsam/linear/JacobianFactor.h.
    A diagonal/isotropic model is folded in by the backend (whitening)."""
    args = list(args)
    model = args.pop() if isinstance(args[-1], _Noise) else None
    b = np.asarray(args.pop(), dtype=float).reshape(-1)
    keys, blocks = [], []
    for i in range(0, len(args), 2):
        keys.append(int(args[i]))
        blocks.append(np.asarray(args[i + 1], dtype=float).reshape(b.size, -1))
    Ab = np.concatenate(blocks + [b.reshape(-1, 1)], axis=1)
    f = _Factor(A.F_LINEAR, keys, b.size, Ab.reshape(-1, order="F"), model)
    f.block_dims = [blk.shape[1] for blk in blocks]
    return f


# ---- containers ----------------------------------------------------------------------------
class Values:
    def __init__(self, other: Optional["Values"] = None):
        self._v: Dict[Key, object] = dict(other._v) if other is not None else {}

    def insert(self, key, value):
        if int(key) in self._v:
            raise KeyError(f"ValuesKeyAlreadyExists: {key}")
        self._v[int(key)] = _wrap_value(value)

    def update(self, key, value):
        self._v[int(key)] = _wrap_value(value)

    def exists(self, key):
        return int(key) in self._v

    def keys(self):
        return sorted(self._v)

    def size(self):
        return len(self._v)

    def at(self, key):
        if int(key) not in self._v:
            raise KeyError(f"ValuesKeyDoesNotExist: {key}")
        v = self._v[int(key)]
        return v.v.copy() if isinstance(v, _Vector) else v

    atPose2 = atPose3 = atPoint2 = atPoint3 = atVector = at

    def dims(self):
        return {k: self._v[k].dim for k in self.keys()}

    def pack(self):
        ks = self.keys()
        return (np.concatenate([self._v[k].state() for k in ks]) if ks else np.zeros(0))

    @staticmethod
    def unpack(keys, types, dims, packed):
        out, off = Values(), 0
        for k, t, d in zip(keys, types, dims):
            sd = A.STATE_DIM.get(int(t), int(d))
            out._v[int(k)] = _wrap_value(_unwrap(int(t), packed[off:off + sd]))
            off += sd
        return out


class Ordering(list):
    """gtsam/inference/Ordering.h — a list of keys in elimination order."""

    def push_back(self, k):
        self.append(int(k))


class NonlinearFactorGraph:
    def __init__(self):
        self.factors: List[Optional[_Factor]] = []

    def add(self, f):
        self.factors.append(f)

    push_back = emplace_shared = add

    def addPrior(self, key, prior, model=None):
        self.add(PriorFactor(key, prior, model))

    def size(self):
        return len(self.factors)

    def keys(self):
        return sorted({k for f in self.factors if f is not None for k in f.keys_})

    # -- lowering to include/gsx.h arrays -----------------------------------------
    def to_arrays(self, values: Optional[Values] = None, var_dims: Optional[Dict[Key, int]] = None) -> A.ProblemArrays:
        facs = [f for f in self.factors if f is not None]  # null factors are skipped (NonlinearFactorGraph.cpp:239-278)
        if values is not None:
            keys = values.keys()
            types = [values._v[k].type_code for k in keys]
            dims = [values._v[k].dim for k in keys]
        else:  # linear graph: dims from the Jacobian blocks
            dd: Dict[Key, int] = dict(var_dims or {})
            for f in facs:
                for k, d in zip(f.keys_, getattr(f, "block_dims", [])):
                    dd[k] = d
            keys = sorted(dd)
            types = [A.VAR_VECTOR] * len(keys)
            dims = [dd[k] for k in keys]
        index = {k: i for i, k in enumerate(keys)}
        key_ptr, meas_ptr, noise_ptr = [0], [0], [0]
        fvars: List[int] = []
        meas: List[np.ndarray] = []
        noise: List[np.ndarray] = []
        for f in facs:
            for k in f.keys_:
                if k not in index:
                    raise KeyError(f"ValuesKeyDoesNotExist: {k}")
                fvars.append(index[k])
            key_ptr.append(len(fvars))
            meas.append(f.meas)
            meas_ptr.append(meas_ptr[-1] + f.meas.size)
            noise.append(f.noise.params)
            noise_ptr.append(noise_ptr[-1] + f.noise.params.size)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0)
        return A.ProblemArrays(
            var_keys=np.array(keys, dtype=np.uint64), var_types=types, var_dims=dims,
            f_type=[f.ftype for f in facs], f_rows=[f.rows for f in facs],
            f_key_ptr=key_ptr, f_vars=fvars, f_meas_ptr=meas_ptr, meas=cat(meas),
            f_noise_kind=[f.noise.kind for f in facs], f_noise_ptr=noise_ptr, noise=cat(noise),
            values=values.pack() if values is not None else None)

    def error(self, values: Values, backend_factory=None) -> float:
        be = _make_backend(self.to_arrays(values), backend_factory)
        try:
            return be.error()
        finally:
            be.close()


class GaussianFactorGraph(NonlinearFactorGraph):
    """Linear graph of JacobianFactors; optimize() = the NonlinearOptimizer::solve seam."""

    def optimize(self, ordering: Optional[Sequence[Key]] = None, backend_factory=None) -> Dict[Key, np.ndarray]:
        arrays = self.to_arrays(None)
        arrays.values = np.zeros(int(arrays.var_dims.sum()))
        be = _make_backend(arrays, backend_factory)
        try:
            be.set_ordering(ordering if ordering is not None else be.compute_ordering(A.ORDER_MINDEGREE))
            be.linearize()
            delta = be.solve(0.0)
        finally:
            be.close()
        off = arrays.tangent_offsets()
        return {int(k): delta[off[i]:off[i + 1]] for i, k in enumerate(arrays.var_keys)}


def _make_backend(arrays, backend_factory):
    if backend_factory is None:
        from ._lib import product_backend
        backend_factory = product_backend
    return backend_factory(arrays)


# ---- optimizers ---------------------------------------------------------------------------------
class LevenbergMarquardtParams:
    """gtsam/nonlinear/LevenbergMarquardtParams.h:61-98 + NonlinearOptimizerParams.h:42-108."""
    SILENT, SUMMARY = 0, 1

    def __init__(self):
        self._set(A.lm_params_legacy())
        # the class default lambdaInitial/lambdaFactor etc. ARE the legacy values;
        self.ordering: Optional[Ordering] = None
        self.orderingType = "COLAMD"

    def _set(self, p: A.LMParams):
        self.maxIterations, self.relativeErrorTol = p.max_iterations, p.relative_error_tol
        self.absoluteErrorTol, self.errorTol = p.absolute_error_tol, p.error_tol
        self.lambdaInitial, self.lambdaFactor = p.lambda_initial, p.lambda_factor
        self.lambdaUpperBound, self.lambdaLowerBound = p.lambda_upper_bound, p.lambda_lower_bound
        self.minModelFidelity = p.min_model_fidelity
        self.diagonalDamping = bool(p.diagonal_damping)
        self.useFixedLambdaFactor = bool(p.use_fixed_lambda_factor)
        self.minDiagonal, self.maxDiagonal = p.min_diagonal, p.max_diagonal
        self.verbosityLM = p.verbosity

    @staticmethod
    def LegacyDefaults():
        return LevenbergMarquardtParams()

    @staticmethod
    def CeresDefaults():
        p = LevenbergMarquardtParams()
        p._set(A.lm_params_ceres())
        return p

    def c_params(self) -> A.LMParams:
        return A.LMParams(self.maxIterations, self.relativeErrorTol, self.absoluteErrorTol, self.errorTol,
                          self.lambdaInitial, self.lambdaFactor, self.lambdaUpperBound, self.lambdaLowerBound,
                          self.minModelFidelity, int(self.diagonalDamping), int(self.useFixedLambdaFactor),
                          self.minDiagonal, self.maxDiagonal, int(self.verbosityLM))


class _OptimizerBase:
    def __init__(self, graph, initialValues, ordering, orderingType, backend_factory, ordering_fn):
        self.graph_ = graph
        self.arrays = graph.to_arrays(initialValues)
        self.backend = _make_backend(self.arrays, backend_factory)
        if ordering is None:
            # EnsureHasOrdering (LevenbergMarquardtParams.h:112-117): the reference calls
            # Ordering::Create(orderingType, graph) here.  A caller-supplied ordering_fn lets the
            # tests feed the reference's own CCOLAMD result; otherwise the library's own ordering.
            if ordering_fn is not None:
                ordering = ordering_fn(self.arrays)
            else:
                kind = {"COLAMD": A.ORDER_MINDEGREE, "METIS": A.ORDER_ND, "NATURAL": A.ORDER_NATURAL,
                        "SCHUR": A.ORDER_SCHUR, "SCHUR_ND": A.ORDER_SCHUR_ND}[orderingType]
                ordering = self.backend.compute_ordering(kind)
        self.ordering = Ordering(int(k) for k in ordering)
        self.backend.set_ordering(self.ordering)
        self.result = None

    def values(self) -> Values:
        return Values.unpack(self.arrays.var_keys, self.arrays.var_types, self.arrays.var_dims,
                             self.backend.get_values())

    def error(self) -> float:
        return self.backend.error()

    def iterations(self) -> int:
        return self.result["iterations"] if self.result else self._iterations


class LevenbergMarquardtOptimizer(_OptimizerBase):
    """LevenbergMarquardtOptimizer(graph, initialValues[, ordering][, params])."""

    def __init__(self, graph, initialValues, *args, backend_factory=None, ordering_fn=None):
        params, ordering = LevenbergMarquardtParams(), None
        for a in args:
            if isinstance(a, LevenbergMarquardtParams):
                params = a
            elif a is not None:
                ordering = a
        if ordering is None:
            ordering = params.ordering
        self.params_ = params
        super().__init__(graph, initialValues, ordering, params.orderingType, backend_factory, ordering_fn)
        self.backend.lm_reset(params.c_params())
        self._iterations = 0
        self._lambda = params.lambdaInitial

    def optimize(self) -> Values:
        self.result = self.backend.lm_optimize(self.params_.c_params())
        self._lambda = self.result["final_lambda"]
        return self.values()

    def iterate(self):
        err, lam = self.backend.lm_iterate(self.params_.c_params())
        self._iterations += 1
        self._lambda = lam
        return err

    def lambda_(self) -> float:
        return self._lambda


class DoglegOptimizer(_OptimizerBase):
    """gtsam/nonlinear/DoglegOptimizer.h: DoglegParams::deltaInitial = 1.0, NonlinearOptimizerParams defaults."""

    def __init__(self, graph, initialValues, ordering=None, deltaInitial=1.0, maxIterations=100, relativeErrorTol=1e-5,
                 absoluteErrorTol=1e-5, errorTol=0.0, backend_factory=None, ordering_fn=None, orderingType="COLAMD"):
        super().__init__(graph, initialValues, ordering, orderingType, backend_factory, ordering_fn)
        self._p = (deltaInitial, maxIterations, relativeErrorTol, absoluteErrorTol, errorTol)

    def optimize(self) -> Values:
        self.result = self.backend.dogleg_optimize(*self._p)
        return self.values()

    def getDelta(self) -> float:
        return float(self.result["final_lambda"])


class GaussNewtonOptimizer(_OptimizerBase):
    def __init__(self, graph, initialValues, ordering=None, maxIterations=100, relativeErrorTol=1e-5,
                 absoluteErrorTol=1e-5, errorTol=0.0, backend_factory=None, ordering_fn=None, orderingType="COLAMD"):
        super().__init__(graph, initialValues, ordering, orderingType, backend_factory, ordering_fn)
        self._p = (maxIterations, relativeErrorTol, absoluteErrorTol, errorTol)
        self._iterations = 0

    def optimize(self) -> Values:
        self.result = self.backend.gn_optimize(*self._p)
        return self.values()


class JointMarginal:
    """gtsam/nonlinear/Marginals.h:140-190: blocks of a joint covariance by key; keys sorted as the reference returns them."""

    def __init__(self, matrix, keys, dims):
        self._m, self._keys = matrix, list(keys)
        self._off = dict(zip(self._keys, np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(int)))
        self._dim = dict(zip(self._keys, dims))

    def at(self, iVariable, jVariable):
        i, j = self._off[iVariable], self._off[jVariable]
        return self._m[i:i + self._dim[iVariable], j:j + self._dim[jVariable]]

    __call__ = at

    def fullMatrix(self):
        return self._m

    def keys(self):
        return list(self._keys)


class Marginals:
    """gtsam/nonlinear/Marginals.h:32-138 on the device factorization: Marginals(graph, solution).marginalCovariance(key),
    .marginalInformation(key), .jointMarginalCovariance(keys), .jointMarginalInformation(keys).  (The reference's
    CHOLESKY / QR switch has no counterpart: the Bayes tree here is always the Cholesky one.)"""

    def __init__(self, graph, solution, ordering=None, backend_factory=None, orderingType="COLAMD"):
        self.arrays = graph.to_arrays(solution)
        self.backend = _make_backend(self.arrays, backend_factory)
        if ordering is None:
            kind = {"COLAMD": A.ORDER_MINDEGREE, "METIS": A.ORDER_ND, "NATURAL": A.ORDER_NATURAL}[orderingType]
            ordering = self.backend.compute_ordering(kind)
        self.backend.set_ordering([int(k) for k in ordering])
        self.backend.linearize()

    def marginalCovariance(self, key):
        return self.backend.marginal_covariance(key)

    def marginalInformation(self, key):
        return np.linalg.inv(self.marginalCovariance(key))

    def jointMarginalCovariance(self, variables) -> JointMarginal:
        keys = sorted(int(k) for k in variables)
        dims = [int(self.arrays.var_dims[int(np.searchsorted(self.arrays.var_keys, np.uint64(k)))]) for k in keys]
        return JointMarginal(self.backend.joint_marginal_covariance(keys), keys, dims)

    def jointMarginalInformation(self, variables) -> JointMarginal:
        j = self.jointMarginalCovariance(variables)
        return JointMarginal(np.linalg.inv(j.fullMatrix()), j.keys(), [j._dim[k] for k in j.keys()])
