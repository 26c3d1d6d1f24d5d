"""gtsam_petercdev_amd — MI355X-native (gfx950, HIP) backend for the sparse nonlinear
least-squares hot path of GTSAM: linearize -> block-sparse Hessian -> multifrontal
Cholesky down the Bayes-tree cliques -> back-substitution, behind the
NonlinearFactorGraph / Values / LevenbergMarquardtOptimizer interface.

The compute lives in csrc/ (HIP kernels + C-ABI, include/gsx.h); this package is the
host-side mirror of the reference interface and the ctypes binding."""
from ._abi import (GsxError, IndeterminantLinearSystemException, ProblemArrays, LMParams,
                   lm_params_legacy, lm_params_ceres)
from .graph import *  # noqa: F401,F403

__version__ = "0.1.0"
