/*
 * gsx.h — C ABI of the MI355X-native sparse nonlinear-least-squares backend.
 *
 * One shared library (libgsx.so, built by hipcc for gfx950) exports exactly the
 * entry points below.  Plain pointers and sizes only: no C++ types, no torch
 * types, no exceptions across the boundary.  Every function returns a
 * gsx_status (0 = OK).  All floating point is FP64, keys are uint64_t
 * (gtsam::Key, gtsam/base/types.h:97), indices are int32_t.
 *
 * What each entry point replaces in the reference (paths relative to
 * /root/reference/):
 *
 *   gsx_create / gsx_destroy     lowering of NonlinearFactorGraph + Values
 *                                (gtsam/nonlinear/NonlinearFactorGraph.h,
 *                                 gtsam/nonlinear/Values.h:74-79)
 *   gsx_set_ordering             params.ordering / Ordering::Create
 *                                (gtsam/nonlinear/LevenbergMarquardtParams.h:112-117),
 *                                symbolic analysis = EliminationTree ctor
 *                                (gtsam/inference/EliminationTree-inst.h:78-156) +
 *                                JunctionTree ctor (gtsam/inference/JunctionTree-inst.h:51-153)
 *   gsx_compute_ordering         Ordering::Create for the stand-alone harness
 *                                (gtsam/inference/Ordering.h:217-236); NOT ccolamd — own
 *                                minimum-degree / nested-dissection / Schur orderings
 *   gsx_set_values/get_values    Values insert/at (gtsam/nonlinear/Values.h)
 *   gsx_error                    NonlinearFactorGraph::error
 *                                (gtsam/nonlinear/NonlinearFactorGraph.cpp:170-179)
 *   gsx_linearize                NonlinearFactorGraph::linearize
 *                                (gtsam/nonlinear/NonlinearFactorGraph.cpp:239-278)
 *   gsx_get_jacobians            the JacobianFactor [A b] blocks
 *                                (gtsam/linear/JacobianFactor-inl.h:62-100)
 *   gsx_hessian_diagonal         GaussianFactorGraph::hessianDiagonal
 *                                (gtsam/linear/GaussianFactorGraph.cpp:279-287)
 *   gsx_solve                    buildDampedSystem + NonlinearOptimizer::solve
 *                                (gtsam/nonlinear/internal/LevenbergMarquardtState.h:125-156,
 *                                 gtsam/nonlinear/NonlinearOptimizer.cpp:132-179) =
 *                                eliminateMultifrontal(EliminatePreferCholesky) +
 *                                GaussianBayesTree::optimize
 *                                (gtsam/inference/EliminateableFactorGraph-inst.h:123-146,
 *                                 gtsam/linear/HessianFactor.cpp:515-551,
 *                                 gtsam/base/cholesky.cpp:108-159,
 *                                 gtsam/linear/linearAlgorithms-inst.h:49-117)
 *   gsx_linear_error             GaussianFactorGraph::error
 *                                (gtsam/linear/GaussianFactorGraph.cpp:71-78)
 *   gsx_retract                  Values::retract (gtsam/nonlinear/Values.cpp:53-64,99-101)
 *   gsx_lm_optimize / _iterate   LevenbergMarquardtOptimizer::optimize / iterate
 *                                (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:121-308,
 *                                 gtsam/nonlinear/NonlinearOptimizer.cpp:62-117,182-231)
 *   gsx_gn_optimize              GaussNewtonOptimizer::iterate
 *                                (gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66)
 *   gsx_solve_gfg                GaussianFactorGraph::optimize(ordering, EliminatePreferCholesky)
 *                                (gtsam/linear/GaussianFactorGraph.cpp:316-319) — the
 *                                NonlinearOptimizer::solve seam
 *                                (gtsam/nonlinear/NonlinearOptimizer.h:129-130)
 *
 * Threading: a handle is NOT thread-safe; distinct handles are independent.
 * One handle = one device + one HIP stream.  The library fails with
 * GSX_E_NO_DEVICE when no gfx950 device is usable: there is no CPU fallback.
 */
#ifndef GSX_H_
#define GSX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsx_context* gsx_handle;

/* ---- status codes ------------------------------------------------------- */
typedef enum gsx_status {
  GSX_OK = 0,
  GSX_E_INVALID = 1,        /* bad argument / malformed description            */
  GSX_E_NO_DEVICE = 2,      /* no usable gfx950 device / HIP error             */
  GSX_E_BAD_ORDERING = 3,   /* ordering is not a permutation of the variables  */
                            /* (EliminationTree-inst.h:137-141)                */
  GSX_E_INDETERMINATE = 4,  /* IndeterminantLinearSystemException              */
                            /* (gtsam/linear/linearExceptions.h:94-97)         */
  GSX_E_STATE = 5,          /* call sequence error (e.g. solve before linearize)*/
  GSX_E_NOMEM = 6
} gsx_status;

/* ---- variable types (state layout / tangent dim) ------------------------ */
enum {
  GSX_VAR_VECTOR = 0, /* R^d: state d, tangent d (Point2/Point3/generic)       */
  GSX_VAR_POSE2 = 1,  /* state (x,y,theta), tangent (x,y,theta)                */
  GSX_VAR_POSE3 = 2,  /* state R row-major 9 + t 3, tangent (omega3, v3)       */
  GSX_VAR_CAMERA = 3  /* PinholeCamera<Cal3Bundler>: Pose3 state 12 +          */
                      /* (f,k1,k2,u0,v0); tangent (pose 6, f, k1, k2)          */
};

/* ---- factor types -------------------------------------------------------- */
enum {
  GSX_F_LINEAR = 0,         /* JacobianFactor given directly: meas = [A b] col-major,
                               rows = f_rows[i]; keys: any number                 */
  GSX_F_PRIOR = 1,          /* PriorFactor<T> (gtsam/nonlinear/PriorFactor.h:98-102);
                               meas = prior value in the variable's state layout  */
  GSX_F_BETWEEN = 2,        /* BetweenFactor<T> (gtsam/slam/BetweenFactor.h:111-124);
                               meas = measured T in state layout; T in
                               {VECTOR, POSE2, POSE3}                              */
  GSX_F_SFM = 3,            /* GeneralSFMFactor<PinholeCamera<Cal3Bundler>,Point3>
                               (gtsam/slam/GeneralSFMFactor.h:141-177); keys
                               (camera, point); meas = (u,v)                       */
  GSX_F_PROJECTION = 4,     /* GenericProjectionFactor<Pose3,Point3,Cal3_S2>
                               (gtsam/slam/ProjectionFactor.h:138-166) with a fixed
                               calibration and no body_P_sensor; keys (POSE3,
                               VECTOR(3)); meas = (u, v, fx, fy, s, u0, v0).
                               Cheirality (default flags): zero Jacobians and the
                               constant error (2 fx, 2 fx)                         */
  GSX_F_BEARINGRANGE = 5    /* BearingRangeFactor<Pose2,Point2> (gtsam/sam/
                               BearingRangeFactor.h; Pose2::bearing / range,
                               gtsam/geometry/Pose2.cpp:246-285, Rot2.cpp:119-130);
                               keys (POSE2, VECTOR(2)); meas = (bearing angle,
                               range); error = (wrapped bearing difference,
                               range difference)                                   */
};

/* ---- noise model kinds (gtsam/linear/NoiseModel.cpp) --------------------- */
enum {
  GSX_NOISE_UNIT = 0,       /* no parameters                                     */
  GSX_NOISE_ISOTROPIC = 1,  /* 1 parameter: sigma                                */
  GSX_NOISE_DIAGONAL = 2,   /* m parameters: sigmas.  A sigma of exactly 0 makes
                               the row a HARD CONSTRAINT — noiseModel::Constrained::
                               MixedSigmas(sigmas), mu = 1000 (NoiseModel.h:389-500) */
  GSX_NOISE_GAUSSIAN = 3,   /* m*m parameters: upper-triangular sqrt information R,
                               row-major (whitened = R * unwhitened)             */
  GSX_NOISE_CONSTRAINED = 4,/* 2m parameters: sigmas (0 = hard constraint), then the
                               weights mu the violated constraint rows get in the
                               error functions — Constrained::MixedSigmas(mu, sigmas)
                               (NoiseModel.cpp:371-381,438-444).  See "Hard
                               constraints" below.                                */
  /* noiseModel::Robust (gtsam/linear/NoiseModel.cpp:709-735, LossFunctions.cpp): OR one of these onto the base
   * kind above and append ONE parameter (k / c) after the base model's parameters.  Linearization whitens with
   * the base model and then scales [A b] by sqrt(weight(|b|)) (Block reweighting); the factor's error is
   * loss(|whitened e|) instead of |e|^2 / 2. */
  GSX_NOISE_ROBUST_HUBER = 1 << 4,   /* mEstimator::Huber  :179-191 */
  GSX_NOISE_ROBUST_TUKEY = 2 << 4,   /* mEstimator::Tukey  :250-267 */
  GSX_NOISE_ROBUST_CAUCHY = 3 << 4,  /* mEstimator::Cauchy :217-224 */
  GSX_NOISE_BASE_MASK = 15
};

/* ---- Hard constraints ------------------------------------------------------
 * A row of sigma 0 (GSX_NOISE_DIAGONAL / GSX_NOISE_CONSTRAINED) is a hard constraint: the linear step satisfies
 * A_row delta = b_row exactly.  The reference eliminates a clique that holds such a row with EliminateQR and
 * Constrained::QR instead of Cholesky (gtsam/linear/HessianFactor.cpp:538-551, JacobianFactor.cpp:804-842,
 * NoiseModel.cpp:503-620); this library turns the row into a pivot of its clique before the clique's Cholesky
 * (csrc/constraint.hip) — the same step delta, the same LM trace.  What a caller sees:
 *   - gsx_error / gsx_linear_error weigh a violated constraint row by mu, as Constrained::squaredMahalanobisDistance does
 *     (NoiseModel.cpp:438-444; mu = 1000 for GSX_NOISE_DIAGONAL);
 *   - gsx_get_jacobians returns a constraint row scaled by sqrt(mu) (the reference keeps it unwhitened and applies mu in
 *     its error functions); gsx_hessian_diagonal counts it unwhitened, as JacobianFactor::hessianDiagonalAdd does;
 *   - gsx_stats.n_constraint_rows / n_constrained_fronts; a clique that takes constraint rows in is always a blocked front;
 *   - gsx_get_conditional of a clique with constraint pivots gives an EQUIVALENT conditional (a pivot row has unit
 *     diagonal and comes first in its clique's elimination), not the reference's row for row;
 *   - NOT available on such a problem: marginal covariances and Dogleg (GSX_E_STATE), sharding (gsx_set_ordering fails);
 *   - a constraint row whose entries in its clique's frontal variables all vanish NUMERICALLY although its factor
 *     holds them (a rank-deficient constraint Jacobian) is reported as GSX_E_INDETERMINATE by the solve — the reference
 *     passes such a row on to the parent clique. */

/* ---- ordering kinds for gsx_compute_ordering ----------------------------- */
enum {
  GSX_ORDER_NATURAL = 0,    /* ascending key                                      */
  GSX_ORDER_MINDEGREE = 1,  /* own approximate-minimum-degree (host)              */
  GSX_ORDER_ND = 2,         /* own nested dissection (BFS level-set separators)   */
  GSX_ORDER_SCHUR = 3,      /* VECTOR(3) landmarks first, then the rest by
                               minimum degree on the reduced graph                */
  GSX_ORDER_SCHUR_ND = 4    /* landmarks first, the rest by nested dissection of
                               the reduced graph (shallow tree; METIS analogue)   */
};

/*
 * Immutable problem structure.  Variables are listed in ASCENDING KEY ORDER
 * (the iteration order of gtsam::Values); everything else refers to variables
 * by their index in this list.  Factors are listed in graph order.
 */
typedef struct gsx_problem_desc {
  int32_t n_vars;
  const uint64_t* var_keys;   /* [n_vars] strictly ascending                     */
  const int32_t* var_types;   /* [n_vars] GSX_VAR_*                              */
  const int32_t* var_dims;    /* [n_vars] tangent dim (checked for typed vars)   */

  int32_t n_factors;
  const int32_t* f_type;      /* [n_factors] GSX_F_*                             */
  const int32_t* f_rows;      /* [n_factors] residual dim m                      */
  const int32_t* f_key_ptr;   /* [n_factors+1] CSR into f_vars                   */
  const int32_t* f_vars;      /* variable indices                                */
  const int64_t* f_meas_ptr;  /* [n_factors+1] CSR into meas                     */
  const double* meas;
  const int32_t* f_noise_kind;/* [n_factors] GSX_NOISE_*                         */
  const int64_t* f_noise_ptr; /* [n_factors+1] CSR into noise                    */
  const double* noise;
} gsx_problem_desc;

/* LevenbergMarquardtParams (gtsam/nonlinear/LevenbergMarquardtParams.h:61-98) +
 * NonlinearOptimizerParams (gtsam/nonlinear/NonlinearOptimizerParams.h:42-108). */
typedef struct gsx_lm_params {
  int32_t max_iterations;        /* 100 (legacy) / 50 (ceres)                  */
  double relative_error_tol;     /* 1e-5 / 1e-6                                */
  double absolute_error_tol;     /* 1e-5 / 0                                   */
  double error_tol;              /* 0                                          */
  double lambda_initial;         /* 1e-5 / 1e-4                                */
  double lambda_factor;          /* 10 / 2                                     */
  double lambda_upper_bound;     /* 1e5 / 1e32                                 */
  double lambda_lower_bound;     /* 0 / 1e-16                                  */
  double min_model_fidelity;     /* 1e-3                                       */
  int32_t diagonal_damping;      /* 0 / 1                                      */
  int32_t use_fixed_lambda_factor; /* 1 / 0                                    */
  double min_diagonal;           /* 1e-6                                       */
  double max_diagonal;           /* 1e32                                       */
  int32_t verbosity;             /* 0 silent, 1 SUMMARY lines to stdout        */
} gsx_lm_params;

/* Result + per-inner-iteration trace (the CSV the reference's logFile holds:
 * LevenbergMarquardtOptimizer.cpp:104-118). */
typedef struct gsx_lm_result {
  double initial_error;
  double final_error;
  double final_lambda;
  int32_t iterations;            /* outer (accepted) iterations                */
  int32_t inner_iterations;      /* tryLambda calls                            */
  int32_t n_solve_failures;      /* indeterminate systems met on the way       */
  int32_t trace_len;             /* entries written to the trace arrays        */
  /* caller-owned, each [trace_cap] or NULL */
  int32_t trace_cap;
  double* trace_error;           /* error after the inner iteration            */
  double* trace_lambda;          /* lambda tried                               */
  int32_t* trace_accepted;       /* 1 accepted, 0 rejected, -1 solve failed    */
} gsx_lm_result;

/* Symbolic-analysis and timing counters (gsx_get_stats). */
typedef struct gsx_stats {
  int64_t n_fronts;              /* Bayes-tree cliques                         */
  int64_t n_levels;              /* height of the assembly tree                */
  int64_t max_front_dim;         /* max frontal scalar dim                     */
  int64_t max_front_rows;        /* max frontal + separator scalar dim         */
  int64_t n_small_fronts;        /* fronts eliminated in LDS                   */
  int64_t n_big_fronts;          /* fronts eliminated by the blocked path      */
  double factor_flops;           /* sum f^3/3 + f^2 (s+1) + f (s+1)^2          */
  double front_bytes;            /* sum 8 (f+s+1)^2                            */
  double lpanel_bytes;           /* sum 8 f (f+s+1)  (back-substitution reads) */
  double jacobian_bytes;         /* bytes of all [A b] blocks                  */
  double hessian_bytes;          /* bytes of the block-sparse H panels         */
  double total_dim;              /* scalar dimension of the system             */
  /* accumulated device times (ms, HIP events on the handle's stream) and call
   * counts since the last gsx_reset_stats, named like the reference's gttic
   * labels (gtsam/base/timing.h) */
  double ms_linearize, ms_assemble_hessian, ms_factorize, ms_backsolve,
      ms_linear_error, ms_retract, ms_error;
  int64_t n_linearize, n_factorize, n_backsolve, n_error;
  int64_t n_cheirality;          /* SFM factors zeroed by cheirality in the last linearize */
  double amalgamation_relax;     /* the amalgamation in effect (the library's choice under GSX_AMALGAMATION_AUTO) */
  int64_t amalgamation_max_frontal_dim;
  int64_t n_medium_fronts;       /* of the LDS fronts: frontal panel in LDS, trailing block in HBM */
  int64_t n_tree_fronts;         /* fronts eliminated dependency-driven, one launch per tier, instead of level by level */
  int64_t n_upper_levels;        /* launch rounds of the fronts that are neither: levelled among themselves (longest chain) */
  int64_t n_constraint_rows;     /* hard-constraint (zero-sigma) rows of the graph */
  int64_t n_constrained_fronts;  /* fronts that take constraint rows in (always blocked; constraint pivots before Cholesky) */
} gsx_stats;

/* ---- on-disk formats (host only) -------------------------
 * Native readers of the files the reference's drivers start from, lowered to a
 * gsx_problem_desc + packed initial Values owned by an opaque gsx_dataset.
 *   gsx_read_g2o   readG2o / parse3DFactors: gtsam/slam/dataset.cpp:216-296,505-633
 *                  (2-D), :756-863 (3-D, information permuted from (t,R) to (R,t));
 *                  noise = Gaussian::Information (gtsam/linear/NoiseModel.cpp:98-112);
 *                  + the anchoring prior of examples/Pose2SLAMExample_g2o.cpp:65-67 /
 *                  Pose3SLAMExample_g2o.cpp:42-48, appended last
 *   gsx_read_bal   SfmData::FromBalFile gtsam/sfm/SfmData.cpp:189-245 + openGL2gtsam
 *                  :79-85; graph of examples/SFMExample_bal.cpp:55-68 (add_priors != 0
 *                  appends its two priors)
 *                  3-D also takes TORO's VERTEX3 / EDGE3 (roll pitch yaw, information as written: :741-764,
 *                  :829-840); poses of a 3-D file that have no VERTEX line are chained from the successive odometry
 *                  edges starting at the origin, as the reference's MATLAB loader does for such files
 *                  (matlab/+gtsam/load3D.m:21-53; examples/Data/sphere2500.txt)
 *   gsx_load2d     load2D gtsam/slam/dataset.cpp:505-570: TORO / "graph" 2-D files — VERTEX2|VERTEX_SE2|VERTEX,
 *                  VERTEX_XY landmarks (keys L(j)), EDGE2|EDGE|EDGE_SE2|ODOMETRY between factors, BR / LANDMARK
 *                  bearing-range factors (:452-496); undeclared variables are created from the odometry / the first
 *                  sighting (:540-563); no prior is added.  noise_format as enum NoiseFormat (dataset.h:65-71, AUTO
 *                  guesses GRAPH or COV, :219-231); smart != 0 lets a diagonal matrix become a Diagonal / Isotropic /
 *                  Unit model (NoiseModel.cpp:98-131); kernel 0/1/2 = none / Huber(1.345) / Tukey(4.6851) (:276-292);
 *                  model_sigmas (3 doubles or NULL) replaces the models of the file (:348-349); max_index as there
 *                  (0 = all).  The sampler (addNoise) is not offered.
 * The pointers handed out by gsx_dataset_get stay valid until gsx_dataset_free. */
enum { GSX_NOISE_FORMAT_G2O = 0, GSX_NOISE_FORMAT_TORO = 1, GSX_NOISE_FORMAT_GRAPH = 2, GSX_NOISE_FORMAT_COV = 3,
       GSX_NOISE_FORMAT_AUTO = 4 };
typedef struct gsx_dataset gsx_dataset;
gsx_status gsx_read_g2o(const char* path, int32_t is3d, gsx_dataset** out);
gsx_status gsx_load2d(const char* path, const double* model_sigmas, int64_t max_index, int32_t smart,
                      int32_t noise_format, int32_t kernel, gsx_dataset** out);
gsx_status gsx_read_bal(const char* path, int32_t add_priors, gsx_dataset** out);
gsx_status gsx_dataset_get(const gsx_dataset* d, gsx_problem_desc* desc, const double** values, int64_t* n_values);
void gsx_dataset_free(gsx_dataset* d);
/* writeG2o (gtsam/slam/dataset.cpp:636-735): the Pose2 / Pose3 variables and the between factors of a problem with the
 * given packed Values, in g2o format (17 significant digits). */
gsx_status gsx_write_g2o(const gsx_problem_desc* desc, const double* values, int64_t n_values, const char* path);
/* save2D (gtsam/slam/dataset.cpp:587-617): VERTEX2 lines for the Pose2 values; for every BetweenFactor<Pose2> an EDGE2
 * line with the keys swapped and the measurement inverted, carrying the information of the ONE Diagonal model handed in
 * (3 sigmas) in TORO order — as the reference, which ignores the factors' own models there. */
gsx_status gsx_save2d(const gsx_problem_desc* desc, const double* values, int64_t n_values, const double* model_sigmas,
                      const char* path);
/* writeBAL / writeBALfromValues (gtsam/sfm/SfmData.cpp:249-377): cameras and points of `values`, the observations of the
 * GSX_F_SFM factors grouped by point; poses back in the OpenGL convention (gtsam2openGL :88-99), rotations as Rodrigues
 * vectors, measurements (u, -v); 17 significant digits. */
gsx_status gsx_write_bal(const gsx_problem_desc* desc, const double* values, int64_t n_values, const char* path);

/* ---- lifecycle ------------------------------------------------------------ */
gsx_status gsx_create(const gsx_problem_desc* desc, int32_t device, gsx_handle* out);
gsx_status gsx_destroy(gsx_handle h);
const char* gsx_last_error(gsx_handle h);          /* human-readable, may be "" */
const char* gsx_version(void);
int32_t gsx_device_count(void);                    /* 0 when no GPU is visible  */

/* ---- ordering / symbolic analysis (host only; works without a GPU) -------- */
gsx_status gsx_set_ordering(gsx_handle h, const uint64_t* keys, int32_t n);
gsx_status gsx_compute_ordering(gsx_handle h, int32_t kind, uint64_t* keys_out);
/* Relaxed clique amalgamation, applied by the NEXT gsx_set_ordering.  relax = 0 builds exactly the
 * reference's Bayes tree: a child cluster is merged into its parent only when that adds no structural zero
 * (gtsam/inference/JunctionTree-inst.h:120-149).  relax > 0 also merges a child when the explicit zeros padded
 * into its columns are at most relax x its own conditional's size and the merged frontal dimension stays
 * <= max_frontal_dim: fewer, larger cliques, i.e. fewer levels and kernel launches on the latency-bound chains of
 * the elimination tree.  The solution is unchanged (zeros are factored as zeros); only gsx_get_tree differs.
 * relax = GSX_AMALGAMATION_AUTO (the default of a new handle; max_frontal_dim ignored): the symbolic analysis picks
 * (relax, max_frontal_dim) itself from a cost model of the level schedule — levels, pivot chains, padded flops —
 * and reports its choice in gsx_stats.  Every reference clique still survives whole inside one of the library's. */
#define GSX_AMALGAMATION_AUTO (-1.0)
gsx_status gsx_set_amalgamation(gsx_handle h, double relax, int32_t max_frontal_dim);
/* ---- one problem over the GPUs of a node -------------------------------------------------------
 * The reference has no distributed solver; the seam is the same as everywhere else in this header — the handle
 * stands in for NonlinearOptimizer::solve()/iterate() — and the partition is the multifrontal one: subtrees of the
 * Bayes tree are independent until their Schur complements meet in a common ancestor
 * (gtsam/inference/ClusterTree-inst.h:256-265), and the solve hands only x_S down
 * (gtsam/linear/linearAlgorithms-inst.h:60-62).
 *
 * One process per GPU; every rank creates its handle from the SAME problem, values and ordering and calls
 * gsx_set_shard before gsx_set_ordering.  The next gsx_set_ordering then splits the tree into a CAP (the top fronts,
 * whose subtrees are too heavy to give to one rank) and the forest below it, dealt to the ranks by cost.  A rank
 * linearizes the factors of its subtrees, assembles and eliminates their fronts, and adds its share into the cap
 * fronts; ONE sum all-reduce of the cap's contiguous arena block per factorization completes them, after which every
 * rank factors and back-substitutes the cap alike (identical bits in, identical bits out) and solves its own
 * subtrees.  Errors and status counters are summed the same way before the host reads them, so the LM policy takes
 * the same decisions on every rank.  A rank keeps only its own variables (and the cap's) up to date between
 * iterations; gsx_get_values and gsx_solve(delta_out) complete the vector with one more all-reduce.
 *
 * allreduce(user, device_buffer, count): in-place SUM of `count` doubles of device memory over all ranks, complete
 * (device-visible) when it returns; non-zero return = failure.  The library synchronizes its own stream before the
 * call.  With torch.distributed this is dist.all_reduce on a tensor aliasing the pointer (backend "nccl" = RCCL over
 * xGMI); a C++ host would call ncclAllReduce.  In a sharded handle EVERY entry point is collective: all ranks call it
 * in the same order.  gsx_dogleg_optimize and gsx_marginal_covariance are not available on a sharded handle
 * (GSX_E_STATE).  world = 1 (the default) is the single-GPU path, bit for bit. */
typedef int32_t (*gsx_allreduce_fn)(void* user, double* device_buffer, int64_t count);
gsx_status gsx_set_shard(gsx_handle h, int32_t rank, int32_t world, gsx_allreduce_fn allreduce, void* user);
typedef struct gsx_shard_info {
  int32_t rank, world;
  int32_t n_cap_fronts;          /* fronts of the cap (processed by every rank)            */
  int32_t n_own_fronts;          /* fronts of this rank's subtrees                         */
  int32_t cap_level0;            /* first level of the cap in the schedule, -1: no cap     */
  int32_t n_own_factors;         /* factors this rank linearizes                           */
  int64_t cap_doubles;           /* size of the per-factorization all-reduce               */
  double cap_flops, own_flops, total_flops;  /* elimination flop estimates: cap, own subtrees, whole tree */
} gsx_shard_info;
/* partition of the current ordering: front_owner[n_fronts] = owning rank or -1 (cap), factor_owned[n_factors] = 1
 * when this rank linearizes the factor; either array may be NULL */
gsx_status gsx_get_shard(gsx_handle h, gsx_shard_info* info, int32_t* front_owner, int32_t* factor_owned);
/* A device buffer the library allocated itself (hipMalloc) that holds nothing between calls: what a host probes its
 * all-reduce on before trusting it with the cap (the hazard is a collective library reducing IN PLACE on memory it did
 * not allocate; gtsam_petercdev_amd/distributed.py: checked_allreduce).  *count doubles at *device_ptr; the host may
 * overwrite them freely between gsx calls. */
gsx_status gsx_scratch_buffer(gsx_handle h, double** device_ptr, int64_t* count);
/* how each front of the current tree is eliminated: bits 0-1 = 0 leaf kernel (panel only), 1 whole front in LDS,
 * 2 blocked path in HBM, 3 medium (frontal panel in LDS, trailing block in HBM); bit 2 = tree front (dependency-driven
 * launch); bit 3 = lean leaf (no Schur complement stored: its blocked parent's gather forms it) */
gsx_status gsx_get_front_classes(gsx_handle h, int32_t* classes);
gsx_status gsx_get_ordering(gsx_handle h, uint64_t* keys_out);
/* Bayes-tree structure for parity checks: per front the frontal / separator
 * variable indices (CSR) and the parent front (-1 = root).  Pass NULL arrays to
 * query sizes (*n_fronts, *n_sep_total = length of sep_vars). */
gsx_status gsx_get_tree(gsx_handle h, int32_t* n_fronts, int64_t* n_sep_total, int32_t* parent,
                        int32_t* frontal_ptr, int32_t* frontal_vars,
                        int32_t* sep_ptr, int32_t* sep_vars);

/* ---- state ----------------------------------------------------------------- */
int64_t gsx_state_size(gsx_handle h);              /* doubles in packed Values  */
int64_t gsx_tangent_size(gsx_handle h);            /* doubles in packed delta   */
int64_t gsx_jacobian_size(gsx_handle h);           /* doubles in all [A b]      */
gsx_status gsx_set_values(gsx_handle h, const double* packed, int64_t n);
gsx_status gsx_get_values(gsx_handle h, double* packed, int64_t n);

/* ---- the hot path, step by step -------------------------------------------- */
gsx_status gsx_error(gsx_handle h, double* out);
gsx_status gsx_linearize(gsx_handle h);
/* [A b] of every factor, graph order, each dense column-major m x (sum d + 1) */
gsx_status gsx_get_jacobians(gsx_handle h, double* out, int64_t n);
gsx_status gsx_hessian_diagonal(gsx_handle h, double* out, int64_t n);
/* Damped solve (J'J + lambda D) delta = J'b with D = I or clamp(diag J'J).
 * delta_out may be NULL (delta stays on the device for gsx_retract).
 * On GSX_E_INDETERMINATE *bad_key holds a frontal key of a failing clique. */
gsx_status gsx_solve(gsx_handle h, double lambda, int32_t diagonal_damping,
                     double min_diagonal, double max_diagonal, double* delta_out,
                     int64_t n, uint64_t* bad_key);
/* 0.5 * sum |A_i delta - b_i|^2 on the UNDAMPED linear graph, at 0 and at the
 * delta of the last gsx_solve. */
gsx_status gsx_linear_error(gsx_handle h, double* err_at_zero, double* err_at_delta);
/* trial = values (+) delta (device delta of the last solve when delta==NULL);
 * *trial_error (may be NULL) = graph error at trial; commit != 0 makes trial
 * the current values. */
gsx_status gsx_retract(gsx_handle h, const double* delta, int64_t n, int32_t commit,
                       double* trial_error);

/* ---- optimizers ------------------------------------------------------------- */
void gsx_lm_params_legacy(gsx_lm_params* p);       /* SetLegacyDefaults          */
void gsx_lm_params_ceres(gsx_lm_params* p);        /* SetCeresDefaults           */
gsx_status gsx_lm_optimize(gsx_handle h, const gsx_lm_params* p, gsx_lm_result* r);
/* one outer iteration (LevenbergMarquardtOptimizer::iterate); the LM state
 * (lambda, factor, error) lives in the handle; reset by gsx_lm_reset. */
gsx_status gsx_lm_reset(gsx_handle h, const gsx_lm_params* p);
gsx_status gsx_lm_iterate(gsx_handle h, const gsx_lm_params* p, double* error,
                          double* lambda);
/* The trust policy alone, as a pure host function (csrc/lm_policy.cpp; usable without a GPU): what one damped trial means
 * for the controller.  Inputs: whether the damped system could be factored, the quadratic model's cost at zero and at the
 * step (undamped linearized graph), the true cost at the retracted point.  `state` is updated in place.  Decision-equivalent
 * to LevenbergMarquardtOptimizer::tryLambda (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:121-270) +
 * LevenbergMarquardtState::increaseLambda / decreaseLambda (gtsam/nonlinear/internal/LevenbergMarquardtState.h:70-94);
 * gsx_lm_optimize / gsx_lm_iterate call exactly this function. */
typedef struct gsx_lm_state {
  double lambda;                 /* damping weight of the next trial            */
  double factor;                 /* its growth multiplier                       */
  double cost;                   /* nonlinear cost at the current values        */
  int32_t outer_iterations;      /* accepted steps                              */
  int32_t inner_iterations;      /* trials that changed the controller          */
} gsx_lm_state;
enum { GSX_LM_TAKE = 1,          /* step accepted: commit the trial values, relinearize           */
       GSX_LM_RETRY = 0,         /* step rejected: same linearization, larger damping             */
       GSX_LM_SETTLE = 2,        /* change below the relative tolerance: stop searching lambda    */
       GSX_LM_GIVE_UP = 3 };     /* damping passed lambda_upper_bound                             */
typedef struct gsx_lm_decision {
  int32_t verdict;               /* GSX_LM_*                                    */
  int32_t solved;                /* the damped system was factored              */
  double gain_ratio;             /* actual / predicted decrease (0 when not formed) */
  double cost_change;            /* cost(now) - cost(trial)                     */
  double trial_cost;             /* +inf when the trial cost was not consulted  */
  double lambda_tried;
} gsx_lm_decision;
gsx_status gsx_lm_decide(const gsx_lm_params* p, gsx_lm_state* state, int32_t solved, double model_at_zero,
                         double model_at_step, double trial_cost, gsx_lm_decision* out);
/* One LM trial without the accept/reject policy — the numeric body of LevenbergMarquardtOptimizer::tryLambda
 * (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:121-230), preceded when relinearize != 0 by the linearization
 * iterate() does (:252-262): damped solve, linearized error at 0 and at delta, retract into the trial values (the
 * current values stay), nonlinear error there — with ONE host synchronisation, as inside gsx_lm_optimize.
 * GSX_E_INDETERMINATE when the damped system cannot be factored. */
gsx_status gsx_lm_trial(gsx_handle h, int32_t relinearize, double lambda, int32_t diagonal_damping,
                        double min_diagonal, double max_diagonal, double* linear_error_0,
                        double* linear_error_delta, double* trial_error);
gsx_status gsx_gn_optimize(gsx_handle h, int32_t max_iterations, double relative_error_tol,
                           double absolute_error_tol, double error_tol, gsx_lm_result* r);
/* DoglegOptimizer (gtsam/nonlinear/DoglegOptimizer.cpp:84-121; DoglegOptimizerImpl::Iterate in
 * ONE_STEP_PER_ITERATION mode, DoglegOptimizerImpl.h:137-252) on the same kernels.  The result's
 * final_lambda / trace_lambda carry the trust-region radius delta.  DoglegParams::deltaInitial = 1.0. */
gsx_status gsx_dogleg_optimize(gsx_handle h, double delta_initial, int32_t max_iterations, double relative_error_tol,
                               double absolute_error_tol, double error_tol, gsx_lm_result* result);
/* Marginals::marginalCovariance(key) (gtsam/nonlinear/Marginals.cpp:107-136): the dA x dA block of H^-1 of the current
 * linearization in the variable's tangent space (column-major), from the undamped factorization on the device. */
gsx_status gsx_marginal_covariance(gsx_handle h, uint64_t key, double* out, int64_t n_out);
/* Marginals::jointMarginalCovariance(variables) (gtsam/nonlinear/Marginals.cpp:138-189): the joint covariance of up to 64
 * variables, D x D row-major with D = sum of their dimensions, blocks in the order of `keys` (the reference returns
 * them sorted by key: pass sorted keys for the same layout). */
gsx_status gsx_joint_marginal_covariance(gsx_handle h, const uint64_t* keys, int32_t n_keys, double* out, int64_t n_out);

/* ---- partial relinearization / re-elimination on a fixed graph (first step) ----------------------
 * The numeric core of an iSAM2 update (gtsam/nonlinear/ISAM2.cpp:419-484: relinearize the factors of the variables
 * whose linearization point moved; :725-783: re-eliminate the part of the Bayes tree that contains them) on a FIXED
 * structure: graph, ordering and Bayes tree stay as they are; the factorization and every untouched subtree's Schur
 * complement stay resident in HBM.  Given the moved variables (`keys`; `states` = their new packed states one after the
 * other in that order, or NULL when the handle's values are already current) the call re-linearizes only the factors
 * touching them, re-assembles only the H panels of those factors' variables, and re-eliminates only the cliques holding
 * those variables and their ancestors.  The result is bit-identical to a full gsx_linearize + gsx_solve(lambda = 0)
 * factorization at the same values.  Needs the resident undamped factorization of the current linearization
 * (GSX_E_STATE otherwise); follow with gsx_solve(h, 0, ...) — which then only back-substitutes — or the marginal queries.
 * When most of the tree is dirty anyway the call takes the full path (same bits; the stats then report everything).
 * Adding / removing factors and variables: gsx_update below; the partial ("wildfire") back-substitution:
 * gsx_backsubstitute_wildfire. */
typedef struct gsx_partial_stats {
  int32_t n_factors_relinearized, n_panels_reassembled, n_fronts_reeliminated, n_fronts;
} gsx_partial_stats;
gsx_status gsx_relinearize_partial(gsx_handle h, const uint64_t* keys, int32_t n_keys, const double* states,
                                   int64_t n_states, gsx_partial_stats* out);
/* ---- ISAM2's partial ("wildfire") back-substitution --------------------------------------------------------------------
 * DeltaImpl::UpdateGaussNewtonDelta (gtsam/nonlinear/ISAM2-impl.cpp:48-77) -> optimizeWildfireNonRecursive
 * (gtsam/nonlinear/ISAM2Clique.cpp:261-287) on the resident undamped factorization: starting at the roots, a clique is
 * back-substituted when it was re-eliminated since the handle's last complete solution ("replaced": the cliques redone
 * by gsx_relinearize_partial, or all of them after a full factorization) or when one of its separator variables changed
 * (ISAM2Clique::isDirty, :68-90); its new frontal solution is kept — and its frontal variables count as changed — when
 * the clique was replaced or the solution moved by at least `threshold` in the infinity norm, and is put back to the old
 * values otherwise (valuesChanged / restoreFromOriginals, :175-201); children are visited only through a dirty parent.
 * threshold <= 0, or no complete undamped solution resident yet (gsx_solve with lambda = 0 leaves one, and so does
 * this call): every clique is back-substituted, as the reference does.  delta_out (tangent order of gsx_solve) may be
 * NULL; n_vars_solved = the reference's lastBacksubVariableCount (frontal variables of the cliques back-substituted).
 * A front of a relaxed tree (gsx_set_amalgamation) is visited or skipped as a whole, so the reference's clique-by-clique
 * pattern is reproduced exactly at relax = 0.  The dirty rule is evaluated at the top of the back-substitution kernels
 * (a clean clique's workgroup returns at once); one small bookkeeping launch follows every level.  Needs the resident
 * undamped factorization (GSX_E_STATE otherwise); not on a sharded handle. */
gsx_status gsx_backsubstitute_wildfire(gsx_handle h, double threshold, double* delta_out, int64_t n,
                                       int64_t* n_vars_solved, uint64_t* bad_key);

/* ---- structural update of a live handle: ISAM2::update(newFactors, newTheta, removeFactorIndices) ------------------------
 * (gtsam/nonlinear/ISAM2.cpp:395-484; the reference then detaches the affected part of the Bayes tree, re-orders it with
 * constrained COLAMD and re-eliminates it, ISAM2.cpp:117-362.)
 * `desc` describes the WHOLE graph after the update: the variables and factors that stay plus the new ones, in any
 * position.  factor_origin[i] = index, in the handle's current graph, of new factor i, or -1 for a factor that is new;
 * current factors that no entry names are removed.  A variable is kept when its key exists on both sides (type and
 * dimension must agree); new_values = the states of the NEW variables, packed in ascending-key order.
 * Semantics (iSAM2's): kept variables keep their current linearization point and kept factors their current [A b]
 * (copied on the device, not re-linearized); only the new factors are linearized.  Elimination order: the unaffected
 * variables keep their relative order, the affected ones — the variables of added / removed factors and the new
 * variables — are moved to the end, least-connected first (the constrained ordering of ISAM2.cpp:265-299, with a static
 * degree rule in place of CCOLAMD); the amalgamation setting of the current tree is kept; gsx_set_ordering afterwards
 * replaces the ordering like on any handle.
 * What is NOT incremental: the symbolic analysis is redone for the whole graph on the host (O(graph), not O(affected)),
 * and the next solve re-eliminates the whole tree on the device (milliseconds; gsx_relinearize_partial remains
 * available afterwards for moved variables).  The stats say what the update cost.  Not on a sharded handle. */
typedef struct gsx_update_stats {
  int32_t n_vars_added, n_vars_removed, n_factors_added, n_factors_removed, n_vars_affected, n_fronts;
  double host_symbolic_s;        /* ordering + symbolic analysis + upload of the tables */
  double device_s;               /* copies + linearization of the new factors (stream time, synchronised) */
} gsx_update_stats;
gsx_status gsx_update(gsx_handle h, const gsx_problem_desc* desc, const int32_t* factor_origin, const double* new_values,
                      int64_t n_new_values, gsx_update_stats* out);

/* DoglegOptimizerImpl::ComputeDoglegPoint (DoglegOptimizerImpl.cpp:26-86) on plain vectors (host). */
gsx_status gsx_dogleg_point(double delta, const double* dx_u, const double* dx_n, int64_t n, double* out);

/* ---- factors the backend does not know (seam S5: NonlinearFactor::linearize, gtsam/nonlinear/NonlinearFactor.h:145-146) ----
 * A graph may carry GSX_F_LINEAR slots for factor types outside the table above: the caller linearizes those on the CPU
 * at every new linearization point (factor->linearize(values) -> JacobianFactor) and refreshes their [A b] blocks in
 * place with this call — values = the factors' blocks one after the other, each m x (sum d + 1) column-major exactly as
 * `meas` at gsx_create (the slot's noise model is folded in here, as at creation).  gsx_linearize leaves these blocks
 * alone; the Hessian panels are re-assembled at the next solve.  gsx_error counts 0 for such a slot: the caller adds
 * its factors' nonlinear error itself. */
gsx_status gsx_set_block_jacobians(gsx_handle h, int32_t first_factor, int32_t n_factors, const double* values,
                                   int64_t n_values);
/* [R S d] of clique `front` (numbering of gsx_get_tree) from the factorization resident after a solve — the
 * GaussianConditional the reference's elimination stores in the Bayes-tree clique (R() / S() / d(),
 * gtsam/linear/GaussianConditional.h:243-252; BayesTreeCliqueBase-inst.h:27-31): n_frontal x n_cols column-major, the
 * columns in the order frontal variables, separator variables (both as gsx_get_tree lists them), rhs.  With the
 * reference's cliques (gsx_set_amalgamation(h, 0, .)) it is the reference's conditional entry for entry; out = NULL
 * queries the sizes. */
gsx_status gsx_get_conditional(gsx_handle h, int32_t front, int32_t* n_frontal, int32_t* n_cols, double* out,
                               int64_t n_out);

/* ---- the linear seam (NonlinearOptimizer::solve override) ------------------- */
/* desc must contain only GSX_F_LINEAR factors and GSX_VAR_VECTOR variables;
 * ordering may be NULL (own minimum degree).  delta_out in variable-index order. */
gsx_status gsx_solve_gfg(const gsx_problem_desc* desc, const uint64_t* ordering,
                         int32_t device, double* delta_out, int64_t n, uint64_t* bad_key);
/* The same seam with the structure kept: NonlinearOptimizer::solve (gtsam/nonlinear/NonlinearOptimizer.h:129-130) is
 * called once per LM trial on graphs of ONE structure (same keys, dims, ordering; only the numbers and the damping
 * priors' sigmas change).  Create the handle once (gsx_create on the GSX_F_LINEAR description + gsx_set_ordering: the
 * symbolic analysis and every device table are built once), then per call hand over only the [A b] blocks of all
 * factors, one after the other as in `meas` (blocks = NULL: solve what the handle holds).  delta_out as above. */
gsx_status gsx_solve_gfg_h(gsx_handle h, const double* blocks, int64_t n_blocks, double* delta_out, int64_t n,
                           uint64_t* bad_key);

/* ---- dense kernel exposed for unit parity (gtsam/base/cholesky.cpp:108-159) -- */
/* In-place partial Cholesky of an n x n column-major symmetric matrix (upper
 * triangle significant, like the reference): on return the first nfrontal rows
 * hold [R S], the trailing block holds C - S'S (upper).  *ok = 0 when the
 * reference would have reported failure. */
gsx_status gsx_cholesky_partial(double* abc, int32_t n, int32_t nfrontal, int32_t device,
                                int32_t* ok);

/* ---- stats / stream ---------------------------------------------------------- */
gsx_status gsx_get_stats(gsx_handle h, gsx_stats* out);
gsx_status gsx_reset_stats(gsx_handle h);
gsx_status gsx_synchronize(gsx_handle h);
/* level -1 (default): no timers — the ms_* / n_* fields of gsx_stats stay 0; level 0: HIP-event timers around the phases
 * (an event record costs ~5 us of stream time, ~70 us per LM iteration); level 1: additionally every factor_small /
 * factor_big launch (and one stream only).  bench.py times with -1 and reads the phase times from a level-0 pass. */
gsx_status gsx_set_profiling(gsx_handle h, int32_t level);
/* average duration (ms) of the named kernel class over launches since the last
 * gsx_reset_stats, measured with HIP events on the handle's stream; names:
 * "linearize", "assemble_hessian", "factor_leaf", "factor_small", "factor_big", "backsolve",
 * "linear_error", "retract", "error". *launches (may be NULL) = count. */
gsx_status gsx_kernel_time(gsx_handle h, const char* name, double* avg_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H_ */
