/*
 * multical_hip.h -- C ABI of the MI355X (gfx950) backend for the aslam_backend
 * Gauss-Newton / Levenberg-Marquardt inner loop of zhixy/multical.
 *
 * The reference has no C ABI for this path: its plug-in point is the C++ class
 * aslam::backend::LinearSystemSolver (aslam_optimizer/aslam_backend/include/aslam/backend/
 * LinearSystemSolver.hpp:16-109), selected through Optimizer2Options::linearSystemSolver
 * (.../Optimizer2Options.hpp:9-42; Python name `linearSolver`,
 * aslam_optimizer/aslam_backend_python/src/OptimizerOptions.cpp:37).  Every entry point below
 * names the reference member function(s) it stands in for; the C++ adapter with the
 * LinearSystemSolver surface is multical_amd/csrc/hip_linear_system_solver.hpp and the
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; the caller owns every host buffer it passes and
 * the pointers are borrowed for the duration of the call; the context owns all device memory.
 * All arithmetic is IEEE fp64, indices are 32-bit int.  Every function returns 0 on success and
 * a negative code on failure unless stated otherwise; mc_last_error() gives the message.
 *
 * Threads and streams.  A context is not thread-safe and not re-entrant, and it is SINGLE-STREAM: the `_device` entry
 * points enqueue on the caller's stream and return at once; the library's per-context scratch (per-observation records,
 * cost partials, replicas of contended entries, chi^2) is shared by all of them, so two `_device` calls of one context must
 * not be in flight on different streams.  Calls without a `_device` suffix run on the null stream and synchronise before
 * returning; they first wait for the context's last `_device` call to finish (an event recorded at its end), so
 * mc_apply_update / mc_revert_update never overwrite the state under running kernels and mc_device_status never reads the
 * flag before the kernels that set it -- also when the caller's stream does not synchronise with the null stream (torch's
 * pool streams).  When mc_device_status reports a non-zero status, the outputs of the call that raised it are undefined
 * (partial sums): the solver classes throw before consuming them, as the reference throws out of evaluateError.
 */
#ifndef MULTICAL_HIP_H
#define MULTICAL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_context mc_context;

/* design-variable kinds (reference classes in aslam_backend_expressions / aslam_splines) */
enum {
  MC_DV_EUCLIDEAN_POINT = 0,     /* EuclideanPoint        : 3 params, additive            */
  MC_DV_ROTATION_QUATERNION = 1, /* RotationQuaternion    : 4 params [x y z w], minimal 3 */
  MC_DV_SCALAR = 2,              /* Scalar                : 1 param                        */
  MC_DV_EUCLIDEAN_DIRECTION = 3, /* EuclideanDirection    : given as a 3-vector, minimal 2 */
  MC_DV_SPLINE_COEFF = 4,        /* DesignVariableMappedVector<D> (created by mc_add_spline) */
  MC_DV_PROJECTION = 5,          /* DesignVariableAdapter<Projection>: 4/5/6 params         */
  MC_DV_DISTORTION = 6,          /* DesignVariableAdapter<Distortion>: 4 params             */
  MC_DV_MATRIX_BASIC = 7,        /* MatrixBasic (created by mc_add_matrix_design_variable)  */
  MC_DV_HOMOGENEOUS_POINT = 8    /* HomogeneousPoint      : 4 params, kept at unit norm, minimal 3 (aslam_backend_expressions/src/HomogeneousPoint.cpp) */
};
enum { MC_PROJ_PINHOLE = 0, MC_PROJ_OMNI = 1, MC_PROJ_DS = 2 };
enum { MC_DIST_NONE = 0, MC_DIST_RADTAN = 1, MC_DIST_EQUI = 2 };
/* M-estimators, aslam_backend/src/MEstimatorPolicies.cpp:17-129.  param: Huber/Bz k; Cauchy/
 * Geman-McClure sigma^2; Blake-Zisserman epsilon (= (1-w)/w * exp(-chi2inv(p, df)), computed by
 * the caller -- the reference evaluates it once in the constructor). */
enum { MC_MEST_NONE = 0, MC_MEST_HUBER = 1, MC_MEST_CAUCHY = 2, MC_MEST_GEMAN_MCCLURE = 3, MC_MEST_BZ = 4,
       MC_MEST_BLAKE_ZISSERMAN = 5 };

/* ---- lifetime ------------------------------------------------------------------------- */
/* One context drives one GPU (one process per GPU).  device < 0 keeps the current device. */
mc_context* mc_create(int device);
void mc_destroy(mc_context* ctx);
const char* mc_last_error(const mc_context* ctx); /* valid until the next call on ctx; NULL ctx: creation error */

/* ---- problem description (replaces OptimizationProblem::addDesignVariable / addErrorTerm as
 *      seen by Optimizer2::initialize, aslam_backend/src/Optimizer2.cpp:95-151) ------------- */
/* Design variables are numbered in call order; that order must be the problem's design-variable
 * order (incremental_calibration/src/core/OptimizationProblem.cpp:272-292) because it defines the
 * block indices.  Returns the dv id (>= 0). */
int mc_add_design_variable(mc_context* ctx, int kind, const double* params, int n_params, int active, double scaling);
/* BSplinePoseDesignVariable (dim 6, rotation-vector parametrisation) or EuclideanBSplineDesignVariable
 * (dim 3): adds n_coeff consecutive MC_DV_SPLINE_COEFF design variables.  coeffs is dim x n_coeff,
 * column major (coefficient i contiguous).  Returns the spline id. */
int mc_add_spline(mc_context* ctx, int order, int dim, const double* knots, int n_knots, const double* coeffs,
                  int n_coeff, int active);
int mc_spline_first_dv(const mc_context* ctx, int spline_id);
/* Camera geometry whose parameters live in two design variables (active or not). dv_dist < 0 for MC_DIST_NONE. */
int mc_add_camera(mc_context* ctx, int projection_model, int distortion_model, int dv_projection, int dv_distortion);

/* Error-term batches.  Terms are numbered in call order (row base = running sum of dimensions,
 * Optimizer2.cpp:130-135).  sqrt_inv_r[k] is the scalar s with sqrtInvR = s*I (Multical uses
 * isotropic covariances throughout; a general sqrtInvR follows the batch with mc_set_last_batch_sqrt_inv_r).
 * All return 0 or a negative error. */

/* SimpleReprojectionError over T_c_b * T_w_b(t + d)^-1 * T_p_w^-1 * p  (sensors_and_targets.py:556-630;
 * aslam_cv_error_terms/.../SimpleReprojectionError.hpp:43-64; time-offset node
 * aslam_splines/src/BSplineExpressions.cpp:406-555).  Terms of observation o are
 * obs_first[o] .. obs_first[o+1]-1.  dv_time_offset < 0: constant time. */
int mc_add_reprojection_batch(mc_context* ctx, int camera, int pose_spline, int dv_q_c_b, int dv_t_c_b,
                              int dv_time_offset, int n_obs, const double* obs_time, const int* obs_dv_q_p_w,
                              const int* obs_dv_t_p_w, const int* obs_first, const double* y /* [n,2] */,
                              const double* p_target /* [n,3] */, const double* sqrt_inv_r /* [n] */,
                              double time_padding, int mest_kind, double mest_param);
/* ReprojectionError with camera design variables over B_{n-1}*..*B_0 * T_target_camera^-1 * p
 * (kalibr_camera_calibration/CameraCalibrator.py:195-219; ReprojectionError.hpp:50-77;
 * aslam_cv_backend/.../CameraDesignVariable.hpp:39-54). */
int mc_add_static_reprojection_batch(mc_context* ctx, int camera, int n_chain, const int* chain_dv_q,
                                     const int* chain_dv_t, int n_obs, const int* obs_dv_q_t_c,
                                     const int* obs_dv_t_t_c, const int* obs_first, const double* y,
                                     const double* p_target, const double* sqrt_inv_r, int mest_kind,
                                     double mest_param);
/* The front-end's initialisation problems (small; a handful of global design variables, no spline).
 * kalibr GyroscopeError(measurement, invR, angularVelocity, bias) (kalibr/src/GyroscopeError.cpp:5-17) as
 * findExtrinsicPriorSensorsToCamera builds it (kalibr_sensor_calibration/sensors_and_targets.py:858-879):
 * e = C(q) * w_pred + bias - meas with w_pred a constant per term; dv_q ROTATION_QUATERNION, dv_bias EUCLIDEAN_POINT. */
int mc_add_gyroscope_prior_batch(mc_context* ctx, int dv_q, int dv_bias, int n, const double* w_pred /* [n,3] */,
                                 const double* meas /* [n,3] */, const double* sqrt_inv_r /* [n] */, int mest_kind,
                                 double mest_param);
/* aslam::backend::ErrorTermTransformation(T, prior, weightRotation, weightTranslation)
 * (aslam_backend_expressions/src/ErrorTermTransformation.cpp:19-81) with T = TransformationBasic(q_a, t_a) or, when
 * dv_q_b / dv_t_b >= 0, T = TransformationBasic(q_a, t_a) * inverse(TransformationBasic(q_b, t_b))
 * (findTargetPosePrior / findExtrinsicPriorSensorsToCamera, sensors_and_targets.py:812-817, 886-919).
 * prior_q [n,4] quaternions [x y z w], prior_t [n,3]. */
int mc_add_transformation_batch(mc_context* ctx, int dv_q_a, int dv_t_a, int dv_q_b, int dv_t_b, int n,
                                const double* prior_q, const double* prior_t, double weight_rotation,
                                double weight_translation, int mest_kind, double mest_param);
/* kalibr EuclideanError, accelerometer form (sensors_and_targets.py:1133-1179; kalibr/src/EuclideanError.cpp:5-27).
 * bias_spline < 0: static bias dv_bias.  dv_gravity: EUCLIDEAN_DIRECTION or EUCLIDEAN_POINT. */
int mc_add_accelerometer_batch(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_gravity,
                               int dv_q_i_b, int dv_r_b, int n, const double* t, const double* meas /* [n,3] */,
                               const double* sqrt_inv_r, int mest_kind, double mest_param);
/* kalibr EuclideanError, gyroscope form (sensors_and_targets.py:1181-1221). */
int mc_add_gyroscope_batch(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_q_i_b, int n,
                           const double* t, const double* meas, const double* sqrt_inv_r, int mest_kind,
                           double mest_param);
/* MatrixBasic (aslam_backend_expressions/src/MatrixBasic.cpp:10-47): 3x3 matrix A (9 values, row major) whose entries
 * with a non-zero update_pattern (9 ints, row major) are the minimal parameters, numbered in row-major order of the
 * pattern.  A matrix without free entries is a constant (the front-end's Ix / Iy / Iz selectors).  Returns the dv id. */
int mc_add_matrix_design_variable(mc_context* ctx, const double* A, const int* update_pattern, int active, double scaling);
/* The other two IMU models of multical_calibrate_sensors --imu-models.
 * ScaledMisalignedImu accelerometer (sensors_and_targets.py:1449-1497):  M_accel * (C_i_b * (...)) + b_a, and with
 * size_effect != NULL the ScaledMisalignedSizeEffectImu form (:1608-1661) with per-axis lever arms; size_effect lists the
 * six design variables rx_i, ry_i, rz_i (EUCLIDEAN_POINT) and Ix, Iy, Iz (MATRIX_BASIC). */
int mc_add_accelerometer_batch_sm(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_gravity, int dv_q_i_b,
                                  int dv_r_b, int dv_M_accel, const int* size_effect, int n, const double* t,
                                  const double* meas, const double* sqrt_inv_r, int mest_kind, double mest_param);
/* ScaledMisalignedImu gyroscope (:1499-1550):  M_gyro * (C_gyro_b * w_b) + M_accel_gyro * (C_gyro_b * a_b) + b_g with
 * C_gyro_b = C_gyro_i * C_i_b and a_b the body-frame specific force at the lever arm r_b. */
int mc_add_gyroscope_batch_sm(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_q_i_b, int dv_q_gyro_i,
                              int dv_M_gyro, int dv_M_accel_gyro, int dv_gravity, int dv_r_b, int n, const double* t,
                              const double* meas, const double* sqrt_inv_r, int mest_kind, double mest_param);
/* Multical LiDAR point-to-plane ScalarError (sensors_and_targets.py:274-305; kalibr/src/ScalarError.cpp:6-31). */
int mc_add_lidar_batch(mc_context* ctx, int pose_spline, int dv_time_offset, int dv_q_l_b, int dv_t_l_b, int n,
                       const double* t, const double* dir /* [n,3] unit */, const double* range,
                       const double* sqrt_inv_r, const int* dv_q_p_w, const int* dv_t_p_w, double time_padding,
                       int mest_kind, double mest_param);
/* Landmarks that are design variables (kalibr_camera_calibration with estimateLandmarks: the target corners become
 * HomogeneousPointDv and ReprojectionError's point is T * landmark; aslam_backend_expressions/src/HomogeneousPoint.cpp:44-52,
 * HomogeneousExpressionNode.cpp:77-82).  For the static reprojection batch added last: per term the dv id of its
 * MC_DV_HOMOGENEOUS_POINT, or -1 to keep the constant p_target.  The state of a landmark must keep w >= 0. */
int mc_set_last_batch_landmarks(mc_context* ctx, int n, const int* dv_landmark);
/* ErrorTermFs<D>::setSqrtInvR for the terms of the batch added last (aslam_backend/include/aslam/backend/ErrorTerm.hpp:246,
 * implementation/ErrorTerm.hpp:96-192): S is [n, dim, dim] row major, one general sqrtInvR per term (invR = S S^T; the weighted
 * error is sqrt(w) S^T e, the weighted Jacobian sqrt(w) S^T J, chi2 = e^T invR e); it replaces the scalars given with the batch.
 * dim and n must be the batch's error dimension and term count.  Not for the motion / transformation terms, which carry their
 * own weighting.  A lowering reads S from the term's sqrtInvR(); callers that hold invR use mc_compute_matrix_sqrt. */
int mc_set_last_batch_sqrt_inv_r(mc_context* ctx, int dim, int n, const double* S);
/* sm::eigen::computeMatrixSqrt (Schweizer-Messer/sm_eigen/include/sm/eigen/matrix_sqrt.hpp:21-40), what ErrorTermFs<D>::setInvR
 * applies to invR: S = P^T L sqrt(D) of the pivoted LDL^T, A = S S^T.  A and S are n x n row major, n <= 16.  Host only. */
int mc_compute_matrix_sqrt(int n, const double* A, double* S);
/* BSplineMotionError (aslam_splines/include/aslam/backend/implementation/BSplineMotionError.hpp:28-166).  W is dim x dim. */
int mc_add_motion_term(mc_context* ctx, int spline, const double* W, int error_term_order);

/* Optional, before mc_finalize: this process handles shard `rank` of `world` (terms are split into
 * `world` contiguous time windows of equal estimated cost; pattern and slot numbering stay global so
 * partial results from different ranks add element-wise).  No counterpart in the reference, whose
 * only parallelism is boost threads over term ranges (LinearSystemSolver.cpp:50-78). */
int mc_set_shard(mc_context* ctx, int rank, int world);
/* the shard this context was given (0 of 1 without mc_set_shard) */
int mc_shard_info(const mc_context* ctx, int* rank, int* world);
/* With shards a rank writes only its own window of the (global) Cx and e: written_only = 1 makes every later pass clear
 * just those values (one launch over a table built at mc_finalize) instead of the whole arrays -- at 8 ranks the whole
 * Cx is 1.25 GB, 0.19 ms of a 1.5 ms build.  The caller then owns the rest of the contract: every d_Cx / d_e buffer it
 * passes starts out zero (one hipMemset when it is allocated) and between passes only values this rank writes are
 * changed in it (mc_unpack_shared does exactly that; an all-reduce of the whole array does not).  0 (default): every
 * pass clears everything, any buffer may be passed.  Without shards the call changes nothing.  mc_optimize_sharded sums
 * whole arrays and therefore switches the mode off on its context.  No counterpart in the reference. */
int mc_set_clear_mode(mc_context* ctx, int written_only);

/* With shards, only the value entries that more than one rank's terms can write need summing over ranks: the halos of
 * the coefficient band at the window boundaries and the blocks between global design variables (a few 10^5 values,
 * against 10^7..10^8 in the whole array).  After mc_finalize: their count, their (sorted) indices into Cx -- identical
 * on every rank --, and gather / scatter between a device Cx array and a compact device buffer, enqueued on `stream`:
 * pack -> all-reduce(buffer) -> unpack.  Every other entry of Cx is non-zero on exactly one rank (its owner).  No
 * counterpart in the reference. */
int mc_shared_count(const mc_context* ctx);
int mc_get_shared_indices(const mc_context* ctx, int* out);
int mc_pack_shared(mc_context* ctx, const double* d_Cx, double* d_buf, void* stream);
int mc_unpack_shared(mc_context* ctx, const double* d_buf, double* d_Cx, void* stream);
/* The same for rhs: the scalar columns more than one rank writes (calibration parameters, coefficients at the window
 * boundaries; sorted, identical on every rank) and gather / scatter between a device rhs array and a compact buffer.  With
 * them one all-reduce of [shared Cx entries | shared rhs entries | J] per iteration sums what needs summing; every other
 * entry of rhs is complete on the rank that owns its window and zero elsewhere. */
int mc_shared_rhs_count(const mc_context* ctx);
int mc_get_shared_rhs_indices(const mc_context* ctx, int* out);
int mc_pack_shared_rhs(mc_context* ctx, const double* d_rhs, double* d_buf, void* stream);
int mc_unpack_shared_rhs(mc_context* ctx, const double* d_buf, double* d_rhs, void* stream);

/* LinearSystemSolver::initMatrixStructure + BlockCholeskyLinearSystemSolver::initMatrixStructureImplementation
 * (LinearSystemSolver.cpp:117-138, BlockCholeskyLinearSystemSolver.cpp:34-55) plus the symbolic part of
 * SparseBlockMatrix::fillCCS (sparse_block_matrix.hpp:704-729): assigns block indices / column bases to
 * active design variables, row bases to terms, and builds the upper-triangular block pattern (including
 * every diagonal block and the structurally-zero blocks of padded spline coefficients). */
int mc_finalize(mc_context* ctx);
int mc_dims(const mc_context* ctx, int* j_rows, int* j_cols, int* n_blocks, int* n_terms, int* nnz);
int mc_dv_block(const mc_context* ctx, int dv, int* block_index, int* column_base);
/* scalar CCS pattern of the upper triangle: Cp[j_cols+1], Ci[nnz] -- bit-exact fillCCS ordering */
int mc_get_pattern(const mc_context* ctx, int* Cp, int* Ci);

/* ---- the hot path --------------------------------------------------------------------- */
/* LinearSystemSolver::evaluateError (LinearSystemSolver.cpp:81-92): returns J = sum w(chi2) chi2 in
 * *J_out and, if e_out != NULL, e (j_rows doubles, = -sqrt(w) sqrtInvR^T e per term). */
int mc_evaluate_error(mc_context* ctx, int use_mestimator, double* J_out, double* e_out);
/* BlockCholeskyLinearSystemSolver::buildSystem (BlockCholeskyLinearSystemSolver.cpp:58-72) followed by
 * fillCCS values: Cx_out[nnz] in the order of mc_get_pattern, rhs_out[j_cols].  Either may be NULL. */
int mc_build_system(mc_context* ctx, int use_mestimator, double* Cx_out, double* rhs_out);
/* Same two operations writing into caller-owned DEVICE buffers, enqueued on `stream` (a hipStream_t)
 * without synchronising: d_J (1 double), d_e (j_rows or NULL), d_Cx (nnz), d_rhs (j_cols).  The buffers
 * are zeroed by the call.  With shards, the caller sums d_J/d_Cx/d_rhs over ranks (RCCL all-reduce). */
int mc_evaluate_error_device(mc_context* ctx, int use_mestimator, double* d_J, double* d_e, void* stream);
int mc_build_system_device(mc_context* ctx, int use_mestimator, double* d_Cx, double* d_rhs, void* stream);
/* Both in ONE pass over the terms.  Optimizer2::optimize (Optimizer2.cpp:183-275) calls evaluateError at the updated state
 * and, when the step is accepted, buildSystem at that same state in the next iteration; the build kernels already form
 * every residual, so a caller that builds speculatively (into a second value buffer: a rejected Levenberg-Marquardt step
 * keeps the previous system, LevenbergMarquardtTrustRegionPolicy.cpp:61-72) gets J, e, Cx and rhs from one set of
 * launches.  Results are those of the two calls above made one after the other.  d_e may be NULL. */
int mc_evaluate_error_and_build_system_device(mc_context* ctx, int use_mestimator, double* d_J, double* d_e, double* d_Cx,
                                              double* d_rhs, void* stream);
/* Allocation-free variant of mc_build_system_device.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * kernels, same results and same arguments as mc_build_system_device, but the scratch the build needs (the 16 replicas of
 * the contended global x global Hessian / rhs entries) lives in a file-scope __device__ array of the library (4.05 MiB,
 * held by every GPU the library runs on until the process ends) whose address is looked up once per device in mc_create,
 * every launch stays on `stream` (no side streams, no events are created; while `stream` is being captured the call
 * records no event either, and the graph's owner orders its replays against the null-stream entry points such as
 * mc_apply_update and mc_device_status itself), and the call makes no hipMalloc / hipFree / allocating library call.  Limit: at most 256 scalar columns of global (non-spline) design variables; more is rejected
 * with -1 and a message (problems with more than 512 such columns use no replicas and are accepted).  All contexts of a
 * process on one GPU share that GPU's array and the call clears what it uses: calls that overlap in time on one GPU
 * (different streams or threads) race on it -- serialise them. */
int mc_build_system_device_noalloc(mc_context* ctx, int use_mestimator, double* d_Cx, double* d_rhs, void* stream);
/* ---- second backend: J^T for SparseCholeskyLinearSystemSolver -------------------------------------------------
 * SparseCholeskyLinearSystemSolver::initMatrixStructureImplementation / buildSystem
 * (aslam_backend/src/SparseCholeskyLinearSystemSolver.cpp:15-45) keep the Jacobian itself: J^T (j_cols x j_rows) in
 * compressed-column form, one column per residual row whose entries are the term's active design variables in
 * block-index order, structural zeros of padded spline coefficients included
 * (CompressedColumnJacobianTransposeBuilder.hpp:20-56, CompressedColumnMatrix.hpp:236-304).  mc_jt_nnz: entry count
 * (-2: the problem holds a BSplineMotionError, whose evaluateJacobians is "currently unsupported" in the reference too);
 * mc_get_jt_pattern: col_ptr[j_rows + 1], row_ind[nnz]; mc_build_jacobian_transpose(_device): the weighted values.
 * The caller forms rhs = J^T e with e from mc_evaluate_error, like the reference (rightMultiply). */
int mc_jt_nnz(const mc_context* ctx, long long* nnz);
int mc_get_jt_pattern(mc_context* ctx, int* col_ptr, int* row_ind);
int mc_build_jacobian_transpose(mc_context* ctx, int use_mestimator, double* values_out);
int mc_build_jacobian_transpose_device(mc_context* ctx, int use_mestimator, double* d_values, void* stream);

/* per-term raw chi^2 of the last mc_evaluate_error (ErrorTerm::getRawSquaredError) */
int mc_get_term_chi2(mc_context* ctx, double* chi2_out);

/* Optimizer2::applyStateUpdate / revertLastStateUpdate (Optimizer2.cpp:292-320): dx has j_cols entries. */
int mc_apply_update(mc_context* ctx, const double* dx);
int mc_revert_update(mc_context* ctx);
/* current parameters of a design variable (DesignVariable::getParameters); returns the count.
 * EUCLIDEAN_DIRECTION reports its 3x3 rotation (row major) followed by the magnitude. */
int mc_get_dv_params(mc_context* ctx, int dv, double* out);
/* LiDAR.findPointsOnTarget (kalibr_sensor_calibration/sensors_and_targets.py:199-235), the data-association step the
 * front-end re-runs before every LiDAR optimisation (calibrator.py:70-89): transformMeasurementsToWorldFrame + _onPlane,
 * a per-reading Python loop there.  Evaluated at the CURRENT state held by the context (after mc_finalize / mc_optimize).
 * points[n,4] = x y z t in the sensor frame; plane_min / plane_max [n_planes,3] = the targets' grid ranges
 * (CalibrationTarget.range, :111-114); threshold = slack along the plane normal.  mask_out[n]: bit p set when reading i
 * lies inside target p's box, bit 31 set when its time lies strictly inside the spline (the readings the reference
 * keeps; others get no target bits).  world_out (optional, [n,3]): the readings in the world frame (LiDAR.pointCloud). */
int mc_lidar_find_points_on_targets(mc_context* ctx, int pose_spline, int dv_time_offset, int dv_q_l_b, int dv_t_l_b,
                                    int n_planes, const int* dv_q_p_w, const int* dv_t_p_w, const double* plane_min,
                                    const double* plane_max, double threshold, int n, const double* points,
                                    unsigned* mask_out, double* world_out);
/* asynchronous device errors raised by the last kernels (e.g. "Spline Coefficient Buffer Exceeded",
 * BSplineExpressions.cpp:447): 0 = none. */
int mc_device_status(mc_context* ctx);

/* Measurement aids.  The reference's phase timers ("Optimizer2: build Hessian", "...evaluate error",
 * aslam_backend/src/Optimizer2.cpp:185-189) are compiled out (Optimizer2.hpp:53-55).  Here every term family is ONE
 * kernel launch shared by all its batches: launch 0 reprojection (F1), 1 static reprojection (F2), 2 IMU (F3 + F4),
 * 3 lidar (F5), 4 motion (F7), 5 IMU with the scale-misalignment models.  mc_build_system_timed runs one evaluateError (build = 0) or buildSystem (build = 1)
 * on the internal buffers with a HIP event pair around every launch and writes mc_num_launches() milliseconds.
 * mc_launch_counts: SURVEY.md 8(d) accounting of this rank's share -- buildSystem flops with every structurally
 * present block (padded zeros included, as the reference computes them), with the live blocks only, and the
 * compulsory record bytes. */
int mc_num_launches(const mc_context* ctx);
int mc_launch_info(const mc_context* ctx, int launch, int* n_batches, int* n_terms, int* n_workgroups);
int mc_launch_counts(const mc_context* ctx, int launch, double* flops_dense, double* flops_live, double* record_bytes);
int mc_build_system_timed(mc_context* ctx, int use_mestimator, int build, float* ms_per_launch);
/* In-kernel phase timing: one pass with a shader-clock mark after every phase of k_reproj (slots 0..6) and k_stream
 * (lidar 8..12, IMU 16..20), summed over workgroups (wave 0 of each).  out24: 24 cycle sums. */
int mc_phase_cycles(mc_context* ctx, int use_mestimator, int build, unsigned long long* out24);

/* number of multiprocessors etc. for reports; returns 0 and fills what it can */
int mc_device_info(mc_context* ctx, char* name, int name_len, int* n_cu, double* hbm_gib);

/* ---- host side above the device entry points (plain C++, multical_amd/csrc/mc_host.cpp) ------------------------
 * The optimiser loop that calls the hot path and the sparse Cholesky that consumes its result stay on the host
 * (north_star).  They are restated here so the backend runs end to end without the reference's build tree; an
 * integrator who keeps aslam::backend::Optimizer2 and CHOLMOD needs none of this section. */

/* scalar column base of every design-variable block in block order, n_blocks + 1 entries (what
 * BlockCholeskyLinearSystemSolver::initMatrixStructureImplementation passes to SparseBlockMatrix as cumulative block
 * sizes, BlockCholeskyLinearSystemSolver.cpp:47-54, shifted to start at 0) */
int mc_get_block_bases(const mc_context* ctx, int* base);
/* per block: the end of the support of the spline coefficient it belongs to, NaN for every other block (n_blocks entries).
 * The coordinate along which the normal equations are banded; sorting by it eliminates a bias coefficient right after the
 * last pose coefficient it is coupled to.  mc_cholesky_set_block_coordinates uses it. */
int mc_get_block_times(const mc_context* ctx, double* t);
/* per block: index of the spline (order of the mc_add_spline calls) whose coefficient the block is, -1 for every other
 * block (n_blocks entries).  The factorisation's GPU stage cuts the band of the spline with the most blocks. */
int mc_get_block_splines(const mc_context* ctx, int* spline);

/* sparse_block_matrix::LinearSolverCholmod<MatrixXd> (sparse_block_matrix/linear_solver_cholmod.h:55-112): analyze once
 * per structure, then factorize + solve per call.  The matrix is the upper-triangular scalar CCS that fillCCS emits;
 * block_base (n_blocks + 1 entries, may be NULL) lets the ordering and the factor work on whole blocks.
 * mc_cholesky_factorize returns 0, 1 = not positive definite (LinearSolverCholmod::solve returns false, :91-100), or
 * -1 on misuse. */
typedef struct mc_cholesky mc_cholesky;
mc_cholesky* mc_cholesky_create(void);
void mc_cholesky_destroy(mc_cholesky* h);
const char* mc_cholesky_last_error(const mc_cholesky* h);
int mc_cholesky_analyze(mc_cholesky* h, int n, const int* Cp, const int* Ci, int n_blocks, const int* block_base);
/* Optional, after mc_cholesky_analyze: a coordinate per block along which the matrix is banded (NaN: none) and the number
 * of threads.  The blocks are cut into `threads` intervals; what couples two intervals, the blocks without a coordinate
 * and the nearly dense blocks form a small top system, the rest falls into independent parts that are factorised and
 * solved by one thread each (std::thread).  Without this call everything runs on the calling thread.  The symbolic work
 * is done at the first mc_cholesky_factorize, on the blocks that hold a non-zero value there (the assembled pattern
 * includes the numerically zero blocks of padded spline coefficients); it is redone if a later matrix has a value
 * outside them. */
int mc_cholesky_set_block_coordinates(mc_cholesky* h, const double* coord /* n_blocks */, int threads);
int mc_cholesky_factorize(mc_cholesky* h, const double* Cx);
int mc_cholesky_solve(mc_cholesky* h, const double* b, double* x);
/* LinearSolverCholmod::solvePattern (linear_solver_cholmod.h:114-165), what BlockCholeskyLinearSystemSolver::
 * computeCovarianceBlocks (BlockCholeskyLinearSystemSolver.cpp:122-148) calls for --recover-covariance: blocks
 * (block_row[k], block_col[k]) of the inverse of the last factorised matrix, block k row major at out + offsets[k]. */
int mc_cholesky_covariance_blocks(mc_cholesky* h, int n_pairs, const int* block_row, const int* block_col,
                                  const long long* offsets, double* out);
int mc_cholesky_stats(const mc_cholesky* h, int* n_dense_blocks, long long* nnz_factor, double* flops);
int mc_cholesky_block_permutation(const mc_cholesky* h, int* perm /* n_blocks */);

/* The factorisation's GPU stage (no counterpart in the reference, whose linear_solver_cholmod.h:70-112 factorises the whole
 * matrix on one host core; INTEGRATION.md "Solving on the device").  The coefficients of the spline with the most blocks are
 * cut into `parts` intervals (0: a default) with r separator blocks between neighbours; the interiors are eliminated on the
 * device (banded Cholesky + dense products per interior), the reduced system -- separators, the other splines'
 * coefficients, calibration parameters -- goes to the host factorisation above, the interiors are back-substituted on the
 * device.  mc_schur_create plans on the pattern of ctx and the NON-ZERO blocks of d_Cx (device, nnz values, copied to the
 * host once); it returns a handle even when the problem has no such band (then mc_schur_planned is 0 and
 * mc_schur_last_error says why: the caller keeps mc_cholesky_*), NULL only on misuse.
 * mc_schur_solve: (H + diag_shift I) x = rhs with H, rhs in device memory, x (JCols) to the host.  Returns 0; 1 = not
 * positive definite; 2 = a block that was zero when the plan was made holds a value now (destroy, create again on the new
 * values); -1 on errors.  The call synchronises the stream.
 * mc_schur_info: counts = [interiors, r, block dim, columns, reduced columns, reduced nnz, watched zero blocks, largest
 * boundary, largest interior, nnz of the reduced factor] (10 ints), seconds = [gather, eliminate, products, read-back, host
 * factor + solve, back-substitution, planning, bytes of device work memory, flops of the reduced factorisation] (9 doubles)
 * of the last solve. */
typedef struct mc_schur mc_schur;
mc_schur* mc_schur_create(mc_context* ctx, const double* d_Cx, int parts);
void mc_schur_destroy(mc_schur* h);
const char* mc_schur_last_error(const mc_schur* h);
int mc_schur_planned(const mc_schur* h);
int mc_schur_solve(mc_schur* h, const double* d_Cx, const double* d_rhs, double diag_shift, double* x_host, void* stream);
int mc_schur_info(const mc_schur* h, int* counts, double* seconds);

/* aslam::backend::Optimizer2Options (aslam_backend/include/aslam/backend/Optimizer2Options.hpp:9-42) with the two
 * shared_ptr members replaced by selectors; mc_optimizer_options_default fills the reference's defaults. */
enum { MC_POLICY_LEVENBERG_MARQUARDT = 0, MC_POLICY_GAUSS_NEWTON = 1 };
enum { MC_SOLVER_BLOCK_CHOLESKY = 0, MC_SOLVER_SPARSE_CHOLESKY = 1 };
typedef struct mc_optimizer_options {
  double convergenceJDescentRatioThreshold; /* 1e-5 */
  double convergenceDeltaX;                 /* 1e-3 */
  int maxIterations;                        /* 20 */
  int verbose;                              /* 0 */
  int nThreads;                             /* 4; a hint the GPU backend ignores */
  int trustRegionPolicy;                    /* MC_POLICY_*: LevenbergMarquardt / GaussNewton TrustRegionPolicy */
  double levenbergMarquardtLambdaInit;      /* 1e-3 (LevenbergMarquardtTrustRegionPolicy.cpp:7-13); Multical uses 10 */
  int linearSolver;                         /* MC_SOLVER_*: BlockCholesky / SparseCholesky LinearSystemSolver */
} mc_optimizer_options;
/* SolutionReturnValue (aslam_backend_python/src/Optimizer.cpp:30-39) plus wall-clock per phase and the size of the host
 * factorisation.  The minimiser (multical_amd/csrc/host/lm_driver.hpp) evaluates every trial state with ONE fused pass
 * (cost + normal equations into a spare device buffer) and copies a system to the host only when it is adopted:
 * seconds_evaluate_error = the fused trial passes, seconds_build_system = adoptions (device-to-host copies),
 * seconds_solve_system = host factorisation + solve. */
typedef struct mc_solution_return_value {
  double JStart, JFinal;
  int iterations, failedIterations;
  double dXFinal, dJFinal;
  int linearSolverFailure;
  double seconds_initialize, seconds_evaluate_error, seconds_build_system, seconds_solve_system, seconds_state_update;
  double factor_nnz, factor_flops;
} mc_solution_return_value;
void mc_optimizer_options_default(mc_optimizer_options* o);
/* What Optimizer2::optimize (Optimizer2.cpp:183-275) computes with a Levenberg-Marquardt or Gauss-Newton policy -- same
 * iterates, same stopping rule, same counters -- on the problem held by ctx; the state in ctx is the result
 * (mc_get_dv_params).  Returns 0, or -1 with the message in mc_host_last_error(). */
int mc_optimize(mc_context* ctx, const mc_optimizer_options* options, mc_solution_return_value* out);
/* The same loop over time-window shards (mc_set_shard: one process per GPU, every context holds its window's terms, the
 * state is replicated): Optimizer2::optimize (Optimizer2.cpp:183-275) with applyStateUpdate / revertLastStateUpdate
 * (:292-320) applied by every rank to its replica.  The caller supplies the two collectives: `allreduce` sums n doubles of
 * DEVICE memory over the ranks in place (RCCL; the Python binding passes torch.distributed.all_reduce), `broadcast`
 * sends n doubles of HOST memory from rank `root` to all; both return 0 on success and are called in the same order on
 * every rank.  Per trial step the cost (1 double) is summed; the partial value array and gradient are summed only when a
 * trial's linearisation is adopted; rank 0 factorises and solves, the step is broadcast, every rank applies it.  Same
 * accepted / rejected sequence and estimates as mc_optimize on the unsharded problem (tests/test_sharding_gloo.py).  With
 * world == 1 the callbacks are not used and may be NULL. */
typedef int (*mc_allreduce_sum_fn)(void* user, double* device_buf, long long n);
typedef int (*mc_broadcast_fn)(void* user, double* host_buf, long long n, int root);
int mc_optimize_sharded(mc_context* ctx, const mc_optimizer_options* options, mc_solution_return_value* out,
                        mc_allreduce_sum_fn allreduce, mc_broadcast_fn broadcast, void* user);
const char* mc_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MULTICAL_HIP_H */
