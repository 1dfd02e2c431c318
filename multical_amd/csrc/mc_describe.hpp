// mc_describe.hpp -- validation + recording of error-term batches (host only, no HIP).
// Mirrors what the Python front-end feeds to problem.addErrorTerm one term at a time
// (sensors_and_targets.py:274-305, 556-630, 1133-1246; CameraCalibrator.py:195-219).
#pragma once
#include "mc_core.hpp"

namespace mc {

inline void requireOpen(const Problem& P) {
  if (P.finalized) throw std::logic_error("the problem structure is frozen after mc_finalize");
}
inline void checkMest(int kind) {
  if (kind < 0 || kind > 5) throw std::invalid_argument("unknown M-estimator kind");
}

inline void describeReprojection(Problem& P, int cam, int spline, int dv_q_cb, int dv_t_cb, int dv_toff, int n_obs,
                                 const double* obs_time, const int* obs_q, const int* obs_t, const int* obs_first,
                                 const double* y, const double* p, const double* sinv, double pad, int mest, double mest_a) {
  requireOpen(P);
  if (cam < 0 || cam >= (int)P.cams.size()) throw std::invalid_argument("camera id out of range");
  P.checkSpline(spline, 6);
  P.checkDv(dv_q_cb, DV_ROTATION_QUATERNION); P.checkDv(dv_t_cb, DV_EUCLIDEAN_POINT);
  if (dv_toff >= 0) P.checkDv(dv_toff, DV_SCALAR);
  checkMest(mest);
  if (n_obs < 0 || obs_first[0] != 0) throw std::invalid_argument("obs_first must start at 0");
  Batch b;
  b.kind = KIND_REPROJ; b.dim = 2; b.cam = cam; b.spline = spline;
  b.dv[0] = dv_q_cb; b.dv[1] = dv_t_cb; b.dv[2] = dv_toff;
  b.pad = pad; b.mest = mest; b.mest_a = mest_a;
  for (int o = 0; o < n_obs; ++o) {
    if (obs_first[o + 1] < obs_first[o]) throw std::invalid_argument("obs_first must be nondecreasing");
    P.checkDv(obs_q[o], DV_ROTATION_QUATERNION); P.checkDv(obs_t[o], DV_EUCLIDEAN_POINT);
  }
  int n = obs_first[n_obs];
  b.n_terms = n;
  b.t.assign(obs_time, obs_time + n_obs);
  b.obs_dv_q.assign(obs_q, obs_q + n_obs); b.obs_dv_t.assign(obs_t, obs_t + n_obs);
  b.obs_first.assign(obs_first, obs_first + n_obs + 1);
  b.y.assign(y, y + 2 * (size_t)n); b.p.assign(p, p + 3 * (size_t)n); b.sinv.assign(sinv, sinv + n);
  P.batches.push_back(std::move(b));
}

inline void describeStaticReprojection(Problem& P, int cam, int n_chain, const int* chain_q, const int* chain_t, int n_obs,
                                       const int* obs_q, const int* obs_t, const int* obs_first, const double* y,
                                       const double* p, const double* sinv, int mest, double mest_a) {
  requireOpen(P);
  if (cam < 0 || cam >= (int)P.cams.size()) throw std::invalid_argument("camera id out of range");
  if (n_chain < 0 || n_chain > F2_MAX_CHAIN) throw std::invalid_argument("baseline chain longer than 4 links");
  checkMest(mest);
  if (n_obs < 0 || obs_first[0] != 0) throw std::invalid_argument("obs_first must start at 0");
  Batch b;
  b.kind = KIND_STATIC_REPROJ; b.dim = 2; b.cam = cam; b.mest = mest; b.mest_a = mest_a;
  for (int j = 0; j < n_chain; ++j) {
    P.checkDv(chain_q[j], DV_ROTATION_QUATERNION); P.checkDv(chain_t[j], DV_EUCLIDEAN_POINT);
    b.chain_q.push_back(chain_q[j]); b.chain_t.push_back(chain_t[j]);
  }
  int n = obs_first[n_obs];
  b.n_terms = n;
  b.term_obs.resize(n);
  for (int o = 0; o < n_obs; ++o) {
    if (obs_first[o + 1] < obs_first[o]) throw std::invalid_argument("obs_first must be nondecreasing");
    P.checkDv(obs_q[o], DV_ROTATION_QUATERNION); P.checkDv(obs_t[o], DV_EUCLIDEAN_POINT);
    for (int k = obs_first[o]; k < obs_first[o + 1]; ++k) b.term_obs[k] = o;
  }
  b.obs_dv_q.assign(obs_q, obs_q + n_obs); b.obs_dv_t.assign(obs_t, obs_t + n_obs);
  b.obs_first.assign(obs_first, obs_first + n_obs + 1);
  b.y.assign(y, y + 2 * (size_t)n); b.p.assign(p, p + 3 * (size_t)n); b.sinv.assign(sinv, sinv + n);
  P.batches.push_back(std::move(b));
}

// GyroscopeError over constants (findExtrinsicPriorSensorsToCamera, sensors_and_targets.py:858-879)
inline void describeGyroPrior(Problem& P, int dv_q, int dv_bias, int n, const double* w_pred, const double* meas, const double* sinv,
                              int mest, double mest_a) {
  requireOpen(P);
  P.checkDv(dv_q, DV_ROTATION_QUATERNION); P.checkDv(dv_bias, DV_EUCLIDEAN_POINT);
  checkMest(mest);
  if (n < 0) throw std::invalid_argument("negative term count");
  Batch b;
  b.kind = KIND_GYRO_PRIOR; b.dim = 3; b.n_terms = n; b.mest = mest; b.mest_a = mest_a;
  b.dv[0] = dv_q; b.dv[1] = dv_bias;
  b.p.assign(w_pred, w_pred + 3 * (size_t)n); b.meas.assign(meas, meas + 3 * (size_t)n); b.sinv.assign(sinv, sinv + n);
  P.batches.push_back(std::move(b));
}
// ErrorTermTransformation (ABE/src/ErrorTermTransformation.cpp:19-33) on T_a or T_a * inverse(T_b); prior as quaternion
// [x y z w] + translation per term, like sm::kinematics::Transformation holds it
inline void describeTransformation(Problem& P, int dv_qa, int dv_ta, int dv_qb, int dv_tb, int n, const double* prior_q,
                                   const double* prior_t, double weight_rotation, double weight_translation, int mest, double mest_a) {
  requireOpen(P);
  P.checkDv(dv_qa, DV_ROTATION_QUATERNION); P.checkDv(dv_ta, DV_EUCLIDEAN_POINT);
  if ((dv_qb >= 0) != (dv_tb >= 0)) throw std::invalid_argument("the second transformation needs both its rotation and its translation");
  if (dv_qb >= 0) { P.checkDv(dv_qb, DV_ROTATION_QUATERNION); P.checkDv(dv_tb, DV_EUCLIDEAN_POINT); }
  checkMest(mest);
  if (n < 0) throw std::invalid_argument("negative term count");
  if (!(weight_rotation > 0.0) || !(weight_translation > 0.0)) throw std::invalid_argument("weights must be positive");
  Batch b;
  b.kind = KIND_TRANSFORMATION; b.dim = 6; b.n_terms = n; b.mest = mest; b.mest_a = mest_a;
  b.dv[0] = dv_qa; b.dv[1] = dv_ta; b.dv[2] = dv_qb; b.dv[3] = dv_tb;
  b.w_t = weight_translation; b.w_r = weight_rotation;
  b.y.resize(12 * (size_t)n);
  for (int k = 0; k < n; ++k) {   // inverse(prior): rotation C^T, translation -C^T t
    Rigid Tp, Ti;
    quat2r(prior_q + 4 * (size_t)k, Tp.C);
    Tp.r[0] = prior_t[3 * (size_t)k]; Tp.r[1] = prior_t[3 * (size_t)k + 1]; Tp.r[2] = prior_t[3 * (size_t)k + 2];
    rigid_inv(Tp, Ti);
    for (int i = 0; i < 9; ++i) b.y[12 * (size_t)k + i] = Ti.C[i];
    for (int i = 0; i < 3; ++i) b.y[12 * (size_t)k + 9 + i] = Ti.r[i];
  }
  b.sinv.assign(n, 1.0);
  P.batches.push_back(std::move(b));
}

inline void describeImu(Problem& P, bool accel, int pose_spline, int bias_spline, int dv_bias, int dv_g, int dv_q_ib, int dv_r_b,
                        int n, const double* t, const double* meas, const double* sinv, int mest, double mest_a) {
  requireOpen(P);
  P.checkSpline(pose_spline, 6);
  if (bias_spline >= 0) P.checkSpline(bias_spline, 3);
  else P.checkDv(dv_bias, DV_EUCLIDEAN_POINT);
  P.checkDv(dv_q_ib, DV_ROTATION_QUATERNION);
  if (accel) {
    P.checkDv(dv_g, -1);
    if (P.dvs[dv_g].kind != DV_EUCLIDEAN_DIRECTION && P.dvs[dv_g].kind != DV_EUCLIDEAN_POINT)
      throw std::invalid_argument("gravity must be an EuclideanDirection or an EuclideanPoint");
    P.checkDv(dv_r_b, DV_EUCLIDEAN_POINT);
  }
  checkMest(mest);
  Batch b;
  b.kind = accel ? KIND_ACCEL : KIND_GYRO; b.dim = 3; b.spline = pose_spline; b.bias_spline = bias_spline;
  b.dv[0] = bias_spline >= 0 ? -1 : dv_bias; b.dv[1] = accel ? dv_g : -1; b.dv[2] = dv_q_ib; b.dv[3] = accel ? dv_r_b : -1;
  b.mest = mest; b.mest_a = mest_a; b.n_terms = n;
  b.t.assign(t, t + n); b.meas.assign(meas, meas + 3 * (size_t)n); b.sinv.assign(sinv, sinv + n);
  P.batches.push_back(std::move(b));
}

// ScaledMisalignedImu / ScaledMisalignedSizeEffectImu (sensors_and_targets.py:1393-1661).  size_effect: NULL or the six
// design variables rx_i, ry_i, rz_i (EuclideanPoint) and Ix, Iy, Iz (MatrixBasic).
inline void describeImuSm(Problem& P, bool accel, int pose_spline, int bias_spline, int dv_bias, int dv_g, int dv_q_ib, int dv_r_b,
                          int dv_M, int dv_q_gi, int dv_Ma, const int* size_effect, int n, const double* t, const double* meas,
                          const double* sinv, int mest, double mest_a) {
  requireOpen(P);
  P.checkSpline(pose_spline, 6);
  if (bias_spline >= 0) P.checkSpline(bias_spline, 3);
  else P.checkDv(dv_bias, DV_EUCLIDEAN_POINT);
  P.checkDv(dv_q_ib, DV_ROTATION_QUATERNION);
  P.checkDv(dv_g, -1);
  if (P.dvs[dv_g].kind != DV_EUCLIDEAN_DIRECTION && P.dvs[dv_g].kind != DV_EUCLIDEAN_POINT)
    throw std::invalid_argument("gravity must be an EuclideanDirection or an EuclideanPoint");
  P.checkDv(dv_r_b, DV_EUCLIDEAN_POINT);
  P.checkDv(dv_M, DV_MATRIX_BASIC);
  if (!accel) { P.checkDv(dv_q_gi, DV_ROTATION_QUATERNION); P.checkDv(dv_Ma, DV_MATRIX_BASIC); }
  if (size_effect) {
    if (!accel) throw std::invalid_argument("the size-effect model changes the accelerometer terms only");
    for (int j = 0; j < 3; ++j) { P.checkDv(size_effect[j], DV_EUCLIDEAN_POINT); P.checkDv(size_effect[3 + j], DV_MATRIX_BASIC); }
  }
  checkMest(mest);
  Batch b;
  b.kind = accel ? KIND_ACCEL_SM : KIND_GYRO_SM; b.dim = 3; b.spline = pose_spline; b.bias_spline = bias_spline;
  b.dv[0] = bias_spline >= 0 ? -1 : dv_bias; b.dv[1] = dv_g; b.dv[2] = dv_q_ib; b.dv[3] = dv_r_b; b.dv[4] = dv_M;
  if (!accel) { b.dv[5] = dv_q_gi; b.dv[6] = dv_Ma; }
  if (size_effect) { b.size_effect = 1; for (int j = 0; j < 6; ++j) b.dv[7 + j] = size_effect[j]; }
  b.mest = mest; b.mest_a = mest_a; b.n_terms = n;
  b.t.assign(t, t + n); b.meas.assign(meas, meas + 3 * (size_t)n); b.sinv.assign(sinv, sinv + n);
  P.batches.push_back(std::move(b));
}

inline void describeLidar(Problem& P, int spline, int dv_toff, int dv_q_lb, int dv_t_lb, int n, const double* t,
                          const double* dir, const double* range, const double* sinv, const int* dv_q_pw, const int* dv_t_pw,
                          double pad, int mest, double mest_a) {
  requireOpen(P);
  P.checkSpline(spline, 6);
  P.checkDv(dv_q_lb, DV_ROTATION_QUATERNION); P.checkDv(dv_t_lb, DV_EUCLIDEAN_POINT);
  if (dv_toff >= 0) P.checkDv(dv_toff, DV_SCALAR);
  checkMest(mest);
  Batch b;
  b.kind = KIND_LIDAR; b.dim = 1; b.spline = spline;
  b.dv[0] = dv_q_lb; b.dv[1] = dv_t_lb; b.dv[2] = dv_toff;
  b.pad = pad; b.mest = mest; b.mest_a = mest_a; b.n_terms = n;
  for (int k = 0; k < n; ++k) { P.checkDv(dv_q_pw[k], DV_ROTATION_QUATERNION); P.checkDv(dv_t_pw[k], DV_EUCLIDEAN_POINT); }
  b.t.assign(t, t + n); b.p.assign(dir, dir + 3 * (size_t)n); b.range.assign(range, range + n); b.sinv.assign(sinv, sinv + n);
  b.term_dv_q.assign(dv_q_pw, dv_q_pw + n); b.term_dv_t.assign(dv_t_pw, dv_t_pw + n);
  P.batches.push_back(std::move(b));
}

// BSplineMotionError: Q = sum_seg M^T (W (x) V^(d)) M scattered into (seg+i, seg+j) blocks
// (BSP/src/BSpline.cpp:1276-1302, 1483-1548, 1585-1622).  Because M only mixes the S basis functions, block
// (i, j) of a segment's Q equals  (B^T V_d B)(i, j) * W  with B the segment's S x S basis matrix.
// ErrorTermFs<D>::setSqrtInvR for the terms of the batch added last (general, per term)
inline void describeSqrtInvR(Problem& P, int dim, int n, const double* S) {
  requireOpen(P);
  if (P.batches.empty()) throw std::invalid_argument("no error-term batch has been added");
  Batch& b = P.batches.back();
  if (b.kind == KIND_MOTION || b.kind == KIND_TRANSFORMATION)
    throw std::invalid_argument("this term type carries its own weighting (W / weightRotation, weightTranslation)");
  if (dim != b.dim || n != b.n_terms) throw std::invalid_argument("The covariance matrix does not match the size of the error");
  if (!S) throw std::invalid_argument("S is NULL");
  if (dim == 1) b.sinv.assign(S, S + n);
  else b.S.assign(S, S + (size_t)n * dim * dim);
}
// ReprojectionError over landmarks that are HomogeneousPoint design variables (kalibr_camera_calibration with
// estimateLandmarks: the target corners become HomogeneousPointDv, CameraCalibrator.py:195-219): for the static
// reprojection batch added last, per term the landmark's dv id, or -1 to keep the constant p_target.
inline void describeLandmarks(Problem& P, int n, const int* dv_lm) {
  requireOpen(P);
  if (P.batches.empty() || P.batches.back().kind != KIND_STATIC_REPROJ)
    throw std::invalid_argument("landmark design variables follow a static reprojection batch");
  Batch& b = P.batches.back();
  if (n != b.n_terms || !dv_lm) throw std::invalid_argument("one landmark id (or -1) per term of the batch");
  // the kernels carry ONE landmark scaling per batch (StaticReprojBatchView::lm_scale scales every landmark column of the
  // batch), k_update applies each design variable's own: landmarks of one batch must share their DesignVariable::scaling()
  double scaling = 0.0;
  bool have = false;
  for (int k = 0; k < n; ++k) {
    if (dv_lm[k] < 0) continue;
    P.checkDv(dv_lm[k], DV_HOMOGENEOUS_POINT);
    const double s = P.dvs[dv_lm[k]].scaling;
    if (have && s != scaling) throw std::invalid_argument("the landmark design variables of one batch must share one scaling()");
    scaling = s; have = true;
  }
  b.term_dv_lm.assign(dv_lm, dv_lm + n);
}
inline void describeMotion(Problem& P, int spline, const double* W, int order) {
  requireOpen(P);
  if (spline < 0 || spline >= (int)P.splines.size()) throw std::invalid_argument("spline id out of range");
  const SplineRec& s = P.splines[spline];
  int S = s.order, D = s.dim;
  if (S <= order && S >= 2) order = S - 1;
  if (order < 0) throw std::invalid_argument("negative error term order");
  for (int r = 0; r < D; ++r) for (int c = 0; c < D; ++c)
    if (std::fabs(W[r * D + c] - W[c * D + r]) > 1e-14) throw std::invalid_argument("W must be symmetric");
  Batch b;
  b.kind = KIND_MOTION; b.dim = 1; b.n_terms = 1; b.spline = spline; b.motion_order = order;
  b.Qband.assign((size_t)s.n_coeff * S * D * D, 0.0);
  std::vector<double> V((size_t)S * S), Dm((size_t)S * S), T((size_t)S * S), q((size_t)S * S);
  for (int seg = 0; seg < s.n_seg; ++seg) {
    double dt = P.knots[s.knots_off + S + seg] - P.knots[s.knots_off + S + seg - 1];
    double recip = dt > 0 ? 1.0 / dt : 0.0;
    for (int r = 0; r < S; ++r) for (int c = 0; c < S; ++c) { V[r * S + c] = dt / (r + c + 1.0); Dm[r * S + c] = 0.0; }
    for (int i = 0; i < S - 1; ++i) Dm[i * S + i + 1] = (i + 1.0) * recip;
    for (int it = 0; it < order; ++it) {  // V <- Dm^T V Dm
      for (int r = 0; r < S; ++r) for (int c = 0; c < S; ++c) { double a = 0; for (int k = 0; k < S; ++k) a += Dm[k * S + r] * V[k * S + c]; T[r * S + c] = a; }
      for (int r = 0; r < S; ++r) for (int c = 0; c < S; ++c) { double a = 0; for (int k = 0; k < S; ++k) a += T[r * S + k] * Dm[k * S + c]; V[r * S + c] = a; }
    }
    const double* B = &P.basis[s.basis_off + (size_t)seg * S * S];
    for (int r = 0; r < S; ++r) for (int c = 0; c < S; ++c) { double a = 0; for (int k = 0; k < S; ++k) a += V[r * S + k] * B[k * S + c]; T[r * S + c] = a; }
    for (int r = 0; r < S; ++r) for (int c = 0; c < S; ++c) { double a = 0; for (int k = 0; k < S; ++k) a += B[k * S + r] * T[k * S + c]; q[r * S + c] = a; }
    for (int i = 0; i < S; ++i)
      for (int j = 0; j <= i; ++j) {  // lower band: Q(seg+i, seg+j), offset d = i - j
        double* blk = &b.Qband[((size_t)(seg + i) * S + (i - j)) * D * D];
        for (int r = 0; r < D; ++r) for (int c = 0; c < D; ++c) blk[r * D + c] += q[i * S + j] * W[r * D + c];
      }
  }
  P.batches.push_back(std::move(b));
}

}  // namespace mc
