// mc_terms.h -- closed-form residual + Jacobian of every Multical error-term kind.
//
// The reference evaluates a term by walking an expression DAG twice (value, then reverse chain rule
// with heap-allocated MatrixXd products: SURVEY.md section 3.5).  Here each kind is lowered once, by
// hand, into straight-line fp64 code: the chain of boxTimes / boxMinus / cross products collapses into
// a few 3-vector operations per Jacobian row.  Comments give the DAG the formulas come from.
//
// Output convention.  A term writes its weighted Jacobian rows into `J` (row stride `ld`) at column
// `rolecol[role] + c`; a negative rolecol drops the block (inactive design variable -- what
// JacobianContainer::add does, aslam_backend/include/aslam/backend/JacobianContainer.hpp:124-126).
// "Weighted" means  sqrt(w(chi2)) * sqrtInvR^T * J * dv.scaling()  exactly as
// ErrorTermFs::buildHessianImplementation + JacobianContainer::evaluateHessian apply it
// (aslam_backend/include/aslam/backend/implementation/ErrorTerm.hpp:96-109, src/JacobianContainer.cpp:151-167).
// The kernels point `J` at an LDS tile whose columns follow the group's sorted block order; the
// CPU-side probe used by the tests points it at a dense row of the global Jacobian.
#pragma once
#include "mc_math.h"

namespace mc {

enum { MC_MAX_ROLES = 13 };
constexpr int S6 = 6;  // spline order handled by the kernels (Multical: splineOrder = 6 everywhere)

struct DvRef {
  int off;      // parameter offset in the state vector (-1: not present)
  int block;    // block index (-1: inactive)
  int col;      // column base (-1: inactive)
  double scale; // DesignVariable::scaling()
};

struct TermOut {
  double chi2;   // raw squared error  e^T invR e
  double w;      // M-estimator weight w(chi2): always applied to the cost (ErrorTerm::evaluateError, ErrorTerm.cpp:20-24);
                 // the useMEstimator flag only gates the sqrt(w) on the residual / Jacobian rows
};

// ------------------------------------------------------------------------------------------------
// pose spline sample shared by the kinds: value, derivatives, rotation, S matrix
// (BSplinePose::transformation / curveValueToTransformationAndJacobian, BSP/src/BSplinePose.cpp:21-24, 394-412)
struct PoseSample {
  double w[3][S6];   // basis weights for derivative 0,1,2
  double c[6], cd[6], cdd[6];
  double C[9], S[9];  // C_w_b(theta), S(theta)
  int seg;
};
template <int ND>
MC_HD bool pose_sample(const SplineView& sp, const double* x, double t, PoseSample& ps) {
  ps.seg = spline_weights<S6, ND>(sp, t, ps.w);
  if (ps.seg < 0) return false;
  const double* cf = x + sp.param_off + (size_t)ps.seg * 6;
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int i = 0; i < S6; ++i) {
      double v = cf[i * 6 + d];
      a0 += v * ps.w[0][i];
      if (ND > 1) a1 += v * ps.w[1][i];
      if (ND > 2) a2 += v * ps.w[2][i];
    }
    ps.c[d] = a0; ps.cd[d] = a1; ps.cdd[d] = a2;
  }
  rotvec_C_S(ps.c + 3, ps.C, ps.S);
  return true;
}
// [u v] * JT,  JT = [I, -r^ S; 0, S]  (BSplinePose.cpp:402-407)  ->  [u, (v - u x r) S]
MC_HD void chain_JT(const double* u, const double* v, const double* r, const double* S, double* B) {
  double c[3], wv[3];
  v3_cross(u, r, c);
  wv[0] = v[0] - c[0]; wv[1] = v[1] - c[1]; wv[2] = v[2] - c[2];
  B[0] = u[0]; B[1] = u[1]; B[2] = u[2];
  m3_tmulv(S, wv, B + 3);
}
// TransformationBasic leaves (ABE/src/TransformationBasic.cpp:48-67): chain [u v] ->
//   translation block = u ;  rotation block = v - u x r
MC_HD void chain_basic(const double* u, const double* v, const double* r, double* Jq, double* Jt) {
  double c[3];
  v3_cross(u, r, c);
  Jq[0] = v[0] - c[0]; Jq[1] = v[1] - c[1]; Jq[2] = v[2] - c[2];
  Jt[0] = u[0]; Jt[1] = u[1]; Jt[2] = u[2];
}
MC_HD void load_rigid(const double* x, int qoff, int toff, Rigid& T) {
  quat2r(x + qoff, T.C);
  T.r[0] = x[toff]; T.r[1] = x[toff + 1]; T.r[2] = x[toff + 2];
}

// ------------------------------------------------------------------------------------------------
// F1: SimpleReprojectionError with time-offset pose spline
//   p_c = T_c_b * inverse(T_w_b(t0 + d)) * inverse(T_p_w) * p      (sensors_and_targets.py:571-611)
// Core columns (19): [0..5] B = chain into one spline coefficient before the basis weight,
//                    [6..8] q_c_b, [9..11] t_c_b, [12..14] q_p_w, [15..17] t_p_w, [18] time offset.
// The Jacobian wrt local spline coefficient i is w_i * B (TransformationTimeOffsetExpressionNode,
// aslam_splines/src/BSplineExpressions.cpp:495-544) -- every corner of one image shares the w_i, which
// is what lets the kernel reduce a 19x19 core per image instead of an 85x85 block matrix per corner.
enum { F1_CORE = 19 };
// General sqrtInvR of ErrorTermFs<D> (aslam_backend/include/aslam/backend/implementation/ErrorTerm.hpp:96-192):
// weighted error S^T e, weighted Jacobian S^T J, chi2 = e^T (S S^T) e.  S is row major.
MC_HD void sinv_mix2(const double* S, double& a0, double& a1) {
  const double t0 = S[0] * a0 + S[2] * a1, t1 = S[1] * a0 + S[3] * a1;
  a0 = t0; a1 = t1;
}
MC_HD void sinv_mix3(const double* S, double* a, int stride) {
  const double a0 = a[0], a1 = a[stride], a2 = a[2 * stride];
  a[0] = S[0] * a0 + S[3] * a1 + S[6] * a2;
  a[stride] = S[1] * a0 + S[4] * a1 + S[7] * a2;
  a[2 * stride] = S[2] * a0 + S[5] * a1 + S[8] * a2;
}
enum { F1_ROLE_COEF = 0, F1_ROLE_QCB = 1, F1_ROLE_TCB = 2, F1_ROLE_QPW = 3, F1_ROLE_TPW = 4, F1_ROLE_TOFF = 5 };
struct ReprojBatchView {
  CameraView cam;
  SplineView sp;
  DvRef q_cb, t_cb, toff;
  double coef_scale, qpw_scale, tpw_scale;
  const double* obs_time;
  const int* obs_first;
  const int* obs_qpw_off;  // per observation: parameter offsets of the target's q / t
  const int* obs_tpw_off;
  const double* y;
  const double* p;
  const double* sinv;
  const double* S = nullptr;   // optional general sqrtInvR, n x D x D row major (then sinv is not used)
  int mest; double mest_a;
  int n_obs, n_terms;
  int row_base;   // row base of term 0 of this batch
  int term_base_global;  // global index of term 0 of this batch
};
// what every corner of one image observation shares (computed once per observation by k_obs_common)
struct F1Common {
  Rigid Tcp, X;        // T_c_p = X * T_w_p, X = T_c_b * T_b_w
  double rcb[3];       // translation of T_c_b
  double rpw[3];       // translation of T_p_w
  double r[3], S[9];   // body position and S(theta) at t + d
  double cd[6];        // time derivative of the spline value
  double w[S6];        // basis weights of the 6 live coefficients
  double pp[6], dk[4]; // projection / distortion parameters of the camera (copied here so that the corner loop reads
                       // them from the staged record instead of issuing dependent global loads)
  int seg, ok;
};
MC_HD void f1_common(const ReprojBatchView& b, const double* x, int obs, F1Common& cm) {
  double t = b.obs_time[obs] + (b.toff.off >= 0 ? x[b.toff.off] : 0.0);
  PoseSample ps;
  cm.ok = pose_sample<2>(b.sp, x, t, ps) ? 1 : 0;
  cm.seg = ps.seg;
#pragma unroll
  for (int i = 0; i < 6; ++i) cm.pp[i] = i < b.cam.proj_dim ? x[b.cam.proj_off + i] : 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) cm.dk[i] = b.cam.dist_off >= 0 ? x[b.cam.dist_off + i] : 0.0;
  if (!cm.ok) return;
  Rigid Tcb, Twb, Tbw, Tpw, Twp;
  load_rigid(x, b.q_cb.off, b.t_cb.off, Tcb);
#pragma unroll
  for (int i = 0; i < 9; ++i) { Twb.C[i] = ps.C[i]; cm.S[i] = ps.S[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { Twb.r[i] = ps.c[i]; cm.r[i] = ps.c[i]; cm.rcb[i] = Tcb.r[i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) { cm.cd[i] = ps.cd[i]; cm.w[i] = ps.w[0][i]; }
  rigid_inv(Twb, Tbw);
  load_rigid(x, b.obs_qpw_off[obs], b.obs_tpw_off[obs], Tpw);
  cm.rpw[0] = Tpw.r[0]; cm.rpw[1] = Tpw.r[1]; cm.rpw[2] = Tpw.r[2];
  rigid_inv(Tpw, Twp);
  rigid_mul(Tcb, Tbw, cm.X);
  rigid_mul(cm.X, Twp, cm.Tcp);
}
// One corner, first half: camera-frame point, projection (+ 2x3 Jacobian), weighted residual e[2], sw = sqrt(w) * s.
template <bool EQUI = true, bool EXPW = true>
MC_HD void f1_project(const ReprojBatchView& b, const double* x, const F1Common& cm, int k, bool useM, bool wantJ,
                      double* pc, double* Jc, double* e, double& sw, TermOut& out) {
  const double* pt = b.p + 3 * (size_t)k;
  m3_mulv(cm.Tcp.C, pt, pc);
  pc[0] += cm.Tcp.r[0]; pc[1] += cm.Tcp.r[1]; pc[2] += cm.Tcp.r[2];
  double kp[2];
  (void)x;
  project<EQUI>(b.cam.model, b.cam.dist, cm.pp, cm.dk, pc, kp, wantJ ? Jc : nullptr);
  double e0 = b.y[2 * (size_t)k] - kp[0], e1 = b.y[2 * (size_t)k + 1] - kp[1];
  double s;
  if (b.S) {   // general sqrtInvR: mix the residual and the two rows of the projection Jacobian once, everything below is linear in them
    const double* S = b.S + 4 * (size_t)k;
    sinv_mix2(S, e0, e1);
    if (wantJ) { sinv_mix2(S, Jc[0], Jc[3]); sinv_mix2(S, Jc[1], Jc[4]); sinv_mix2(S, Jc[2], Jc[5]); }
    s = 1.0;
  } else s = b.sinv[k];
  out.chi2 = (s * s) * (e0 * e0 + e1 * e1);
  out.w = mest_weight<EXPW>(b.mest, b.mest_a, out.chi2);
  sw = (useM ? sqrt(out.w) : 1.0) * s;
  e[0] = sw * e0; e[1] = sw * e1;
}
// The chain below the camera is LINEAR in z = [a, a x p_c] (a = one weighted row of -J_cam): every node between the
// homogeneous point and the leaves multiplies the incoming 1 x 6 row by a matrix that only depends on the observation
// (T_c_p, X, r, S, the lever arms).  f1_chain_lin is that linear map, z -> 19 core columns.
MC_HD void f1_chain_lin(const ReprojBatchView& b, const F1Common& cm, const double* a, const double* ap, double* row) {
  // lhs of (X * T_w_p), lhs of (T_c_b * T_b_w): TransformationBasic(q_c_b, t_c_b)
  chain_basic(a, ap, cm.rcb, row + 6, row + 9);
#pragma unroll
  for (int d = 0; d < 3; ++d) { row[6 + d] *= b.q_cb.scale; row[9 + d] *= b.t_cb.scale; }
  MC_SCHED_FENCE();
  // rhs T_w_p = inverse(T_p_w): chain * boxTimes(X) * (-boxTimes(T_w_p)) = -chain * boxTimes(T_c_p)
  double u[3], v[3];
  chain_boxtimes(a, ap, cm.Tcp, u, v);
  u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2];
  chain_basic(u, v, cm.rpw, row + 12, row + 15);
#pragma unroll
  for (int d = 0; d < 3; ++d) { row[12 + d] *= b.qpw_scale; row[15 + d] *= b.tpw_scale; }
  MC_SCHED_FENCE();
  // rhs T_b_w = inverse(T_w_b): chain * boxTimes(T_c_b) * (-boxTimes(T_b_w)) = -chain * boxTimes(X)
  chain_boxtimes(a, ap, cm.X, u, v);
  u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2];
  MC_SCHED_FENCE();
  double B[6];
  chain_JT(u, v, cm.r, cm.S, B);
  MC_SCHED_FENCE();
  // time offset: B * (Phi_dot c)  (BSplineExpressions.cpp:538-541)
  double jd = 0;
#pragma unroll
  for (int d = 0; d < 6; ++d) jd += B[d] * cm.cd[d];
  row[18] = jd * b.toff.scale;
#pragma unroll
  for (int d = 0; d < 6; ++d) row[d] = B[d] * b.coef_scale;
}
// One corner, second half: one weighted core row (19 values) from one row jc[3] of the projection Jacobian.
MC_HD void f1_chain_row(const ReprojBatchView& b, const F1Common& cm, const double* jc, double sw, const double* pc, double* row) {
  // A0 = -J_cam; through HomogeneousExpressionNodeMultiply: A1 = A0 * boxMinus(p_c) = [a, a x p_c]
  double a[3] = {-sw * jc[0], -sw * jc[1], -sw * jc[2]};
  double ap[3];
  v3_cross(a, pc, ap);
  f1_chain_lin(b, cm, a, ap, row);
}
// The matrix of f1_chain_lin, K (6 x 19, row major): core row = z K.  With it the 19 x 19 core Gram of an observation is
// K^T (sum_rows z^T z) K, so the build kernel reduces a 7 x 7 Gram of [z | residual] per observation and applies K once
// instead of pushing every residual row through the chain.
struct F1Chain { double K[6 * F1_CORE]; };
MC_HD void f1_chain_matrix(const ReprojBatchView& b, const F1Common& cm, F1Chain& ck) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double z[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    z[i] = 1.0;
    f1_chain_lin(b, cm, z, z + 3, ck.K + i * F1_CORE);
  }
}
// One corner: weighted core rows (2 x 19, row stride ld) and weighted residual e[2].
MC_HD void f1_term(const ReprojBatchView& b, const double* x, const F1Common& cm, int k, bool useM, bool wantJ,
                   double* core, int ld, double* e, TermOut& out) {
  double pc[3], Jc[6], sw;
  f1_project(b, x, cm, k, useM, wantJ, pc, Jc, e, sw, out);
  if (!wantJ) return;
  f1_chain_row(b, cm, Jc, sw, pc, core);
  f1_chain_row(b, cm, Jc + 3, sw, pc, core + ld);
}
// expand one core row into a (sorted / dense) Jacobian row
MC_HD void f1_expand_row(const double* core, const double* w, const int* rolecol, double* Jrow) {
  if (rolecol[F1_ROLE_COEF] >= 0)
    for (int i = 0; i < S6; ++i)
      for (int d = 0; d < 6; ++d) Jrow[rolecol[F1_ROLE_COEF] + 6 * i + d] = w[i] * core[d];
  for (int g = 0; g < 4; ++g)
    if (rolecol[1 + g] >= 0)
      for (int d = 0; d < 3; ++d) Jrow[rolecol[1 + g] + d] = core[6 + 3 * g + d];
  if (rolecol[F1_ROLE_TOFF] >= 0) Jrow[rolecol[F1_ROLE_TOFF]] = core[18];
}

// ------------------------------------------------------------------------------------------------
// F5: Multical LiDAR point-to-plane ScalarError (sensors_and_targets.py:274-305)
//   T_p_l = T_p_w * (T_w_b(t + d) * inverse(T_l_b));  pred = -(n . t_p) / (n . C_p_l dir),  n = e_z
enum { F5_ROLE_COEF = 0, F5_ROLE_QLB = 1, F5_ROLE_TLB = 2, F5_ROLE_QPW = 3, F5_ROLE_TPW = 4, F5_ROLE_TOFF = 5 };
struct LidarBatchView {
  SplineView sp;
  DvRef q_lb, t_lb, toff;
  double coef_scale, qpw_scale, tpw_scale;
  const double* t;
  const double* dir;
  const double* range;
  const double* sinv;
  const double* S = nullptr;   // optional general sqrtInvR, n x D x D row major (then sinv is not used)
  const int* qpw_off;  // per term
  const int* tpw_off;
  int mest; double mest_a;
  int n_terms, row_base, term_base_global;
};
// COMPACT: the 36 coefficient columns of the row are w_i * B (rank one), so only B (6 values, at rolecol[COEF]) and the 6
// basis weights (behind it) are written; the stream kernel's Gram forms w_i w_j B^T B from them (12 columns instead of 36
// in the LDS tile)
template <bool COMPACT = false>
MC_HD bool f5_term(const LidarBatchView& b, const double* x, int k, bool useM, bool wantJ, const int* rolecol,
                   double* J, int ld, double* e, TermOut& out) {
  double t = b.t[k] + (b.toff.off >= 0 ? x[b.toff.off] : 0.0);
  PoseSample ps;
  if (!pose_sample<2>(b.sp, x, t, ps)) return false;
  Rigid Twb, Tlb, Tbl, Tpw, Twl, Tpl;
#pragma unroll
  for (int i = 0; i < 9; ++i) Twb.C[i] = ps.C[i];
  Twb.r[0] = ps.c[0]; Twb.r[1] = ps.c[1]; Twb.r[2] = ps.c[2];
  load_rigid(x, b.q_lb.off, b.t_lb.off, Tlb);
  rigid_inv(Tlb, Tbl);
  load_rigid(x, b.qpw_off[k], b.tpw_off[k], Tpw);
  rigid_mul(Twb, Tbl, Twl);
  rigid_mul(Tpw, Twl, Tpl);
  const double* dir = b.dir + 3 * (size_t)k;
  double Cd[3];
  m3_mulv(Tpl.C, dir, Cd);
  double d = Tpl.r[2], th = Cd[2];
  double pred = (d / th) * -1.0;
  double e0 = pred - b.range[k];
  double s = b.sinv[k];
  out.chi2 = (s * e0) * (s * e0);
  out.w = mest_weight(b.mest, b.mest_a, out.chi2);
  double sw = (useM ? sqrt(out.w) : 1.0) * s;
  e[0] = sw * e0;
  if (!wantJ) return true;
  // scalar chain: Multiply(., -1) -> -1 ; Divide(d, theta): a_d = -1/theta, a_th = d/theta^2
  double a_d = sw * (-1.0 / th), a_th = sw * (d / (th * th));
  // chain into T_p_l: a_d n^T [I | t_p^]  +  a_th n^T (C dir)^ [0 | I]       (row x matrix^ = row cross vector)
  double n[3] = {0.0, 0.0, 1.0};
  double u[3] = {a_d * n[0], a_d * n[1], a_d * n[2]};
  double v[3], tmp[3];
  v3_cross(u, Tpl.r, v);
  double un[3] = {a_th * n[0], a_th * n[1], a_th * n[2]};
  v3_cross(un, Cd, tmp);
  v[0] += tmp[0]; v[1] += tmp[1]; v[2] += tmp[2];
  double Jq[3], Jt[3];
  // lhs T_p_w: TransformationBasic(q_p_w, t_p_w)
  chain_basic(u, v, Tpw.r, Jq, Jt);
  if (rolecol[F5_ROLE_QPW] >= 0) for (int c = 0; c < 3; ++c) J[rolecol[F5_ROLE_QPW] + c] = Jq[c] * b.qpw_scale;
  if (rolecol[F5_ROLE_TPW] >= 0) for (int c = 0; c < 3; ++c) J[rolecol[F5_ROLE_TPW] + c] = Jt[c] * b.tpw_scale;
  // rhs T_w_l: chain * boxTimes(T_p_w)
  double u2[3], v2[3];
  chain_boxtimes(u, v, Tpw, u2, v2);
  // lhs T_w_b: time-offset spline node
  double B[6];
  chain_JT(u2, v2, ps.c, ps.S, B);
  if (rolecol[F5_ROLE_COEF] >= 0) {
    if (COMPACT) {
      for (int c = 0; c < 6; ++c) J[rolecol[F5_ROLE_COEF] + c] = B[c] * b.coef_scale;
      for (int i = 0; i < S6; ++i) J[rolecol[F5_ROLE_COEF] + 6 + i] = ps.w[0][i];
    } else {
      for (int i = 0; i < S6; ++i)
        for (int c = 0; c < 6; ++c) J[rolecol[F5_ROLE_COEF] + 6 * i + c] = ps.w[0][i] * B[c] * b.coef_scale;
    }
  }
  if (rolecol[F5_ROLE_TOFF] >= 0) {
    double jd = 0;
    for (int c = 0; c < 6; ++c) jd += B[c] * ps.cd[c];
    J[rolecol[F5_ROLE_TOFF]] = jd * b.toff.scale;
  }
  // rhs T_b_l = inverse(T_l_b): chain * boxTimes(T_w_b) * (-boxTimes(T_b_l)) = -chain * boxTimes(T_w_l)
  double u3[3], v3[3];
  chain_boxtimes(u2, v2, Twl, u3, v3);
  u3[0] = -u3[0]; u3[1] = -u3[1]; u3[2] = -u3[2]; v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2];
  chain_basic(u3, v3, Tlb.r, Jq, Jt);
  if (rolecol[F5_ROLE_QLB] >= 0) for (int c = 0; c < 3; ++c) J[rolecol[F5_ROLE_QLB] + c] = Jq[c] * b.q_lb.scale;
  if (rolecol[F5_ROLE_TLB] >= 0) for (int c = 0; c < 3; ++c) J[rolecol[F5_ROLE_TLB] + c] = Jt[c] * b.t_lb.scale;
  (void)ld;
  return true;
}

// ------------------------------------------------------------------------------------------------
// F3 / F4: kalibr EuclideanError, accelerometer and gyroscope forms (sensors_and_targets.py:1152-1170, 1198-1212)
//   gyro : C_i_b * w_b(t) + b_g(t) - w_meas
//   accel: C_i_b * (C_b_w(t) (a_w(t) - g_w) + wdot_b(t) x r_b + w_b x (w_b x r_b)) + b_a(t) - a_meas
// with w_b = -C_b_w S(theta) theta_dot and the reference's "angular acceleration" -C_b_w S(theta) theta_ddot
// (BSP/src/BSplinePose.cpp:207-253, 288-331: the S_dot theta_dot term is dropped there and here).
enum { IMU_ROLE_POSE = 0, IMU_ROLE_BIAS = 1, IMU_ROLE_G = 2, IMU_ROLE_QIB = 3, IMU_ROLE_RB = 4 };
struct ImuBatchView {
  SplineView pose;
  SplineView bias;      // order 0 => static bias in dv_bias
  DvRef bias_dv;
  DvRef g, q_ib, r_b;
  int g_is_direction;
  double pose_scale, bias_scale;
  const double* t;
  const double* meas;
  const double* sinv;
  const double* S = nullptr;   // optional general sqrtInvR, n x D x D row major (then sinv is not used)
  int mest; double mest_a;
  int is_accel;
  int n_terms, row_base, term_base_global;
};
// body-frame rate -C_b_w S v and the two 3x3 factors of its coefficient Jacobian:
//   d rate / d coef_i (rotation part) = w_i * G0 + wd_i * G1,  G1 = -C_b_w S,
//   G0 = -C_b_w D(theta, v) - rate^ C_b_w S    (BSplinePose.cpp:243-248 with Jr from :64-83)
MC_HD void body_rate(const PoseSample& ps, const double* v, double* rate, double* G0, double* G1) {
  double CbwS[9];
  m3_tmul(ps.C, ps.S, CbwS);  // C_w_b^T S
  double t[3];
  m3_mulv(CbwS, v, t);
  rate[0] = -t[0]; rate[1] = -t[1]; rate[2] = -t[2];
  if (!G0) return;
  double D[9], CD[9], K[9], KC[9];
  rotvec_dSv_dtheta(ps.c + 3, v, D);
  m3_tmul(ps.C, D, CD);
  m3_skew(rate, K);
  m3_mul(K, CbwS, KC);
#pragma unroll
  for (int i = 0; i < 9; ++i) { G0[i] = -CD[i] - KC[i]; G1[i] = -CbwS[i]; }
}
// CB (stream kernel, no general sqrtInvR): the bias columns of the three rows hold one value per live bias coefficient
// (w_i on the diagonal of a 3 x 3 block), so it is written once, behind the rows: J[3 ld + rolecol[BIAS] + i]
// CP (stream kernel, with CB): the pose-coefficient columns are rank three in the basis weights -- block i of residual row r is
// [ wdd_i P_r | w_i Q0_r + wd_i Q1_r + wdd_i Q2_r ] -- and are stored that way: the row keeps [P_r Q0_r Q1_r Q2_r] (12 values,
// scaled) at rolecol[POSE], and the term keeps w, wd, wdd (3 x 6) once behind its rows at J[3 ld + wofs]: 54 instead of 108
// doubles per term, which is what lets a chunk hold half as many terms again.
template <bool CB = false, bool CP = false>
MC_HD bool imu_term(const ImuBatchView& b, const double* x, int k, bool useM, bool wantJ, const int* rolecol, double* J,
                    int ld, double* e, TermOut& out, int wofs = 0) {
  double t = b.t[k];
  PoseSample ps;
  bool ok = b.is_accel ? pose_sample<3>(b.pose, x, t, ps) : pose_sample<2>(b.pose, x, t, ps);
  if (!ok) return false;
  // bias
  double bw[1][S6];
  double bias[3];
  int bseg = 0;
  if (b.bias.order > 0) {
    bseg = spline_weights<S6, 1>(b.bias, t, bw);
    if (bseg < 0) return false;
    const double* cf = x + b.bias.param_off + (size_t)bseg * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      double a = 0;
#pragma unroll
      for (int i = 0; i < S6; ++i) a += cf[3 * i + d] * bw[0][i];
      bias[d] = a;
    }
  } else {
    bias[0] = x[b.bias_dv.off]; bias[1] = x[b.bias_dv.off + 1]; bias[2] = x[b.bias_dv.off + 2];
  }
  double Cib[9];
  quat2r(x + b.q_ib.off, Cib);
  double om[3], G0w[9], G1[9];
  body_rate(ps, ps.cd + 3, om, wantJ ? G0w : nullptr, G1);
  double inner[3];
  double al[3], G0a[9], v1[3], rb[3] = {0, 0, 0}, omxr[3];
  if (b.is_accel) {
    double G1b[9];
    body_rate(ps, ps.cdd + 3, al, wantJ ? G0a : nullptr, G1b);
    double g[3];
    if (b.g_is_direction) {  // EuclideanDirection::toEuclidean: magnitude * C.col(2)  (state: C row-major 9 + magnitude)
      const double* Cg = x + b.g.off;
      g[0] = Cg[9] * Cg[2]; g[1] = Cg[9] * Cg[5]; g[2] = Cg[9] * Cg[8];
    } else { g[0] = x[b.g.off]; g[1] = x[b.g.off + 1]; g[2] = x[b.g.off + 2]; }
    double ag[3] = {ps.cdd[0] - g[0], ps.cdd[1] - g[1], ps.cdd[2] - g[2]};
    m3_tmulv(ps.C, ag, v1);  // C_b_w (a_w - g)
    rb[0] = x[b.r_b.off]; rb[1] = x[b.r_b.off + 1]; rb[2] = x[b.r_b.off + 2];
    double alxr[3], oo[3];
    v3_cross(al, rb, alxr);
    v3_cross(om, rb, omxr);
    v3_cross(om, omxr, oo);
#pragma unroll
    for (int d = 0; d < 3; ++d) inner[d] = v1[d] + alxr[d] + oo[d];
  } else {
    inner[0] = om[0]; inner[1] = om[1]; inner[2] = om[2];
  }
  double pr[3];
  m3_mulv(Cib, inner, pr);
  double ev[3];
  const double* m = b.meas + 3 * (size_t)k;
#pragma unroll
  for (int d = 0; d < 3; ++d) ev[d] = pr[d] + bias[d] - m[d];
  double s = b.sinv[k];
  if (b.S) { sinv_mix3(b.S + 9 * (size_t)k, ev, 1); s = 1.0; }
  out.chi2 = (s * s) * (ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
  out.w = mest_weight(b.mest, b.mest_a, out.chi2);
  double sw = (useM ? sqrt(out.w) : 1.0) * s;
  e[0] = sw * ev[0]; e[1] = sw * ev[1]; e[2] = sw * ev[2];
  if (!wantJ) return true;

  // q_i_b : crossMx(C_i_b * inner)   (EuclideanExpressionNodeMultiply, ABE/src/EuclideanExpressionNode.cpp:82-86)
  if (rolecol[IMU_ROLE_QIB] >= 0) {
    double K[9];
    m3_skew(pr, K);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_QIB] + c] = sw * b.q_ib.scale * K[3 * r + c];
  }
  // bias: w^b_i * I  (BSplineEuclideanExpressionNode, BSplineExpressions.cpp:589-599) or I (static EuclideanPoint)
  if (rolecol[IMU_ROLE_BIAS] >= 0) {
    if (CB) {
      if (b.bias.order > 0) { for (int i = 0; i < S6; ++i) J[3 * ld + rolecol[IMU_ROLE_BIAS] + i] = sw * b.bias_scale * bw[0][i]; }
      else J[3 * ld + rolecol[IMU_ROLE_BIAS]] = sw * b.bias_scale;
    } else if (b.bias.order > 0) {
      for (int i = 0; i < S6; ++i)
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_BIAS] + 3 * i + c] = (r == c) ? sw * b.bias_scale * bw[0][i] : 0.0;
    } else {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_BIAS] + c] = (r == c) ? sw * b.bias_scale : 0.0;
    }
  }
  // pose coefficients: block i = [ wdd_i * P | w_i Q0 + wd_i Q1 + wdd_i Q2 ]
  double P[9], Q0[9], Q1[9], Q2[9];
  if (!b.is_accel) {
    m3_mul(Cib, G0w, Q0);
    m3_mul(Cib, G1, Q1);
#pragma unroll
    for (int i = 0; i < 9; ++i) { P[i] = 0.0; Q2[i] = 0.0; }
  } else {
    // K = C_i_b is the chain into `inner`
    double CbwS[9], Kv1[9], KC[9], T1[9], T2[9], Kr[9], KrG[9], Mw[9], So[9], Sor[9], Sr[9], tmp[9];
    m3_mult(Cib, ps.C, P);              // C_i_b C_b_w : a_w chain  (acceleration node block = [wdd_i I | 0])
    m3_tmul(ps.C, ps.S, CbwS);
    // rotation node through RotationInverse: -K (v1)^ C_b_w S w_i
    m3_skew(v1, Kv1);
    m3_mul(Cib, Kv1, KC);
    m3_mul(KC, CbwS, T1);               // K v1^ C_b_w S
    // Cross(wdot_b, r_b) lhs: -K r^  ->  alpha node (w_i G0a + wdd_i G1)
    m3_skew(rb, Sr);
    m3_mul(Cib, Sr, Kr);                // K r^
    m3_mul(Kr, G0a, KrG);
    // Cross(w_b, Cross(w_b, r_b)): lhs -K (w x r)^ ; rhs K w^ then inner lhs -(K w^) r^
    m3_skew(omxr, Sor);
    m3_skew(om, So);
    m3_mul(So, Sr, tmp);
#pragma unroll
    for (int i = 0; i < 9; ++i) tmp[i] = -(Sor[i] + tmp[i]);
    m3_mul(Cib, tmp, Mw);               // Mw = -K((w x r)^ + w^ r^)
    m3_mul(Mw, G0w, T2);
#pragma unroll
    for (int i = 0; i < 9; ++i) Q0[i] = -T1[i] - KrG[i] + T2[i];
    m3_mul(Mw, G1, Q1);
    m3_mul(Kr, G1, Q2);
#pragma unroll
    for (int i = 0; i < 9; ++i) Q2[i] = -Q2[i];
    // gravity: -K C_b_w * Jg
    if (rolecol[IMU_ROLE_G] >= 0) {
      if (b.g_is_direction) {  // Jg = m [ -C[:,0]  C[:,1] ]  (ABE/src/EuclideanDirection.cpp:70-77)
        const double* Cg = x + b.g.off;
        double mg = Cg[9];
        double c0[3] = {-mg * Cg[0], -mg * Cg[3], -mg * Cg[6]}, c1[3] = {mg * Cg[1], mg * Cg[4], mg * Cg[7]};
        double o0[3], o1[3];
        m3_mulv(P, c0, o0);
        m3_mulv(P, c1, o1);
        for (int r = 0; r < 3; ++r) {
          J[r * ld + rolecol[IMU_ROLE_G]] = -sw * b.g.scale * o0[r];
          J[r * ld + rolecol[IMU_ROLE_G] + 1] = -sw * b.g.scale * o1[r];
        }
      } else {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_G] + c] = -sw * b.g.scale * P[3 * r + c];
      }
    }
    // r_b: K (alpha^ + w^ w^)
    if (rolecol[IMU_ROLE_RB] >= 0) {
      double Sa[9], SS[9], R[9];
      m3_skew(al, Sa);
      m3_mul(So, So, SS);
#pragma unroll
      for (int i = 0; i < 9; ++i) SS[i] += Sa[i];
      m3_mul(Cib, SS, R);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_RB] + c] = sw * b.r_b.scale * R[3 * r + c];
    }
  }
  if (rolecol[IMU_ROLE_POSE] >= 0) {
    double f = sw * b.pose_scale;
    if (CP) {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double* q = J + r * ld + rolecol[IMU_ROLE_POSE] + c;
          q[0] = f * P[3 * r + c]; q[3] = f * Q0[3 * r + c]; q[6] = f * Q1[3 * r + c]; q[9] = f * Q2[3 * r + c];
        }
      for (int i = 0; i < S6; ++i) {
        J[3 * ld + wofs + i] = ps.w[0][i];
        J[3 * ld + wofs + S6 + i] = ps.w[1][i];
        J[3 * ld + wofs + 2 * S6 + i] = b.is_accel ? ps.w[2][i] : 0.0;
      }
    } else
    for (int i = 0; i < S6; ++i) {
      double w0 = ps.w[0][i], w1 = ps.w[1][i], w2 = b.is_accel ? ps.w[2][i] : 0.0;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          J[r * ld + rolecol[IMU_ROLE_POSE] + 6 * i + c] = f * w2 * P[3 * r + c];
          J[r * ld + rolecol[IMU_ROLE_POSE] + 6 * i + 3 + c] = f * (w0 * Q0[3 * r + c] + w1 * Q1[3 * r + c] + w2 * Q2[3 * r + c]);
        }
    }
  }
  if (b.S) {   // general sqrtInvR: J_w = S^T J, column by column over the whole row (columns the term does not write hold zeros or nothing the caller reads)
    const double* S = b.S + 9 * (size_t)k;
    for (int c = 0; c < ld; ++c) sinv_mix3(S, J + c, ld);
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// F3 / F4 with the front-end's other two IMU models (multical_calibrate_sensors --imu-models):
//   ScaledMisalignedImu            (sensors_and_targets.py:1393-1550)
//     accel: M_a (C_i_b (v1 + c)) + b_a - a_meas
//     gyro : M_g (C_gyro_b w_b) + M_ag (C_gyro_b (v1 + c)) + b_g - w_meas,   C_gyro_b = C_gyro_i C_i_b
//   ScaledMisalignedSizeEffectImu  (:1553-1661), accelerometer only (the gyroscope is inherited):
//     accel: M_a (C_i_b v1 + sum_x I_x (C_i_b c_x)) + b_a - a_meas,   r_x = r_b + C_i_b^T rx_i
// with v1 = C_b_w (a_w - g_w),  c(r) = wdot_b x r + w_b x (w_b x r).  M_* are MatrixBasic design variables
// (ABE/src/MatrixBasic.cpp:10-47): 3x3, the entries flagged in the update pattern are the minimal parameters, in
// row-major order of the pattern; d(M p)/d(entry (r, c)) = p[c] e_r (EuclideanExpressionNodeMatrixMultiply,
// ABE/src/EuclideanExpressionNode.cpp:120-134, times MatrixBasic's selection matrix B).
// Everything below the leading matrices has one shape: chain Kw into w_b, K0 into v1 and K_j into c(r_j) for every
// lever arm j; the calibrated model of imu_term is the special case Kw = C_i_b (gyro) or K0 = K_1 = C_i_b (accel).
enum { IMU_ROLE_M = 5, IMU_ROLE_QGI = 6, IMU_ROLE_MA = 7, IMU_ROLE_RX = 8 /* RY = 9, RZ = 10 */ };
struct ImuSmBatchView : ImuBatchView {
  DvRef M;          // accel: M_accel, gyro: M_gyro
  DvRef q_gi, Ma;   // gyro: q_gyro_i, M_accel_gyro
  DvRef arm[3];     // size effect: rx_i, ry_i, rz_i
  int I_off[3];     // size effect: parameter offsets of Ix, Iy, Iz (MatrixBasic without free entries, 9 row-major values)
  int M_mask, Ma_mask;   // update patterns, bit 3 * row + col
  int size_effect;
};
// columns of a MatrixBasic design variable multiplying `vec`; f = sqrt(w) sqrtInvR scaling
MC_HD void matrix_dv_columns(const double* vec, int mask, double f, double* J, int ld) {
  int k = 0;
  for (int idx = 0; idx < 9; ++idx)
    if ((mask >> idx) & 1) { J[(idx / 3) * ld + k] = f * vec[idx % 3]; ++k; }
}
MC_HD void m3_add_skew_product(const double* A, const double* v, double* acc) {  // acc += A v^
  double K[9], T[9];
  m3_skew(v, K);
  m3_mul(A, K, T);
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] += T[i];
}
template <bool CB = false>
MC_HD bool imu_sm_term(const ImuSmBatchView& b, const double* x, int k, bool useM, bool wantJ, const int* rolecol, double* J,
                       int ld, double* e, TermOut& out) {
  const double t = b.t[k];
  PoseSample ps;
  if (!pose_sample<3>(b.pose, x, t, ps)) return false;
  double bw[1][S6];
  double bias[3];
  if (b.bias.order > 0) {
    int bseg = spline_weights<S6, 1>(b.bias, t, bw);
    if (bseg < 0) return false;
    const double* cf = x + b.bias.param_off + (size_t)bseg * 3;
    for (int d = 0; d < 3; ++d) {
      double a = 0;
      for (int i = 0; i < S6; ++i) a += cf[3 * i + d] * bw[0][i];
      bias[d] = a;
    }
  } else {
    bias[0] = x[b.bias_dv.off]; bias[1] = x[b.bias_dv.off + 1]; bias[2] = x[b.bias_dv.off + 2];
  }
  double Cib[9], R[9];
  quat2r(x + b.q_ib.off, Cib);
  double Cgi[9];
  if (!b.is_accel) { quat2r(x + b.q_gi.off, Cgi); m3_mul(Cgi, Cib, R); }
  else { for (int i = 0; i < 9; ++i) R[i] = Cib[i]; }
  double om[3], al[3], G0w[9], G0a[9], G1[9], G1b[9];
  body_rate(ps, ps.cd + 3, om, wantJ ? G0w : nullptr, G1);
  body_rate(ps, ps.cdd + 3, al, wantJ ? G0a : nullptr, G1b);
  double g[3];
  if (b.g_is_direction) { const double* Cg = x + b.g.off; g[0] = Cg[9] * Cg[2]; g[1] = Cg[9] * Cg[5]; g[2] = Cg[9] * Cg[8]; }
  else { g[0] = x[b.g.off]; g[1] = x[b.g.off + 1]; g[2] = x[b.g.off + 2]; }
  double ag[3] = {ps.cdd[0] - g[0], ps.cdd[1] - g[1], ps.cdd[2] - g[2]}, v1[3];
  m3_tmulv(ps.C, ag, v1);
  // lever arms
  const int na = (b.is_accel && b.size_effect) ? 3 : 1;
  double rb[3] = {x[b.r_b.off], x[b.r_b.off + 1], x[b.r_b.off + 2]};
  double rj[3][3], rho[3][3], cj[3][3], omxr[3][3];
  for (int j = 0; j < na; ++j) {
    rho[j][0] = rho[j][1] = rho[j][2] = 0.0;
    if (na == 3) m3_tmulv(Cib, x + b.arm[j].off, rho[j]);     // C_i_b^T rx_i
    for (int d = 0; d < 3; ++d) rj[j][d] = rb[d] + rho[j][d];
    double alxr[3], oo[3];
    v3_cross(al, rj[j], alxr);
    v3_cross(om, rj[j], omxr[j]);
    v3_cross(om, omxr[j], oo);
    for (int d = 0; d < 3; ++d) cj[j][d] = alxr[d] + oo[d];
  }
  // value
  const double* Mm = x + b.M.off;
  double u1[3] = {0, 0, 0}, u2[3] = {0, 0, 0}, Tsum[3], s0[3], sj[3][3], pr[3];
  if (b.is_accel) {
    m3_mulv(R, v1, s0);
    for (int d = 0; d < 3; ++d) Tsum[d] = s0[d];
    for (int j = 0; j < na; ++j) {
      m3_mulv(R, cj[j], sj[j]);
      if (na == 3) { double t3[3]; m3_mulv(x + b.I_off[j], sj[j], t3); for (int d = 0; d < 3; ++d) Tsum[d] += t3[d]; }
      else for (int d = 0; d < 3; ++d) Tsum[d] += sj[j][d];
    }
    m3_mulv(Mm, Tsum, pr);
  } else {
    double ab[3] = {v1[0] + cj[0][0], v1[1] + cj[0][1], v1[2] + cj[0][2]}, p1[3], p2[3];
    m3_mulv(R, om, u1);
    m3_mulv(R, ab, u2);
    m3_mulv(Mm, u1, p1);
    m3_mulv(x + b.Ma.off, u2, p2);
    for (int d = 0; d < 3; ++d) pr[d] = p1[d] + p2[d];
  }
  double ev[3];
  const double* m = b.meas + 3 * (size_t)k;
  for (int d = 0; d < 3; ++d) ev[d] = pr[d] + bias[d] - m[d];
  double s = b.sinv[k];
  if (b.S) { sinv_mix3(b.S + 9 * (size_t)k, ev, 1); s = 1.0; }
  out.chi2 = (s * s) * (ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
  out.w = mest_weight(b.mest, b.mest_a, out.chi2);
  const double sw = (useM ? sqrt(out.w) : 1.0) * s;
  e[0] = sw * ev[0]; e[1] = sw * ev[1]; e[2] = sw * ev[2];
  if (!wantJ) return true;

  // bias
  if (rolecol[IMU_ROLE_BIAS] >= 0) {
    if (CB) {
      if (b.bias.order > 0) { for (int i = 0; i < S6; ++i) J[3 * ld + rolecol[IMU_ROLE_BIAS] + i] = sw * b.bias_scale * bw[0][i]; }
      else J[3 * ld + rolecol[IMU_ROLE_BIAS]] = sw * b.bias_scale;
    } else if (b.bias.order > 0) {
      for (int i = 0; i < S6; ++i)
        for (int r = 0; r < 3; ++r) J[r * ld + rolecol[IMU_ROLE_BIAS] + 3 * i + r] = sw * b.bias_scale * bw[0][i];
    } else {
      for (int r = 0; r < 3; ++r) J[r * ld + rolecol[IMU_ROLE_BIAS] + r] = sw * b.bias_scale;
    }
  }
  // leading matrices and the chains below them (sw folded in)
  double Kw[9], K0[9], Kx[3][9], Qrot[9];   // Qrot: chain into the rotation R (accel: C_i_b; gyro: C_gyro_i C_i_b)
  for (int i = 0; i < 9; ++i) { Kw[i] = 0.0; Qrot[i] = 0.0; }
  if (b.is_accel) {
    if (rolecol[IMU_ROLE_M] >= 0) matrix_dv_columns(Tsum, b.M_mask, sw * b.M.scale, J + rolecol[IMU_ROLE_M], ld);
    double A[9];
    for (int i = 0; i < 9; ++i) A[i] = sw * Mm[i];
    m3_add_skew_product(A, s0, Qrot);
    m3_mul(A, R, K0);
    for (int j = 0; j < na; ++j) {
      double Aj[9];
      if (na == 3) m3_mul(A, x + b.I_off[j], Aj);
      else for (int i = 0; i < 9; ++i) Aj[i] = A[i];
      m3_add_skew_product(Aj, sj[j], Qrot);
      m3_mul(Aj, R, Kx[j]);
    }
  } else {
    if (rolecol[IMU_ROLE_M] >= 0) matrix_dv_columns(u1, b.M_mask, sw * b.M.scale, J + rolecol[IMU_ROLE_M], ld);
    if (rolecol[IMU_ROLE_MA] >= 0) matrix_dv_columns(u2, b.Ma_mask, sw * b.Ma.scale, J + rolecol[IMU_ROLE_MA], ld);
    double Aw[9], Aa[9];
    const double* Ma = x + b.Ma.off;
    for (int i = 0; i < 9; ++i) { Aw[i] = sw * Mm[i]; Aa[i] = sw * Ma[i]; }
    m3_add_skew_product(Aw, u1, Qrot);
    m3_add_skew_product(Aa, u2, Qrot);
    m3_mul(Aw, R, Kw);
    m3_mul(Aa, R, K0);
    for (int i = 0; i < 9; ++i) Kx[0][i] = K0[i];
  }
  // aggregates over the lever arms
  double KR[9], MW[9], KOm[9], Om[9], So[9], Sa[9];
  m3_skew(om, So);
  m3_skew(al, Sa);
  m3_mul(So, So, Om);
  for (int i = 0; i < 9; ++i) { Om[i] += Sa[i]; KR[i] = 0.0; MW[i] = Kw[i]; KOm[i] = 0.0; }
  double Qib_extra[9];
  for (int i = 0; i < 9; ++i) Qib_extra[i] = 0.0;
  for (int j = 0; j < na; ++j) {
    double Sr[9], Sor[9], tmp[9], T[9], KjOm[9];
    m3_skew(rj[j], Sr);
    m3_mul(Kx[j], Sr, T);
    for (int i = 0; i < 9; ++i) KR[i] += T[i];
    m3_skew(omxr[j], Sor);
    m3_mul(So, Sr, tmp);
    for (int i = 0; i < 9; ++i) tmp[i] += Sor[i];
    m3_mul(Kx[j], tmp, T);
    for (int i = 0; i < 9; ++i) MW[i] -= T[i];
    m3_mul(Kx[j], Om, KjOm);                 // chain into r_j
    for (int i = 0; i < 9; ++i) KOm[i] += KjOm[i];
    if (na == 3) {
      // r_j = r_b + inverse(C_i_b) * rj_i : the arm gets KjOm C_i_b^T, q_i_b gets -KjOm (C_i_b^T rj_i)^ C_i_b^T
      double KC[9];
      m3_mult(KjOm, Cib, KC);                // KjOm * C_i_b^T
      const int col = rolecol[IMU_ROLE_RX + j];
      if (col >= 0)
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) J[r * ld + col + c] = b.arm[j].scale * KC[3 * r + c];
      double Srho[9], T2[9], T3[9];
      m3_skew(rho[j], Srho);
      m3_mul(KjOm, Srho, T2);
      m3_mult(T2, Cib, T3);
      for (int i = 0; i < 9; ++i) Qib_extra[i] -= T3[i];
    }
  }
  // rotations: accel q_i_b <- Qrot ; gyro q_gyro_i <- Qrot, q_i_b <- Qrot C_gyro_i   (RotationExpressionNodeMultiply)
  if (b.is_accel) {
    if (rolecol[IMU_ROLE_QIB] >= 0)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_QIB] + c] = b.q_ib.scale * (Qrot[3 * r + c] + Qib_extra[3 * r + c]);
  } else {
    if (rolecol[IMU_ROLE_QGI] >= 0)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_QGI] + c] = b.q_gi.scale * Qrot[3 * r + c];
    if (rolecol[IMU_ROLE_QIB] >= 0) {
      double QC[9];
      m3_mul(Qrot, Cgi, QC);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_QIB] + c] = b.q_ib.scale * QC[3 * r + c];
    }
  }
  // r_b
  if (rolecol[IMU_ROLE_RB] >= 0)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_RB] + c] = b.r_b.scale * KOm[3 * r + c];
  // gravity: -K0 C_b_w Jg
  double P[9];
  m3_mult(K0, ps.C, P);
  if (rolecol[IMU_ROLE_G] >= 0) {
    if (b.g_is_direction) {
      const double* Cg = x + b.g.off;
      const double mg = Cg[9];
      double c0[3] = {-mg * Cg[0], -mg * Cg[3], -mg * Cg[6]}, c1[3] = {mg * Cg[1], mg * Cg[4], mg * Cg[7]}, o0[3], o1[3];
      m3_mulv(P, c0, o0);
      m3_mulv(P, c1, o1);
      for (int r = 0; r < 3; ++r) {
        J[r * ld + rolecol[IMU_ROLE_G]] = -b.g.scale * o0[r];
        J[r * ld + rolecol[IMU_ROLE_G] + 1] = -b.g.scale * o1[r];
      }
    } else {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[r * ld + rolecol[IMU_ROLE_G] + c] = -b.g.scale * P[3 * r + c];
    }
  }
  // pose coefficients: block i = [ wdd_i P | w_i Q0 + wd_i Q1 + wdd_i Q2 ]
  if (rolecol[IMU_ROLE_POSE] >= 0) {
    double CbwS[9], Kv1[9], KC[9], T1[9], T2[9], T3[9], Q0[9], Q1[9], Q2[9];
    m3_tmul(ps.C, ps.S, CbwS);
    m3_skew(v1, Kv1);
    m3_mul(K0, Kv1, KC);
    m3_mul(KC, CbwS, T1);       // K0 v1^ C_b_w S
    m3_mul(KR, G0a, T2);
    m3_mul(MW, G0w, T3);
    for (int i = 0; i < 9; ++i) Q0[i] = -T1[i] - T2[i] + T3[i];
    m3_mul(MW, G1, Q1);
    m3_mul(KR, G1, Q2);
    for (int i = 0; i < 9; ++i) Q2[i] = -Q2[i];
    const double f = b.pose_scale;
    for (int i = 0; i < S6; ++i) {
      const double w0 = ps.w[0][i], w1 = ps.w[1][i], w2 = ps.w[2][i];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          J[r * ld + rolecol[IMU_ROLE_POSE] + 6 * i + c] = f * w2 * P[3 * r + c];
          J[r * ld + rolecol[IMU_ROLE_POSE] + 6 * i + 3 + c] = f * (w0 * Q0[3 * r + c] + w1 * Q1[3 * r + c] + w2 * Q2[3 * r + c]);
        }
    }
  }
  if (b.S) {   // general sqrtInvR: J_w = S^T J, column by column over the whole row (columns the term does not write hold zeros or nothing the caller reads)
    const double* S = b.S + 9 * (size_t)k;
    for (int c = 0; c < ld; ++c) sinv_mix3(S, J + c, ld);
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// F2: ReprojectionError with camera design variables over a static chain
//   p_c = B_{n-1} * ... * B_0 * inverse(T_t_c) * p      (CameraCalibrator.py:195-212)
// roles: 0 q_t_c, 1 t_t_c, 2+2j / 3+2j = q / t of baseline j (j < 4), 10 projection, 11 distortion, 12 landmark
// (a HomogeneousPoint design variable in place of the constant target point)
enum { F2_ROLE_QTC = 0, F2_ROLE_TTC = 1, F2_ROLE_CHAIN = 2, F2_ROLE_PROJ = 10, F2_ROLE_DIST = 11, F2_ROLE_LM = 12, F2_MAX_CHAIN = 4 };
struct StaticReprojBatchView {
  CameraView cam;
  int n_chain;
  DvRef chain_q[F2_MAX_CHAIN], chain_t[F2_MAX_CHAIN];
  double qtc_scale, ttc_scale, proj_scale, dist_scale;
  const int* obs_first;
  const int* obs_qtc_off;
  const int* obs_ttc_off;
  const int* term_obs;   // observation of each term
  const double* y;
  const double* p;
  const double* sinv;
  const double* S = nullptr;   // optional general sqrtInvR, n x D x D row major (then sinv is not used)
  const int* lm_off = nullptr;   // optional, per term: parameter offset of the landmark (HomogeneousPoint), or -1
  double lm_scale = 1.0;
  const int* lm_slots = nullptr;       // with lm_off, per term: Batch::lm_slots (4 ints: where its landmark x target-pose blocks sit)
  const int* lm_terms = nullptr;       // terms listed by landmark (work items of the landmark-major pass)
  const int* lm_obs_common = nullptr;  // per observation: index of its pre-pass record
  int mest; double mest_a;
  int n_obs, n_terms, row_base, term_base_global;
};
// what every corner of one observation (one camera's view of the target) shares
struct F2Common {
  Rigid Tinv;               // inverse(T_t_c)
  Rigid T;                  // B_{n-1} ... B_0 inverse(T_t_c)
  Rigid Bj[F2_MAX_CHAIN];   // the baselines
  double rtc[3];            // translation of T_t_c
  double pp[6], dk[4];      // projection / distortion parameters
};
// loops over the chain are unrolled to F2_MAX_CHAIN with a guard so that every array index is a compile-time
// constant (a run-time bound would put the temporaries into scratch memory on the GPU)
MC_HD void f2_common(const StaticReprojBatchView& b, const double* x, int obs, F2Common& cm) {
  Rigid Ttc;
  load_rigid(x, b.obs_qtc_off[obs], b.obs_ttc_off[obs], Ttc);
  cm.rtc[0] = Ttc.r[0]; cm.rtc[1] = Ttc.r[1]; cm.rtc[2] = Ttc.r[2];
  rigid_inv(Ttc, cm.Tinv);  // T_c0_t
  Rigid T = cm.Tinv;
#pragma unroll
  for (int j = 0; j < F2_MAX_CHAIN; ++j) {
    if (j < b.n_chain) {
      load_rigid(x, b.chain_q[j].off, b.chain_t[j].off, cm.Bj[j]);
      Rigid N;
      rigid_mul(cm.Bj[j], T, N);
      T = N;
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) cm.Bj[j].C[i] = 0.0;
      cm.Bj[j].r[0] = cm.Bj[j].r[1] = cm.Bj[j].r[2] = 0.0;
    }
  }
  cm.T = T;
#pragma unroll
  for (int i = 0; i < 6; ++i) cm.pp[i] = i < b.cam.proj_dim ? x[b.cam.proj_off + i] : 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) cm.dk[i] = b.cam.dist_off >= 0 ? x[b.cam.dist_off + i] : 0.0;
}
// rmask: bit r set = write Jacobian row r (to J + r * ld); the row-per-thread kernel asks for one row only
MC_HD bool f2_term_cm(const StaticReprojBatchView& b, const F2Common& cm, const double* x, int k, bool useM, bool wantJ,
                      const int* rolecol, double* J, int ld, double* e, TermOut& out, int rmask = 3) {
  const Rigid& T = cm.T;
  const Rigid& Tinv = cm.Tinv;
  const Rigid* Bj = cm.Bj;
  const double* pp = cm.pp;
  const double* dk = cm.dk;
  // the target point: a constant (x, y, z, 1) or the state of a HomogeneousPoint design variable (unit 4-vector)
  const int lo = b.lm_off ? b.lm_off[k] : -1;
  double ph[4];
  if (lo >= 0) { ph[0] = x[lo]; ph[1] = x[lo + 1]; ph[2] = x[lo + 2]; ph[3] = x[lo + 3]; }
  else { const double* pt = b.p + 3 * (size_t)k; ph[0] = pt[0]; ph[1] = pt[1]; ph[2] = pt[2]; ph[3] = 1.0; }
  double pc[3];
  m3_mulv(T.C, ph, pc);
  pc[0] += ph[3] * T.r[0]; pc[1] += ph[3] * T.r[1]; pc[2] += ph[3] * T.r[2];
  double kp[2], Jc[6];
  // homogeneousToKeypoint: a point with w < 0 is projected through -p.  The pinhole Jacobian overload returns before
  // that branch and differentiates at +p (aslam_cameras/.../implementation/PinholeProjection.hpp:149-198); the omni and
  // double-sphere overloads negate p and J (OmniProjection.hpp:219-228, DoubleSphereProjection.hpp:259-268).  The
  // intrinsics / distortion Jacobians take the sign branch in every model (:391-416).
  const bool negw = ph[3] < 0.0;
  double ps[3] = {pc[0], pc[1], pc[2]};
  if (!negw) {
    project(b.cam.model, b.cam.dist, pp, dk, pc, kp, wantJ ? Jc : nullptr);
  } else {
    ps[0] = -pc[0]; ps[1] = -pc[1]; ps[2] = -pc[2];
    if (b.cam.model == 0) {   // pinhole
      project(b.cam.model, b.cam.dist, pp, dk, ps, kp, nullptr);
      double unused[2];
      if (wantJ) project(b.cam.model, b.cam.dist, pp, dk, pc, unused, Jc);
    } else {
      project(b.cam.model, b.cam.dist, pp, dk, ps, kp, wantJ ? Jc : nullptr);
      if (wantJ) for (int c = 0; c < 6; ++c) Jc[c] = -Jc[c];
    }
  }
  double e0 = b.y[2 * (size_t)k] - kp[0], e1 = b.y[2 * (size_t)k + 1] - kp[1];
  const double* Sk = b.S ? b.S + 4 * (size_t)k : nullptr;
  double s = Sk ? 1.0 : b.sinv[k];
  if (Sk) sinv_mix2(Sk, e0, e1);
  out.chi2 = (s * s) * (e0 * e0 + e1 * e1);
  out.w = mest_weight(b.mest, b.mest_a, out.chi2);
  double sw = (useM ? sqrt(out.w) : 1.0) * s;
  e[0] = sw * e0; e[1] = sw * e1;
  if (!wantJ) return true;
  double Ji[12], Jd[8];
  if (rolecol[F2_ROLE_PROJ] >= 0) intrinsics_jacobian(b.cam.model, b.cam.dist, pp, dk, ps, Ji);
  if (rolecol[F2_ROLE_DIST] >= 0) distortion_jacobian(b.cam.model, b.cam.dist, pp, ps, Jd);
  const int pd = b.cam.proj_dim;
  if (Sk) {   // general sqrtInvR: every row below is linear in the rows of Jc / Ji / Jd
    for (int c = 0; c < 3; ++c) sinv_mix2(Sk, Jc[c], Jc[3 + c]);
    if (rolecol[F2_ROLE_PROJ] >= 0) for (int c = 0; c < pd; ++c) sinv_mix2(Sk, Ji[c], Ji[pd + c]);
    if (rolecol[F2_ROLE_DIST] >= 0) for (int c = 0; c < 4; ++c) sinv_mix2(Sk, Jd[c], Jd[4 + c]);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (!((rmask >> r) & 1)) continue;
    double* row = J + r * ld;
    double a[3] = {-sw * Jc[3 * r], -sw * Jc[3 * r + 1], -sw * Jc[3 * r + 2]};
    // through HomogeneousExpressionNodeMultiply (ABE/src/HomogeneousExpressionNode.cpp:77-82):
    //   lhs: chain * boxMinus(p_c) = [w a, a x p_c]    rhs (the landmark): chain * T, then quatInvS(p) (HomogeneousPoint.cpp:44-52)
    double u[3] = {ph[3] * a[0], ph[3] * a[1], ph[3] * a[2]}, v[3];
    v3_cross(a, pc, v);
    if (rolecol[F2_ROLE_LM] >= 0) {
      double g[4];
      m3_tmulv(T.C, a, g);
      g[3] = a[0] * T.r[0] + a[1] * T.r[1] + a[2] * T.r[2];
      const double h = 0.5 * b.lm_scale;
      row[rolecol[F2_ROLE_LM]] = h * (g[0] * ph[3] + g[1] * ph[2] - g[2] * ph[1] - g[3] * ph[0]);
      row[rolecol[F2_ROLE_LM] + 1] = h * (-g[0] * ph[2] + g[1] * ph[3] + g[2] * ph[0] - g[3] * ph[1]);
      row[rolecol[F2_ROLE_LM] + 2] = h * (g[0] * ph[1] - g[1] * ph[0] + g[2] * ph[3] - g[3] * ph[2]);
    }
    // top-down through Multiply(B_j, rest): lhs gets the chain, rhs gets chain * boxTimes(B_j)
#pragma unroll
    for (int jj = 0; jj < F2_MAX_CHAIN; ++jj) {
      const int j = F2_MAX_CHAIN - 1 - jj;
      if (j < b.n_chain) {
        double Jq[3], Jt[3];
        chain_basic(u, v, Bj[j].r, Jq, Jt);
        int cq = rolecol[F2_ROLE_CHAIN + 2 * j], ct = rolecol[F2_ROLE_CHAIN + 2 * j + 1];
        if (cq >= 0) for (int c = 0; c < 3; ++c) row[cq + c] = Jq[c] * b.chain_q[j].scale;
        if (ct >= 0) for (int c = 0; c < 3; ++c) row[ct + c] = Jt[c] * b.chain_t[j].scale;
        double u2[3], v2[3];
        chain_boxtimes(u, v, Bj[j], u2, v2);
        u[0] = u2[0]; u[1] = u2[1]; u[2] = u2[2]; v[0] = v2[0]; v[1] = v2[1]; v[2] = v2[2];
      }
    }
    // inverse node: chain * (-boxTimes(inverse(T_t_c)))
    double u2[3], v2[3], Jq[3], Jt[3];
    chain_boxtimes(u, v, Tinv, u2, v2);
    u2[0] = -u2[0]; u2[1] = -u2[1]; u2[2] = -u2[2]; v2[0] = -v2[0]; v2[1] = -v2[1]; v2[2] = -v2[2];
    chain_basic(u2, v2, cm.rtc, Jq, Jt);
    if (rolecol[F2_ROLE_QTC] >= 0) for (int c = 0; c < 3; ++c) row[rolecol[F2_ROLE_QTC] + c] = Jq[c] * b.qtc_scale;
    if (rolecol[F2_ROLE_TTC] >= 0) for (int c = 0; c < 3; ++c) row[rolecol[F2_ROLE_TTC] + c] = Jt[c] * b.ttc_scale;
    // camera design variables: -Jp, -Jd   (CameraDesignVariable.hpp:45-52); Ji is 2 x proj_dim, row-major
    if (rolecol[F2_ROLE_PROJ] >= 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double jv;
        if (pd == 4) jv = c < 4 ? Ji[r * 4 + (c & 3)] : 0.0;
        else if (pd == 5) jv = c < 5 ? Ji[r * 5 + (c < 5 ? c : 0)] : 0.0;
        else jv = Ji[r * 6 + c];
        if (c < pd) row[rolecol[F2_ROLE_PROJ] + c] = -sw * b.proj_scale * jv;
      }
    }
    if (rolecol[F2_ROLE_DIST] >= 0)
      for (int c = 0; c < 4; ++c) row[rolecol[F2_ROLE_DIST] + c] = -sw * b.dist_scale * Jd[r * 4 + c];
  }
  return true;
}
// ------------------------------------------------------------------------------------------------
// F6: the error terms of the front-end's initialisation problems (small, a few global design variables each).
//
// kalibr GyroscopeError (KAL/src/GyroscopeError.cpp:5-17) as findExtrinsicPriorSensorsToCamera builds it
// (sensors_and_targets.py:858-879):  e = C(q_i_c) * w_c + b_g - w_meas,  w_c a constant (the camera spline's body rate),
// design variables q_i_c (RotationQuaternion) and the gyroscope bias (EuclideanPoint).
enum { GP_ROLE_Q = 0, GP_ROLE_BIAS = 1 };
struct GyroPriorBatchView {
  DvRef q, bias;
  const double* w_pred;   // n x 3: predicted rate in the camera frame (constant)
  const double* meas;     // n x 3
  const double* sinv;
  const double* S = nullptr;   // optional general sqrtInvR, n x D x D row major (then sinv is not used)
  int mest; double mest_a;
  int n_terms, row_base, term_base_global;
};
MC_HD bool gyro_prior_term(const GyroPriorBatchView& b, const double* x, int k, bool useM, bool wantJ, const int* rolecol,
                           double* J, int ld, double* e, TermOut& out) {
  double C[9], p[3];
  quat2r(x + b.q.off, C);
  m3_mulv(C, b.w_pred + 3 * (size_t)k, p);
  const double* bg = x + b.bias.off;
  const double* m = b.meas + 3 * (size_t)k;
  double ev[3] = {p[0] + bg[0] - m[0], p[1] + bg[1] - m[1], p[2] + bg[2] - m[2]};
  double s = b.sinv[k];
  if (b.S) { sinv_mix3(b.S + 9 * (size_t)k, ev, 1); s = 1.0; }
  out.chi2 = (s * s) * (ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
  out.w = mest_weight(b.mest, b.mest_a, out.chi2);
  const double sw = (useM ? sqrt(out.w) : 1.0) * s;
  e[0] = sw * ev[0]; e[1] = sw * ev[1]; e[2] = sw * ev[2];
  if (!wantJ) return true;
  // EuclideanExpressionNodeMultiply (ABE/src/EuclideanExpressionNode.cpp:82-86): rotation branch crossMx(C p); bias: I
  double K[9];
  m3_skew(p, K);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      if (rolecol[GP_ROLE_Q] >= 0) J[r * ld + rolecol[GP_ROLE_Q] + c] = sw * b.q.scale * K[3 * r + c];
      if (rolecol[GP_ROLE_BIAS] >= 0) J[r * ld + rolecol[GP_ROLE_BIAS] + c] = (r == c) ? sw * b.bias.scale : 0.0;
    }
  if (b.S) {   // general sqrtInvR: J_w = S^T J, column by column over the whole row (columns the term does not write hold zeros or nothing the caller reads)
    const double* S = b.S + 9 * (size_t)k;
    for (int c = 0; c < ld; ++c) sinv_mix3(S, J + c, ld);
  }
  return true;
}
// ErrorTermTransformation (ABE/src/ErrorTermTransformation.cpp:19-81) over the two expression shapes the front-end
// builds: T = TransformationDv_a (findExtrinsicPriorSensorsToCamera, sensors_and_targets.py:886-919) or
// T = TransformationDv_a * inverse(TransformationDv_b) (findTargetPosePrior, :812-817).
//   E = T * inverse(prior);  e = [ E.t ; R2AxisAngle(E.C) ];  invR = diag(wt wt wt wr wr wr)
//   chain J6 = [ I  crossMx(E.t) ; 0  invS(axis-angle) ]  into T.evaluateJacobians
enum { TE_ROLE_QA = 0, TE_ROLE_TA = 1, TE_ROLE_QB = 2, TE_ROLE_TB = 3 };
struct TransformBatchView {
  DvRef qa, ta, qb, tb;    // qb.off < 0: no second transformation
  const double* prior_inv; // n x 12: rotation (row-major 9) and translation of inverse(prior)
  double sw_t, sw_r;       // sqrt of the translation / rotation weights (sqrtInvR of the diagonal invR)
  int mest; double mest_a;
  int n_terms, row_base, term_base_global;
};
MC_HD bool transform_term(const TransformBatchView& b, const double* x, int k, bool useM, bool wantJ, const int* rolecol,
                          double* J, int ld, double* e, TermOut& out) {
  Rigid Ta, Tb, Tbi, T, Pi, E;
  load_rigid(x, b.qa.off, b.ta.off, Ta);
  const bool two = b.qb.off >= 0;
  if (two) {
    load_rigid(x, b.qb.off, b.tb.off, Tb);
    rigid_inv(Tb, Tbi);
    rigid_mul(Ta, Tbi, T);
  } else {
    T = Ta; Tb = Ta; Tbi = Ta;
  }
  const double* pi = b.prior_inv + 12 * (size_t)k;
#pragma unroll
  for (int i = 0; i < 9; ++i) Pi.C[i] = pi[i];
  Pi.r[0] = pi[9]; Pi.r[1] = pi[10]; Pi.r[2] = pi[11];
  rigid_mul(T, Pi, E);
  double aa[3];
  r_to_axis_angle(E.C, aa);
  const double ev[6] = {E.r[0], E.r[1], E.r[2], aa[0], aa[1], aa[2]};
  double chi2 = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) { const double s = r < 3 ? b.sw_t : b.sw_r; chi2 += (s * ev[r]) * (s * ev[r]); }
  out.chi2 = chi2;
  out.w = mest_weight(b.mest, b.mest_a, chi2);
  const double sm = useM ? sqrt(out.w) : 1.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) e[r] = sm * (r < 3 ? b.sw_t : b.sw_r) * ev[r];
  if (!wantJ) return true;
  double Kt[9], iS[9];
  m3_skew(E.r, Kt);
  rotvec_inv_S(aa, iS);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    // row r of J6, weighted: [u | v]
    const double f = sm * (r < 3 ? b.sw_t : b.sw_r);
    double u[3] = {0.0, 0.0, 0.0}, v[3];
    if (r < 3) { u[r] = f; v[0] = f * Kt[3 * r]; v[1] = f * Kt[3 * r + 1]; v[2] = f * Kt[3 * r + 2]; }
    else { v[0] = f * iS[3 * (r - 3)]; v[1] = f * iS[3 * (r - 3) + 1]; v[2] = f * iS[3 * (r - 3) + 2]; }
    double* row = J + r * ld;
    double Jq[3], Jt[3];
    // lhs of Multiply (or the single leaf): TransformationBasic(q_a, t_a)
    chain_basic(u, v, Ta.r, Jq, Jt);
    if (rolecol[TE_ROLE_QA] >= 0) for (int c = 0; c < 3; ++c) row[rolecol[TE_ROLE_QA] + c] = Jq[c] * b.qa.scale;
    if (rolecol[TE_ROLE_TA] >= 0) for (int c = 0; c < 3; ++c) row[rolecol[TE_ROLE_TA] + c] = Jt[c] * b.ta.scale;
    if (two) {
      // rhs: chain * boxTimes(T_a), through the inverse node * (-boxTimes(inverse(T_b)))
      double u2[3], v2[3], u3[3], v3[3];
      chain_boxtimes(u, v, Ta, u2, v2);
      chain_boxtimes(u2, v2, Tbi, u3, v3);
      u3[0] = -u3[0]; u3[1] = -u3[1]; u3[2] = -u3[2]; v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2];
      chain_basic(u3, v3, Tb.r, Jq, Jt);
      if (rolecol[TE_ROLE_QB] >= 0) for (int c = 0; c < 3; ++c) row[rolecol[TE_ROLE_QB] + c] = Jq[c] * b.qb.scale;
      if (rolecol[TE_ROLE_TB] >= 0) for (int c = 0; c < 3; ++c) row[rolecol[TE_ROLE_TB] + c] = Jt[c] * b.tb.scale;
    }
  }
  return true;
}

MC_HD bool f2_term(const StaticReprojBatchView& b, const double* x, int k, bool useM, bool wantJ, const int* rolecol,
                   double* J, int ld, double* e, TermOut& out, int rmask = 3) {
  F2Common cm;
  f2_common(b, x, b.term_obs[k], cm);
  return f2_term_cm(b, cm, x, k, useM, wantJ, rolecol, J, ld, e, out, rmask);
}

}  // namespace mc
