// mc_math.h -- fixed-size fp64 building blocks for the gfx950 kernels: 3-vectors, row-major 3x3
// matrices, the RotationVector / quaternion kinematics, B-spline basis weights and the camera
// projection models.  Everything is straight-line code on plain arrays so that it stays in
// registers; no dynamic sizes, no expression DAG.  Each function cites the reference routine whose
// result it reproduces (paths relative to the reference root; SM = Schweizer-Messer,
// BSP = aslam_nonparametric_estimation/bsplines, CAM = aslam_cv/aslam_cameras/include/aslam/cameras/implementation).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MC_HD __host__ __device__ __forceinline__
#else
#define MC_HD inline
#endif
// Scheduling fence for the device code: keeps the compiler from hoisting every operand load of a long straight-line
// chain to its top (which costs ~2 VGPRs per hoisted double and, past the occupancy target, scratch spills).
#if defined(__HIP_DEVICE_COMPILE__)
#define MC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MC_SCHED_FENCE() ((void)0)
#endif

namespace mc {

// ---------------------------------------------------------------- 3-vectors / 3x3 (row major)
MC_HD void v3_cross(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
MC_HD double v3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// o = M v
MC_HD void m3_mulv(const double* M, const double* v, double* o) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
// o = M^T v  (= row vector v^T M)
MC_HD void m3_tmulv(const double* M, const double* v, double* o) {
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  double y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  double z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
MC_HD void m3_mul(const double* A, const double* B, double* O) {  // O = A B  (O must not alias)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) O[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
MC_HD void m3_tmul(const double* A, const double* B, double* O) {  // O = A^T B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) O[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
MC_HD void m3_mult(const double* A, const double* B, double* O) {  // O = A B^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) O[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
// O = crossMx(v)   SM/sm_kinematics/src/rotations.cpp:78-84
MC_HD void m3_skew(const double* v, double* O) {
  O[0] = 0; O[1] = -v[2]; O[2] = v[1];
  O[3] = v[2]; O[4] = 0; O[5] = -v[0];
  O[6] = -v[1]; O[7] = v[0]; O[8] = 0;
}

// rigid transform (C, r):  x -> C x + r
struct Rigid { double C[9]; double r[3]; };
MC_HD void rigid_mul(const Rigid& A, const Rigid& B, Rigid& O) {  // O = A * B
  m3_mul(A.C, B.C, O.C);
  double t[3];
  m3_mulv(A.C, B.r, t);
  O.r[0] = t[0] + A.r[0]; O.r[1] = t[1] + A.r[1]; O.r[2] = t[2] + A.r[2];
}
// Inverse of a rigid transform.  The reference inverts with a general 4x4 Matrix4d::inverse()
// (ABE/src/TransformationExpressionNode.cpp:81,88); for a valid transform the two agree to rounding.
MC_HD void rigid_inv(const Rigid& A, Rigid& O) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) O.C[3 * i + j] = A.C[3 * j + i];
  double t[3];
  m3_mulv(O.C, A.r, t);
  O.r[0] = -t[0]; O.r[1] = -t[1]; O.r[2] = -t[2];
}
// Row-vector chain through boxTimes(T) = [C, -r^ C; 0, C]  (SM/sm_kinematics/src/transformations.cpp:132-142):
// [u v] * boxTimes(T) = [u C, (v - u x r) C]
MC_HD void chain_boxtimes(const double* u, const double* v, const Rigid& T, double* uo, double* vo) {
  double c[3], w[3];
  v3_cross(u, T.r, c);
  w[0] = v[0] - c[0]; w[1] = v[1] - c[1]; w[2] = v[2] - c[2];
  double a[3];
  m3_tmulv(T.C, u, a);
  m3_tmulv(T.C, w, vo);
  uo[0] = a[0]; uo[1] = a[1]; uo[2] = a[2];
}

// ---------------------------------------------------------------- quaternions
// SM/sm_kinematics/src/quaternion_algebra.cpp:77-101 (q = [x y z w])
MC_HD void quat2r(const double* q, double* R) {
  R[0] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
  R[1] = q[0] * q[1] * 2.0 + q[2] * q[3] * 2.0;
  R[2] = q[0] * q[2] * 2.0 - q[1] * q[3] * 2.0;
  R[3] = q[0] * q[1] * 2.0 - q[2] * q[3] * 2.0;
  R[4] = -q[0] * q[0] + q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
  R[5] = q[0] * q[3] * 2.0 + q[1] * q[2] * 2.0;
  R[6] = q[0] * q[2] * 2.0 + q[1] * q[3] * 2.0;
  R[7] = q[0] * q[3] * (-2.0) + q[1] * q[2] * 2.0;
  R[8] = -q[0] * q[0] - q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
}
// updateQuat (:302-318) with axisAngle2quat (:200-221; small-angle branch theta < eps^(1/4))
MC_HD void update_quat(const double* q, const double* dq, double* o) {
  double theta = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
  double na;
  if (theta < 1.220703125e-4) na = 0.5 + (theta * theta) * (1.0 / 48.0);  // 2^-13 = eps^(1/4)
  else na = sin(theta * 0.5) / theta;
  double d0 = dq[0] * na, d1 = dq[1] * na, d2 = dq[2] * na, ca = cos(theta * 0.5);
  double r0 = q[0] * ca + d0 * q[3] - d1 * q[2] + d2 * q[1];
  double r1 = q[1] * ca + d0 * q[2] + d1 * q[3] - d2 * q[0];
  double r2 = q[2] * ca - d0 * q[1] + d1 * q[0] + d2 * q[3];
  double r3 = q[3] * ca - d0 * q[0] - d1 * q[1] - d2 * q[2];
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
// EulerAnglesYawPitchRoll::parametersToRotationMatrix with yaw = 0 (SM/.../EulerAnglesYawPitchRoll.cpp:12-35)
MC_HD void ypr0_to_r(double pitch, double roll, double* C) {
  double cx = cos(roll), sx = sin(roll), cy = cos(pitch), sy = sin(pitch);
  C[0] = cy;  C[1] = sy * sx; C[2] = sy * cx;
  C[3] = 0.0; C[4] = cx;      C[5] = -sx;
  C[6] = -sy; C[7] = cy * sx; C[8] = cy * cx;
}

// ---------------------------------------------------------------- RotationVector (SM/sm_kinematics/src/RotationVector.cpp)
// C(theta) (:10-55) and S(theta) (:82-106); identity below |theta| < 1e-14 like the reference.
MC_HD void rotvec_C_S(const double* th, double* C, double* S) {
  double angle = sqrt(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
  if (angle < 1e-14) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { C[i] = (i % 4 == 0) ? 1.0 : 0.0; if (S) S[i] = C[i]; }
    return;
  }
  double recip = 1.0 / angle;
  double ax = th[0] * recip, ay = th[1] * recip, az = th[2] * recip;
  double sa = sin(angle), ca = cos(angle);
  double ax2 = ax * ax, ay2 = ay * ay, az2 = az * az;
  C[0] = ax2 + ca * (1.0 - ax2);            C[1] = ax * ay - ca * ax * ay + sa * az; C[2] = ax * az - ca * ax * az - sa * ay;
  C[3] = ax * ay - ca * ax * ay - sa * az;  C[4] = ay2 + ca * (1.0 - ay2);           C[5] = ay * az - ca * ay * az + sa * ax;
  C[6] = ax * az - ca * ax * az + sa * ay;  C[7] = ay * az - ca * ay * az - sa * ax; C[8] = az2 + ca * (1.0 - az2);
  if (S) {
    double st2 = sin(angle * 0.5);
    double c1 = -2.0 * st2 * st2 * recip;
    double c2 = (angle - sa) * recip;
    // S = I + c1 a^ + c2 a^ a^ ,  a^ a^ = a a^T - I
    S[0] = 1.0 + c2 * (ax2 - 1.0); S[1] = -c1 * az + c2 * ax * ay; S[2] = c1 * ay + c2 * ax * az;
    S[3] = c1 * az + c2 * ax * ay; S[4] = 1.0 + c2 * (ay2 - 1.0);  S[5] = -c1 * ax + c2 * ay * az;
    S[6] = -c1 * ay + c2 * ax * az; S[7] = c1 * ax + c2 * ay * az; S[8] = 1.0 + c2 * (az2 - 1.0);
  }
}
// D = d( S(theta) v ) / d theta, the 3x3 the reference carries as machine-generated code in
// RotationVector::angularVelocityAndJacobian (:109-183).  Closed form in theta:
//   S v = v + k1 (theta x v) + k2 theta x (theta x v),  k1 = -(1-cos phi)/phi^2,  k2 = (phi - sin phi)/phi^3
//   D   = (theta x v)(k1'/phi) theta^T - k1 v^ + (theta x (theta x v))(k2'/phi) theta^T
//         + k2 ((theta.v) I + theta v^T - 2 v theta^T)
// Same function of (theta, v); singular at theta = 0 exactly like the reference (division by phi).
MC_HD void rotvec_dSv_dtheta(const double* th, const double* v, double* D) {
  double phi2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  double phi = sqrt(phi2);
  double s = sin(phi), sh = sin(0.5 * phi);
  double omc = 2.0 * sh * sh;  // 1 - cos(phi)
  double k1 = -omc / phi2;
  double k2 = (phi - s) / (phi2 * phi);
  double dk1 = (-s / phi2 + 2.0 * omc / (phi2 * phi)) / phi;            // k1'/phi
  double dk2 = (omc / (phi2 * phi) - 3.0 * (phi - s) / (phi2 * phi2)) / phi;  // k2'/phi
  double txv[3], ttxv[3];
  v3_cross(th, v, txv);
  v3_cross(th, txv, ttxv);
  double tv = v3_dot(th, v);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      D[3 * i + j] = txv[i] * dk1 * th[j] + ttxv[i] * dk2 * th[j] + k2 * (th[i] * v[j] - 2.0 * v[i] * th[j]);
  D[0] += k2 * tv; D[4] += k2 * tv; D[8] += k2 * tv;
  // - k1 v^
  D[1] += k1 * v[2];  D[2] -= k1 * v[1];
  D[3] -= k1 * v[2];  D[5] += k1 * v[0];
  D[6] += k1 * v[1];  D[7] -= k1 * v[0];
}

// ---------------------------------------------------------------- B-spline basis weights
// Device view of one spline (uniform or not): knots and the per-segment basis matrices that the
// reference precomputes in BSpline::initializeBasisMatrices (BSP/src/BSpline.cpp:58-121).
// R2AxisAngle (SM/sm_kinematics/src/rotations.cpp:133-157): axis-angle vector of a rotation matrix, trace clamped into
// acos' range, zero below 1e-10 (angle or sine).  C row-major.
MC_HD void r_to_axis_angle(const double* C, double* a) {
  double tr = (C[0] + C[4] + C[8] - 1.0) * 0.5;
  tr = tr < -1.0 ? -1.0 : (tr > 1.0 ? 1.0 : tr);
  const double ang = acos(tr);
  a[0] = a[1] = a[2] = 0.0;
  if (fabs(ang) < 1e-10) return;
  const double ax = C[7] - C[5], ay = C[2] - C[6], az = C[3] - C[1];
  const double n2 = sqrt(ax * ax + ay * ay + az * az);
  if (fabs(n2) < 1e-10) return;
  const double sc = -ang / n2;
  a[0] = sc * ax; a[1] = sc * ay; a[2] = sc * az;
}
// RotationVector::parametersToInverseSMatrix (SM/sm_kinematics/src/RotationVector.cpp:186-201): I + 1/2 p^ + a1 p^ p^
MC_HD void rotvec_inv_S(const double* p, double* O) {
  const double phi = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
#pragma unroll
  for (int i = 0; i < 9; ++i) O[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (phi == 0.0) return;
  const double cot = -sin(phi) / (cos(phi) - 1.0);
  const double a1 = 1.0 / (phi * phi) * (1.0 - 0.5 * phi * cot);
  double K[9], KK[9];
  m3_skew(p, K);
  m3_mul(K, K, KK);
#pragma unroll
  for (int i = 0; i < 9; ++i) O[i] += 0.5 * K[i] + a1 * KK[i];
}

struct SplineView {
  const double* knots;   // n_knots
  const double* basis;   // n_seg x S x S, row major: basis[seg][k][j] = M_seg(k, j)
  int n_knots, order, dim, n_coeff, n_seg;
  int param_off;         // offset of coefficient 0 in the state vector (coefficient i at param_off + i*dim)
  int first_block;       // block index of coefficient 0 (-1 if the spline is inactive)
  int first_col;         // column base of coefficient 0
  double t_min, t_max;
};
// BSpline::computeTIndex / computeUAndTIndex / computeU + Bt_u (BSP/src/BSpline.cpp:237-318, 331-349).
// Writes, for derivative orders 0..ND-1, w[d][j] = sum_k M(k, j) * u_d[k]; returns the segment index
// (= index of the first local coefficient), or -1 when t is outside [t_min, t_max] (the reference throws).
template <int S, int ND>
MC_HD int spline_weights(const SplineView& sp, double t, double (*w)[S]) {
  if (!(t >= sp.t_min)) return -1;
  if (fabs(sp.t_max - t) < 1e-10) t = sp.t_max;
  if (!(t <= sp.t_max)) return -1;
  int idx;
  if (t == sp.t_max) idx = sp.n_knots - S - 1;
  else {  // std::upper_bound(knots, t) - 1
    int lo = 0, hi = sp.n_knots;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (sp.knots[mid] <= t) lo = mid + 1; else hi = mid;
    }
    idx = lo - 1;
  }
  double width = sp.knots[idx + 1] - sp.knots[idx];
  double u = width > 0.0 ? (t - sp.knots[idx]) / width : 0.0;
  int seg = idx - S + 1;
  const double* M = sp.basis + (size_t)seg * S * S;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    double uv[S];
    double mult = width > 0.0 ? 1.0 : 0.0;
    for (int q = 0; q < d; ++q) mult /= width;  // 1/pow(delta_t, d)
    double uu = 1.0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      if (i < d) uv[i] = 0.0;
      else {
        int dm = 1;
        for (int q = 0; q < d; ++q) dm *= (i - q);
        uv[i] = mult * uu * (double)dm;
        uu *= u;
      }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < S; ++k) acc += M[k * S + j] * uv[k];
      w[d][j] = acc;
    }
  }
  return seg;
}
// first segment index containing t (BSpline::segmentIndex, :1440-1444), host-side helper too
MC_HD int spline_segment(const SplineView& sp, double t) {
  if (fabs(sp.t_max - t) < 1e-10) t = sp.t_max;
  if (t == sp.t_max) return sp.n_seg - 1;
  int lo = 0, hi = sp.n_knots;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (sp.knots[mid] <= t) lo = mid + 1; else hi = mid;
  }
  return lo - 1 - sp.order + 1;
}

// ---------------------------------------------------------------- camera models
struct CameraView {
  int model, dist;
  int proj_off, dist_off;   // offsets of the projection / distortion parameters in the state vector
  int proj_block, dist_block, proj_col, dist_col;  // block index / column base (-1 if inactive)
  int proj_dim;
};

// distort(y) and d distort/dy (row-major 2x2 in Jd), CAM/RadialTangentialDistortion.hpp:4-65,
// CAM/EquidistantDistortion.hpp:4-183 (Jacobian = derivative of y*thetad(atan r)/r, which is what the
// reference's MATLAB-generated expression evaluates; no r>1e-8 guard on it, like the reference).
// EQUI = false compiles the equidistant branch (sqrt, atan) out: for launches that hold no such camera
template <bool EQUI = true>
MC_HD void distort(int kind, const double* k, double* y, double* Jd) {
  if (kind == 1) {
    double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
    double mx2 = y[0] * y[0], my2 = y[1] * y[1], mxy = y[0] * y[1];
    double rho2 = mx2 + my2;
    double rad = k1 * rho2 + k2 * rho2 * rho2;
    if (Jd) {
      Jd[0] = 1 + rad + k1 * 2.0 * mx2 + k2 * rho2 * 4 * mx2 + 2.0 * p1 * y[1] + 6 * p2 * y[0];
      Jd[2] = k1 * 2.0 * y[0] * y[1] + k2 * 4 * rho2 * y[0] * y[1] + p1 * 2.0 * y[0] + 2.0 * p2 * y[1];
      Jd[1] = Jd[2];
      Jd[3] = 1 + rad + k1 * 2.0 * my2 + k2 * rho2 * 4 * my2 + 6 * p1 * y[1] + 2.0 * p2 * y[0];
    }
    double y0 = y[0], y1 = y[1];
    y[0] = y0 + y0 * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2);
    y[1] = y1 + y1 * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
  } else if (EQUI && kind == 2) {
    double r2 = y[0] * y[0] + y[1] * y[1];
    double r = sqrt(r2);
    double th = atan(r);
    double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    double thd = th * (1 + k[0] * th2 + k[1] * th4 + k[2] * th6 + k[3] * th8);
    if (Jd) {
      double dthd = 1 + 3 * k[0] * th2 + 5 * k[1] * th4 + 7 * k[2] * th6 + 9 * k[3] * th8;
      double s = thd / r;
      double g = dthd / (r2 * (1.0 + r2)) - thd / (r2 * r);
      Jd[0] = s + y[0] * y[0] * g; Jd[1] = y[0] * y[1] * g;
      Jd[2] = Jd[1];               Jd[3] = s + y[1] * y[1] * g;
    }
    double sc = (r > 1e-8) ? thd / r : 1.0;
    y[0] *= sc; y[1] *= sc;
  } else if (Jd) {
    Jd[0] = 1.0; Jd[1] = 0.0; Jd[2] = 0.0; Jd[3] = 1.0;
  }
}
// d distort / d k, row-major 2x4 (CAM/RadialTangentialDistortion.hpp:152-182, EquidistantDistortion.hpp:243-273)
MC_HD void distort_param_jacobian(int kind, const double* y, double* J) {
  if (kind == 1) {
    double y0 = y[0], y1 = y[1], r2 = y0 * y0 + y1 * y1, r4 = r2 * r2;
    J[0] = y0 * r2; J[1] = y0 * r4; J[2] = 2.0 * y0 * y1;      J[3] = r2 + 2.0 * y0 * y0;
    J[4] = y1 * r2; J[5] = y1 * r4; J[6] = r2 + 2.0 * y1 * y1; J[7] = 2.0 * y0 * y1;
  } else {
    double r = sqrt(y[0] * y[0] + y[1] * y[1]);
    double th = atan(r);
    double pw = th * th * th;
#pragma unroll
    for (int j = 0; j < 4; ++j) { J[j] = y[0] * pw / r; J[4 + j] = y[1] * pw / r; pw *= th * th; }
  }
}

// euclideanToKeypoint of the three projection models (+ optional row-major 2x3 Jacobian wrt the point).
// pp = projection parameters in update() order: pinhole (fu fv cu cv), omni (xi fu fv cu cv),
// ds (xi alpha fu fv cu cv); dk = distortion parameters.  Where the reference bails out of the omni / ds
// field-of-view test before writing anything (the error terms ignore the flag and read an uninitialised
// keypoint), keypoint and Jacobian are defined as zero.
// CAM/PinholeProjection.hpp:73-145, OmniProjection.hpp:72-180, DoubleSphereProjection.hpp:90-221.
template <bool EQUI = true>
MC_HD void project(int model, int dist, const double* pp, const double* dk, const double* p, double* kp, double* J) {
  if (model == 0) {
    double fu = pp[0], fv = pp[1], cu = pp[2], cv = pp[3];
    double rz = 1.0 / p[2], rz2 = rz * rz;
    double y[2] = {p[0] * rz, p[1] * rz};
    double Jd[4];
    distort<EQUI>(dist, dk, y, J ? Jd : nullptr);
    if (J) {
      J[0] = fu * Jd[0] * rz; J[1] = fu * Jd[1] * rz; J[2] = -fu * (p[0] * Jd[0] + p[1] * Jd[1]) * rz2;
      J[3] = fv * Jd[2] * rz; J[4] = fv * Jd[3] * rz; J[5] = -fv * (p[0] * Jd[2] + p[1] * Jd[3]) * rz2;
    }
    kp[0] = fu * y[0] + cu; kp[1] = fv * y[1] + cv;
  } else if (model == 1) {
    double xi = pp[0], fu = pp[1], fv = pp[2], cu = pp[3], cv = pp[4];
    double fov = (xi <= 1.0) ? xi : 1 / xi;
    double d = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    kp[0] = kp[1] = 0.0;
    if (J) { J[0] = J[1] = J[2] = J[3] = J[4] = J[5] = 0.0; }
    if (p[2] <= -(fov * d)) return;
    double rz = 1.0 / (p[2] + xi * d);
    double y[2] = {p[0] * rz, p[1] * rz};
    double Jn[6];
    if (J) {
      double q = rz * rz / d;
      Jn[0] = q * (d * p[2] + xi * (p[1] * p[1] + p[2] * p[2]));
      Jn[3] = -q * xi * p[0] * p[1];
      Jn[1] = Jn[3];
      Jn[4] = q * (d * p[2] + xi * (p[0] * p[0] + p[2] * p[2]));
      q = q * (-xi * p[2] - d);
      Jn[2] = p[0] * q; Jn[5] = p[1] * q;
    }
    double Jd[4];
    distort<EQUI>(dist, dk, y, J ? Jd : nullptr);
    if (J) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        J[c] = fu * (Jd[0] * Jn[c] + Jd[1] * Jn[3 + c]);
        J[3 + c] = fv * (Jd[2] * Jn[c] + Jd[3] * Jn[3 + c]);
      }
    }
    kp[0] = fu * y[0] + cu; kp[1] = fv * y[1] + cv;
  } else {
    double xi = pp[0], alpha = pp[1], fu = pp[2], fv = pp[3], cu = pp[4], cv = pp[5];
    double x = p[0], y = p[1], z = p[2];
    double xx = x * x, yy = y * y;
    double r2 = xx + yy;
    double d1 = sqrt(r2 + z * z), d1_inv = 1.0 / d1;
    double temp = alpha <= 0.5 ? alpha / (1 - alpha) : (1 - alpha) / alpha;
    double fov = (temp + xi) / sqrt(2 * temp * xi + xi * xi + 1);
    kp[0] = kp[1] = 0.0;
    if (J) { J[0] = J[1] = J[2] = J[3] = J[4] = J[5] = 0.0; }
    if (z <= -(fov * d1)) return;
    double k = xi * d1 + z;
    double d2 = sqrt(r2 + k * k), d2_inv = 1.0 / d2;
    double norm = alpha * d2 + (1 - alpha) * k;
    double norm_inv = 1.0 / norm, norm_inv2 = norm_inv * norm_inv;
    kp[0] = fu * (x * norm_inv) + cu; kp[1] = fv * (y * norm_inv) + cv;
    if (J) {
      double xy = x * y;
      double tt2 = xi * z * d1_inv + 1;
      double dn = (xi * (1 - alpha) * d1_inv + alpha * (xi * k * d1_inv + 1) * d2_inv) * norm_inv2;
      double tmp2 = ((1 - alpha) * tt2 + alpha * k * tt2 * d2_inv) * norm_inv2;
      J[0] = fu * (norm_inv - xx * dn); J[1] = -fu * xy * dn;             J[2] = -fu * x * tmp2;
      J[3] = -fv * xy * dn;             J[4] = fv * (norm_inv - yy * dn); J[5] = -fv * y * tmp2;
    }
  }
}
// homogeneousToKeypointIntrinsicsJacobian (row-major 2 x proj_dim) and ...DistortionJacobian (2x4) for a point
// with positive homogeneous weight.  CAM/PinholeProjection.hpp:324-378, OmniProjection.hpp:381-445,
// DoubleSphereProjection.hpp:443-503.
MC_HD void intrinsics_jacobian(int model, int dist, const double* pp, const double* dk, const double* p, double* J) {
  if (model == 0) {
    double rz = 1.0 / p[2];
    double y[2] = {p[0] * rz, p[1] * rz};
    distort(dist, dk, y, nullptr);
    J[0] = y[0]; J[1] = 0; J[2] = 1; J[3] = 0;
    J[4] = 0; J[5] = y[1]; J[6] = 0; J[7] = 1;
  } else if (model == 1) {
    double xi = pp[0], fu = pp[1], fv = pp[2];
    double d = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    double rz = 1.0 / (p[2] + xi * d);
    double y[2] = {p[0] * rz, p[1] * rz};
    double jx0 = -y[0] * d * rz, jx1 = -y[1] * d * rz;
    double Jd[4];
    distort(dist, dk, y, Jd);
    J[0] = fu * (Jd[0] * jx0 + Jd[1] * jx1); J[1] = y[0]; J[2] = 0; J[3] = 1; J[4] = 0;
    J[5] = fv * (Jd[2] * jx0 + Jd[3] * jx1); J[6] = 0; J[7] = y[1]; J[8] = 0; J[9] = 1;
  } else {
    double xi = pp[0], alpha = pp[1], fu = pp[2], fv = pp[3];
    double x = p[0], y = p[1], z = p[2];
    double r2 = x * x + y * y;
    double d1 = sqrt(r2 + z * z);
    double k = xi * d1 + z;
    double d2 = sqrt(r2 + k * k), d2_inv = 1.0 / d2;
    double norm = alpha * d2 + (1 - alpha) * k;
    double norm_inv = 1.0 / norm, norm_inv2 = norm_inv * norm_inv;
    double tmp4 = (alpha - 1 - alpha * k * d2_inv) * d1 * norm_inv2;
    double tmp5 = (k - d2) * norm_inv2;
    J[0] = fu * x * tmp4; J[1] = fu * x * tmp5; J[2] = x * norm_inv; J[3] = 0; J[4] = 1; J[5] = 0;
    J[6] = fv * y * tmp4; J[7] = fv * y * tmp5; J[8] = 0; J[9] = y * norm_inv; J[10] = 0; J[11] = 1;
  }
}
MC_HD void distortion_jacobian(int model, int dist, const double* pp, const double* p, double* J) {
  double rz, fu, fv;
  if (model == 0) { rz = 1.0 / p[2]; fu = pp[0]; fv = pp[1]; }
  else {
    double d = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    rz = 1.0 / (p[2] + pp[0] * d); fu = pp[1]; fv = pp[2];
  }
  double y[2] = {p[0] * rz, p[1] * rz};
  distort_param_jacobian(dist, y, J);
#pragma unroll
  for (int j = 0; j < 4; ++j) { J[j] *= fu; J[4 + j] *= fv; }
}

// M-estimator weight, AB/src/MEstimatorPolicies.cpp:17-129
// EXPW = false compiles the Blake-Zisserman weight (exp) out: for launches that hold no such batch
template <bool EXPW = true>
MC_HD double mest_weight(int kind, double a, double e2) {
  switch (kind) {
    case 1: return e2 < a * a ? 1.0 : a / sqrt(e2);
    case 2: return 1.0 / (1.0 + e2 / a);
    case 3: { double se = a + e2; return a / (se * se); }
    case 4: return e2 < a * a ? 1.0 : (a * a) / e2;
    case 5: if (EXPW) { double ex = exp(-e2); return ex / (ex + a); } return 1.0;
    default: return 1.0;
  }
}

}  // namespace mc
