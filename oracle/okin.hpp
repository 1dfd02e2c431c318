// TEST INFRASTRUCTURE ONLY (see omat.hpp).
// okin.hpp: kinematics helpers restated from Schweizer-Messer/sm_kinematics.
#pragma once
#include "omat.hpp"

namespace orc {

// SM/sm_kinematics/src/rotations.cpp:78-84  crossMx
inline Mat crossMx(double x, double y, double z) {
  Mat C(3, 3);
  C(0, 1) = -z; C(0, 2) = y;
  C(1, 0) = z;  C(1, 2) = -x;
  C(2, 0) = -y; C(2, 1) = x;
  return C;
}
inline Mat crossMx(const Mat& p) { return crossMx(p[0], p[1], p[2]); }

// SM/sm_kinematics/src/transformations.cpp:45-53  boxMinus (4x6)
inline Mat boxMinus(const Mat& p) {
  Mat B(4, 6);
  for (int i = 0; i < 3; ++i) B(i, i) = p[3];
  B.setBlock(0, 3, crossMx(p[0], p[1], p[2]));
  return B;
}

// SM/sm_kinematics/src/transformations.cpp:132-142  boxTimes (6x6)
inline Mat boxTimes(const Mat& T) {
  Mat C = T.block(0, 0, 3, 3);
  Mat B(6, 6);
  B.setBlock(0, 0, C);
  B.setBlock(3, 3, C);
  B.setBlock(0, 3, -(crossMx(T(0, 3), T(1, 3), T(2, 3)) * C));
  return B;
}

// General 4x4 inverse by Gauss-Jordan with partial pivoting: the reference
// inverts transformation matrices with Eigen's general Matrix4d::inverse()
// (ABE/src/TransformationExpressionNode.cpp:81,88), not the rigid shortcut.
inline Mat inverse(const Mat& A) {
  int n = A.r;
  Mat a = A, inv = Mat::Identity(n);
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int i = col + 1; i < n; ++i)
      if (std::fabs(a(i, col)) > std::fabs(a(piv, col))) piv = i;
    if (piv != col)
      for (int j = 0; j < n; ++j) {
        std::swap(a(col, j), a(piv, j));
        std::swap(inv(col, j), inv(piv, j));
      }
    double d = 1.0 / a(col, col);
    for (int j = 0; j < n; ++j) { a(col, j) *= d; inv(col, j) *= d; }
    for (int i = 0; i < n; ++i) {
      if (i == col) continue;
      double f = a(i, col);
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { a(i, j) -= f * a(col, j); inv(i, j) -= f * inv(col, j); }
    }
  }
  return inv;
}

// SM/sm_kinematics/src/quaternion_algebra.cpp:77-101  quat2r  (q = [x y z w])
inline Mat quat2r(const Mat& q) {
  Mat R(3, 3);
  R(0, 0) = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
  R(0, 1) = q[0] * q[1] * 2.0 + q[2] * q[3] * 2.0;
  R(0, 2) = q[0] * q[2] * 2.0 - q[1] * q[3] * 2.0;
  R(1, 0) = q[0] * q[1] * 2.0 - q[2] * q[3] * 2.0;
  R(1, 1) = -q[0] * q[0] + q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
  R(1, 2) = q[0] * q[3] * 2.0 + q[1] * q[2] * 2.0;
  R(2, 0) = q[0] * q[2] * 2.0 + q[1] * q[3] * 2.0;
  R(2, 1) = q[0] * q[3] * (-2.0) + q[1] * q[2] * 2.0;
  R(2, 2) = -q[0] * q[0] - q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  return R;
}

// SM/sm_kinematics/src/quaternion_algebra.cpp:200-221  axisAngle2quat
inline Mat axisAngle2quat(const Mat& a) {
  double theta = a.norm();
  static const double eps4 = std::pow(2.220446049250313e-16, 0.25);
  double na;
  if (theta < eps4) na = 0.5 + (theta * theta) * (1.0 / 48.0);
  else na = std::sin(theta * 0.5) / theta;
  Mat q(4, 1);
  q[0] = a[0] * na; q[1] = a[1] * na; q[2] = a[2] * na;
  q[3] = std::cos(theta * 0.5);
  return q;
}

// SM/sm_kinematics/src/quaternion_algebra.cpp:355-372  quatInvS (4x3)
inline Mat quatInvS(const Mat& qq) {
  double q0 = 0.5 * qq[0], q1 = 0.5 * qq[1], q2 = 0.5 * qq[2], q3 = 0.5 * qq[3];
  Mat S(4, 3);
  S(0, 0) = q3;  S(0, 1) = -q2; S(0, 2) = q1;
  S(1, 0) = q2;  S(1, 1) = q3;  S(1, 2) = -q0;
  S(2, 0) = -q1; S(2, 1) = q0;  S(2, 2) = q3;
  S(3, 0) = -q0; S(3, 1) = -q1; S(3, 2) = -q2;
  return S;
}
// SM/sm_kinematics/src/quaternion_algebra.cpp:302-318  updateQuat
inline Mat updateQuat(const Mat& q, const Mat& dq) {
  Mat d = axisAngle2quat(dq);
  double ca = d[3];
  Mat r(4, 1);
  r[0] = q[0] * ca + d[0] * q[3] - d[1] * q[2] + d[2] * q[1];
  r[1] = q[1] * ca + d[0] * q[2] + d[1] * q[3] - d[2] * q[0];
  r[2] = q[2] * ca - d[0] * q[1] + d[1] * q[0] + d[2] * q[3];
  r[3] = q[3] * ca - d[0] * q[0] - d[1] * q[1] - d[2] * q[2];
  return r;
}

// SM/sm_kinematics/src/EulerAnglesYawPitchRoll.cpp:12-35 (parameters = [yaw pitch roll])
inline Mat yprToRotationMatrix(double yaw, double pitch, double roll) {
  double cx = std::cos(roll), sx = std::sin(roll);
  double cy = std::cos(pitch), sy = std::sin(pitch);
  double cz = std::cos(yaw), sz = std::sin(yaw);
  Mat C(3, 3);
  C(0, 0) = cz * cy; C(0, 1) = -sz * cx + cz * sy * sx; C(0, 2) = sz * sx + cz * sy * cx;
  C(1, 0) = sz * cy; C(1, 1) = cz * cx + sz * sy * sx;  C(1, 2) = -cz * sx + sz * sy * cx;
  C(2, 0) = -sy;     C(2, 1) = cy * sx;                 C(2, 2) = cy * cx;
  return C;
}

// SM/sm_kinematics/src/rotations.cpp:133-157  R2AxisAngle
inline Mat R2AxisAngle(const Mat& C) {
  double tr = std::max(-1.0, std::min((C(0, 0) + C(1, 1) + C(2, 2) - 1.0) * 0.5, 1.0));
  double a = std::acos(tr);
  Mat axis(3, 1);
  if (std::fabs(a) < 1e-10) return axis;
  axis[0] = C(2, 1) - C(1, 2);
  axis[1] = C(0, 2) - C(2, 0);
  axis[2] = C(1, 0) - C(0, 1);
  double n2 = axis.norm();
  if (std::fabs(n2) < 1e-10) return Mat(3, 1);
  return axis * (-a / n2);
}

// ---- RotationVector: SM/sm_kinematics/src/RotationVector.cpp ----

// :186-201 parametersToInverseSMatrix
inline Mat rvInverseSMatrix(const Mat& p) {
  double phi = std::sqrt(p.dot(p));
  if (phi == 0) return Mat::Identity(3);
  double cot = -std::sin(phi) / (std::cos(phi) - 1);
  double a1 = 1 / (phi * phi) * (1 - 0.5 * phi * cot);
  Mat K = crossMx(p);
  return Mat::Identity(3) + (0.5 * K) + (a1 * (K * K));
}

// :82-106 parametersToSMatrix
inline Mat rvSMatrix(const Mat& p) {
  double angle = p.norm();
  if (angle < 1e-14) return Mat::Identity(3);
  double recip = 1.0 / angle;
  Mat axis = p * recip;
  double st2 = std::sin(angle * 0.5);
  double st = std::sin(angle);
  double c1 = -2.0 * st2 * st2 * recip;
  double c2 = (angle - st) * recip;
  Mat cA = crossMx(axis);
  return Mat::Identity(3) + (c1 * cA) + (c2 * (cA * cA));
}

// :10-55 parametersToRotationMatrix (Rodrigues, explicit element form)
inline Mat rvRotationMatrix(const Mat& p, Mat* S = nullptr) {
  double angle = p.norm();
  Mat C(3, 3);
  if (angle < 1e-14) {
    C = Mat::Identity(3);
    if (S) *S = Mat::Identity(3);
    return C;
  }
  double recip = 1.0 / angle;
  double ax = p[0] * recip, ay = p[1] * recip, az = p[2] * recip;
  double sa = std::sin(angle), ca = std::cos(angle);
  double ax2 = ax * ax, ay2 = ay * ay, az2 = az * az;
  C(0, 0) = ax2 + ca * (1.0 - ax2);          C(0, 1) = ax * ay - ca * ax * ay + sa * az; C(0, 2) = ax * az - ca * ax * az - sa * ay;
  C(1, 0) = ax * ay - ca * ax * ay - sa * az; C(1, 1) = ay2 + ca * (1.0 - ay2);           C(1, 2) = ay * az - ca * ay * az + sa * ax;
  C(2, 0) = ax * az - ca * ax * az + sa * ay; C(2, 1) = ay * az - ca * ay * az - sa * ax; C(2, 2) = az2 + ca * (1.0 - az2);
  if (S) *S = rvSMatrix(p);
  return C;
}

// :109-183 angularVelocityAndJacobian.  omega = S(p) pdot;  J = [ d(S pdot)/dp , S ].
// The reference carries a machine-generated 3x3 for d(S(p) pdot)/dp (divides by
// |p|^5, no small-angle branch).  Restated here analytically in axis/angle form
// (axis a = p/|p|, S = I + c1(phi) a^ + c2(phi) a^ a^, exactly the factors of
// parametersToSMatrix), differentiating through a(p) and phi(p).  It is the same
// function of (p, pdot); like the reference it is singular at p = 0.
inline Mat rvAngularVelocityAndJacobian(const Mat& p, const Mat& pdot, Mat* J) {
  Mat S = rvSMatrix(p);
  Mat omega = S * pdot;
  if (J) {
    *J = Mat::Zero(3, 6);
    J->setBlock(0, 3, S);
    double phi = p.norm();
    Mat a = p * (1.0 / phi);
    double s = std::sin(phi), c = std::cos(phi), sh = std::sin(0.5 * phi);
    double c1 = -2.0 * sh * sh / phi;
    double c2 = (phi - s) / phi;
    double dc1 = -s / phi + 2.0 * sh * sh / (phi * phi);
    double dc2 = (1.0 - c) / phi - (phi - s) / (phi * phi);
    Mat axv = crossMx(a) * pdot;            // a x v
    Mat aaxv = crossMx(a) * axv;            // a x (a x v)
    Mat da = (Mat::Identity(3) - a * a.t()) * (1.0 / phi);   // d a / d p
    double adv = a.dot(pdot);
    Mat dAaxv = adv * Mat::Identity(3) + a * pdot.t() - 2.0 * (pdot * a.t());  // d(a x (a x v))/da
    Mat D = (axv * a.t()) * dc1 + (c1 * (-1.0 * crossMx(pdot))) * da
          + (aaxv * a.t()) * dc2 + (c2 * dAaxv) * da;
    J->setBlock(0, 0, D);
  }
  return omega;
}

}  // namespace orc
