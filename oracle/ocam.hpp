// TEST INFRASTRUCTURE ONLY (see omat.hpp).
// ocam.hpp: camera projection / distortion models restated from
// aslam_cv/aslam_cameras/include/aslam/cameras/implementation/*.hpp  (CAM/).
#pragma once
#include "omat.hpp"

namespace orc {

enum { PROJ_PINHOLE = 0, PROJ_OMNI = 1, PROJ_DS = 2 };
enum { DIST_NONE = 0, DIST_RADTAN = 1, DIST_EQUI = 2 };

struct Distortion {
  int kind = DIST_NONE;
  double k[4] = {0, 0, 0, 0};
  int dims() const { return kind == DIST_NONE ? 0 : 4; }

  // y <- distort(y); optional J = d distort / d y (2x2)
  void distort(double* y, Mat* J) const {
    if (kind == DIST_NONE) {
      if (J) *J = Mat::Identity(2);
      return;
    }
    if (kind == DIST_RADTAN) {  // CAM/RadialTangentialDistortion.hpp:4-65
      double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
      double mx2 = y[0] * y[0], my2 = y[1] * y[1], mxy = y[0] * y[1];
      double rho2 = mx2 + my2;
      double rad = k1 * rho2 + k2 * rho2 * rho2;
      if (J) {
        *J = Mat(2, 2);
        (*J)(0, 0) = 1 + rad + k1 * 2.0 * mx2 + k2 * rho2 * 4 * mx2 + 2.0 * p1 * y[1] + 6 * p2 * y[0];
        (*J)(1, 0) = k1 * 2.0 * y[0] * y[1] + k2 * 4 * rho2 * y[0] * y[1] + p1 * 2.0 * y[0] + 2.0 * p2 * y[1];
        (*J)(0, 1) = (*J)(1, 0);
        (*J)(1, 1) = 1 + rad + k1 * 2.0 * my2 + k2 * rho2 * 4 * my2 + 6 * p1 * y[1] + 2.0 * p2 * y[0];
      }
      double y0 = y[0], y1 = y[1];
      y[0] += y0 * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2);
      y[1] += y1 * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
      return;
    }
    // DIST_EQUI: CAM/EquidistantDistortion.hpp:4-183.  The reference Jacobian is a MATLAB
    // dump of d/dy [ y * thetad(atan r)/r ]; restated here as that derivative:
    //   s = thetad/r,  J = s I + (y y^T) * (thetad'/(r^2 (1+r^2)) - thetad/r^3)
    // (no r>1e-8 guard on the Jacobian, exactly like the reference: NaN at the centre).
    double r = std::sqrt(y[0] * y[0] + y[1] * y[1]);
    double th = std::atan(r);
    double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    double poly = 1 + k[0] * th2 + k[1] * th4 + k[2] * th6 + k[3] * th8;
    double thd = th * poly;
    if (J) {
      double dthd = 1 + 3 * k[0] * th2 + 5 * k[1] * th4 + 7 * k[2] * th6 + 9 * k[3] * th8;
      double s = thd / r;
      double g = dthd / (r * r * (1.0 + r * r)) - thd / (r * r * r);
      *J = Mat(2, 2);
      (*J)(0, 0) = s + y[0] * y[0] * g;
      (*J)(0, 1) = y[0] * y[1] * g;
      (*J)(1, 0) = y[0] * y[1] * g;
      (*J)(1, 1) = s + y[1] * y[1] * g;
    }
    double scaling = (r > 1e-8) ? thd / r : 1.0;
    y[0] *= scaling;
    y[1] *= scaling;
  }

  // d distort(y) / d k   (2x4)
  Mat parameterJacobian(const double* y) const {
    Mat J(2, 4);
    if (kind == DIST_RADTAN) {  // CAM/RadialTangentialDistortion.hpp:152-182
      double y0 = y[0], y1 = y[1], r2 = y0 * y0 + y1 * y1, r4 = r2 * r2;
      J(0, 0) = y0 * r2; J(0, 1) = y0 * r4; J(0, 2) = 2.0 * y0 * y1;      J(0, 3) = r2 + 2.0 * y0 * y0;
      J(1, 0) = y1 * r2; J(1, 1) = y1 * r4; J(1, 2) = r2 + 2.0 * y1 * y1; J(1, 3) = 2.0 * y0 * y1;
    } else if (kind == DIST_EQUI) {  // CAM/EquidistantDistortion.hpp:243-273
      double r = std::sqrt(y[0] * y[0] + y[1] * y[1]);
      double th = std::atan(r);
      double pw = th * th * th;
      for (int j = 0; j < 4; ++j) {
        J(0, j) = y[0] * pw / r;
        J(1, j) = y[1] * pw / r;
        pw *= th * th;
      }
    }
    return J;
  }
};

struct Camera {
  int model = PROJ_PINHOLE;
  // projection parameter vector in the reference's update() order:
  // pinhole (fu fv cu cv) | omni (xi fu fv cu cv) | ds (xi alpha fu fv cu cv)
  std::vector<double> proj;
  Distortion dist;

  int projDims() const { return model == PROJ_PINHOLE ? 4 : (model == PROJ_OMNI ? 5 : 6); }
  double fu() const { return proj[projDims() - 4]; }
  double fv() const { return proj[projDims() - 3]; }
  double cu() const { return proj[projDims() - 2]; }
  double cv() const { return proj[projDims() - 1]; }

  // euclideanToKeypoint (+ optional 2x3 Jacobian).  Returns false where the reference
  // bails out before writing the keypoint (omni/ds field-of-view test); the error terms
  // ignore the flag and the reference then reads an uninitialised keypoint -- here the
  // keypoint and Jacobian are defined as zero in that case.
  bool e2k(const double* p, double* out, Mat* J) const {
    out[0] = out[1] = 0.0;
    if (J) *J = Mat(2, 3);
    if (model == PROJ_PINHOLE) {  // CAM/PinholeProjection.hpp:73-145
      double rz = 1.0 / p[2], rz2 = rz * rz;
      out[0] = p[0] * rz; out[1] = p[1] * rz;
      Mat Jd;
      dist.distort(out, J ? &Jd : nullptr);
      if (J) {
        (*J)(0, 0) = fu() * Jd(0, 0) * rz;
        (*J)(0, 1) = fu() * Jd(0, 1) * rz;
        (*J)(0, 2) = -fu() * (p[0] * Jd(0, 0) + p[1] * Jd(0, 1)) * rz2;
        (*J)(1, 0) = fv() * Jd(1, 0) * rz;
        (*J)(1, 1) = fv() * Jd(1, 1) * rz;
        (*J)(1, 2) = -fv() * (p[0] * Jd(1, 0) + p[1] * Jd(1, 1)) * rz2;
      }
      out[0] = fu() * out[0] + cu();
      out[1] = fv() * out[1] + cv();
      return p[2] > 0;
    }
    if (model == PROJ_OMNI) {  // CAM/OmniProjection.hpp:72-180
      double xi = proj[0];
      double fov = (xi <= 1.0) ? xi : 1 / xi;  // :632
      double d = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      if (p[2] <= -(fov * d)) return false;
      double rz = 1.0 / (p[2] + xi * d);
      out[0] = p[0] * rz; out[1] = p[1] * rz;
      Mat Jn(2, 3);
      if (J) {
        rz = rz * rz / d;
        Jn(0, 0) = rz * (d * p[2] + xi * (p[1] * p[1] + p[2] * p[2]));
        Jn(1, 0) = -rz * xi * p[0] * p[1];
        Jn(0, 1) = Jn(1, 0);
        Jn(1, 1) = rz * (d * p[2] + xi * (p[0] * p[0] + p[2] * p[2]));
        rz = rz * (-xi * p[2] - d);
        Jn(0, 2) = p[0] * rz;
        Jn(1, 2) = p[1] * rz;
      }
      Mat Jd;
      dist.distort(out, J ? &Jd : nullptr);
      if (J) {
        Mat F(2, 2);
        F(0, 0) = fu(); F(1, 1) = fv();
        *J = F * Jd * Jn;
      }
      out[0] = fu() * out[0] + cu();
      out[1] = fv() * out[1] + cv();
      return true;
    }
    // PROJ_DS: CAM/DoubleSphereProjection.hpp:90-221 (no distortion, :63)
    double xi = proj[0], alpha = proj[1];
    double x = p[0], y = p[1], z = p[2];
    double xx = x * x, yy = y * y, zz = z * z;
    double r2 = xx + yy;
    double d1 = std::sqrt(r2 + zz), d1_inv = 1.0 / d1;
    double temp = alpha <= 0.5 ? alpha / (1 - alpha) : (1 - alpha) / alpha;  // :704-705
    double fov = (temp + xi) / std::sqrt(2 * temp * xi + xi * xi + 1);
    if (z <= -(fov * d1)) return false;
    double k = xi * d1 + z;
    double d2 = std::sqrt(r2 + k * k), d2_inv = 1.0 / d2;
    double norm = alpha * d2 + (1 - alpha) * k;
    double norm_inv = 1.0 / norm, norm_inv2 = norm_inv * norm_inv;
    out[0] = fu() * (x * norm_inv) + cu();
    out[1] = fv() * (y * norm_inv) + cv();
    if (J) {
      double xy = x * y;
      double tt2 = xi * z * d1_inv + 1;
      double d_norm_d_r2 = (xi * (1 - alpha) * d1_inv + alpha * (xi * k * d1_inv + 1) * d2_inv) * norm_inv2;
      double tmp2 = ((1 - alpha) * tt2 + alpha * k * tt2 * d2_inv) * norm_inv2;
      (*J)(0, 0) = fu() * (norm_inv - xx * d_norm_d_r2);
      (*J)(1, 0) = -fv() * xy * d_norm_d_r2;
      (*J)(0, 1) = -fu() * xy * d_norm_d_r2;
      (*J)(1, 1) = fv() * (norm_inv - yy * d_norm_d_r2);
      (*J)(0, 2) = -fu() * x * tmp2;
      (*J)(1, 2) = -fv() * y * tmp2;
    }
    return true;
  }

  // homogeneousToKeypoint value path: sign branch on ph[3] for every model
  // (CAM/PinholeProjection.hpp:159-162, OmniProjection.hpp:194-197, DoubleSphereProjection.hpp:234-237)
  void h2k(const Mat& ph, double* out) const {
    double p[3] = {ph[0], ph[1], ph[2]};
    if (ph[3] < 0) { p[0] = -p[0]; p[1] = -p[1]; p[2] = -p[2]; }
    e2k(p, out, nullptr);
  }
  // homogeneousToKeypoint with Jacobian (2x4, last column zero).  Pinhole returns early
  // WITHOUT the sign branch (CAM/PinholeProjection.hpp:185-186; :188-197 is dead code);
  // omni / ds negate p and J (OmniProjection.hpp:219-228; DoubleSphereProjection.hpp:259-268).
  void h2kJ(const Mat& ph, double* out, Mat* J4) const {
    double p[3] = {ph[0], ph[1], ph[2]};
    bool neg = (model != PROJ_PINHOLE) && ph[3] < 0;
    if (neg) { p[0] = -p[0]; p[1] = -p[1]; p[2] = -p[2]; }
    Mat J3;
    e2k(p, out, &J3);
    *J4 = Mat(2, 4);
    J4->setBlock(0, 0, neg ? -J3 : J3);
  }

  // homogeneousToKeypointIntrinsicsJacobian (sign branch for all models)
  Mat intrinsicsJacobian(const Mat& ph) const {
    double p[3] = {ph[0], ph[1], ph[2]};
    if (ph[3] < 0.0) { p[0] = -p[0]; p[1] = -p[1]; p[2] = -p[2]; }
    Mat J(2, projDims());
    if (model == PROJ_PINHOLE) {  // CAM/PinholeProjection.hpp:324-353
      double rz = 1.0 / p[2];
      double kp[2] = {p[0] * rz, p[1] * rz};
      dist.distort(kp, nullptr);
      J(0, 0) = kp[0]; J(0, 2) = 1; J(1, 1) = kp[1]; J(1, 3) = 1;
    } else if (model == PROJ_OMNI) {  // CAM/OmniProjection.hpp:381-420
      double xi = proj[0];
      double d = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      double rz = 1.0 / (p[2] + xi * d);
      double kp[2] = {p[0] * rz, p[1] * rz};
      Mat Jxi(2, 1);
      Jxi[0] = -kp[0] * d * rz; Jxi[1] = -kp[1] * d * rz;
      Mat Jd;
      dist.distort(kp, &Jd);
      Mat F(2, 2);
      F(0, 0) = fu(); F(1, 1) = fv();
      J.setBlock(0, 0, F * Jd * Jxi);
      J(0, 1) = kp[0]; J(0, 3) = 1; J(1, 2) = kp[1]; J(1, 4) = 1;
    } else {  // CAM/DoubleSphereProjection.hpp:443-503
      double xi = proj[0], alpha = proj[1];
      double x = p[0], y = p[1], z = p[2];
      double r2 = x * x + y * y;
      double d1 = std::sqrt(r2 + z * z);
      double k = xi * d1 + z;
      double d2 = std::sqrt(r2 + k * k), d2_inv = 1.0 / d2;
      double norm = alpha * d2 + (1 - alpha) * k;
      double norm_inv = 1.0 / norm, norm_inv2 = norm_inv * norm_inv;
      double tmp4 = (alpha - 1 - alpha * k * d2_inv) * d1 * norm_inv2;
      double tmp5 = (k - d2) * norm_inv2;
      J(0, 0) = fu() * x * tmp4; J(1, 0) = fv() * y * tmp4;
      J(0, 1) = fu() * x * tmp5; J(1, 1) = fv() * y * tmp5;
      J(0, 2) = x * norm_inv; J(0, 4) = 1; J(1, 3) = y * norm_inv; J(1, 5) = 1;
    }
    return J;
  }

  // homogeneousToKeypointDistortionJacobian (CAM/PinholeProjection.hpp:355-378, OmniProjection.hpp:422-445)
  Mat distortionJacobian(const Mat& ph) const {
    double p[3] = {ph[0], ph[1], ph[2]};
    if (ph[3] < 0.0) { p[0] = -p[0]; p[1] = -p[1]; p[2] = -p[2]; }
    double rz;
    if (model == PROJ_PINHOLE) rz = 1.0 / p[2];
    else {
      double d = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      rz = 1.0 / (p[2] + proj[0] * d);
    }
    double kp[2] = {p[0] * rz, p[1] * rz};
    Mat J = dist.parameterJacobian(kp);
    for (int j = 0; j < J.c; ++j) { J(0, j) *= fu(); J(1, j) *= fv(); }
    return J;
  }
};

}  // namespace orc
