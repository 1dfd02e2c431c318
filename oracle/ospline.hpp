// TEST INFRASTRUCTURE ONLY (see omat.hpp).
// ospline.hpp: restatement of bsplines::BSpline / BSplinePose (only what the
// Multical hot path reaches).  BSP = aslam_nonparametric_estimation/bsplines.
#pragma once
#include <algorithm>
#include "okin.hpp"

namespace orc {

struct BSpline {
  int order = 0;
  std::vector<double> knots;
  Mat coeff;  // D x ncoef, column major: coefficient i is the contiguous D-vector coeff.v[i*D..]
  std::vector<Mat> basis;

  int numValidTimeSegments() const { return std::max((int)knots.size() - 2 * order + 1, 0); }  // BSP/src/BSpline.cpp:198-202
  int numVvCoefficients() const { return coeff.c; }
  int dim() const { return coeff.r; }
  double t_min() const { return knots[order - 1]; }                  // :225-229
  double t_max() const { return knots[knots.size() - order]; }       // :231-235

  // BSP/src/BSpline.cpp:123-149
  double d_0(int k, int i, int j) const {
    double denom = knots[j + k - 1] - knots[j];
    if (denom <= 0.0) return 0.0;
    return (knots[i] - knots[j]) / denom;
  }
  double d_1(int k, int i, int j) const {
    double denom = knots[j + k - 1] - knots[j];
    if (denom <= 0.0) return 0.0;
    return (knots[i + 1] - knots[i]) / denom;
  }
  // BSP/src/BSpline.cpp:70-121  recursive basis matrix  M_k = [M_km1;0] A + [0;M_km1] B
  Mat M(int k, int i) const {
    if (k == 1) { Mat m(1, 1); m(0, 0) = 1; return m; }
    Mat Mk1 = M(k - 1, i);
    Mat M1(Mk1.r + 1, Mk1.c), M2(Mk1.r + 1, Mk1.c);
    M1.setBlock(0, 0, Mk1);
    M2.setBlock(1, 0, Mk1);
    Mat A(k - 1, k), B(k - 1, k);
    for (int idx = 0; idx < k - 1; ++idx) {
      int j = i - k + 2 + idx;
      double d0 = d_0(k, i, j);
      A(idx, idx) = 1.0 - d0; A(idx, idx + 1) = d0;
      double d1 = d_1(k, i, j);
      B(idx, idx) = -d1; B(idx, idx + 1) = d1;
    }
    return M1 * A + M2 * B;
  }
  // :58-67
  void initializeBasisMatrices() {
    basis.resize(numValidTimeSegments());
    for (size_t i = 0; i < basis.size(); ++i) basis[i] = M(order, (int)i + order - 1);
  }
  void set(int order_, const std::vector<double>& k, const Mat& c) {
    order = order_; knots = k; coeff = c;
    if (numValidTimeSegments() + order - 1 != c.c) throw std::runtime_error("BSpline: knots/coefficients mismatch");
    initializeBasisMatrices();
  }
  // :237-264  (segment width, knot index)
  std::pair<double, int> computeTIndex(double t) const {
    if (t < t_min()) throw std::runtime_error("BSpline: time out of range (low)");
    if (std::fabs(t_max() - t) < 1e-10) t = t_max();
    if (t > t_max()) throw std::runtime_error("BSpline: time out of range (high)");
    std::vector<double>::const_iterator i;
    if (t == t_max()) i = knots.end() - order;
    else i = std::upper_bound(knots.begin(), knots.end(), t);
    return std::make_pair(*i - *(i - 1), (int)(i - knots.begin()) - 1);
  }
  // :266-287
  std::pair<double, int> computeUAndTIndex(double t) const {
    std::pair<double, int> ui = computeTIndex(t);
    if (ui.first <= 0.0) return std::make_pair(0.0, ui.second);
    return std::make_pair((t - knots[ui.second]) / ui.first, ui.second);
  }
  static int dmul(int i, int d) { return d == 0 ? 1 : (d == 1 ? i : i * dmul(i - 1, d - 1)); }  // :289-297
  // :300-318
  Mat computeU(double uval, int segmentIndex, int derivativeOrder) const {
    Mat u(order, 1);
    double delta_t = knots[segmentIndex + 1] - knots[segmentIndex];
    double multiplier = 0.0;
    if (delta_t > 0.0) multiplier = 1.0 / std::pow(delta_t, derivativeOrder);
    double uu = 1.0;
    for (int i = derivativeOrder; i < order; ++i) {
      u[i] = multiplier * uu * dmul(i, derivativeOrder);
      uu = uu * uval;
    }
    return u;
  }
  // :331-349
  Mat evalD(double t, int derivativeOrder) const {
    std::pair<double, int> ui = computeUAndTIndex(t);
    Mat u = computeU(ui.first, ui.second, derivativeOrder);
    int bidx = ui.second - order + 1;
    return coeff.block(0, bidx, coeff.r, order) * (basis[bidx].t() * u);
  }
  Mat eval(double t) const { return evalD(t, 0); }
  // :351-387
  Mat evalDAndJacobian(double t, int derivativeOrder, Mat* J) const {
    std::pair<double, int> ui = computeUAndTIndex(t);
    Mat u = computeU(ui.first, ui.second, derivativeOrder);
    int bidx = ui.second - order + 1;
    Mat Bt_u = basis[bidx].t() * u;
    Mat v = coeff.block(0, bidx, coeff.r, order) * Bt_u;
    if (J) {
      int D = coeff.r;
      *J = Mat::Zero(D, order * D);
      for (int i = 0; i < order; ++i) J->setBlock(0, i * D, Mat::Identity(D) * Bt_u[i]);
    }
    return v;
  }
  // :451-456 first local coefficient index
  int localCoefficientStart(double t) const { return computeTIndex(t).second - order + 1; }
  // :1440-1444
  int segmentIndex(double t) const { return computeTIndex(t).second - order + 1; }
  // :544-549
  std::pair<double, double> timeInterval(int i) const { return std::make_pair(knots[order + i - 1], knots[order + i]); }

  // ---- quadratic integral (BSP/src/BSpline.cpp:1276-1302, 1483-1500, 1512-1548, 1585-1622) ----
  Mat Vi(int seg) const {
    Mat V(order, order);
    for (int r = 0; r < order; ++r)
      for (int c = 0; c < order; ++c) V(r, c) = 1.0 / (r + c + 1.0);
    std::pair<double, double> ti = timeInterval(seg);
    V *= (ti.second - ti.first);
    return V;
  }
  Mat Dii(int seg) const {
    std::pair<double, double> ti = timeInterval(seg);
    double dt = ti.second - ti.first;
    double recip = dt > 0 ? 1.0 / dt : 0.0;
    Mat D(order, order);
    for (int i = 0; i < order - 1; ++i) D(i, i + 1) = (i + 1.0) * recip;
    return D;
  }
  Mat Mi(int seg) const {
    int D = coeff.r;
    Mat Mm(order * D, order * D);
    for (int j = 0; j < order; ++j)
      for (int i = 0; i < D; ++i)
        for (int k = 0; k < order; ++k) Mm(i * order + k, j * D + i) = basis[seg](k, j);
    return Mm;
  }
  Mat segmentQuadraticIntegral(const Mat& W, int seg, int derivativeOrder) const {
    int D = coeff.r;
    Mat Dm = Dii(seg), V = Vi(seg), Mm = Mi(seg);
    for (int i = 0; i < derivativeOrder; ++i) V = Dm.t() * V * Dm;
    Mat WV(order * D, order * D);
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) WV.setBlock(order * r, order * c, W(r, c) * V);
    return Mm.t() * WV * Mm;
  }
};

// BSP/src/BSplinePose.cpp -- rotation parametrisation = RotationVector (sensors_and_targets.py:691)
struct BSplinePose : BSpline {
  // :394-412
  Mat curveValueToTransformationAndJacobian(const Mat& p, Mat* J) const {
    Mat T = Mat::Identity(4);
    Mat S;
    T.setBlock(0, 0, rvRotationMatrix(p.tail(3), &S));
    T.setBlock(0, 3, p.head(3));
    if (J) {
      *J = Mat::Identity(6);
      J->setBlock(0, 3, -(crossMx(p.head(3)) * S));
      J->setBlock(3, 3, S);
    }
    return T;
  }
  Mat transformation(double t) const { return curveValueToTransformationAndJacobian(eval(t), nullptr); }  // :21-24
  // :43-62
  Mat orientationAndJacobian(double t, Mat* J) const {
    Mat JS;
    Mat p = evalDAndJacobian(t, 0, &JS);
    Mat S;
    Mat C = rvRotationMatrix(p.tail(3), &S);
    if (J) {
      Mat JO(3, 6);
      JO.setBlock(0, 3, S);
      *J = JO * JS;
    }
    return C;
  }
  Mat orientation(double t) const { return rvRotationMatrix(eval(t).tail(3)); }
  // :64-83
  Mat inverseOrientationAndJacobian(double t, Mat* J) const {
    Mat JS;
    Mat p = evalDAndJacobian(t, 0, &JS);
    Mat S;
    Mat C = rvRotationMatrix(p.tail(3), &S).t();
    if (J) {
      Mat JO(3, 6);
      JO.setBlock(0, 3, S);
      *J = -(C * JO * JS);
    }
    return C;
  }
  // :207-219 / :288-300 (derivative order 1: omega_b, 2: the reference's "angular acceleration")
  Mat angularRateBodyFrame(double t, int dorder) const {
    Mat r = evalD(t, 0), v = evalD(t, dorder);
    Mat S;
    Mat C_w_b = rvRotationMatrix(r.tail(3), &S);
    return -(C_w_b.t() * S * v.tail(3));
  }
  // :222-253 / :303-331
  Mat angularRateBodyFrameAndJacobian(double t, int dorder, Mat* J) const {
    Mat Jp, Jpdot;
    Mat p = evalDAndJacobian(t, 0, &Jp).tail(3);
    Mat pdot = evalDAndJacobian(t, dorder, &Jpdot).tail(3);
    Mat Jr;
    Mat C_b_w = inverseOrientationAndJacobian(t, &Jr);
    Jpdot.setBlock(0, 0, Jp.block(3, 0, 3, Jp.c));
    Mat Jo;
    Mat omega = -(C_b_w * rvAngularVelocityAndJacobian(p, pdot, &Jo));
    Jo = -(C_b_w * Jo);
    if (J) *J = Jo * Jpdot + crossMx(omega) * Jr;
    return omega;
  }
};

}  // namespace orc
