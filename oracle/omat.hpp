// TEST INFRASTRUCTURE ONLY -- CPU oracle for the multical_amd parity tests.
// Nothing under oracle/ is linked, imported or executed by the product path
// (multical_amd/, include/); only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py use it, and only as the checker.
//
// omat.hpp: a deliberately plain dynamic-size dense matrix (heap storage,
// column major like Eigen::MatrixXd).  The reference does every chain-rule
// product with heap-allocated Eigen::MatrixXd temporaries; keeping that cost
// shape here is what makes this oracle usable as the "reference-shaped" CPU
// baseline.
#pragma once
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <initializer_list>

namespace orc {

struct Mat {
  int r = 0, c = 0;
  std::vector<double> v;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
  static Mat Zero(int r, int c) { return Mat(r, c); }
  static Mat Identity(int r, int c) {
    Mat m(r, c);
    for (int i = 0; i < r && i < c; ++i) m(i, i) = 1.0;
    return m;
  }
  static Mat Identity(int n) { return Identity(n, n); }
  static Mat Vec(std::initializer_list<double> l) {
    Mat m((int)l.size(), 1);
    int i = 0;
    for (double x : l) m.v[i++] = x;
    return m;
  }
  inline double& operator()(int i, int j) { return v[(size_t)j * r + i]; }
  inline double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
  inline double& operator[](int i) { return v[i]; }
  inline double operator[](int i) const { return v[i]; }
  int rows() const { return r; }
  int cols() const { return c; }
  int size() const { return r * c; }
  void setZero() { for (auto& x : v) x = 0.0; }
  Mat t() const {
    Mat m(c, r);
    for (int j = 0; j < c; ++j)
      for (int i = 0; i < r; ++i) m(j, i) = (*this)(i, j);
    return m;
  }
  Mat block(int i0, int j0, int p, int q) const {
    Mat m(p, q);
    for (int j = 0; j < q; ++j)
      for (int i = 0; i < p; ++i) m(i, j) = (*this)(i0 + i, j0 + j);
    return m;
  }
  void setBlock(int i0, int j0, const Mat& b) {
    for (int j = 0; j < b.c; ++j)
      for (int i = 0; i < b.r; ++i) (*this)(i0 + i, j0 + j) = b(i, j);
  }
  Mat col(int j) const { return block(0, j, r, 1); }
  Mat head(int n) const { return block(0, 0, n, 1); }
  Mat tail(int n) const { return block(r - n, 0, n, 1); }
  double norm() const {
    double s = 0;
    for (double x : v) s += x * x;
    return std::sqrt(s);
  }
  double dot(const Mat& o) const {
    double s = 0;
    for (size_t i = 0; i < v.size(); ++i) s += v[i] * o.v[i];
    return s;
  }
  double maxAbs() const {
    double s = 0;
    for (double x : v) s = std::fmax(s, std::fabs(x));
    return s;
  }
  Mat& operator+=(const Mat& o) {
    if (o.r != r || o.c != c) throw std::runtime_error("Mat += shape");
    for (size_t i = 0; i < v.size(); ++i) v[i] += o.v[i];
    return *this;
  }
  Mat& operator-=(const Mat& o) {
    if (o.r != r || o.c != c) throw std::runtime_error("Mat -= shape");
    for (size_t i = 0; i < v.size(); ++i) v[i] -= o.v[i];
    return *this;
  }
  Mat& operator*=(double s) {
    for (auto& x : v) x *= s;
    return *this;
  }
};

inline Mat operator*(const Mat& a, const Mat& b) {
  if (a.c != b.r) {
    char buf[128];
    snprintf(buf, sizeof buf, "Mat * shape (%dx%d)*(%dx%d)", a.r, a.c, b.r, b.c);
    throw std::runtime_error(buf);
  }
  Mat m(a.r, b.c);
  for (int j = 0; j < b.c; ++j)
    for (int k = 0; k < a.c; ++k) {
      double bkj = b(k, j);
      for (int i = 0; i < a.r; ++i) m(i, j) += a(i, k) * bkj;
    }
  return m;
}
inline Mat operator+(Mat a, const Mat& b) { a += b; return a; }
inline Mat operator-(Mat a, const Mat& b) { a -= b; return a; }
inline Mat operator*(Mat a, double s) { a *= s; return a; }
inline Mat operator*(double s, Mat a) { a *= s; return a; }
inline Mat operator-(Mat a) { a *= -1.0; return a; }

}  // namespace orc
