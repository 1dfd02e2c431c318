// TEST INFRASTRUCTURE ONLY (see omat.hpp).
// oexpr.hpp: design variables, JacobianContainer and the expression-DAG nodes that the
// Multical front-end composes, restated
// node by node so that the chain rule is applied exactly where the reference applies it.
// AB  = aslam_optimizer/aslam_backend, ABE = aslam_optimizer/aslam_backend_expressions,
// ASP = aslam_nonparametric_estimation/aslam_splines.
#pragma once
#include <map>
#include <memory>
#include <set>
#include "ocam.hpp"
#include "ospline.hpp"

namespace orc {

// ---------------------------------------------------------------- design variables
struct DesignVariable {
  bool active = true;
  int blockIndex = -1;
  int columnBase = -1;
  double scaling = 1.0;
  int id = -1;  // registration id (oracle bookkeeping only)
  virtual ~DesignVariable() {}
  virtual int minimalDimensions() const = 0;
  virtual void update(const double* dp) = 0;
  virtual void revertUpdate() = 0;
  virtual void getParameters(std::vector<double>& out) const = 0;
};

// ABE/src/EuclideanPoint.cpp:16-53
struct EuclideanPoint : DesignVariable {
  Mat p, pp;
  explicit EuclideanPoint(const double* v) : p(3, 1), pp(3, 1) { for (int i = 0; i < 3; ++i) p[i] = pp[i] = v[i]; }
  int minimalDimensions() const override { return 3; }
  void update(const double* dp) override { pp = p; for (int i = 0; i < 3; ++i) p[i] += dp[i]; }
  void revertUpdate() override { p = pp; }
  void getParameters(std::vector<double>& o) const override { o = p.v; }
};
// ABE/src/Scalar.cpp:17-52
struct ScalarDv : DesignVariable {
  double s, ps;
  explicit ScalarDv(double v) : s(v), ps(v) {}
  int minimalDimensions() const override { return 1; }
  void update(const double* dp) override { ps = s; s += dp[0]; }
  void revertUpdate() override { s = ps; }
  void getParameters(std::vector<double>& o) const override { o = {s}; }
};
// ABE/src/RotationQuaternion.cpp:10-61
struct RotationQuaternion : DesignVariable {
  Mat q, pq, C;
  explicit RotationQuaternion(const double* v) : q(4, 1), pq(4, 1) {
    for (int i = 0; i < 4; ++i) q[i] = pq[i] = v[i];
    C = quat2r(q);
  }
  int minimalDimensions() const override { return 3; }
  void update(const double* dp) override {
    pq = q;
    Mat d(3, 1);
    for (int i = 0; i < 3; ++i) d[i] = dp[i];
    q = updateQuat(q, d);
    C = quat2r(q);
  }
  void revertUpdate() override { q = pq; C = quat2r(q); }
  void getParameters(std::vector<double>& o) const override { o = q.v; }
};
// ABE/src/HomogeneousPoint.cpp:7-81
struct HomogeneousPointDv : DesignVariable {
  Mat p, pp;
  explicit HomogeneousPointDv(const double* v) : p(4, 1), pp(4, 1) {
    double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    double recip = 1.0 / n;
    for (int i = 0; i < 4; ++i) p[i] = pp[i] = v[i] * recip;
  }
  int minimalDimensions() const override { return 3; }
  void update(const double* dp) override {
    pp = p;
    Mat d(3, 1);
    for (int i = 0; i < 3; ++i) d[i] = dp[i];
    p = updateQuat(p, d);
  }
  void revertUpdate() override { p = pp; }
  void getParameters(std::vector<double>& o) const override { o = p.v; }
};
// ABE/src/EuclideanDirection.cpp:5-77
struct EuclideanDirection : DesignVariable {
  Mat C, pC;
  double magnitude;
  explicit EuclideanDirection(const double* dir) {
    magnitude = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    double u0 = dir[0] / magnitude, u1 = dir[1] / magnitude, u2 = dir[2] / magnitude;
    double y = std::asin(-u0);
    double x = std::atan2(u1, u2);
    C = yprToRotationMatrix(0.0, y, x).t();
    pC = C;
  }
  int minimalDimensions() const override { return 2; }
  void update(const double* dp) override {
    pC = C;
    C = C * yprToRotationMatrix(0.0, dp[0], dp[1]).t();
  }
  void revertUpdate() override { C = pC; }
  void getParameters(std::vector<double>& o) const override {
    // row-major C followed by the magnitude (same layout the product reports)
    o.clear();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.push_back(C(i, j));
    o.push_back(magnitude);
  }
  Mat toEuclidean() const { return magnitude * C.col(2); }
};
// ABE/include/aslam/backend/implementation/DesignVariableMappedVector.hpp:48-93 -- a view into
// the spline's coefficient storage; additive update in place.
struct MappedVector : DesignVariable {
  double* ptr;
  int D;
  std::vector<double> prev;
  MappedVector(double* p_, int D_) : ptr(p_), D(D_), prev(p_, p_ + D_) {}
  int minimalDimensions() const override { return D; }
  void update(const double* dp) override { for (int i = 0; i < D; ++i) { prev[i] = ptr[i]; ptr[i] += dp[i]; } }
  void revertUpdate() override { for (int i = 0; i < D; ++i) ptr[i] = prev[i]; }
  void getParameters(std::vector<double>& o) const override { o.assign(ptr, ptr + D); }
};
// AB/include/aslam/backend/implementation/DesignVariableAdapter.hpp:26-55 over a camera's
// projection or distortion parameter block (additive update(v), CAM/...Projection.hpp update()).
struct ParamVectorDv : DesignVariable {
  std::vector<double> p, pp;
  ParamVectorDv(const double* v, int n) : p(v, v + n), pp(v, v + n) {}
  int minimalDimensions() const override { return (int)p.size(); }
  void update(const double* dp) override { pp = p; for (size_t i = 0; i < p.size(); ++i) p[i] += dp[i]; }
  void revertUpdate() override { p = pp; }
  void getParameters(std::vector<double>& o) const override { o = p; }
};

// ABE/src/MatrixBasic.cpp:10-77: 3x3 matrix; _B (9 x n) selects the entries flagged in the update pattern, column k
// of _B being the k-th flagged entry in the row-major scan of the pattern, at row col*3+row of the column-major vec(A)
struct MatrixBasic : DesignVariable {
  Mat A, pA, B;
  MatrixBasic(const double* a /* row major */, const int* pattern /* row major */) : A(3, 3), pA(3, 3) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A(r, c) = pA(r, c) = a[3 * r + c];
    int n = 0;
    for (int i = 0; i < 9; ++i) if (pattern[i] != 0) ++n;
    B = Mat(9, n);
    int k = 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        if (pattern[3 * r + c] != 0) { B(c * 3 + r, k) = 1.0; ++k; }
  }
  int minimalDimensions() const override { return B.c; }
  void update(const double* dp) override {
    pA = A;
    for (int i = 0; i < 9; ++i) {       // dV = B d, A += unvec(dV) (column major)
      double dv = 0.0;
      for (int k = 0; k < B.c; ++k) dv += B(i, k) * dp[k];
      A(i % 3, i / 3) += dv;
    }
  }
  void revertUpdate() override { A = pA; }
  void getParameters(std::vector<double>& o) const override {
    o.clear();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o.push_back(A(r, c));   // row major, like the product reports
  }
};

// ---------------------------------------------------------------- JacobianContainer
// AB/include/aslam/backend/JacobianContainer.hpp:119-135, AB/src/JacobianContainer.cpp.
// Keyed by design variable, ordered by block index; inactive variables are dropped by add().
struct DvBlockLess {
  bool operator()(const DesignVariable* a, const DesignVariable* b) const { return a->blockIndex < b->blockIndex; }
};
struct JacobianContainer {
  typedef std::map<DesignVariable*, Mat, DvBlockLess> map_t;
  map_t jm;
  int rows;
  explicit JacobianContainer(int r) : rows(r) {}
  void clear() { jm.clear(); }
  void add(DesignVariable* dv, const Mat& J) {
    if (!dv->active) return;
    map_t::iterator it = jm.find(dv);
    if (it == jm.end()) jm.insert(std::make_pair(dv, J));
    else it->second += J;
  }
  void applyChainRule(const Mat& df_dx) {
    for (auto& kv : jm) kv.second = df_dx * kv.second;
    rows = df_dx.r;
  }
};

// ---------------------------------------------------------------- expression nodes
struct RotationNode {
  virtual ~RotationNode() {}
  virtual Mat toRotationMatrix() = 0;
  virtual void evalJ(JacobianContainer& jc, const Mat* chain) = 0;
  virtual void getDvs(std::set<DesignVariable*>& s) = 0;
};
struct EuclideanNode {
  virtual ~EuclideanNode() {}
  virtual Mat toEuclidean() = 0;
  virtual void evalJ(JacobianContainer& jc, const Mat* chain) = 0;
  virtual void getDvs(std::set<DesignVariable*>& s) = 0;
};
struct TransformationNode {
  virtual ~TransformationNode() {}
  virtual Mat toT() = 0;
  virtual void evalJ(JacobianContainer& jc, const Mat* chain) = 0;
  virtual void getDvs(std::set<DesignVariable*>& s) = 0;
};
struct HomogeneousNode {
  virtual ~HomogeneousNode() {}
  virtual Mat toHomogeneous() = 0;
  virtual void evalJ(JacobianContainer& jc, const Mat* chain) = 0;
  virtual void getDvs(std::set<DesignVariable*>& s) = 0;
};
struct ScalarNode {
  virtual ~ScalarNode() {}
  virtual double toScalar() = 0;
  virtual void evalJ(JacobianContainer& jc, const Mat* chain) = 0;
  virtual void getDvs(std::set<DesignVariable*>& s) = 0;
};
typedef std::shared_ptr<RotationNode> RotP;
typedef std::shared_ptr<EuclideanNode> EucP;
typedef std::shared_ptr<TransformationNode> TraP;
typedef std::shared_ptr<HomogeneousNode> HomP;
typedef std::shared_ptr<ScalarNode> ScaP;

inline Mat chainOr(const Mat* chain, const Mat& local) { return chain ? (*chain) * local : local; }

// --- leaves
struct RotationQuaternionNode : RotationNode {  // ABE/src/RotationQuaternion.cpp:43-56
  RotationQuaternion* dv;
  explicit RotationQuaternionNode(RotationQuaternion* d) : dv(d) {}
  Mat toRotationMatrix() override { return dv->C; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override { jc.add(dv, chain ? *chain : Mat::Identity(3)); }
  void getDvs(std::set<DesignVariable*>& s) override { s.insert(dv); }
};
struct HomogeneousPointNode : HomogeneousNode {  // ABE/src/HomogeneousPoint.cpp:39-57
  HomogeneousPointDv* dv;
  explicit HomogeneousPointNode(HomogeneousPointDv* d) : dv(d) {}
  Mat toHomogeneous() override { return dv->p; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override { jc.add(dv, chainOr(chain, quatInvS(dv->p))); }
  void getDvs(std::set<DesignVariable*>& s) override { s.insert(dv); }
};
struct EuclideanPointNode : EuclideanNode {  // ABE/src/EuclideanPoint.cpp
  EuclideanPoint* dv;
  explicit EuclideanPointNode(EuclideanPoint* d) : dv(d) {}
  Mat toEuclidean() override { return dv->p; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override { jc.add(dv, chain ? *chain : Mat::Identity(3)); }
  void getDvs(std::set<DesignVariable*>& s) override { s.insert(dv); }
};
struct EuclideanDirectionNode : EuclideanNode {  // ABE/src/EuclideanDirection.cpp:54-77
  EuclideanDirection* dv;
  explicit EuclideanDirectionNode(EuclideanDirection* d) : dv(d) {}
  Mat toEuclidean() override { return dv->toEuclidean(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J(3, 2);
    J.setBlock(0, 0, -1.0 * dv->C.col(0) * dv->magnitude);
    J.setBlock(0, 1, dv->C.col(1) * dv->magnitude);
    jc.add(dv, chainOr(chain, J));
  }
  void getDvs(std::set<DesignVariable*>& s) override { s.insert(dv); }
};
struct EuclideanConstantNode : EuclideanNode {  // ABE/src/EuclideanExpressionNode.cpp:258-282
  Mat p;
  explicit EuclideanConstantNode(const Mat& p_) : p(p_) {}
  Mat toEuclidean() override { return p; }
  void evalJ(JacobianContainer&, const Mat*) override {}
  void getDvs(std::set<DesignVariable*>&) override {}
};
struct ScalarDvNode : ScalarNode {  // ABE/src/Scalar.cpp
  ScalarDv* dv;
  explicit ScalarDvNode(ScalarDv* d) : dv(d) {}
  double toScalar() override { return dv->s; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override { jc.add(dv, chain ? *chain : Mat::Identity(1)); }
  void getDvs(std::set<DesignVariable*>& s) override { s.insert(dv); }
};
struct ScalarConstantNode : ScalarNode {  // ABE/src/ScalarExpressionNode.cpp:166-174
  double s;
  explicit ScalarConstantNode(double v) : s(v) {}
  double toScalar() override { return s; }
  void evalJ(JacobianContainer&, const Mat*) override {}
  void getDvs(std::set<DesignVariable*>&) override {}
};
struct HomogeneousConstantNode : HomogeneousNode {  // ABE/src/HomogeneousExpressionNode.cpp:93-100
  Mat p;
  explicit HomogeneousConstantNode(const Mat& p_) : p(p_) {}
  Mat toHomogeneous() override { return p; }
  void evalJ(JacobianContainer&, const Mat*) override {}
  void getDvs(std::set<DesignVariable*>&) override {}
};

// --- rotation nodes (ABE/src/RotationExpressionNode.cpp)
struct RotationMultiplyNode : RotationNode {  // :32-64
  RotP l, r;
  Mat Cl, Cr;
  RotationMultiplyNode(RotP l_, RotP r_) : l(l_), r(r_) {}
  Mat toRotationMatrix() override { Cl = l->toRotationMatrix(); Cr = r->toRotationMatrix(); return Cl * Cr; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat cr = chainOr(chain, Cl);
    r->evalJ(jc, &cr);
    l->evalJ(jc, chain);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct RotationInverseNode : RotationNode {  // :71-97
  RotP a;
  Mat C;
  explicit RotationInverseNode(RotP a_) : a(a_) {}
  Mat toRotationMatrix() override { C = a->toRotationMatrix(); return C.t(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat c = chain ? -1.0 * ((*chain) * C.t()) : -1.0 * C.t();
    a->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { a->getDvs(s); }
};
struct RotationFromTransformationNode : RotationNode {  // :100-130
  TraP T;
  explicit RotationFromTransformationNode(TraP t) : T(t) {}
  Mat toRotationMatrix() override { return T->toT().block(0, 0, 3, 3); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J(3, 6);
    J.setBlock(0, 3, Mat::Identity(3));
    Mat c = chainOr(chain, J);
    T->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { T->getDvs(s); }
};

// --- transformation nodes (ABE/src/TransformationExpressionNode.cpp, TransformationBasic.cpp)
struct TransformationBasicNode : TransformationNode {  // ABE/src/TransformationBasic.cpp:19-67
  RotP rot;
  EucP tra;
  TransformationBasicNode(RotP r, EucP t) : rot(r), tra(t) {}
  Mat toT() override {
    Mat T = Mat::Identity(4);
    T.setBlock(0, 0, rot->toRotationMatrix());
    T.setBlock(0, 3, tra->toEuclidean());
    return T;
  }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat r = tra->toEuclidean();
    Mat crR(6, 3);
    crR.setBlock(0, 0, -crossMx(r));
    crR.setBlock(3, 0, Mat::Identity(3));
    Mat c1 = chainOr(chain, crR);
    rot->evalJ(jc, &c1);
    Mat crT(6, 3);
    crT.setBlock(0, 0, Mat::Identity(3));
    Mat c2 = chainOr(chain, crT);
    tra->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { rot->getDvs(s); tra->getDvs(s); }
};
struct TransformationMultiplyNode : TransformationNode {  // :38-72
  TraP l, r;
  Mat Tl, Tr;
  TransformationMultiplyNode(TraP l_, TraP r_) : l(l_), r(r_) {}
  Mat toT() override { Tl = l->toT(); Tr = r->toT(); return Tl * Tr; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat cr = chainOr(chain, boxTimes(Tl));
    r->evalJ(jc, &cr);
    l->evalJ(jc, chain);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct TransformationInverseNode : TransformationNode {  // :79-106 (general 4x4 inverse)
  TraP a;
  Mat T;
  explicit TransformationInverseNode(TraP a_) : a(a_) {}
  Mat toT() override { T = inverse(a->toT()); return T; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat c = chainOr(chain, -boxTimes(T));
    a->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { a->getDvs(s); }
};

// --- homogeneous nodes (ABE/src/HomogeneousExpressionNode.cpp:48-88)
struct HomogeneousMultiplyNode : HomogeneousNode {
  TraP l;
  HomP r;
  Mat Tl, pr;
  HomogeneousMultiplyNode(TraP l_, HomP r_) : l(l_), r(r_) {}
  Mat toHomogeneous() override { Tl = l->toT(); pr = r->toHomogeneous(); return Tl * pr; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat c1 = chainOr(chain, boxMinus(Tl * pr));
    l->evalJ(jc, &c1);
    Mat c2 = chainOr(chain, Tl);
    r->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};

// --- euclidean nodes (ABE/src/EuclideanExpressionNode.cpp)
struct EuclideanRotateNode : EuclideanNode {  // EuclideanExpressionNodeMultiply :47-86
  RotP l;
  EucP r;
  Mat Cl, pr;
  EuclideanRotateNode(RotP l_, EucP r_) : l(l_), r(r_) {}
  Mat toEuclidean() override { Cl = l->toRotationMatrix(); pr = r->toEuclidean(); return Cl * pr; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat c1 = chainOr(chain, crossMx(Cl * pr));
    l->evalJ(jc, &c1);
    Mat c2 = chainOr(chain, Cl);
    r->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct EuclideanRotateConstantNode : EuclideanNode {  // EuclideanExpressionNodeMultiplyConstant :551-579
  RotP l;
  Mat p;
  EuclideanRotateConstantNode(RotP l_, const Mat& p_) : l(l_), p(p_) {}
  Mat toEuclidean() override { return l->toRotationMatrix() * p; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat c = chainOr(chain, crossMx(l->toRotationMatrix() * p));
    l->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); }
};
struct EuclideanCrossNode : EuclideanNode {  // :140-174
  EucP l, r;
  EuclideanCrossNode(EucP l_, EucP r_) : l(l_), r(r_) {}
  Mat toEuclidean() override { return crossMx(l->toEuclidean()) * r->toEuclidean(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat cl = chain ? -1.0 * (*chain) * crossMx(r->toEuclidean()) : -crossMx(r->toEuclidean());
    l->evalJ(jc, &cl);
    Mat cr = chainOr(chain, crossMx(l->toEuclidean()));
    r->evalJ(jc, &cr);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct EuclideanAddNode : EuclideanNode {  // :180-214
  EucP l, r;
  EuclideanAddNode(EucP l_, EucP r_) : l(l_), r(r_) {}
  Mat toEuclidean() override { return l->toEuclidean() + r->toEuclidean(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override { l->evalJ(jc, chain); r->evalJ(jc, chain); }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct EuclideanSubtractNode : EuclideanNode {  // :220-254
  EucP l, r;
  EuclideanSubtractNode(EucP l_, EucP r_) : l(l_), r(r_) {}
  Mat toEuclidean() override { return l->toEuclidean() - r->toEuclidean(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    l->evalJ(jc, chain);
    Mat c = chain ? -(*chain) : -Mat::Identity(3);
    r->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct EuclideanTranslationNode : EuclideanNode {  // :416-446
  TraP T;
  explicit EuclideanTranslationNode(TraP t) : T(t) {}
  Mat toEuclidean() override { return T->toT().block(0, 3, 3, 1); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J = Mat::Identity(3, 6);
    Mat p = T->toT().block(0, 3, 3, 1);
    J.setBlock(0, 3, crossMx(p));
    Mat c = chainOr(chain, J);
    T->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { T->getDvs(s); }
};

// MatrixBasic as an expression leaf + EuclideanExpressionNodeMatrixMultiply (ABE/src/EuclideanExpressionNode.cpp:91-134)
struct EuclideanMatrixMultiplyNode : EuclideanNode {
  MatrixBasic* M;
  EucP r;
  Mat Al, pr;
  EuclideanMatrixMultiplyNode(MatrixBasic* m, EucP r_) : M(m), r(r_) {}
  Mat toEuclidean() override { Al = M->A; pr = r->toEuclidean(); return Al * pr; }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat Jfull(3, 9);
    for (int c = 0; c < 3; ++c) for (int d = 0; d < 3; ++d) Jfull(d, 3 * c + d) = pr[c];   // [p0 I, p1 I, p2 I]
    // MatrixBasic::evaluateJacobiansImplementation(chain): add(this, chain * _B) if minimalDimensions != 0
    if (M->minimalDimensions() != 0) jc.add(M, chainOr(chain, Jfull) * M->B);
    Mat c2 = chainOr(chain, Al);
    r->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { if (M->minimalDimensions() != 0) s.insert(M); r->getDvs(s); }
};

// --- scalar nodes (ABE/src/ScalarExpressionNode.cpp)
struct ScalarAddNode : ScalarNode {  // :127-163
  ScaP l, r;
  double mul;
  ScalarAddNode(ScaP l_, ScaP r_, double m = 1.0) : l(l_), r(r_), mul(m) {}
  double toScalar() override { return l->toScalar() + mul * r->toScalar(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    l->evalJ(jc, chain);
    Mat R = Mat::Identity(1) * mul;
    Mat c = chainOr(chain, R);
    r->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct ScalarMultiplyNode : ScalarNode {  // :46-84
  ScaP l, r;
  ScalarMultiplyNode(ScaP l_, ScaP r_) : l(l_), r(r_) {}
  double toScalar() override { return l->toScalar() * r->toScalar(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat L = Mat::Identity(1) * l->toScalar(), R = Mat::Identity(1) * r->toScalar();
    Mat c1 = chainOr(chain, R), c2 = chainOr(chain, L);
    l->evalJ(jc, &c1);
    r->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct ScalarDivideNode : ScalarNode {  // :86-124
  ScaP l, r;
  ScalarDivideNode(ScaP l_, ScaP r_) : l(l_), r(r_) {}
  double toScalar() override { return l->toScalar() / r->toScalar(); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat L = Mat::Identity(1) * (1 / r->toScalar());
    Mat R = Mat::Identity(1) * (-l->toScalar() / (r->toScalar() * r->toScalar()));
    Mat c1 = chainOr(chain, L), c2 = chainOr(chain, R);
    l->evalJ(jc, &c1);
    r->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};
struct ScalarDotNode : ScalarNode {  // :208-240
  EucP l, r;
  ScalarDotNode(EucP l_, EucP r_) : l(l_), r(r_) {}
  double toScalar() override { return l->toEuclidean().dot(r->toEuclidean()); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat c1 = chainOr(chain, r->toEuclidean().t());
    l->evalJ(jc, &c1);
    Mat c2 = chainOr(chain, l->toEuclidean().t());
    r->evalJ(jc, &c2);
  }
  void getDvs(std::set<DesignVariable*>& s) override { l->getDvs(s); r->getDvs(s); }
};

// ---------------------------------------------------------------- spline design variable + nodes
// ASP/src/BSplinePoseDesignVariable.cpp:9-19: one mapped design variable per coefficient.
struct SplineDv {
  BSplinePose spline;  // also used (as plain BSpline) for 3-d bias splines
  std::vector<std::unique_ptr<MappedVector>> dvs;
  void init(int order, const std::vector<double>& knots, const Mat& coeff) {
    spline.set(order, knots, coeff);
    int D = spline.dim();
    for (int i = 0; i < spline.numVvCoefficients(); ++i)
      dvs.emplace_back(new MappedVector(&spline.coeff.v[(size_t)i * D], D));
  }
  std::vector<DesignVariable*> localDvs(double t) {
    int b = spline.localCoefficientStart(t);
    std::vector<DesignVariable*> r;
    for (int i = 0; i < spline.order; ++i) r.push_back(dvs[b + i].get());
    return r;
  }
};

// fixed-time nodes, ASP/src/BSplineExpressions.cpp:62-401, 561-661: the design-variable list
// is frozen at construction; value and Jacobian both re-evaluate the spline.
struct SplineRotationNode : RotationNode {  // :62-116
  SplineDv* s; std::vector<DesignVariable*> dvs; double t;
  SplineRotationNode(SplineDv* s_, double t_) : s(s_), dvs(s_->localDvs(t_)), t(t_) {}
  Mat toRotationMatrix() override { return s->spline.orientation(t); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J;
    s->spline.orientationAndJacobian(t, &J);
    for (size_t i = 0; i < dvs.size(); ++i) jc.add(dvs[i], chainOr(chain, J.block(0, (int)i * 6, 3, 6)));
  }
  void getDvs(std::set<DesignVariable*>& o) override { o.insert(dvs.begin(), dvs.end()); }
};
struct SplineEuclideanDerivNode : EuclideanNode {  // position/velocity/acceleration :121-293 (head<3> of evalD)
  SplineDv* s; std::vector<DesignVariable*> dvs; double t; int order;
  SplineEuclideanDerivNode(SplineDv* s_, double t_, int o) : s(s_), dvs(s_->localDvs(t_)), t(t_), order(o) {}
  Mat toEuclidean() override { return s->spline.evalD(t, order).head(3); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J;
    s->spline.evalDAndJacobian(t, order, &J);
    for (size_t i = 0; i < dvs.size(); ++i) jc.add(dvs[i], chainOr(chain, J.block(0, (int)i * 6, 3, 6)));
  }
  void getDvs(std::set<DesignVariable*>& o) override { o.insert(dvs.begin(), dvs.end()); }
};
struct SplineAngularRateBodyNode : EuclideanNode {  // :348-401 (dorder 1) and :610-661 (dorder 2)
  SplineDv* s; std::vector<DesignVariable*> dvs; double t; int dorder;
  SplineAngularRateBodyNode(SplineDv* s_, double t_, int d) : s(s_), dvs(s_->localDvs(t_)), t(t_), dorder(d) {}
  Mat toEuclidean() override { return s->spline.angularRateBodyFrame(t, dorder); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J;
    s->spline.angularRateBodyFrameAndJacobian(t, dorder, &J);
    for (size_t i = 0; i < dvs.size(); ++i) jc.add(dvs[i], chainOr(chain, J.block(0, (int)i * 6, 3, 6)));
  }
  void getDvs(std::set<DesignVariable*>& o) override { o.insert(dvs.begin(), dvs.end()); }
};
struct SplineEuclidean3Node : EuclideanNode {  // BSplineEuclideanExpressionNode :561-608 (3-d spline)
  SplineDv* s; std::vector<DesignVariable*> dvs; double t; int order;
  SplineEuclidean3Node(SplineDv* s_, double t_, int o) : s(s_), dvs(s_->localDvs(t_)), t(t_), order(o) {}
  Mat toEuclidean() override { return s->spline.evalD(t, order); }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {
    Mat J;
    s->spline.evalDAndJacobian(t, order, &J);
    for (size_t i = 0; i < dvs.size(); ++i) jc.add(dvs[i], chainOr(chain, J.block(0, (int)i * 3, 3, 3)));
  }
  void getDvs(std::set<DesignVariable*>& o) override { o.insert(dvs.begin(), dvs.end()); }
};

// TransformationTimeOffsetExpressionNode, ASP/src/BSplineExpressions.cpp:406-555
struct TimeOffsetTransformationNode : TransformationNode {
  SplineDv* s;
  ScaP time;
  int bufLeft, bufRight;
  double bufTmin, bufTmax;
  std::vector<int> localIdx;
  TimeOffsetTransformationNode(SplineDv* s_, ScaP time_, double bufferTmin, double bufferTmax) : s(s_), time(time_) {
    const BSplinePose& sp = s->spline;
    double initTime = time->toScalar();
    if (bufferTmax + initTime > sp.t_max()) bufRight = sp.numValidTimeSegments() - 1;
    else bufRight = sp.segmentIndex(initTime + bufferTmax);
    if (initTime - bufferTmin < sp.t_min()) bufLeft = 0;
    else bufLeft = sp.segmentIndex(initTime - bufferTmin);
    bufTmax = sp.timeInterval(bufRight).second;
    bufTmin = sp.timeInterval(bufLeft).first;
    std::pair<double, double> il = sp.timeInterval(bufLeft), ir = sp.timeInterval(bufRight);
    int l = sp.localCoefficientStart((il.first + il.second) / 2.0);
    int r = sp.localCoefficientStart((ir.first + ir.second) / 2.0) + sp.order - 1;
    for (int i = l; i <= r; ++i) localIdx.push_back(i);
  }
  void checkBuffer(double t) const {
    if (!(t >= bufTmin && t < bufTmax)) throw std::runtime_error("Spline Coefficient Buffer Exceeded. Set larger buffer margins!");
  }
  Mat toT() override {
    double t = time->toScalar();
    checkBuffer(t);
    return s->spline.transformation(t);
  }
  void evalJ(JacobianContainer& jc, const Mat* chain) override {  // the applyChainRule overload, :495-544
    double t = time->toScalar();
    checkBuffer(t);
    const BSplinePose& sp = s->spline;
    int minIdx = sp.localCoefficientStart(t), maxIdx = minIdx + sp.order - 1;
    Mat JS, JT;
    Mat p = sp.evalDAndJacobian(t, 0, &JS);
    sp.curveValueToTransformationAndJacobian(p, &JT);
    Mat J = JT * JS;
    int j = 0;
    for (size_t i = 0; i < localIdx.size(); ++i) {
      if (localIdx[i] >= minIdx && localIdx[i] <= maxIdx) {
        jc.add(s->dvs[localIdx[i]].get(), chainOr(chain, J.block(0, j * 6, 6, 6)));
        ++j;
      } else {
        jc.add(s->dvs[localIdx[i]].get(), chainOr(chain, Mat::Zero(6, 6)));
      }
    }
    Mat Phi_dot_c = sp.evalD(t, 1);
    Mat c = chainOr(chain, JT * Phi_dot_c);
    time->evalJ(jc, &c);
  }
  void getDvs(std::set<DesignVariable*>& o) override {
    for (int i : localIdx) o.insert(s->dvs[i].get());
    time->getDvs(o);
  }
};

}  // namespace orc
