// TEST INFRASTRUCTURE ONLY (see omat.hpp).
// oterms.hpp: error terms, M-estimators, SparseBlockMatrix, the BlockCholesky linear system
// solver and the Optimizer2 / Levenberg-Marquardt driver, restated from
// aslam_optimizer/aslam_backend (AB), sparse_block_matrix (SBM), kalibr (KAL),
// aslam_cv_error_terms (CVE), aslam_cv_backend (CVB) and aslam_splines (ASP).
#pragma once
#include <thread>
#include "oexpr.hpp"

namespace orc {

// ---------------------------------------------------------------- M-estimators (AB/src/MEstimatorPolicies.cpp:17-129)
enum { MEST_NONE = 0, MEST_HUBER = 1, MEST_CAUCHY = 2, MEST_GEMAN_MCCLURE = 3, MEST_BZ = 4, MEST_BLAKE_ZISSERMAN = 5 };
struct MEstimator {
  int kind = MEST_NONE;
  double a = 0.0;  // Huber/Bz: k ; Cauchy/GM: sigma^2 ; BlakeZisserman: epsilon (precomputed by the caller)
  double getWeight(double e2) const {
    switch (kind) {
      case MEST_HUBER: return e2 < a * a ? 1.0 : a / std::sqrt(e2);
      case MEST_CAUCHY: return 1.0 / (1.0 + e2 / a);
      case MEST_GEMAN_MCCLURE: { double se = a + e2; return a / (se * se); }
      case MEST_BZ: return e2 < a * a ? 1.0 : (a * a) / e2;
      case MEST_BLAKE_ZISSERMAN: return std::exp(-e2) / (std::exp(-e2) + a);
      default: return 1.0;
    }
  }
};

// ---------------------------------------------------------------- SparseBlockMatrix
// SBM/include/sparse_block_matrix/implementation/sparse_block_matrix.hpp:96-143, 677-729:
// one std::map<blockRow, block> per block column; blocks allocated on first touch.
struct SparseBlockMatrix {
  std::vector<int> rowEnd, colEnd;  // cumulative sizes (like _rowBlockIndices/_colBlockIndices)
  std::vector<std::map<int, Mat>> cols;
  void init(const std::vector<int>& dims) {
    rowEnd.clear();
    int s = 0;
    for (int d : dims) { s += d; rowEnd.push_back(s); }
    colEnd = rowEnd;
    cols.assign(dims.size(), std::map<int, Mat>());
  }
  int bRows() const { return (int)rowEnd.size(); }
  int rows() const { return rowEnd.empty() ? 0 : rowEnd.back(); }
  int rowsOfBlock(int r) const { return r ? rowEnd[r] - rowEnd[r - 1] : rowEnd[0]; }
  int colsOfBlock(int c) const { return c ? colEnd[c] - colEnd[c - 1] : colEnd[0]; }
  int rowBaseOfBlock(int r) const { return r ? rowEnd[r - 1] : 0; }
  int colBaseOfBlock(int c) const { return c ? colEnd[c - 1] : 0; }
  Mat* block(int r, int c, bool alloc) {
    auto it = cols[c].find(r);
    if (it == cols[c].end()) {
      if (!alloc) return nullptr;
      it = cols[c].insert(std::make_pair(r, Mat(rowsOfBlock(r), colsOfBlock(c)))).first;
    }
    return &it->second;
  }
  void clearValues() { for (auto& c : cols) for (auto& kv : c) kv.second.setZero(); }  // clear(false)
  // fillCCS(Cp, Ci, Cx, upperTriangle=true), :704-729
  int fillCCS(std::vector<int>& Cp, std::vector<int>& Ci, std::vector<double>& Cx, bool upper) const {
    Cp.clear(); Ci.clear(); Cx.clear();
    int nz = 0;
    for (size_t i = 0; i < cols.size(); ++i) {
      int cstart = colBaseOfBlock((int)i), csize = colsOfBlock((int)i);
      for (int c = 0; c < csize; ++c) {
        Cp.push_back(nz);
        for (auto& kv : cols[i]) {
          const Mat& b = kv.second;
          int rstart = rowBaseOfBlock(kv.first);
          int n = b.r;
          if (upper && rstart == cstart) n = c + 1;
          for (int r = 0; r < n; ++r) { Cx.push_back(b(r, c)); Ci.push_back(rstart + r); ++nz; }
        }
      }
    }
    Cp.push_back(nz);
    return nz;
  }
  // y = H_sym * x using the stored upper block triangle (for rhsJtJrhs; SBM multiplySymmetricUpperTriangle)
  void multiplySymUpper(const std::vector<double>& x, std::vector<double>& y) const {
    y.assign(rows(), 0.0);
    for (size_t c = 0; c < cols.size(); ++c) {
      int cb = colBaseOfBlock((int)c);
      for (auto& kv : cols[c]) {
        int rb = rowBaseOfBlock(kv.first);
        const Mat& b = kv.second;
        bool diag = (kv.first == (int)c);
        for (int j = 0; j < b.c; ++j)
          for (int i = 0; i < b.r; ++i) {
            if (diag) { if (i <= j) { y[rb + i] += b(i, j) * x[cb + j]; if (i != j) y[cb + j] += b(i, j) * x[rb + i]; } }
            else { y[rb + i] += b(i, j) * x[cb + j]; y[cb + j] += b(i, j) * x[rb + i]; }
          }
      }
    }
  }
};

// ---------------------------------------------------------------- error terms
// AB/include/aslam/backend/implementation/ErrorTerm.hpp:96-192, AB/src/ErrorTerm.cpp:20-24
struct ErrorTerm {
  int dim;
  int rowBase = 0;
  double squaredError = 0.0;  // raw chi^2 cached by the last evaluateError()
  // Terms built over the SAME expression nodes (the corners of one image observation share the transformation chain, whose
  // nodes cache their operands like the reference's do) carry the same non-negative group: the threaded evaluateError
  // never cuts a group in two, so no node is ever evaluated by two threads.  -1: shares no caching node.
  int shareGroup = -1;
  MEstimator mest;
  Mat error;       // D x 1
  Mat sqrtInvR;    // D x D
  explicit ErrorTerm(int d) : dim(d), error(d, 1), sqrtInvR(Mat::Identity(d)) {}
  virtual ~ErrorTerm() {}
  virtual double evaluateErrorImplementation() = 0;
  virtual void evaluateJacobiansImplementation(JacobianContainer& jc) = 0;
  virtual void getDvs(std::set<DesignVariable*>& s) = 0;

  double evaluateError() {
    squaredError = evaluateErrorImplementation();
    return mest.getWeight(squaredError) * squaredError;
  }
  double evaluateChiSquaredError() const { Mat e = sqrtInvR.t() * error; return e.dot(e); }
  void evaluateJacobians(JacobianContainer& jc) { jc.clear(); evaluateJacobiansImplementation(jc); }
  virtual void getWeightedError(Mat& e, bool useM) const {
    double sw = useM ? std::sqrt(mest.getWeight(squaredError)) : 1.0;
    e = sqrtInvR.t() * error * sw;
  }
  virtual void getWeightedJacobians(JacobianContainer& jc, bool useM) {
    evaluateJacobians(jc);
    jc.applyChainRule(sqrtInvR.t());
    double sw = useM ? std::sqrt(mest.getWeight(squaredError)) : 1.0;
    for (auto& kv : jc.jm) kv.second *= sw * kv.first->scaling;
  }
  // buildHessianImplementation + JacobianContainer::evaluateHessian (AB/src/JacobianContainer.cpp:103-167)
  virtual void buildHessian(SparseBlockMatrix& H, std::vector<double>& rhs, bool useM) {
    JacobianContainer J(dim);
    evaluateJacobians(J);
    double sw = useM ? std::sqrt(mest.getWeight(squaredError)) : 1.0;
    Mat S = sw * sqrtInvR;
    JacobianContainer::map_t mc = J.jm;  // the reference copies the map too
    for (auto& kv : mc) kv.second = kv.first->scaling * (S.t() * kv.second);
    Mat e = S.t() * error;
    for (auto it = mc.begin(); it != mc.end(); ++it) {
      int b1 = it->first->blockIndex;
      Mat g = it->second.t() * e;
      int rb = H.rowBaseOfBlock(b1);
      for (int k = 0; k < g.r; ++k) rhs[rb + k] -= g[k];
      for (auto jt = it; jt != mc.end(); ++jt) {
        Mat* blk = H.block(b1, jt->first->blockIndex, true);
        *blk += it->second.t() * jt->second;
      }
    }
  }
};

// --- reprojection (CVE/.../SimpleReprojectionError.hpp:43-64, ReprojectionError.hpp:50-77,
//     CVB/.../CameraDesignVariable.hpp:39-54)
struct CameraDv {
  Camera cam;
  ParamVectorDv* projDv = nullptr;  // parameters live in the dvs; cam is refreshed from them
  ParamVectorDv* distDv = nullptr;
  void sync() {   // writes only when a parameter changed (after an update): concurrent evaluateError threads only read
    if (cam.proj != projDv->p) cam.proj = projDv->p;
    if (distDv) for (int i = 0; i < 4; ++i) if (cam.dist.k[i] != distDv->p[i]) cam.dist.k[i] = distDv->p[i];
  }
};
struct ReprojectionError : ErrorTerm {
  Mat y;
  HomP point;
  CameraDv* camera;
  bool withIntrinsics;  // false: SimpleReprojectionError, true: ReprojectionError
  ReprojectionError(const double* y_, double sinv, HomP p, CameraDv* c, bool wi)
      : ErrorTerm(2), y(2, 1), point(p), camera(c), withIntrinsics(wi) {
    y[0] = y_[0]; y[1] = y_[1];
    sqrtInvR = Mat::Identity(2) * sinv;
  }
  double evaluateErrorImplementation() override {
    camera->sync();
    Mat p = point->toHomogeneous();
    double hat[2];
    camera->cam.h2k(p, hat);
    error[0] = y[0] - hat[0]; error[1] = y[1] - hat[1];
    Mat invR = sqrtInvR * sqrtInvR.t();
    return error.dot(invR * error);
  }
  void evaluateJacobiansImplementation(JacobianContainer& jc) override {
    camera->sync();
    Mat p = point->toHomogeneous();
    double hat[2];
    Mat J;
    camera->cam.h2kJ(p, hat, &J);
    Mat c = -J;
    point->evalJ(jc, &c);
    if (withIntrinsics) {
      if (camera->projDv->active) jc.add(camera->projDv, -camera->cam.intrinsicsJacobian(p));
      if (camera->distDv && camera->distDv->active) jc.add(camera->distDv, -camera->cam.distortionJacobian(p));
    }
  }
  void getDvs(std::set<DesignVariable*>& s) override {
    point->getDvs(s);
    if (withIntrinsics) { s.insert(camera->projDv); if (camera->distDv) s.insert(camera->distDv); }
  }
};

// --- KAL/src/EuclideanError.cpp:5-27
struct EuclideanError : ErrorTerm {
  Mat meas;
  EucP pred;
  EuclideanError(const double* m, double sinv, EucP p) : ErrorTerm(3), meas(3, 1), pred(p) {
    for (int i = 0; i < 3; ++i) meas[i] = m[i];
    sqrtInvR = Mat::Identity(3) * sinv;
  }
  double evaluateErrorImplementation() override { error = pred->toEuclidean() - meas; return evaluateChiSquaredError(); }
  void evaluateJacobiansImplementation(JacobianContainer& jc) override { pred->evalJ(jc, nullptr); }
  void getDvs(std::set<DesignVariable*>& s) override { pred->getDvs(s); }
};
// --- ABE/src/ErrorTermTransformation.cpp:19-81 (the weightRotation / weightTranslation constructor)
struct ErrorTermTransformation : ErrorTerm {
  TraP T;
  Mat priorInv;     // inverse(prior).T()
  Mat errorMatrix;
  ErrorTermTransformation(TraP T_, const Mat& priorInv_, double weightRotation, double weightTranslation)
      : ErrorTerm(6), T(T_), priorInv(priorInv_) {
    // invR = diag(wt wt wt wr wr wr); its LDL^T square root is the diagonal of square roots
    for (int i = 0; i < 6; ++i) sqrtInvR(i, i) = std::sqrt(i < 3 ? weightTranslation : weightRotation);
  }
  double evaluateErrorImplementation() override {
    errorMatrix = T->toT() * priorInv;
    error.setBlock(0, 0, errorMatrix.block(0, 3, 3, 1));
    error.setBlock(3, 0, R2AxisAngle(errorMatrix.block(0, 0, 3, 3)));
    return evaluateChiSquaredError();
  }
  void evaluateJacobiansImplementation(JacobianContainer& jc) override {
    Mat J = Mat::Identity(6);
    J.setBlock(0, 3, crossMx(errorMatrix.block(0, 3, 3, 1)));
    J.setBlock(3, 3, rvInverseSMatrix(R2AxisAngle(errorMatrix.block(0, 0, 3, 3))));
    T->evalJ(jc, &J);
  }
  void getDvs(std::set<DesignVariable*>& s) override { T->getDvs(s); }
};
// --- KAL/src/ScalarError.cpp:6-31
struct ScalarError : ErrorTerm {
  double meas;
  ScaP pred;
  ScalarError(double m, double sinv, ScaP p) : ErrorTerm(1), meas(m), pred(p) { sqrtInvR(0, 0) = sinv; }
  double evaluateErrorImplementation() override { error[0] = pred->toScalar() - meas; return evaluateChiSquaredError(); }
  void evaluateJacobiansImplementation(JacobianContainer& jc) override { pred->evalJ(jc, nullptr); }
  void getDvs(std::set<DesignVariable*>& s) override { pred->getDvs(s); }
};

// --- ASP/include/aslam/backend/implementation/BSplineMotionError.hpp:28-166
struct BSplineMotionError : ErrorTerm {
  SplineDv* s;
  SparseBlockMatrix Q;  // full symmetric block band (BSP/src/BSpline.cpp:1585-1622)
  BSplineMotionError(SplineDv* s_, const Mat& W, int errorTermOrder) : ErrorTerm(1), s(s_) {
    const BSpline& sp = s->spline;
    int D = sp.dim(), n = sp.numVvCoefficients();
    if (sp.order <= errorTermOrder && sp.order >= 2) errorTermOrder = sp.order - 1;
    Q.init(std::vector<int>(n, D));
    for (int seg = 0; seg < sp.numValidTimeSegments(); ++seg) {
      Mat Qs = sp.segmentQuadraticIntegral(W, seg, errorTermOrder);
      for (int i = 0; i < sp.order; ++i)
        for (int j = 0; j < sp.order; ++j) *Q.block(seg + i, seg + j, true) += Qs.block(i * D, j * D, D, D);
    }
  }
  void Qc(std::vector<double>& out) const {  // full (non-symmetric-storage) block multiply
    const BSpline& sp = s->spline;
    int D = sp.dim();
    out.assign((size_t)Q.rows(), 0.0);
    for (size_t c = 0; c < Q.cols.size(); ++c)
      for (auto& kv : Q.cols[c]) {
        const Mat& b = kv.second;
        for (int j = 0; j < D; ++j)
          for (int i = 0; i < D; ++i) out[kv.first * D + i] += b(i, j) * sp.coeff.v[c * D + j];
      }
  }
  double evaluateErrorImplementation() override {
    std::vector<double> q;
    Qc(q);
    double r = 0;
    for (size_t i = 0; i < q.size(); ++i) r += s->spline.coeff.v[i] * q[i];
    return r;  // never sets _error (:62-77)
  }
  void evaluateJacobiansImplementation(JacobianContainer&) override { throw std::runtime_error("This is currently unsupported"); }
  void getWeightedError(Mat& e, bool) const override { e = Mat(1, 1); }  // reference leaves garbage; zero here
  void getDvs(std::set<DesignVariable*>& o) override { for (auto& d : s->dvs) o.insert(d.get()); }
  void buildHessian(SparseBlockMatrix& H, std::vector<double>& rhs, bool) override {  // :93-166
    std::vector<double> bu;
    Qc(bu);
    int n = (int)s->dvs.size();
    for (int i = 0; i < n; ++i) {
      DesignVariable* di = s->dvs[i].get();
      if (!di->active) continue;
      int colBlock = di->blockIndex, rows = di->minimalDimensions();
      int rowBase = H.colBaseOfBlock(colBlock);
      for (int j = 0; j <= i; ++j) {
        DesignVariable* dj = s->dvs[j].get();
        if (!dj->active) continue;
        Mat* Qb = Q.block(i, j, false);
        if (Qb) *H.block(dj->blockIndex, colBlock, true) += *Qb;
      }
      for (int k = 0; k < rows; ++k) rhs[rowBase + k] -= bu[(size_t)i * rows + k];
    }
  }
};

// ---------------------------------------------------------------- dense Cholesky (stand-in for CHOLMOD in tests)
// CHOLMOD is outside the GPU parity scope (north_star keeps it on the host); the oracle only
// needs *a* trustworthy factorisation for its small end-to-end LM runs.
inline bool denseCholeskySolve(int n, std::vector<double>& A /* col-major full sym */, std::vector<double>& b) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)j * n + i];
      for (int k = 0; k < j; ++k) s -= A[(size_t)k * n + i] * A[(size_t)k * n + j];
      A[(size_t)j * n + i] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[(size_t)k * n + i] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= A[(size_t)i * n + k] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

// ---------------------------------------------------------------- linear system solver
// AB/src/LinearSystemSolver.cpp:12-138 + AB/src/BlockCholeskyLinearSystemSolver.cpp:34-106,172-176
struct BlockCholeskySolver {
  std::vector<ErrorTerm*> errorTerms;
  SparseBlockMatrix H;
  std::vector<double> rhs, e, diagCond;
  int JRows = 0, JCols = 0;
  bool useDiag = false;

  void initMatrixStructure(const std::vector<DesignVariable*>& dvs, const std::vector<ErrorTerm*>& errs, bool useDiagonalConditioner) {
    errorTerms = errs;
    JRows = 0;
    for (auto* t : errs) JRows += t->dim;
    JCols = 0;
    std::vector<int> dims;
    for (size_t i = 0; i < dvs.size(); ++i) { dvs[i]->blockIndex = (int)i; dims.push_back(dvs[i]->minimalDimensions()); JCols += dims.back(); }
    e.assign(JRows, 0.0);
    rhs.assign(JCols, 0.0);
    diagCond.assign(JCols, 0.0);
    useDiag = useDiagonalConditioner;
    H.init(dims);
  }
  void evaluateRange(size_t a, size_t b, bool useM, double* acc) {
    Mat we;
    for (size_t i = a; i < b; ++i) {
      *acc += errorTerms[i]->evaluateError();
      errorTerms[i]->getWeightedError(we, useM);
      for (int k = 0; k < errorTerms[i]->dim; ++k) e[errorTerms[i]->rowBase + k] = -we[k];
    }
  }
  double evaluateError(size_t nThreads, bool useM) {  // setupThreadedJob: contiguous ranges
    nThreads = std::max<size_t>(1, nThreads);
    std::vector<double> local(nThreads, 0.0);
    if (nThreads <= 1) evaluateRange(0, errorTerms.size(), useM, &local[0]);
    else {
      nThreads = std::min(nThreads, errorTerms.size());
      std::vector<size_t> idx(nThreads + 1, 0);
      size_t per = std::max<size_t>(1, errorTerms.size() / nThreads);
      for (size_t i = 0; i < nThreads; ++i) idx[i + 1] = idx[i] + per;
      idx.back() = errorTerms.size();
      // the reference cuts contiguous ranges blindly (LinearSystemSolver.cpp:50-78); here a cut moves forward to the next
      // boundary between share groups (see ErrorTerm::shareGroup)
      for (size_t i = 1; i < nThreads; ++i) {
        size_t c = std::max(idx[i], idx[i - 1]);
        while (c > 0 && c < errorTerms.size() && errorTerms[c]->shareGroup >= 0 && errorTerms[c]->shareGroup == errorTerms[c - 1]->shareGroup) ++c;
        idx[i] = c;
      }
      std::vector<std::thread> th;
      for (size_t i = 0; i < nThreads; ++i) th.emplace_back([&, i] { evaluateRange(idx[i], idx[i + 1], useM, &local[i]); });
      for (auto& t : th) t.join();
    }
    double J = 0;
    for (double x : local) J += x;
    return J;
  }
  void buildSystem(bool useM) {  // serial, :58-72
    H.clearValues();
    std::fill(rhs.begin(), rhs.end(), 0.0);
    for (auto* t : errorTerms) t->buildHessian(H, rhs, useM);
  }
  void setConstantConditioner(double d) { std::fill(diagCond.begin(), diagCond.end(), d); }
  void augment(bool undo) {
    int rb = 0;
    for (int i = 0; i < H.bRows(); ++i) {
      Mat& b = *H.block(i, i, true);
      for (int k = 0; k < b.r; ++k) b(k, k) += undo ? -diagCond[rb + k] : diagCond[rb + k] * diagCond[rb + k];
      rb += b.r;
    }
  }
  bool solveSystem(std::vector<double>& dx) {  // :74-106 incl. the lambda^2 / lambda un-augment mismatch
    if (useDiag) augment(false);
    std::vector<int> Cp, Ci;
    std::vector<double> Cx;
    H.fillCCS(Cp, Ci, Cx, true);
    int n = H.rows();
    std::vector<double> A((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c)
      for (int k = Cp[c]; k < Cp[c + 1]; ++k) { A[(size_t)c * n + Ci[k]] = Cx[k]; A[(size_t)Ci[k] * n + c] = Cx[k]; }
    dx = rhs;
    bool ok = denseCholeskySolve(n, A, dx);
    if (useDiag) augment(true);
    return ok;
  }
  double rhsJtJrhs() {
    std::vector<double> y;
    H.multiplySymUpper(rhs, y);
    double s = 0;
    for (size_t i = 0; i < y.size(); ++i) s += rhs[i] * y[i];
    return s;
  }
};

// ---------------------------------------------------------------- Optimizer2 + LM policy
// AB/src/Optimizer2.cpp:95-320, AB/src/LevenbergMarquardtTrustRegionPolicy.cpp:38-113, AB/src/TrustRegionPolicy.cpp:29-57
struct OptimizerOptions {
  double convergenceDeltaX = 1e-3, convergenceJDescentRatioThreshold = 1e-3;  // AB/.../Optimizer2Options.hpp defaults
  int maxIterations = 20;
  double lambdaInit = 1e-3;
  int nThreads = 1;
};
struct SolutionReturnValue {
  double JStart = 0, JFinal = 0, dXFinal = 0, dJFinal = 0;
  int iterations = 0, failedIterations = 0;
  bool linearSolverFailure = false;
};
struct Optimizer2 {
  OptimizerOptions opt;
  std::vector<DesignVariable*> problemDvs;   // in problem order
  std::vector<ErrorTerm*> problemTerms;
  std::vector<DesignVariable*> dvs;          // active
  BlockCholeskySolver solver;
  std::vector<double> dx;
  double J = 0, pJ = 0;
  // LM state
  double lambda = 0, gamma = 3, beta = 2, mu = 2; int p = 3;
  double trJ = 0, trPJ = 0, trLastOkJ = 0; bool first = true;
  std::vector<double> lmDx;

  void initialize() {
    dvs.clear();
    for (auto* d : problemDvs) { d->blockIndex = -1; if (d->active) dvs.push_back(d); }
    int cb = 0;
    for (size_t i = 0; i < dvs.size(); ++i) { dvs[i]->blockIndex = (int)i; dvs[i]->columnBase = cb; cb += dvs[i]->minimalDimensions(); }
    int dim = 0;
    for (auto* t : problemTerms) { t->rowBase = dim; dim += t->dim; }
    solver.initMatrixStructure(dvs, problemTerms, true);
  }
  double applyStateUpdate() {
    int s = 0;
    double mx = 0;
    for (auto* d : dvs) {
      int n = d->minimalDimensions();
      std::vector<double> seg(dx.begin() + s, dx.begin() + s + n);
      for (auto& x : seg) x *= d->scaling;
      if (n > 0) d->update(seg.data());
      s += n;
    }
    for (double x : dx) mx = std::fmax(mx, std::fabs(x));
    return mx;
  }
  void revertLastStateUpdate() { for (auto* d : dvs) d->revertUpdate(); }
  bool lmSolve(double Jc, bool prevFailed) {
    if (prevFailed) trJ = Jc;
    else { trPJ = trLastOkJ; trLastOkJ = Jc; trJ = Jc; }
    if (first) solver.buildSystem(true);
    else {
      double d1 = trPJ - trJ;
      double d2 = 0;
      for (size_t i = 0; i < lmDx.size(); ++i) d2 += lmDx[i] * (lambda * lmDx[i] + solver.rhs[i]);
      double rho = d1 / d2;
      if (prevFailed) { mu *= 2; lambda *= mu; }
      else if (rho <= 0) { mu *= 10; lambda *= mu; }
      else {
        solver.buildSystem(true);
        if (lambda > 1e-16) {
          double u1 = 1 / gamma, u2 = 1 - (beta - 1) * std::pow(2 * rho - 1, p);
          lambda *= (u1 > u2) ? u1 : u2;
          mu = beta;
        } else lambda = 1e-15;
      }
    }
    solver.setConstantConditioner(lambda);
    bool ok = solver.solveSystem(lmDx);
    dx = lmDx;
    first = false;
    return ok;
  }
  SolutionReturnValue optimize() {
    initialize();
    SolutionReturnValue srv;
    J = solver.evaluateError(opt.nThreads, true);
    pJ = J;
    srv.JStart = pJ;
    double deltaX = opt.convergenceDeltaX + 1.0, ratio = opt.convergenceJDescentRatioThreshold + 1.0, deltaJ = 0.0;
    bool prevFailed = false, lsFail = false;
    trJ = trPJ = trLastOkJ = J; first = true;
    lambda = opt.lambdaInit; gamma = 3; beta = 2; p = 3; mu = 2;
    while (srv.iterations < opt.maxIterations && srv.failedIterations < opt.maxIterations &&
           ((deltaX > opt.convergenceDeltaX && std::fabs(ratio) > opt.convergenceJDescentRatioThreshold) || lsFail)) {
      bool ok = lmSolve(J, prevFailed);
      if (!ok) { prevFailed = true; lsFail = true; srv.failedIterations++; }
      else {
        deltaX = applyStateUpdate();
        J = solver.evaluateError(opt.nThreads, true);
        deltaJ = pJ - J;
        ratio = deltaJ / pJ;
        if (ratio < 0.0) { revertLastStateUpdate(); srv.failedIterations++; prevFailed = true; }
        else { pJ = J; prevFailed = false; }
        srv.iterations++;
      }
    }
    srv.JFinal = pJ; srv.dXFinal = deltaX; srv.dJFinal = deltaJ; srv.linearSolverFailure = lsFail;
    return srv;
  }
};

}  // namespace orc
