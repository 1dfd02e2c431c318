// hip_linear_system_solver.hpp -- C++ classes with the surface of aslam::backend::LinearSystemSolver
// (aslam_optimizer/aslam_backend/include/aslam/backend/LinearSystemSolver.hpp:16-109) over the C ABI of
// include/multical_hip.h.  Header only, no Eigen / Boost: vectors are std::vector<double>.  INTEGRATION.md shows the
// ~40-line subclass of the real LinearSystemSolver that forwards to these classes inside an aslam_backend build tree.
//
//   HipBlockCholeskyLinearSystemSolver   <- BlockCholeskyLinearSystemSolver  (src/BlockCholeskyLinearSystemSolver.cpp)
//   HipSparseCholeskyLinearSystemSolver  <- SparseCholeskyLinearSystemSolver (src/SparseCholeskyLinearSystemSolver.cpp)
//
// Call order accepted (the one Optimizer2 uses, Optimizer2.cpp:142, 183-275; host/lm_driver.hpp is this repository's own caller): initMatrixStructure once,
// evaluateError before every buildSystem, solveSystem any number of times after one buildSystem (rejected LM steps
// do not rebuild, LevenbergMarquardtTrustRegionPolicy.cpp:61-72).
//
// The residuals, Jacobians and the normal equations come from the GPU; the factorisation stays on the host
// (host/block_cholesky.hpp in place of CHOLMOD, which this image does not have).
#pragma once
#include <chrono>
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/multical_hip.h"
#include "host/block_cholesky.hpp"

namespace multical_amd {

// what SM_DEFINE_EXCEPTION(Exception, std::runtime_error) gives the reference class (LinearSystemSolver.hpp:18)
struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };

// The reference's abstract class.  Differences, both forced by the GPU residency of the problem (SURVEY.md 8b "gap"):
// evaluateError is virtual, and the design variables / error terms are not passed to initMatrixStructure because the
// context already holds them (mc_add_*).
class LinearSystemSolver {
 public:
  explicit LinearSystemSolver(mc_context* ctx) : ctx_(ctx) {
    if (!ctx_) throw Exception("LinearSystemSolver: null context");
  }
  virtual ~LinearSystemSolver() {}

  // LinearSystemSolver::initMatrixStructure (LinearSystemSolver.cpp:117-138)
  void initMatrixStructure(bool useDiagonalConditioner) {
    int nb = 0, nt = 0, nnz = 0;
    // Optimizer2::initialize runs at the start of every optimize(); the structure of a context is frozen by the first one
    if (mc_dims(ctx_, &jRows_, &jCols_, &nb, &nt, &nnz) < 0) check(mc_finalize(ctx_));
    check(mc_dims(ctx_, &jRows_, &jCols_, &nb, &nt, &nnz));
    nBlocks_ = nb; nnz_ = nnz;
    e_.assign(jRows_, 0.0);
    rhs_.assign(jCols_, 0.0);
    diagonalConditioner_.assign(jCols_, 0.0);
    initMatrixStructureImplementation(useDiagonalConditioner);
  }
  // LinearSystemSolver::evaluateError (LinearSystemSolver.cpp:81-92); nThreads has no meaning on the GPU
  // The stacked error vector is j_rows doubles (72 MB on the full Multical problem) and only the J^T backend consumes it
  // (rhs = J^T e): it is copied from the device with the cost when the solver says it needs it, otherwise on the first e().
  virtual double evaluateError(std::size_t /*nThreads*/, bool useMEstimator) {
    double J = 0.0;
    const bool withE = needsErrorVector();
    check(mc_evaluate_error(ctx_, useMEstimator ? 1 : 0, &J, withE ? e_.data() : nullptr));
    raiseDeviceStatus();
    eFresh_ = withE;
    lastUseMEstimator_ = useMEstimator;
    return J;
  }
  virtual void buildSystem(std::size_t nThreads, bool useMEstimator) = 0;
  virtual bool solveSystem(std::vector<double>& outDx) = 0;
  virtual std::string name() const = 0;
  virtual double rhsJtJrhs() = 0;

  // LinearSystemSolver.cpp:105-114 (neither touches _useDiagonalConditioner, which only initMatrixStructure sets)
  virtual void setConditioner(const std::vector<double>& diag) {
    if ((int)diag.size() != jCols_)
      throw Exception("The diagonal conditioner must have the same number of rows as the Hessian matrix");
    diagonalConditioner_ = diag;
  }
  virtual void setConstantConditioner(double diag) { diagonalConditioner_.assign(jCols_, diag); }

  const std::vector<double>& rhs() const { return rhs_; }
  const std::vector<double>& e() {
    if (!eFresh_) {   // same state, same weights as the last evaluateError: the same vector
      double J = 0.0;
      check(mc_evaluate_error(ctx_, lastUseMEstimator_ ? 1 : 0, &J, e_.data()));
      raiseDeviceStatus();
      eFresh_ = true;
    }
    return e_;
  }
  std::size_t JRows() const { return (std::size_t)jRows_; }
  std::size_t JCols() const { return (std::size_t)jCols_; }

  // Optimizer2::applyStateUpdate / revertLastStateUpdate (Optimizer2.cpp:292-320) act on the device-resident state
  void applyStateUpdate(const std::vector<double>& dx) {
    if ((int)dx.size() != jCols_) throw Exception("dx has the wrong size");
    check(mc_apply_update(ctx_, dx.data()));
  }
  void revertLastStateUpdate() { check(mc_revert_update(ctx_)); }
  mc_context* context() const { return ctx_; }
  // wall-clock spent in buildSystem so far ("Optimizer2: build Hessian" of Optimizer2.cpp:185, compiled out there)
  double secondsInBuildSystem() const { return buildSeconds_; }

 protected:
  virtual void initMatrixStructureImplementation(bool useDiagonalConditioner) = 0;
  virtual bool needsErrorVector() const { return false; }
  struct BuildClock {
    explicit BuildClock(double& acc) : acc_(acc), t0_(std::chrono::steady_clock::now()) {}
    ~BuildClock() { acc_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
    double& acc_;
    std::chrono::steady_clock::time_point t0_;
  };
  void check(int rc) const { if (rc < 0) throw Exception(mc_last_error(ctx_)); }
  void raiseDeviceStatus() const { if (mc_device_status(ctx_) != 0) throw Exception(mc_last_error(ctx_)); }

  mc_context* ctx_;
  std::vector<double> e_, rhs_, diagonalConditioner_;
  bool useDiagonalConditioner_ = false, eFresh_ = false, lastUseMEstimator_ = true;
  int jRows_ = 0, jCols_ = 0, nBlocks_ = 0, nnz_ = 0;
  double buildSeconds_ = 0.0;
};

class HipBlockCholeskyLinearSystemSolver : public LinearSystemSolver {
 public:
  using LinearSystemSolver::LinearSystemSolver;
  std::string name() const override { return "hip_block_cholesky"; }

  // BlockCholeskyLinearSystemSolver::buildSystem (BlockCholeskyLinearSystemSolver.cpp:58-72) + the values of fillCCS
  void buildSystem(std::size_t /*nThreads*/, bool useMEstimator) override {
    BuildClock bc(buildSeconds_);
    check(mc_build_system(ctx_, useMEstimator ? 1 : 0, H_.data(), rhs_.data()));
    raiseDeviceStatus();
  }
  // BlockCholeskyLinearSystemSolver::solveSystem (:74-106).  The diagonal is augmented by the SQUARE of the
  // conditioner (:77-84) and afterwards reduced by the conditioner itself (:92-97), so lambda^2 - lambda stays on H until
  // the next buildSystem -- visible after rejected Levenberg-Marquardt steps, which do not rebuild.  Kept on purpose.
  bool solveSystem(std::vector<double>& outDx) override {
    if (useDiagonalConditioner_)
      for (int c = 0; c < jCols_; ++c) H_[Cp_[c + 1] - 1] += diagonalConditioner_[c] * diagonalConditioner_[c];
    outDx.assign(jCols_, 0.0);
    bool ok = chol_.factorize(H_.data());
    if (ok) {
      chol_.solve(rhs_.data(), outDx.data());
      for (double v : outDx) if (!std::isfinite(v)) { ok = false; break; }
    }
    if (useDiagonalConditioner_)
      for (int c = 0; c < jCols_; ++c) H_[Cp_[c + 1] - 1] -= diagonalConditioner_[c];
    return ok;
  }
  // BlockCholeskyLinearSystemSolver::rhsJtJrhs (:172-176): rhs^T H rhs, H symmetric with its upper triangle stored
  double rhsJtJrhs() override {
    double s = 0.0;
    for (int c = 0; c < jCols_; ++c)
      for (int k = Cp_[c]; k < Cp_[c + 1]; ++k) {
        int i = Ci_[k];
        s += (i == c ? 1.0 : 2.0) * rhs_[i] * H_[k] * rhs_[c];
      }
    return s;
  }
  // BlockCholeskyLinearSystemSolver::computeCovarianceBlocks (:122-148): blocks of (H + conditioner^2)^-1 for the given
  // (block row, block column) pairs, with the same augment / un-augment sequence around the factorisation as solveSystem.
  // Block k comes back row major in outP[k].  Throws like the reference's SM_ASSERT_TRUE when H is not positive definite.
  void computeCovarianceBlocks(const std::vector<std::pair<int, int> >& blockIndices, std::vector<std::vector<double> >& outP) {
    if (useDiagonalConditioner_)
      for (int c = 0; c < jCols_; ++c) H_[Cp_[c + 1] - 1] += diagonalConditioner_[c] * diagonalConditioner_[c];
    const bool ok = chol_.factorize(H_.data());
    if (useDiagonalConditioner_)
      for (int c = 0; c < jCols_; ++c) H_[Cp_[c + 1] - 1] -= diagonalConditioner_[c];
    if (!ok) throw Exception("Unable to retrieve covariance");
    std::vector<int> bi, bj;
    std::vector<long long> off;
    long long total = 0;
    for (const auto& p : blockIndices) {
      if (p.first < 0 || p.first >= nBlocks_ || p.second < 0 || p.second >= nBlocks_) throw Exception("block index out of range");
      bi.push_back(p.first); bj.push_back(p.second); off.push_back(total);
      total += (long long)(blockBase_[p.first + 1] - blockBase_[p.first]) * (blockBase_[p.second + 1] - blockBase_[p.second]);
    }
    std::vector<double> flat((std::size_t)total);
    chol_.covarianceBlocks((int)bi.size(), bi.data(), bj.data(), off.data(), flat.data());
    outP.resize(bi.size());
    for (std::size_t k = 0; k < bi.size(); ++k)
      outP[k].assign(flat.begin() + off[k], k + 1 < bi.size() ? flat.begin() + off[k + 1] : flat.end());
  }
  // scalar CCS of the upper triangle, bit-exact fillCCS(upper) ordering
  const std::vector<int>& Cp() const { return Cp_; }
  const std::vector<int>& Ci() const { return Ci_; }
  const std::vector<double>& Cx() const { return H_; }
  const BlockSparseCholesky& factorisation() const { return chol_; }

 protected:
  // BlockCholeskyLinearSystemSolver::initMatrixStructureImplementation (:34-55); the symbolic analysis CHOLMOD does at
  // the first solve (linear_solver_cholmod.h:80-88) happens here because the pattern is already complete
  void initMatrixStructureImplementation(bool useDiagonalConditioner) override {
    useDiagonalConditioner_ = useDiagonalConditioner;
    Cp_.assign(jCols_ + 1, 0);
    Ci_.assign(nnz_, 0);
    H_.assign(nnz_, 0.0);
    check(mc_get_pattern(ctx_, Cp_.data(), Ci_.data()));
    blockBase_.assign(nBlocks_ + 1, 0);
    check(mc_get_block_bases(ctx_, blockBase_.data()));
    chol_.analyze(jCols_, Cp_.data(), Ci_.data(), nBlocks_, blockBase_.data());
    setTimeAxis();
  }

 private:
  // time-axis dissection of the host factorisation (host/block_cholesky.hpp): one thread per independent part
  void setTimeAxis() {
    std::vector<double> when(nBlocks_);
    check(mc_get_block_times(ctx_, when.data()));
    chol_.setBlockCoordinates(when.data(), hostCholeskyThreads());
  }
  std::vector<int> Cp_, Ci_, blockBase_;
  std::vector<double> H_;
  BlockSparseCholesky chol_;
};

class HipSparseCholeskyLinearSystemSolver : public LinearSystemSolver {
 public:
  using LinearSystemSolver::LinearSystemSolver;
  std::string name() const override { return "hip_sparse_cholesky"; }

  // SparseCholeskyLinearSystemSolver::buildSystem (:39-46): J^T values, then rhs = J^T e (rightMultiply)
  void buildSystem(std::size_t /*nThreads*/, bool useMEstimator) override {
    BuildClock bc(buildSeconds_);
    check(mc_build_jacobian_transpose(ctx_, useMEstimator ? 1 : 0, values_.data()));
    raiseDeviceStatus();
    std::fill(rhs_.begin(), rhs_.end(), 0.0);
    for (int c = 0; c < jRows_; ++c) {
      const double ec = e_[c];
      for (long long k = colPtr_[c]; k < colPtr_[c + 1]; ++k) rhs_[rowInd_[k]] += values_[k] * ec;
    }
  }
  // SparseCholeskyLinearSystemSolver::solveSystem (:48-89): CHOLMOD factors A A^T for A = [J^T  diag(conditioner)]
  // (pushDiagonalBlock, CompressedColumnMatrix.hpp), i.e. J^T J + conditioner^2; formed here column by column of J^T.
  bool solveSystem(std::vector<double>& outDx) override {
    std::fill(H_.begin(), H_.end(), 0.0);
    for (int c = 0; c < jRows_; ++c) {
      const long long k0 = colPtr_[c], k1 = colPtr_[c + 1];
      for (long long b = k0; b < k1; ++b) {          // rows ascend inside a column: (rowInd[a], rowInd[b]) is upper
        const int col = rowInd_[b];
        const double vb = values_[b];
        const int* cb = Ci_.data() + Cp_[col];
        const int* ce = Ci_.data() + Cp_[col + 1];
        const int* it = cb;
        for (long long a = k0; a <= b; ++a) {
          it = std::lower_bound(it, ce, rowInd_[a]);
          H_[Cp_[col] + (it - cb)] += values_[a] * vb;
        }
      }
    }
    if (useDiagonalConditioner_)
      for (int c = 0; c < jCols_; ++c) H_[Cp_[c + 1] - 1] += diagonalConditioner_[c] * diagonalConditioner_[c];
    outDx.assign(jCols_, 0.0);
    if (!chol_.factorize(H_.data())) return false;
    chol_.solve(rhs_.data(), outDx.data());
    for (double v : outDx) if (!std::isfinite(v)) return false;
    return true;
  }
  // SparseCholeskyLinearSystemSolver::rhsJtJrhs (:106-111): |J rhs|^2 (leftMultiply)
  double rhsJtJrhs() override {
    double s = 0.0;
    for (int c = 0; c < jRows_; ++c) {
      double d = 0.0;
      for (long long k = colPtr_[c]; k < colPtr_[c + 1]; ++k) d += values_[k] * rhs_[rowInd_[k]];
      s += d * d;
    }
    return s;
  }
  const std::vector<int>& colPtr() const { return colPtr_; }
  const std::vector<int>& rowInd() const { return rowInd_; }
  const std::vector<double>& values() const { return values_; }

 protected:
  bool needsErrorVector() const override { return true; }
  // SparseCholeskyLinearSystemSolver::initMatrixStructureImplementation (:14-36): symbolic J^T
  // (CompressedColumnJacobianTransposeBuilder::initMatrixStructure).  Throws like the reference when the problem holds
  // a BSplineMotionError, whose evaluateJacobians is "currently unsupported" (BSplineMotionError.hpp:93).
  void initMatrixStructureImplementation(bool useDiagonalConditioner) override {
    useDiagonalConditioner_ = useDiagonalConditioner;
    long long nnz = 0;
    int rc = mc_jt_nnz(ctx_, &nnz);
    if (rc == -2) throw Exception("This is currently unsupported");
    check(rc);
    if (nnz > 0x7fffffffLL) throw Exception("J^T has more entries than a 32-bit compressed-column matrix can index");
    colPtr_.assign(jRows_ + 1, 0);
    rowInd_.assign((std::size_t)nnz, 0);
    values_.assign((std::size_t)nnz, 0.0);
    check(mc_get_jt_pattern(ctx_, colPtr_.data(), rowInd_.data()));
    // J^T J has the pattern of the block Hessian (union of the terms' footprints)
    Cp_.assign(jCols_ + 1, 0);
    Ci_.assign(nnz_, 0);
    H_.assign(nnz_, 0.0);
    check(mc_get_pattern(ctx_, Cp_.data(), Ci_.data()));
    std::vector<int> base(nBlocks_ + 1, 0);
    check(mc_get_block_bases(ctx_, base.data()));
    chol_.analyze(jCols_, Cp_.data(), Ci_.data(), nBlocks_, base.data());
    {
      std::vector<double> when(nBlocks_);
      check(mc_get_block_times(ctx_, when.data()));
      chol_.setBlockCoordinates(when.data(), hostCholeskyThreads());
    }
  }

 private:
  std::vector<int> colPtr_, rowInd_, Cp_, Ci_;
  std::vector<double> values_, H_;
  BlockSparseCholesky chol_;
};

}  // namespace multical_amd
