// hip_linear_system_solver.hpp -- C++ adapter with the surface of aslam::backend::LinearSystemSolver
// (aslam_optimizer/aslam_backend/include/aslam/backend/LinearSystemSolver.hpp:16-109) over the C ABI of
// include/multical_hip.h.  Header only, no Eigen / Boost: vectors are std::vector<double>, the factorisation is a
// callback (the reference keeps it in CHOLMOD on the host: BlockCholeskyLinearSystemSolver.cpp:74-99).  INTEGRATION.md
// shows the ~40-line subclass of the real LinearSystemSolver that forwards to this class inside an aslam_backend
// build tree.
//
// Call order accepted (the one Optimizer2 uses, Optimizer2.cpp:142, 183-275): initMatrixStructure once,
// evaluateError before every buildSystem, solveSystem any number of times after one buildSystem (rejected LM steps
// do not rebuild, LevenbergMarquardtTrustRegionPolicy.cpp:20-40).
#pragma once
#include <cmath>
#include <cstddef>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/multical_hip.h"

namespace multical_amd {

// what SM_DEFINE_EXCEPTION(Exception, std::runtime_error) gives the reference class (LinearSystemSolver.hpp:18)
struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };

// Upper-triangular CCS system handed to the host factorisation; Cp/Ci are bit-exact fillCCS(upper) ordering.
struct CcsSystem {
  int n = 0;
  std::vector<int> Cp, Ci;
  std::vector<double> Cx;    // already augmented by the conditioner
  std::vector<double> rhs;
};
// returns false on numerical failure (not positive definite), like LinearSolverCholmod::solve
using HostFactorSolve = std::function<bool(const CcsSystem&, std::vector<double>& dx)>;

class HipLinearSystemSolver {
 public:
  // `ctx` already holds the problem description (mc_add_*): the lowering pass that walks an OptimizationProblem and
  // emits typed batches lives with the caller.  The context is borrowed, not owned.
  HipLinearSystemSolver(mc_context* ctx, HostFactorSolve solve) : ctx_(ctx), solve_(std::move(solve)) {
    if (!ctx_) throw Exception("HipLinearSystemSolver: null context");
  }
  std::string name() const { return "hip_block_cholesky"; }

  // LinearSystemSolver::initMatrixStructure (LinearSystemSolver.cpp:117-138)
  void initMatrixStructure(bool useDiagonalConditioner) {
    check(mc_finalize(ctx_));
    int nb = 0, nt = 0, nnz = 0;
    check(mc_dims(ctx_, &jRows_, &jCols_, &nb, &nt, &nnz));
    sys_.n = jCols_;
    sys_.Cp.assign(jCols_ + 1, 0);
    sys_.Ci.assign(nnz, 0);
    sys_.Cx.assign(nnz, 0.0);
    sys_.rhs.assign(jCols_, 0.0);
    H_.assign(nnz, 0.0);
    e_.assign(jRows_, 0.0);
    check(mc_get_pattern(ctx_, sys_.Cp.data(), sys_.Ci.data()));
    useDiagonalConditioner_ = useDiagonalConditioner;
    diagonalConditioner_.assign(useDiagonalConditioner ? jCols_ : 0, 1.0);
  }
  // LinearSystemSolver::evaluateError (LinearSystemSolver.cpp:81-92)
  double evaluateError(std::size_t /*nThreads*/, bool useMEstimator) {
    double J = 0.0;
    check(mc_evaluate_error(ctx_, useMEstimator ? 1 : 0, &J, e_.data()));
    raise_device_status();
    return J;
  }
  // BlockCholeskyLinearSystemSolver::buildSystem (BlockCholeskyLinearSystemSolver.cpp:58-72)
  void buildSystem(std::size_t /*nThreads*/, bool useMEstimator) {
    check(mc_build_system(ctx_, useMEstimator ? 1 : 0, H_.data(), sys_.rhs.data()));
    raise_device_status();
  }
  // BlockCholeskyLinearSystemSolver::solveSystem (BlockCholeskyLinearSystemSolver.cpp:74-99): augment the diagonal
  // by the *square* of the conditioner, factor/solve on the host, report failure as `false`.
  bool solveSystem(std::vector<double>& outDx) {
    sys_.Cx = H_;
    if (useDiagonalConditioner_)
      for (int c = 0; c < jCols_; ++c) {
        double d = diagonalConditioner_[c];
        sys_.Cx[sys_.Cp[c + 1] - 1] += d * d;  // diagonal entry is the last of its upper-triangular column
      }
    outDx.assign(jCols_, 0.0);
    bool ok = solve_(sys_, outDx);
    if (!ok) return false;
    for (double v : outDx) if (!std::isfinite(v)) return false;
    return true;
  }
  // LinearSystemSolver.hpp:53-60
  void setConstantConditioner(double diag) { diagonalConditioner_.assign(jCols_, diag); useDiagonalConditioner_ = true; }
  void setConditioner(const std::vector<double>& diag) {
    if ((int)diag.size() != jCols_) throw Exception("conditioner has the wrong size");
    diagonalConditioner_ = diag; useDiagonalConditioner_ = true;
  }
  // BlockCholeskyLinearSystemSolver::rhsJtJrhs (BlockCholeskyLinearSystemSolver.cpp:105-119): rhs^T H rhs with the
  // un-augmented H (symmetric, upper triangle stored)
  double rhsJtJrhs() const {
    const std::vector<double>& r = sys_.rhs;
    double s = 0.0;
    for (int c = 0; c < jCols_; ++c)
      for (int k = sys_.Cp[c]; k < sys_.Cp[c + 1]; ++k) {
        int i = sys_.Ci[k];
        s += (i == c ? 1.0 : 2.0) * r[i] * H_[k] * r[c];
      }
    return s;
  }
  const std::vector<double>& rhs() const { return sys_.rhs; }
  const std::vector<double>& e() const { return e_; }
  const CcsSystem& Hessian() const { return sys_; }
  std::size_t JRows() const { return (std::size_t)jRows_; }
  std::size_t JCols() const { return (std::size_t)jCols_; }

  // Optimizer2::applyStateUpdate / revertLastStateUpdate act on the device-resident state
  void applyStateUpdate(const std::vector<double>& dx) {
    if ((int)dx.size() != jCols_) throw Exception("dx has the wrong size");
    check(mc_apply_update(ctx_, dx.data()));
  }
  void revertLastStateUpdate() { check(mc_revert_update(ctx_)); }

 private:
  void check(int rc) const { if (rc < 0) throw Exception(mc_last_error(ctx_)); }
  void raise_device_status() const { if (mc_device_status(ctx_) != 0) throw Exception(mc_last_error(ctx_)); }
  mc_context* ctx_;
  HostFactorSolve solve_;
  CcsSystem sys_;
  std::vector<double> H_, e_, diagonalConditioner_;
  bool useDiagonalConditioner_ = false;
  int jRows_ = 0, jCols_ = 0;
};

}  // namespace multical_amd
