// lm_driver.hpp -- this repository's own minimiser around the hot path.
//
// The reference drives evaluateError / buildSystem / solveSystem from Optimizer2::optimize through a TrustRegionPolicy
// object (aslam_optimizer/aslam_backend/src/Optimizer2.cpp:183-275, LevenbergMarquardtTrustRegionPolicy.cpp:50-113).  That
// shape costs a GPU backend one extra pass per accepted step: the cost at the trial state and, one call later, the normal
// equations at that same state.  This driver is organised around what the GPU path offers instead:
//
//   * a TRIAL is one fused pass (mc_evaluate_error_and_build_system_device): the cost of the trial state and, into a spare
//     device buffer, the normal equations linearised there.  Nothing is copied to the host yet.
//   * the decision about a trial falls in two parts, like the reference's: accept / reject from the cost alone, and -- one
//     iteration later, from the gain ratio -- whether the system linearised at the accepted state replaces the current one
//     (ADOPT: swap of the two device buffers + one device-to-host copy) or the old system is kept with a larger damping.
//   * the factorisation works on a pinned host image of the adopted system, whose diagonal keeps the reference's
//     damping residue (augmented by lambda^2, relieved by lambda: BlockCholeskyLinearSystemSolver.cpp:76-98) until the next
//     adoption overwrites it.
//
// Only the ARITHMETIC of the damping schedule (Nielsen's rule with gamma = 3, beta = 2, p = 3, mu0 = 2), of the gain ratio and
// of the stopping rule is shared with the reference -- the iterates must match it to 1e-7.  Policies: Levenberg-Marquardt
// and undamped Gauss-Newton (every trial accepted and adopted).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../hip_linear_system_solver.hpp"

namespace multical_amd {

struct MinimiserSettings {
  double stepTolerance = 1e-3;         // stop when max |dx| falls to this           (Optimizer2Options::convergenceDeltaX)
  double costRatioTolerance = 1e-5;    // ... or |cost change| / cost falls to this  (convergenceJDescentRatioThreshold)
  int maxSteps = 20;                   // bound on evaluated trials and on failed ones alike (maxIterations)
  bool damped = true;                  // Levenberg-Marquardt; false: Gauss-Newton
  double initialDamping = 1e-3;
  bool transposeBackend = false;       // J^T in compressed columns + host J^T J instead of the fused block build
  bool verbose = false;
};
struct MinimiserReport {
  double costStart = 0, costFinal = 0, lastStep = 0, lastCostChange = 0;
  int trials = 0, failures = 0;
  bool solverFailed = false;
  double secondsSetup = 0, secondsTrials = 0, secondsAdopt = 0, secondsSolve = 0, secondsUpdate = 0;
  double factorNnz = 0, factorFlops = 0;
};

namespace detail {
struct Stopwatch {
  explicit Stopwatch(double& acc) : acc_(acc), t0_(std::chrono::steady_clock::now()) {}
  ~Stopwatch() { acc_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
  double& acc_;
  std::chrono::steady_clock::time_point t0_;
};
inline void hipOk(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Exception(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace detail

// Where cost, normal equations and the step come from.  A trial is evaluated at the state the context holds.
class NormalEquations {
 public:
  virtual ~NormalEquations() {}
  virtual void prepare(bool damped) = 0;                              // sizes, pattern, symbolic factorisation
  virtual double trial() = 0;                                         // cost at the current state (may linearise on the side)
  virtual void adopt() = 0;                                           // the last trial's linearisation becomes the system
  virtual bool step(double damping, std::vector<double>& dx) = 0;     // (H + damping^2 I) dx = rhs on the adopted system
  virtual const std::vector<double>& gradient() const = 0;            // rhs = -J^T e of the adopted system
  virtual void factorStats(double& nnz, double& flops) { nnz = flops = 0; }
};

// The fused GPU path: two device systems (adopted / trial), one pinned host image for the factorisation.
class FusedBlockEquations : public NormalEquations {
 public:
  explicit FusedBlockEquations(mc_context* ctx) : ctx_(ctx) {}
  ~FusedBlockEquations() override {
    for (int k = 0; k < 2; ++k) { if (dH_[k]) (void)hipFree(dH_[k]); if (dG_[k]) (void)hipFree(dG_[k]); }
    if (dCost_) (void)hipFree(dCost_);
    if (hH_) (void)hipHostFree(hH_);
  }
  void prepare(bool damped) override {
    damped_ = damped;
    int rows = 0, blocks = 0, terms = 0;
    if (mc_dims(ctx_, &rows, &n_, &blocks, &terms, &nnz_) < 0) ok(mc_finalize(ctx_));
    ok(mc_dims(ctx_, &rows, &n_, &blocks, &terms, &nnz_));
    colEnd_.assign(n_ + 1, 0);
    std::vector<int> rowIdx(nnz_), base(blocks + 1);
    ok(mc_get_pattern(ctx_, colEnd_.data(), rowIdx.data()));
    ok(mc_get_block_bases(ctx_, base.data()));
    chol_.analyze(n_, colEnd_.data(), rowIdx.data(), blocks, base.data());
    {
      std::vector<double> when(blocks);
      ok(mc_get_block_times(ctx_, when.data()));
      chol_.setBlockCoordinates(when.data(), hostCholeskyThreads());   // time-axis dissection: one thread per part
    }
    for (int k = 0; k < 2; ++k) {
      detail::hipOk(hipMalloc((void**)&dH_[k], sizeof(double) * (size_t)nnz_), "hipMalloc(system values)");
      detail::hipOk(hipMalloc((void**)&dG_[k], sizeof(double) * (size_t)n_), "hipMalloc(gradient)");
      // neither system is ever read before a trial has written it (trialPending_), but a zero system fails loudly where
      // uninitialised memory would not
      detail::hipOk(hipMemset(dH_[k], 0, sizeof(double) * (size_t)nnz_), "hipMemset(system values)");
      detail::hipOk(hipMemset(dG_[k], 0, sizeof(double) * (size_t)n_), "hipMemset(gradient)");
    }
    detail::hipOk(hipMalloc((void**)&dCost_, sizeof(double)), "hipMalloc(cost)");
    detail::hipOk(hipHostMalloc((void**)&hH_, sizeof(double) * (size_t)nnz_, hipHostMallocDefault), "hipHostMalloc(system image)");
    g_.assign(n_, 0.0);
  }
  double trial() override {
    const int t = 1 - cur_;
    ok(mc_evaluate_error_and_build_system_device(ctx_, 1, dCost_, nullptr, dH_[t], dG_[t], nullptr));
    double cost = 0.0;
    detail::hipOk(hipMemcpy(&cost, dCost_, sizeof(double), hipMemcpyDeviceToHost), "cost read-back");
    if (mc_device_status(ctx_) != 0) throw Exception(mc_last_error(ctx_));   // the reference throws out of evaluateError
    trialPending_ = true;
    return cost;
  }
  void adopt() override {
    // Only a trial's linearisation can be adopted.  Without a trial since the last adoption (Gauss-Newton after a failed
    // factorisation: the reference's GaussNewtonTrustRegionPolicy rebuilds at the same state, fails the same way and stops
    // at maxIterations with linearSolverFailure) the adopted system stays: flipping cur_ again would hand the step an
    // older, or never written, device system.
    if (!trialPending_) return;
    trialPending_ = false;
    cur_ = 1 - cur_;
    detail::hipOk(hipMemcpy(hH_, dH_[cur_], sizeof(double) * (size_t)nnz_, hipMemcpyDeviceToHost), "system read-back");
    detail::hipOk(hipMemcpy(g_.data(), dG_[cur_], sizeof(double) * (size_t)n_, hipMemcpyDeviceToHost), "gradient read-back");
  }
  bool step(double damping, std::vector<double>& dx) override {
    // the diagonal entry of column c is the last one of the column (upper triangle in compressed columns)
    if (damped_) for (int c = 0; c < n_; ++c) hH_[colEnd_[c + 1] - 1] += damping * damping;
    dx.assign(n_, 0.0);
    bool good = chol_.factorize(hH_);
    if (good) {
      chol_.solve(g_.data(), dx.data());
      for (double v : dx) if (!std::isfinite(v)) { good = false; break; }
    }
    // the reference relieves the diagonal by the damping, not by its square; the difference stays until the next adoption
    if (damped_) for (int c = 0; c < n_; ++c) hH_[colEnd_[c + 1] - 1] -= damping;
    return good;
  }
  const std::vector<double>& gradient() const override { return g_; }
  void factorStats(double& nnz, double& flops) override { nnz = (double)chol_.stats().nnz_factor; flops = chol_.stats().flops; }

 private:
  void ok(int rc) const { if (rc < 0) throw Exception(mc_last_error(ctx_)); }
  mc_context* ctx_;
  bool damped_ = true, trialPending_ = false;
  int n_ = 0, nnz_ = 0, cur_ = 0;
  double* dH_[2] = {nullptr, nullptr};
  double* dG_[2] = {nullptr, nullptr};
  double* dCost_ = nullptr;
  double* hH_ = nullptr;
  std::vector<int> colEnd_;
  std::vector<double> g_;
  BlockSparseCholesky chol_;
};

// The second backend (J^T in compressed columns): nothing to speculate with -- the trial is the residual pass and the
// linearisation is formed when it is adopted, at the state the trial was evaluated at.
class TransposeEquations : public NormalEquations {
 public:
  explicit TransposeEquations(mc_context* ctx) : s_(ctx) {}
  void prepare(bool damped) override { s_.initMatrixStructure(damped); }
  double trial() override { return s_.evaluateError(1, true); }
  void adopt() override { s_.buildSystem(1, true); }
  bool step(double damping, std::vector<double>& dx) override { s_.setConstantConditioner(damping); return s_.solveSystem(dx); }
  const std::vector<double>& gradient() const override { return s_.rhs(); }

 private:
  HipSparseCholeskyLinearSystemSolver s_;
};

// Nielsen's damping rule as the reference parametrises it (LevenbergMarquardtTrustRegionPolicy.cpp:7-13, 61-84)
struct DampingSchedule {
  double damping, growth = 2.0;
  static constexpr double kShrinkFloor = 1.0 / 3.0, kBeta = 2.0;
  static constexpr int kPower = 3;
  explicit DampingSchedule(double d0) : damping(d0) {}
  void trialRejected() { growth *= 2.0; damping *= growth; }
  void noGain() { growth *= 10.0; damping *= growth; }
  void gained(double ratio) {
    if (damping > 1e-16) {
      const double byGain = 1.0 - (kBeta - 1.0) * std::pow(2.0 * ratio - 1.0, kPower);
      damping *= kShrinkFloor > byGain ? kShrinkFloor : byGain;
      growth = kBeta;
    } else {
      damping = 1e-15;
    }
  }
};

inline MinimiserReport minimise(mc_context* ctx, const MinimiserSettings& cfg) {
  if (!ctx) throw Exception("minimise: null context");
  MinimiserReport rep;
  std::unique_ptr<NormalEquations> eq;
  if (cfg.transposeBackend) eq.reset(new TransposeEquations(ctx));
  else eq.reset(new FusedBlockEquations(ctx));
  { detail::Stopwatch w(rep.secondsSetup); eq->prepare(cfg.damped); }

  double cost;
  { detail::Stopwatch w(rep.secondsTrials); cost = eq->trial(); }
  rep.costStart = cost;
  double accepted = cost;              // cost of the state the context would fall back to
  double gainFrom = cost, gainTo = cost;   // the two most recent accepted costs, as seen at the start of an undisturbed iteration
  double step = cfg.stepTolerance + 1.0, ratio = cfg.costRatioTolerance + 1.0, change = 0.0;
  bool disturbed = false;              // the previous iteration ended in a rejected trial or a failed factorisation
  bool haveSystem = false;
  DampingSchedule lm(cfg.initialDamping);
  std::vector<double> dx;

  auto unfinished = [&]() {
    const bool moving = step > cfg.stepTolerance && std::fabs(ratio) > cfg.costRatioTolerance;
    return rep.trials < cfg.maxSteps && rep.failures < cfg.maxSteps && (moving || rep.solverFailed);
  };
  while (unfinished()) {
    // ---- which system, which damping
    if (!disturbed) { gainFrom = gainTo; gainTo = cost; }
    if (!haveSystem || !cfg.damped) {
      detail::Stopwatch w(rep.secondsAdopt);
      eq->adopt();
      haveSystem = true;
    } else if (disturbed) {
      lm.trialRejected();
    } else {
      // gain ratio of the step just accepted: actual decrease over the decrease the damped model predicted
      const std::vector<double>& g = eq->gradient();
      double predicted = 0.0;
      for (std::size_t i = 0; i < dx.size(); ++i) predicted += dx[i] * (lm.damping * dx[i] + g[i]);
      const double gain = (gainFrom - gainTo) / predicted;
      if (gain <= 0) {
        lm.noGain();                   // accepted, but the old linearisation is kept
      } else {
        { detail::Stopwatch w(rep.secondsAdopt); eq->adopt(); }
        lm.gained(gain);
      }
    }
    // ---- step
    bool solved;
    { detail::Stopwatch w(rep.secondsSolve); solved = eq->step(cfg.damped ? lm.damping : 0.0, dx); }
    if (!solved) {
      if (cfg.verbose) std::printf("[multical_amd] factorisation failed (damping %g)\n", lm.damping);
      disturbed = true;
      rep.solverFailed = true;
      ++rep.failures;
      continue;
    }
    { detail::Stopwatch w(rep.secondsUpdate); if (mc_apply_update(ctx, dx.data()) < 0) throw Exception(mc_last_error(ctx)); }
    step = 0.0;
    for (double v : dx) step = std::fmax(step, std::fabs(v));
    // ---- trial
    { detail::Stopwatch w(rep.secondsTrials); cost = eq->trial(); }
    change = accepted - cost;
    ratio = change / accepted;
    const bool worse = ratio < 0.0;
    if (worse && cfg.damped) {
      detail::Stopwatch w(rep.secondsUpdate);
      if (mc_revert_update(ctx) < 0) throw Exception(mc_last_error(ctx));
      ++rep.failures;
      disturbed = true;
    } else {
      accepted = cost;
      disturbed = false;
    }
    ++rep.trials;
    if (cfg.verbose)
      std::printf("[multical_amd] trial %d: cost %.10g (%+.3g), max|dx| %.3g, damping %.3g%s\n", rep.trials, cost, -change, step,
                  cfg.damped ? lm.damping : 0.0, (worse && cfg.damped) ? "  -> rejected" : "");
  }
  rep.costFinal = accepted;
  rep.lastStep = step;
  rep.lastCostChange = change;
  eq->factorStats(rep.factorNnz, rep.factorFlops);
  return rep;
}

}  // namespace multical_amd
