// lm_core.hpp -- the minimiser's policy, free of any device code: settings, report, the interface it drives
// (NormalEquations) and the loop itself.  host/lm_driver.hpp puts the GPU behind that interface; the tests' host twin
// (tests/hostprobe) puts the g++ build of the same term math behind it, so the N-rank iteration runs on CPU ranks too.
//
// Only the ARITHMETIC of the damping schedule (Nielsen's rule with gamma = 3, beta = 2, p = 3, mu0 = 2), of the gain ratio and
// of the stopping rule is shared with the reference (aslam_optimizer/aslam_backend/src/Optimizer2.cpp:183-275,
// LevenbergMarquardtTrustRegionPolicy.cpp:50-113) -- the iterates must match it to 1e-7.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

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
  virtual void applyUpdate(const std::vector<double>& dx) = 0;       // Optimizer2::applyStateUpdate
  virtual void revertUpdate() = 0;                                   // Optimizer2::revertLastStateUpdate
  virtual void factorStats(double& nnz, double& flops) { nnz = flops = 0; }
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

inline MinimiserReport minimise(NormalEquations* eq, const MinimiserSettings& cfg) {
  MinimiserReport rep;
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
    { detail::Stopwatch w(rep.secondsUpdate); eq->applyUpdate(dx); }
    step = 0.0;
    for (double v : dx) step = std::fmax(step, std::fabs(v));
    // ---- trial
    { detail::Stopwatch w(rep.secondsTrials); cost = eq->trial(); }
    change = accepted - cost;
    ratio = change / accepted;
    const bool worse = ratio < 0.0;
    if (worse && cfg.damped) {
      detail::Stopwatch w(rep.secondsUpdate);
      eq->revertUpdate();
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
