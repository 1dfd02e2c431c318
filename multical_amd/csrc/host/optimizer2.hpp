// optimizer2.hpp -- the caller of the hot path, restated in C++ so that the backend can be driven and checked end to
// end without the reference's build tree: aslam::backend::Optimizer2 (aslam_optimizer/aslam_backend/src/Optimizer2.cpp:
// 95-151, 183-334), Optimizer2Options (include/aslam/backend/Optimizer2Options.hpp:9-42), TrustRegionPolicy
// (src/TrustRegionPolicy.cpp:29-57) and its three policies (src/LevenbergMarquardtTrustRegionPolicy.cpp,
// src/GaussNewtonTrustRegionPolicy.cpp, src/DogLegTrustRegionPolicy.cpp).  Same names, same control flow, same
// defaults; the problem itself lives in the mc_context the solver was created on.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include "../hip_linear_system_solver.hpp"

namespace multical_amd {

class TrustRegionPolicy {
 public:
  virtual ~TrustRegionPolicy() {}
  void setSolver(std::shared_ptr<LinearSystemSolver> solver) { _solver = std::move(solver); }
  std::shared_ptr<LinearSystemSolver> getSolver() { return _solver; }
  // TrustRegionPolicy.cpp:29-36
  void optimizationStarting(double J) {
    _J = J; _p_J = J; _last_successful_J = J;
    _isFirstIteration = true;
    optimizationStartingImplementation(J);
  }
  // TrustRegionPolicy.cpp:39-52
  bool solveSystem(double J, bool previousIterationFailed, int nThreads, std::vector<double>& outDx) {
    if (previousIterationFailed) {
      _J = J;
    } else {
      _p_J = _last_successful_J;
      _last_successful_J = J;
      _J = J;
    }
    bool success = solveSystemImplementation(J, previousIterationFailed, nThreads, outDx);
    _isFirstIteration = false;
    return success;
  }
  virtual std::string name() const = 0;
  virtual bool revertOnFailure() { return true; }
  virtual bool requiresAugmentedDiagonal() const = 0;
  virtual std::string printState() const = 0;
  double get_dJ() const { return _p_J - _J; }
  bool isFirstIteration() const { return _isFirstIteration; }

 protected:
  virtual void optimizationStartingImplementation(double J) = 0;
  virtual bool solveSystemImplementation(double J, bool previousIterationFailed, int nThreads, std::vector<double>& outDx) = 0;
  std::shared_ptr<LinearSystemSolver> _solver;
  double _J = 0, _p_J = 0, _last_successful_J = 0;
  bool _isFirstIteration = true;
};

static inline double dotv(const std::vector<double>& a, const std::vector<double>& b) {
  double s = 0.0;
  for (std::size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

// LevenbergMarquardtTrustRegionPolicy.cpp:7-117 (Nielsen's damping schedule)
class LevenbergMarquardtTrustRegionPolicy : public TrustRegionPolicy {
 public:
  explicit LevenbergMarquardtTrustRegionPolicy(double lambdaInit = 1e-3)
      : _lambdaInit(lambdaInit), _gammaInit(3), _betaInit(2), _pInit(3), _muInit(2) {}
  std::string name() const override { return "levenberg_marquardt"; }
  bool revertOnFailure() override { return true; }
  bool requiresAugmentedDiagonal() const override { return true; }
  std::string printState() const override {
    char buf[96];
    std::snprintf(buf, sizeof buf, "LM - lambda:%g mu:%g", _lambda, _mu);
    return buf;
  }
  double lambda() const { return _lambda; }

 protected:
  void optimizationStartingImplementation(double) override {
    _lambda = _lambdaInit; _gamma = _gammaInit; _beta = _betaInit; _p = _pInit; _mu = _muInit;
  }
  bool solveSystemImplementation(double, bool previousIterationFailed, int nThreads, std::vector<double>& outDx) override {
    if (!_solver) throw Exception("The solver is null");
    if (isFirstIteration()) {
      _solver->buildSystem(nThreads, true);
    } else {
      double rho = getLmRho();
      if (previousIterationFailed) {          // the last step was a regression
        _mu *= 2;
        _lambda *= _mu;
      } else if (rho <= 0) {                  // no rebuild, only a stronger conditioner
        _mu *= 10;
        _lambda *= _mu;
      } else {                                // the last iteration was successful: rebuild
        _solver->buildSystem(nThreads, true);
        if (_lambda > 1e-16) {
          double u1 = 1 / _gamma;
          double u2 = 1 - (_beta - 1) * std::pow(2 * rho - 1, _p);
          _lambda *= (u1 > u2) ? u1 : u2;
          _mu = _beta;
        } else {
          _lambda = 1e-15;
        }
      }
    }
    _solver->setConstantConditioner(_lambda);
    bool success = _solver->solveSystem(_dx);
    outDx = _dx;
    return success;
  }
  double getLmRho() {
    double d1 = get_dJ();
    const std::vector<double>& rhs = _solver->rhs();
    double d2 = 0.0;                          // L(0) - L(h) = dx^T (lambda dx + rhs)
    for (std::size_t i = 0; i < _dx.size(); ++i) d2 += _dx[i] * (_lambda * _dx[i] + rhs[i]);
    return d1 / d2;
  }
  double _lambdaInit, _gammaInit, _betaInit;
  int _pInit;
  double _muInit;
  double _lambda = 0, _gamma = 0, _beta = 0, _mu = 0;
  int _p = 0;
  std::vector<double> _dx;
};

// GaussNewtonTrustRegionPolicy.cpp:12-39
class GaussNewtonTrustRegionPolicy : public TrustRegionPolicy {
 public:
  std::string name() const override { return "gauss_newton"; }
  bool revertOnFailure() override { return false; }
  bool requiresAugmentedDiagonal() const override { return false; }
  std::string printState() const override { return "GN"; }

 protected:
  void optimizationStartingImplementation(double) override {}
  bool solveSystemImplementation(double, bool, int nThreads, std::vector<double>& outDx) override {
    _solver->buildSystem(nThreads, true);
    return _solver->solveSystem(outDx);
  }
};

// DogLegTrustRegionPolicy.cpp:11-161
class DogLegTrustRegionPolicy : public TrustRegionPolicy {
 public:
  std::string name() const override { return "dog_leg"; }
  bool revertOnFailure() override { return true; }
  bool requiresAugmentedDiagonal() const override { return false; }
  std::string printState() const override {
    char buf[128];
    if (_stepType == "DL") std::snprintf(buf, sizeof buf, "DL - delta:%g, DL, beta:%g", _delta, _beta);
    else std::snprintf(buf, sizeof buf, "DL - delta:%g, %s", _delta, _stepType.c_str());
    return buf;
  }

 protected:
  void optimizationStartingImplementation(double) override {
    _dx_sd_norm = 0; _dx_gn_norm = 0; _L0 = 0; _sd_scale = 0; _beta = 0; _delta = 0; _p_delta = 0;
  }
  static double norm(const std::vector<double>& v) { return std::sqrt(dotv(v, v)); }
  bool solveSystemImplementation(double J, bool previousIterationFailed, int nThreads, std::vector<double>& outDx) override {
    if (!_solver) throw Exception("The solver is null");
    double rho = get_dJ() / _L0;
    if (!isFirstIteration()) {
      _p_delta = _delta;
      if (rho > 0.75) {                       // step succeeded
        double dx_norm3 = 3 * norm(_dx);
        if (!(_delta > dx_norm3)) _delta = dx_norm3;
      } else if (rho > 0 && rho < 0.25) {     // step almost failed
        _delta /= 2.0;
      } else if (rho <= 0) {                  // step failed
        if (_stepType == "GN") _delta = _dx_gn_norm / 2.0;
        else _delta /= 2.0;
      }
    }
    bool gnComputed = true;
    if (!previousIterationFailed) {
      _solver->buildSystem(nThreads, true);
      const std::vector<double>& rhs = _solver->rhs();
      _sd_scale = dotv(rhs, rhs) / _solver->rhsJtJrhs();
      _dx_sd.assign(rhs.size(), 0.0);
      for (std::size_t i = 0; i < rhs.size(); ++i) _dx_sd[i] = _sd_scale * rhs[i];
      _dx_sd_norm = norm(_dx_sd);
      gnComputed = false;
    }
    if (_dx_sd_norm >= _delta && _delta != 0) {     // trust region smaller than the steepest-descent step
      _dx.resize(_dx_sd.size());
      for (std::size_t i = 0; i < _dx.size(); ++i) _dx[i] = _delta / _dx_sd_norm * _dx_sd[i];
      _L0 = _delta * (2 * _dx_sd_norm - _delta) / (2 * _sd_scale);
      _stepType = "SD";
    } else {
      if (!gnComputed) {
        if (!_solver->solveSystem(_dx_gn)) return false;
        gnComputed = true;
        _dx_gn_norm = norm(_dx_gn);
      }
      if (_delta == 0) {                      // first step: a trust region that admits the full GN step
        double s = 0.0;
        for (std::size_t i = 0; i < _dx_gn.size(); ++i) { double v = _dx_sd[i] + 0.5 * (_dx_gn[i] - _dx_sd[i]); s += v * v; }
        _delta = std::sqrt(s);
      }
      if (_dx_gn_norm <= _delta) {
        _dx = _dx_gn;
        _L0 = J;
        _stepType = "GN";
      } else {                                // on the line between the Cauchy point and the GN step
        std::vector<double> dgnsd(_dx_gn.size());
        for (std::size_t i = 0; i < dgnsd.size(); ++i) dgnsd[i] = _dx_gn[i] - _dx_sd[i];
        double gdnsd_norm_sqr = dotv(dgnsd, dgnsd);
        double c = dotv(_dx_sd, dgnsd);
        double disc = std::sqrt(c * c + gdnsd_norm_sqr * (_delta * _delta - _dx_sd_norm * _dx_sd_norm));
        if (c <= 0) {
          _beta = (-c + disc) / gdnsd_norm_sqr;
        } else {
          _beta = (_delta * _delta - _dx_sd_norm * _dx_sd_norm) / (c + disc);
        }
        _dx.resize(dgnsd.size());
        for (std::size_t i = 0; i < _dx.size(); ++i) _dx[i] = _dx_sd[i] + _beta * dgnsd[i];
        // the reference writes `1/2 * ...`, an integer division that is 0 (DogLegTrustRegionPolicy.cpp:134): the
        // first summand vanishes there, so it is left out here too
        _L0 = _beta * (2 - _beta) * J;
        _stepType = "DL";
      }
    }
    outDx = _dx;
    return true;
  }
  double _dx_sd_norm = 0, _dx_gn_norm = 0, _L0 = 0, _sd_scale = 0, _beta = 0, _delta = 0, _p_delta = 0;
  std::string _stepType;
  std::vector<double> _dx, _dx_sd, _dx_gn;
};

struct Optimizer2Options {
  double convergenceJDescentRatioThreshold = 1e-5;
  double convergenceDeltaX = 1e-3;
  int maxIterations = 20;
  bool verbose = false;
  int linearSolverMaximumFails = 0;
  int nThreads = 4;
  std::shared_ptr<LinearSystemSolver> linearSystemSolver;
  std::shared_ptr<TrustRegionPolicy> trustRegionPolicy;
};

// aslam_backend_python/src/Optimizer.cpp:30-39
struct SolutionReturnValue {
  double JStart = 0, JFinal = 0;
  int iterations = 0, failedIterations = 0;
  double dXFinal = 0, dJFinal = 0;
  bool linearSolverFailure = false;
};

class Optimizer2 {
 public:
  // wall-clock per phase; the reference's tags (Optimizer2.cpp:185-189) are compiled out there (Optimizer2.hpp:53-55)
  struct Timing { double evaluateError = 0, buildSystem = 0, solveSystem = 0, backSubstitution = 0, initialize = 0; };

  explicit Optimizer2(const Optimizer2Options& options) : _options(options) {}
  Optimizer2Options& options() { return _options; }
  const Timing& timing() const { return _timing; }
  double J() const { return _J; }
  const std::vector<double>& dx() const { return _dx; }

  // Optimizer2::initialize (:95-151): block indices, column and row bases are assigned inside initMatrixStructure
  // (mc_finalize) in the same order -- active design variables in problem order, error terms in problem order
  void initialize() {
    if (!_options.linearSystemSolver)
      throw Exception("Optimizer2Options::linearSystemSolver must be set (the CPU default, sparse_cholesky on the "
                      "problem's own error terms, is what this backend replaces)");
    _solver = _options.linearSystemSolver;
    if (!_options.trustRegionPolicy) _trustRegionPolicy.reset(new LevenbergMarquardtTrustRegionPolicy());
    else _trustRegionPolicy = _options.trustRegionPolicy;
    Clock c;
    _solver->initMatrixStructure(_trustRegionPolicy->requiresAugmentedDiagonal());
    _timing.initialize += c.seconds();
    if (_options.verbose)
      std::printf("The Jacobian matrix is %zu x %zu\n", _solver->JRows(), _solver->JCols());
  }

  // Optimizer2::optimize (:183-275)
  SolutionReturnValue optimize() {
    initialize();
    SolutionReturnValue srv;
    _p_J = 0.0;
    evaluateError(true);
    _p_J = _J;
    srv.JStart = _p_J;
    if (_options.verbose) std::printf("[%d.0]: J: %.10g\n", srv.iterations, _J);
    double deltaX = _options.convergenceDeltaX + 1.0;
    double JDescentRatio = _options.convergenceJDescentRatioThreshold + 1.0;
    double deltaJ = 0.0;
    bool previousIterationFailed = false;
    bool linearSolverFailure = false;
    _trustRegionPolicy->setSolver(_solver);
    _trustRegionPolicy->optimizationStarting(_J);
    while (srv.iterations < _options.maxIterations && srv.failedIterations < _options.maxIterations &&
           ((deltaX > _options.convergenceDeltaX && std::fabs(JDescentRatio) > _options.convergenceJDescentRatioThreshold) ||
            linearSolverFailure)) {
      Clock cs;
      const double b0 = _solver->secondsInBuildSystem();   // buildSystem is called from inside the policy
      bool solutionSuccess = _trustRegionPolicy->solveSystem(_J, previousIterationFailed, _options.nThreads, _dx);
      const double db = _solver->secondsInBuildSystem() - b0;
      _timing.buildSystem += db;
      _timing.solveSystem += cs.seconds() - db;
      if (!solutionSuccess) {
        if (_options.verbose) std::printf("[WARNING] System solution failed\n");
        previousIterationFailed = true;
        linearSolverFailure = true;
        srv.failedIterations++;
      } else {
        Clock cb;
        deltaX = applyStateUpdate();
        _timing.backSubstitution += cb.seconds();
        evaluateError(true);
        deltaJ = _p_J - _J;
        JDescentRatio = deltaJ / _p_J;
        if (_trustRegionPolicy->revertOnFailure()) {
          if (JDescentRatio < 0.0) {
            if (_options.verbose) std::printf("Last step was a regression. Reverting\n");
            revertLastStateUpdate();
            srv.failedIterations++;
            previousIterationFailed = true;
          } else {
            _p_J = _J;
            previousIterationFailed = false;
          }
        } else {
          _p_J = _J;
        }
        srv.iterations++;
        if (_options.verbose)
          std::printf("[%d]: J: %.10g, dJ: %g, deltaX: %g, %s\n", srv.iterations, _J, deltaJ, deltaX,
                      _trustRegionPolicy->printState().c_str());
      }
    }
    srv.JFinal = _p_J;
    srv.dXFinal = deltaX;
    srv.dJFinal = deltaJ;
    srv.linearSolverFailure = linearSolverFailure;
    return srv;
  }

  // Optimizer2::applyStateUpdate (:292-309): the per-design-variable scaling and update run on the device
  double applyStateUpdate() {
    _solver->applyStateUpdate(_dx);
    double deltaX = 0.0;
    for (double v : _dx) deltaX = std::fmax(deltaX, std::fabs(v));
    return deltaX;
  }
  void revertLastStateUpdate() { _solver->revertLastStateUpdate(); }
  double evaluateError(bool useMEstimator) {
    Clock c;
    _J = _solver->evaluateError(_options.nThreads, useMEstimator);
    _timing.evaluateError += c.seconds();
    return _J;
  }

 private:
  struct Clock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  };
  Optimizer2Options _options;
  std::shared_ptr<LinearSystemSolver> _solver;
  std::shared_ptr<TrustRegionPolicy> _trustRegionPolicy;
  double _J = 0, _p_J = 0;
  std::vector<double> _dx;
  Timing _timing;
};

}  // namespace multical_amd
