// mc_host.cpp -- C entry points of the host side that sits ABOVE the device C ABI: the optimiser loop that calls the
// hot path (host/optimizer2.hpp), the two LinearSystemSolver classes (hip_linear_system_solver.hpp) and the host sparse
// Cholesky (host/block_cholesky.hpp).  Plain host C++ (g++), no HIP: everything it needs from the GPU goes through
// include/multical_hip.h, like an integrator's code would.
#include <cstring>
#include <memory>
#include <string>

#include "../../include/multical_hip.h"
#include "host/optimizer2.hpp"

using namespace multical_amd;

struct mc_cholesky {
  BlockSparseCholesky chol;
  std::string err;
};

static thread_local std::string g_host_error;

extern "C" {

const char* mc_host_last_error(void) { return g_host_error.c_str(); }

void mc_optimizer_options_default(mc_optimizer_options* o) {
  Optimizer2Options d;
  o->convergenceJDescentRatioThreshold = d.convergenceJDescentRatioThreshold;
  o->convergenceDeltaX = d.convergenceDeltaX;
  o->maxIterations = d.maxIterations;
  o->verbose = d.verbose ? 1 : 0;
  o->nThreads = d.nThreads;
  o->trustRegionPolicy = MC_POLICY_LEVENBERG_MARQUARDT;
  o->levenbergMarquardtLambdaInit = 1e-3;
  o->linearSolver = MC_SOLVER_BLOCK_CHOLESKY;
}

int mc_optimize(mc_context* ctx, const mc_optimizer_options* opt, mc_solution_return_value* out) {
  try {
    if (!ctx || !opt || !out) throw Exception("mc_optimize: null argument");
    Optimizer2Options o;
    o.convergenceJDescentRatioThreshold = opt->convergenceJDescentRatioThreshold;
    o.convergenceDeltaX = opt->convergenceDeltaX;
    o.maxIterations = opt->maxIterations;
    o.verbose = opt->verbose != 0;
    o.nThreads = opt->nThreads;
    switch (opt->linearSolver) {
      case MC_SOLVER_BLOCK_CHOLESKY: o.linearSystemSolver.reset(new HipBlockCholeskyLinearSystemSolver(ctx)); break;
      case MC_SOLVER_SPARSE_CHOLESKY: o.linearSystemSolver.reset(new HipSparseCholeskyLinearSystemSolver(ctx)); break;
      default: throw Exception("mc_optimize: unknown linearSolver");
    }
    switch (opt->trustRegionPolicy) {
      case MC_POLICY_LEVENBERG_MARQUARDT:
        o.trustRegionPolicy.reset(new LevenbergMarquardtTrustRegionPolicy(opt->levenbergMarquardtLambdaInit)); break;
      case MC_POLICY_GAUSS_NEWTON: o.trustRegionPolicy.reset(new GaussNewtonTrustRegionPolicy()); break;
      case MC_POLICY_DOG_LEG: o.trustRegionPolicy.reset(new DogLegTrustRegionPolicy()); break;
      default: throw Exception("mc_optimize: unknown trustRegionPolicy");
    }
    Optimizer2 optimizer(o);
    SolutionReturnValue srv = optimizer.optimize();
    std::memset(out, 0, sizeof *out);
    out->JStart = srv.JStart; out->JFinal = srv.JFinal;
    out->iterations = srv.iterations; out->failedIterations = srv.failedIterations;
    out->dXFinal = srv.dXFinal; out->dJFinal = srv.dJFinal;
    out->linearSolverFailure = srv.linearSolverFailure ? 1 : 0;
    const Optimizer2::Timing& t = optimizer.timing();
    out->seconds_initialize = t.initialize;
    out->seconds_evaluate_error = t.evaluateError;
    out->seconds_build_system = t.buildSystem;
    out->seconds_solve_system = t.solveSystem;
    out->seconds_state_update = t.backSubstitution;
    if (auto* b = dynamic_cast<HipBlockCholeskyLinearSystemSolver*>(o.linearSystemSolver.get())) {
      out->factor_nnz = (double)b->factorisation().stats().nnz_factor;
      out->factor_flops = b->factorisation().stats().flops;
    }
    return 0;
  } catch (const std::exception& e) {
    g_host_error = e.what();
    return -1;
  }
}

mc_cholesky* mc_cholesky_create(void) {
  try { return new mc_cholesky(); } catch (...) { return nullptr; }
}
void mc_cholesky_destroy(mc_cholesky* h) { delete h; }
const char* mc_cholesky_last_error(const mc_cholesky* h) { return h ? h->err.c_str() : ""; }

int mc_cholesky_analyze(mc_cholesky* h, int n, const int* Cp, const int* Ci, int n_blocks, const int* block_base) {
  try { h->chol.analyze(n, Cp, Ci, n_blocks, block_base); return 0; }
  catch (const std::exception& e) { h->err = e.what(); return -1; }
}
int mc_cholesky_factorize(mc_cholesky* h, const double* Cx) {
  try { return h->chol.factorize(Cx) ? 0 : 1; }
  catch (const std::exception& e) { h->err = e.what(); return -1; }
}
int mc_cholesky_solve(mc_cholesky* h, const double* b, double* x) {
  try { h->chol.solve(b, x); return 0; }
  catch (const std::exception& e) { h->err = e.what(); return -1; }
}
int mc_cholesky_covariance_blocks(mc_cholesky* h, int n_pairs, const int* block_row, const int* block_col, const long long* offsets,
                                  double* out) {
  try { h->chol.covarianceBlocks(n_pairs, block_row, block_col, offsets, out); return 0; }
  catch (const std::exception& e) { h->err = e.what(); return -1; }
}
int mc_cholesky_stats(const mc_cholesky* h, int* n_dense_blocks, long long* nnz_factor, double* flops) {
  const BlockSparseCholesky::Stats& s = h->chol.stats();
  if (n_dense_blocks) *n_dense_blocks = s.n_dense_blocks;
  if (nnz_factor) *nnz_factor = s.nnz_factor;
  if (flops) *flops = s.flops;
  return 0;
}
int mc_cholesky_block_permutation(const mc_cholesky* h, int* perm) {
  const std::vector<int>& p = h->chol.blockPermutation();
  std::memcpy(perm, p.data(), sizeof(int) * p.size());
  return (int)p.size();
}

}  // extern "C"
