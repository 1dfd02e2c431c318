// mc_driver.hip -- mc_optimize: the minimiser of host/lm_driver.hpp behind its C entry point.  Host code only (no kernels);
// compiled with hipcc because the driver owns device buffers and a pinned host image (hipMalloc / hipHostMalloc /
// hipMemcpy).  Everything else it needs from the GPU goes through include/multical_hip.h, like an integrator's code would.
#include <cstring>
#include <string>

#include "../../include/multical_hip.h"
#include "host/lm_driver.hpp"

using namespace multical_amd;

static thread_local std::string g_driver_error;

extern "C" {

const char* mc_host_last_error(void) { return g_driver_error.c_str(); }

void mc_optimizer_options_default(mc_optimizer_options* o) {
  // Optimizer2Options' defaults (aslam_backend/include/aslam/backend/Optimizer2Options.hpp:9-42)
  o->convergenceJDescentRatioThreshold = 1e-5;
  o->convergenceDeltaX = 1e-3;
  o->maxIterations = 20;
  o->verbose = 0;
  o->nThreads = 4;
  o->trustRegionPolicy = MC_POLICY_LEVENBERG_MARQUARDT;
  o->levenbergMarquardtLambdaInit = 1e-3;
  o->linearSolver = MC_SOLVER_BLOCK_CHOLESKY;
}

static int optimizeImpl(mc_context* ctx, const mc_optimizer_options* opt, mc_solution_return_value* out, ShardHooks hooks, bool sharded) {
  try {
    if (!ctx || !opt || !out) throw Exception("mc_optimize: null argument");
    {
      int rank = 0, world = 1;
      mc_shard_info(ctx, &rank, &world);
      if (!sharded && world > 1)
        // a sharded context returns this rank's partial cost, H and rhs: factorising those would apply a wrong step silently
        throw Exception("mc_optimize drives one unsharded context; with mc_set_shard(world > 1) use mc_optimize_sharded, "
                        "which sums the cost, the normal equations and the step over the ranks");
      hooks.rank = rank; hooks.world = world;
      if (sharded && world > 1 && (!hooks.allreduce || !hooks.broadcast)) throw Exception("mc_optimize_sharded: both callbacks are required");
      // the minimiser sums the WHOLE Cx over the ranks into its own buffers: the other ranks' values land in them, which
      // is what mc_set_clear_mode(1) rules out
      if (sharded && world > 1) mc_set_clear_mode(ctx, 0);
    }
    MinimiserSettings cfg;
    cfg.stepTolerance = opt->convergenceDeltaX;
    cfg.costRatioTolerance = opt->convergenceJDescentRatioThreshold;
    cfg.maxSteps = opt->maxIterations;
    cfg.verbose = opt->verbose != 0 && hooks.rank == 0;
    cfg.initialDamping = opt->levenbergMarquardtLambdaInit;
    switch (opt->trustRegionPolicy) {
      case MC_POLICY_LEVENBERG_MARQUARDT: cfg.damped = true; break;
      case MC_POLICY_GAUSS_NEWTON: cfg.damped = false; break;
      default: throw Exception("mc_optimize: trustRegionPolicy must be MC_POLICY_LEVENBERG_MARQUARDT or MC_POLICY_GAUSS_NEWTON");
    }
    switch (opt->linearSolver) {
      case MC_SOLVER_BLOCK_CHOLESKY: cfg.transposeBackend = false; break;
      case MC_SOLVER_SPARSE_CHOLESKY: cfg.transposeBackend = true; break;
      default: throw Exception("mc_optimize: unknown linearSolver");
    }
    const MinimiserReport r = minimise(ctx, cfg, hooks);
    std::memset(out, 0, sizeof *out);
    out->JStart = r.costStart; out->JFinal = r.costFinal;
    out->iterations = r.trials; out->failedIterations = r.failures;
    out->dXFinal = r.lastStep; out->dJFinal = r.lastCostChange;
    out->linearSolverFailure = r.solverFailed ? 1 : 0;
    out->seconds_initialize = r.secondsSetup;
    out->seconds_evaluate_error = r.secondsTrials;
    out->seconds_build_system = r.secondsAdopt;
    out->seconds_solve_system = r.secondsSolve;
    out->seconds_state_update = r.secondsUpdate;
    out->factor_nnz = r.factorNnz;
    out->factor_flops = r.factorFlops;
    return 0;
  } catch (const std::exception& e) {
    g_driver_error = e.what();
    return -1;
  }
}

int mc_optimize(mc_context* ctx, const mc_optimizer_options* opt, mc_solution_return_value* out) {
  return optimizeImpl(ctx, opt, out, ShardHooks(), false);
}

int mc_optimize_sharded(mc_context* ctx, const mc_optimizer_options* opt, mc_solution_return_value* out, mc_allreduce_sum_fn allreduce,
                        mc_broadcast_fn broadcast, void* user) {
  ShardHooks h;
  h.allreduce = allreduce; h.broadcast = broadcast; h.user = user;
  return optimizeImpl(ctx, opt, out, h, true);
}

}  // extern "C"
