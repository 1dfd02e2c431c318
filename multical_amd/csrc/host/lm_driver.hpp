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
//   * the factorisation keeps the reference's damping residue on the diagonal of the adopted system (augmented by lambda^2,
//     relieved by lambda: BlockCholeskyLinearSystemSolver.cpp:76-98) until the next adoption.  Two ways to factorise:
//     the GPU stage (mc_schur.hip: interiors of the pose-spline band eliminated on the device from the adopted device system,
//     the reduced system on the host; the residue is one scalar added to every diagonal entry) when the problem has such a
//     band, else -- or with MC_GPU_SOLVE=0, or with shards -- the host factorisation on a pinned host image of the system,
//     whose diagonal carries the residue.
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
#include "lm_core.hpp"
#include "schur_device.hpp"

namespace multical_amd {

namespace detail {
inline void hipOk(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Exception(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace detail

// What a sharded run (mc_set_shard, one process per GPU) adds: the caller's collectives.  Every rank's context holds the
// terms of its own time window; pattern and numbering are global, so partial results add element-wise.
//   allreduce  in-place sum over the ranks of n doubles in DEVICE memory (RCCL through torch.distributed in the Python
//              binding; anything else that sums a device buffer)
//   broadcast  n doubles of HOST memory from rank `root` to every rank
// Both return 0 on success.  The minimiser calls them in the same order on every rank.
struct ShardHooks {
  int rank = 0, world = 1;
  int (*allreduce)(void* user, double* device_buf, long long n) = nullptr;
  int (*broadcast)(void* user, double* host_buf, long long n, int root) = nullptr;
  void* user = nullptr;
  bool active() const { return world > 1; }
};

// The fused GPU path: two device systems (adopted / trial), one pinned host image for the factorisation.
// With shards (Optimizer2::optimize over time-window shards): a trial's cost is summed over the ranks at once
// (one double); its partial normal equations stay on the device and are summed only if the trial's linearisation is
// adopted (one all-reduce of the value array and the gradient per ADOPTED step, none for a rejected one); rank 0 factorises
// and solves, the step is broadcast, and every rank applies the same update to its replica of the state.
class FusedBlockEquations : public NormalEquations {
 public:
  explicit FusedBlockEquations(mc_context* ctx, ShardHooks hooks = ShardHooks()) : ctx_(ctx), hooks_(hooks) {
    const char* v = std::getenv("MC_GPU_SOLVE");
    // with shards the summed system is on every rank's device; rank 0 solves (fp64 atomics make the stage's result depend on
    // the order of additions in the last bits, and the ranks' states must stay bit-identical: one solve, then the broadcast)
    gpuSolve_ = !(v && v[0] == '0') && (!hooks_.active() || hooks_.rank == 0);
  }
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
    rowIdx_.swap(rowIdx); base_.swap(base);
    when_.assign(blocks, 0.0);
    ok(mc_get_block_times(ctx_, when_.data()));
    chol_.setBlockCoordinates(when_.data(), hostCholeskyThreads());   // time-axis dissection: one thread per part
    for (int k = 0; k < 2; ++k) {
      detail::hipOk(hipMalloc((void**)&dH_[k], sizeof(double) * (size_t)nnz_), "hipMalloc(system values)");
      detail::hipOk(hipMalloc((void**)&dG_[k], sizeof(double) * (size_t)n_), "hipMalloc(gradient)");
      // neither system is ever read before a trial has written it (trialPending_), but a zero system fails loudly where
      // uninitialised memory would not
      detail::hipOk(hipMemset(dH_[k], 0, sizeof(double) * (size_t)nnz_), "hipMemset(system values)");
      detail::hipOk(hipMemset(dG_[k], 0, sizeof(double) * (size_t)n_), "hipMemset(gradient)");
    }
    detail::hipOk(hipMalloc((void**)&dCost_, 2 * sizeof(double)), "hipMalloc(cost)");
    detail::hipOk(hipHostMalloc((void**)&hH_, sizeof(double) * (size_t)nnz_, hipHostMallocDefault), "hipHostMalloc(system image)");
    g_.assign(n_, 0.0);
  }
  double trial() override {
    const int t = 1 - cur_;
    ok(mc_evaluate_error_and_build_system_device(ctx_, 1, dCost_, nullptr, dH_[t], dG_[t], nullptr));
    if (hooks_.active()) {
      // the device status travels with the cost, so that every rank sees a failure of any rank and all of them throw together
      // (a rank that threw alone would leave the others waiting in the next collective)
      const int st = mc_device_status(ctx_);
      const double flag = st != 0 ? 1.0 : 0.0;
      detail::hipOk(hipMemcpy(dCost_ + 1, &flag, sizeof(double), hipMemcpyHostToDevice), "status flag");
      if (hooks_.allreduce(hooks_.user, dCost_, 2) != 0) throw Exception("sharded minimiser: all-reduce of the cost failed");
      double both[2] = {0.0, 0.0};
      detail::hipOk(hipMemcpy(both, dCost_, 2 * sizeof(double), hipMemcpyDeviceToHost), "cost read-back");
      if (both[1] != 0.0) throw Exception(st != 0 ? mc_last_error(ctx_) : "a device error was raised on another rank");
      trialPending_ = true;
      return both[0];
    }
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
    if (hooks_.active()) {
      // the ranks' partial systems become the whole one, identical on every rank
      if (hooks_.allreduce(hooks_.user, dH_[cur_], nnz_) != 0 || hooks_.allreduce(hooks_.user, dG_[cur_], n_) != 0)
        throw Exception("sharded minimiser: all-reduce of the normal equations failed");
    }
    shift_ = 0.0;
    hostImage_ = false;
    // the GPU stage works on the device system (8 nnz bytes stay where they are); with shards only rank 0 factorises at all
    if (!gpuSolve_ && (!hooks_.active() || hooks_.rank == 0)) readBackSystem();
    detail::hipOk(hipMemcpy(g_.data(), dG_[cur_], sizeof(double) * (size_t)n_, hipMemcpyDeviceToHost), "gradient read-back");
  }
  bool step(double damping, std::vector<double>& dx) override {
    dx.assign(n_, 0.0);
    if (gpuSolve_) {
      if (damped_) shift_ += damping * damping;
      int rc = planGpuStage() ? schur_->solve(dH_[cur_], dG_[cur_], shift_, dx.data(), nullptr) : 3;
      if (rc == 2) {                       // a block that was zero when the plan was made holds a value now
        schur_.reset();
        rc = planGpuStage() ? schur_->solve(dH_[cur_], dG_[cur_], shift_, dx.data(), nullptr) : 3;
      }
      if (rc == 0 || rc == 1) {
        bool good = rc == 0;
        if (good) for (double v : dx) if (!std::isfinite(v)) { good = false; break; }
        if (damped_) shift_ -= damping;    // the reference relieves the diagonal by the damping, not by its square
        return shareStep(good, dx);
      }
      // no band the kernels take (or it outgrew them): the host factorisation from here on, on an image with the residue so far
      gpuSolve_ = false;
      if (damped_) shift_ -= damping * damping;
      if (!hostImage_) readBackSystem();
      for (int c = 0; c < n_; ++c) hH_[colEnd_[c + 1] - 1] += shift_;
    }
    // the diagonal entry of column c is the last one of the column (upper triangle in compressed columns)
    if (damped_) for (int c = 0; c < n_; ++c) hH_[colEnd_[c + 1] - 1] += damping * damping;
    bool good = true;
    if (!hooks_.active() || hooks_.rank == 0) {   // with shards: one factorisation, on rank 0
      good = chol_.factorize(hH_);
      if (good) {
        chol_.solve(g_.data(), dx.data());
        for (double v : dx) if (!std::isfinite(v)) { good = false; break; }
      }
    }
    // the reference relieves the diagonal by the damping, not by its square; the difference stays until the next adoption
    if (damped_) for (int c = 0; c < n_; ++c) hH_[colEnd_[c + 1] - 1] -= damping;
    return shareStep(good, dx);
  }
  const std::vector<double>& gradient() const override { return g_; }
  void applyUpdate(const std::vector<double>& dx) override { ok(mc_apply_update(ctx_, dx.data())); }
  void revertUpdate() override { ok(mc_revert_update(ctx_)); }
  void factorStats(double& nnz, double& flops) override {
    if (gpuSolve_ && schur_ && schur_->planned()) { nnz = (double)schur_->info().reduced_factor_nnz; flops = schur_->info().reduced_factor_flops; return; }
    nnz = (double)chol_.stats().nnz_factor; flops = chol_.stats().flops;
  }
  bool usesGpuStage() const { return gpuSolve_ && schur_ && schur_->planned(); }

 private:
  void ok(int rc) const { if (rc < 0) throw Exception(mc_last_error(ctx_)); }
  // with shards: rank 0's step (and whether it has one) goes to every rank
  bool shareStep(bool good, std::vector<double>& dx) {
    if (!hooks_.active()) return good;
    dx.push_back(good ? 1.0 : 0.0);
    if (hooks_.broadcast(hooks_.user, dx.data(), (long long)dx.size(), 0) != 0) throw Exception("sharded minimiser: broadcast of the step failed");
    good = dx.back() != 0.0;
    dx.pop_back();
    return good;
  }
  void readBackSystem() {
    detail::hipOk(hipMemcpy(hH_, dH_[cur_], sizeof(double) * (size_t)nnz_, hipMemcpyDeviceToHost), "system read-back");
    hostImage_ = true;
  }
  // plans the GPU stage on the non-zero blocks of the adopted system (once, and again when the structure grew)
  bool planGpuStage() {
    if (schur_ && schur_->planned()) return true;
    if (schurFailed_) return false;
    if (!hostImage_) readBackSystem();
    std::vector<int> spl(base_.size() - 1);
    ok(mc_get_block_splines(ctx_, spl.data()));
    int band = 0;
    {
      std::vector<int> count;
      for (int s : spl) if (s >= 0) { if ((int)count.size() <= s) count.resize(s + 1, 0); band = std::max(band, ++count[s]); }
    }
    int device = 0;
    detail::hipOk(hipGetDevice(&device), "hipGetDevice");
    schur_.reset(new SchurSolver());
    try {
      if (schur_->plan(device, n_, colEnd_.data(), rowIdx_.data(), (int)base_.size() - 1, base_.data(), spl.data(), when_.data(), hH_, schurPartsFor(band),
                       hostCholeskyThreads()))
        return true;
    } catch (const std::exception& e) {
      throw Exception(std::string("factorisation, GPU stage: ") + e.what());
    }
    schurFailed_ = true;
    return false;
  }
  mc_context* ctx_;
  ShardHooks hooks_;
  bool damped_ = true, trialPending_ = false;
  int n_ = 0, nnz_ = 0, cur_ = 0;
  double* dH_[2] = {nullptr, nullptr};
  double* dG_[2] = {nullptr, nullptr};
  double* dCost_ = nullptr;
  double* hH_ = nullptr;
  std::vector<int> colEnd_, rowIdx_, base_;
  std::vector<double> g_, when_;
  BlockSparseCholesky chol_;
  // the GPU stage of the factorisation
  bool gpuSolve_ = false, schurFailed_ = false, hostImage_ = false;
  double shift_ = 0.0;          // the damping residue on the diagonal (the same for every column)
  std::unique_ptr<SchurSolver> schur_;
};

// The second backend (J^T in compressed columns): nothing to speculate with -- the trial is the residual pass and the
// linearisation is formed when it is adopted, at the state the trial was evaluated at.
class TransposeEquations : public NormalEquations {
 public:
  explicit TransposeEquations(mc_context* ctx) : ctx_(ctx), s_(ctx) {}
  void prepare(bool damped) override { s_.initMatrixStructure(damped); }
  double trial() override { return s_.evaluateError(1, true); }
  void adopt() override { s_.buildSystem(1, true); }
  bool step(double damping, std::vector<double>& dx) override { s_.setConstantConditioner(damping); return s_.solveSystem(dx); }
  const std::vector<double>& gradient() const override { return s_.rhs(); }
  void applyUpdate(const std::vector<double>& dx) override { if (mc_apply_update(ctx_, dx.data()) < 0) throw Exception(mc_last_error(ctx_)); }
  void revertUpdate() override { if (mc_revert_update(ctx_) < 0) throw Exception(mc_last_error(ctx_)); }

 private:
  mc_context* ctx_;
  HipSparseCholeskyLinearSystemSolver s_;
};

// The minimiser on the problem held by ctx: one unsharded context, or -- with hooks -- this rank's shard of it.
inline MinimiserReport minimise(mc_context* ctx, const MinimiserSettings& cfg, ShardHooks hooks = ShardHooks()) {
  if (!ctx) throw Exception("minimise: null context");
  std::unique_ptr<NormalEquations> eq;
  if (cfg.transposeBackend) {
    if (hooks.active()) throw Exception("the J^T backend is single-GPU");
    eq.reset(new TransposeEquations(ctx));
  } else eq.reset(new FusedBlockEquations(ctx, hooks));
  return minimise(eq.get(), cfg);
}

}  // namespace multical_amd
