// schur_device.hpp -- interface of the factorisation's GPU stage (implementation and kernels: mc_schur.hip; the symbolic
// plan: schur_plan.hpp).  Used by the minimiser (lm_driver.hpp) and, through mc_schur_*, by the tests and the benchmark.
#pragma once
#include <memory>
#include <string>

namespace multical_amd {

class SchurSolver {
 public:
  struct Info {
    int parts = 0, r = 0, d = 0;
    int n = 0, reduced_n = 0;
    long long nnz = 0, reduced_nnz = 0;
    int watched_pairs = 0;        // block pairs next to an interior that were zero when the plan was made
    int max_boundary = 0, max_interior = 0;
    double work_bytes = 0;        // device memory of the work arrays
    double plan_seconds = 0;
    // of the last solve: [clear + copy + gather, eliminate (band Cholesky + W), products + scatter, read-back,
    //                     host factor + solve of the reduced system, back-substitution + read-back of x]
    double seconds[6] = {0, 0, 0, 0, 0, 0};
    long long reduced_factor_nnz = 0;
    double reduced_factor_flops = 0;
  };
  SchurSolver();
  ~SchurSolver();
  // Plan on the pattern (scalar CCS upper triangle in fillCCS block layout) and the non-zero blocks of Cx_host.
  // spline_of[b]: spline of a coefficient block or -1; when[b]: time coordinate of a coefficient block or NaN.
  // false: this problem has no band the kernels take (why() says which); the caller keeps the host factorisation.
  bool plan(int device, int n, const int* Cp, const int* Ci, int n_blocks, const int* block_base, const int* spline_of,
            const double* when, const double* Cx_host, int parts, int host_threads);
  // (H + shift I) x = g, H (values of the planned pattern) and g in device memory, x to the host.
  // 0 ok, 1 not positive definite, 2 a watched block has become non-zero: plan again on the new values.  Throws on HIP errors.
  int solve(const double* d_Cx, const double* d_g, double shift, double* x_host, void* stream);
  const Info& info() const;
  const std::string& why() const;
  bool planned() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> p_;
};

// interiors asked for: MC_SCHUR_PARTS, else one per ~90 band blocks, at most 512
int schurPartsFor(int band_blocks);

}  // namespace multical_amd
