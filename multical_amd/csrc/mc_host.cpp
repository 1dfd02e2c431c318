// mc_host.cpp -- C entry points of the host sparse Cholesky (host/block_cholesky.hpp) that consumes the hot path's result.
// Plain host C++ (g++), no HIP.  The minimiser that calls the hot path is mc_driver.hip (host/lm_driver.hpp).
#include <cmath>
#include <cstring>
#include <utility>
#include <memory>
#include <string>

#include "../../include/multical_hip.h"
#include "host/block_cholesky.hpp"

using namespace multical_amd;

struct mc_cholesky {
  BlockSparseCholesky chol;
  std::string err;
};

extern "C" {

mc_cholesky* mc_cholesky_create(void) {
  try { return new mc_cholesky(); } catch (...) { return nullptr; }
}
void mc_cholesky_destroy(mc_cholesky* h) { delete h; }
const char* mc_cholesky_last_error(const mc_cholesky* h) { return h ? h->err.c_str() : ""; }

int mc_cholesky_analyze(mc_cholesky* h, int n, const int* Cp, const int* Ci, int n_blocks, const int* block_base) {
  try { h->chol.analyze(n, Cp, Ci, n_blocks, block_base); return 0; }
  catch (const std::exception& e) { h->err = e.what(); return -1; }
}
int mc_cholesky_set_block_coordinates(mc_cholesky* h, const double* coord, int threads) {
  try { h->chol.setBlockCoordinates(coord, threads); return 0; }
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
  const BlockSparseCholesky::Stats& s = const_cast<mc_cholesky*>(h)->chol.stats();
  if (n_dense_blocks) *n_dense_blocks = s.n_dense_blocks;
  if (nnz_factor) *nnz_factor = s.nnz_factor;
  if (flops) *flops = s.flops;
  return 0;
}
int mc_cholesky_block_permutation(const mc_cholesky* h, int* perm) {
  const std::vector<int>& p = const_cast<mc_cholesky*>(h)->chol.blockPermutation();
  std::memcpy(perm, p.data(), sizeof(int) * p.size());
  return (int)p.size();
}


// sm::eigen::computeMatrixSqrt (Schweizer-Messer/sm_eigen/include/sm/eigen/matrix_sqrt.hpp:21-40), which
// ErrorTermFs<D>::setInvR uses (aslam_backend/include/aslam/backend/implementation/ErrorTerm.hpp:96-192):
// S = P^T L sqrt(D) of Eigen's pivoted LDL^T, so that A = S S^T.  Eigen's factorisation is left-looking: when it picks
// the pivot of step k the trailing diagonal has not been reduced yet, so the pivot order is simply "largest remaining
// diagonal entry of A itself" (first one on ties).  The permuted matrix B = P A P^T is therefore never modified here;
// column k of L and D[k] are computed from B and the finished columns.  Row major, n <= 16; a negative D gives NaN.
int mc_compute_matrix_sqrt(int n, const double* A, double* S) {
  if (n < 1 || n > 16 || !A || !S) return -1;
  double B[256], L[256], D[16];
  int tr[16];
  for (int i = 0; i < n * n; ++i) { B[i] = A[i]; L[i] = 0.0; }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int j = k + 1; j < n; ++j)
      if (std::fabs(B[j * n + j]) > std::fabs(B[piv * n + piv])) piv = j;
    tr[k] = piv;
    if (piv != k) {
      for (int c = 0; c < n; ++c) std::swap(B[k * n + c], B[piv * n + c]);
      for (int r = 0; r < n; ++r) std::swap(B[r * n + k], B[r * n + piv]);
      for (int c = 0; c < k; ++c) std::swap(L[k * n + c], L[piv * n + c]);
    }
    double d = B[k * n + k];
    for (int j = 0; j < k; ++j) d -= L[k * n + j] * L[k * n + j] * D[j];
    D[k] = d;
    L[k * n + k] = 1.0;
    for (int i = k + 1; i < n; ++i) {
      double v = B[i * n + k];
      for (int j = 0; j < k; ++j) v -= L[i * n + j] * D[j] * L[k * n + j];
      L[i * n + k] = d != 0.0 ? v / d : v;
    }
  }
  // rows back to the original order (the transpositions in reverse), then the columns scaled by sqrt(D)
  for (int k = n - 1; k >= 0; --k)
    if (tr[k] != k)
      for (int c = 0; c < n; ++c) std::swap(L[k * n + c], L[tr[k] * n + c]);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) S[i * n + j] = L[i * n + j] * std::sqrt(D[j]);
  return 0;
}
}  // extern "C"
