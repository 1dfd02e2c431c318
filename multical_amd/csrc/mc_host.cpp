// mc_host.cpp -- C entry points of the host sparse Cholesky (host/block_cholesky.hpp) that consumes the hot path's result.
// Plain host C++ (g++), no HIP.  The minimiser that calls the hot path is mc_driver.hip (host/lm_driver.hpp).
#include <cstring>
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
