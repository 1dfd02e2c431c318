// block_cholesky.hpp -- host sparse Cholesky for the normal equations the GPU assembles.
//
// Stands in for sparse_block_matrix::LinearSolverCholmod<MatrixXd>::solve
// (aslam_optimizer/sparse_block_matrix/include/sparse_block_matrix/linear_solver_cholmod.h:70-112): symbolic analysis once
// per matrix structure (cholmod_analyze with a fill-reducing ordering, :80-88), numeric factorisation + triangular solves
// per call, `false` when the matrix is not positive definite (:91-100).  CHOLMOD / SuiteSparse are not part of this image
// (the reference vendors a download that is absent), so the factorisation is written here, with no dependency.
//
// Input is the matrix exactly as SparseBlockMatrix::fillCCS(upper) lays it out (scalar CCS of the upper triangle) plus the
// partition of the columns into design-variable blocks.  Everything below works on those blocks: the ordering is a
// minimum-degree ordering of the block graph (weights = block dimensions, nearly dense blocks -- the calibration
// parameters every measurement touches -- postponed to the end, like AMD's dense-row rule), the factor is stored as one
// dense column-major panel per block column, and the numeric phase is a left-looking panel factorisation whose inner
// loops are unit-stride over the panel height.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

namespace multical_amd {

class BlockSparseCholesky {
 public:
  struct Stats {
    int n = 0, n_blocks = 0, n_dense_blocks = 0;
    long long nnz_factor = 0;     // scalar entries stored in the panels
    double flops = 0;             // of one numeric factorisation
  };

  // Symbolic phase.  block_base has n_blocks + 1 entries (scalar column base of every block, last = n); NULL: every
  // column is its own block.
  void analyze(int n, const int* Cp, const int* Ci, int n_blocks, const int* block_base) {
    n_ = n;
    std::vector<int> own_base;
    if (!block_base) {
      n_blocks = n;
      own_base.resize(n + 1);
      for (int i = 0; i <= n; ++i) own_base[i] = i;
      block_base = own_base.data();
    }
    nb_ = n_blocks;
    if (nb_ <= 0 || block_base[0] != 0 || block_base[nb_] != n) throw std::invalid_argument("block partition does not cover the columns");
    obase_.assign(block_base, block_base + nb_ + 1);
    std::vector<int> blk_of(n);
    for (int b = 0; b < nb_; ++b) {
      if (obase_[b + 1] <= obase_[b]) throw std::invalid_argument("empty block");
      for (int c = obase_[b]; c < obase_[b + 1]; ++c) blk_of[c] = b;
    }
    // ---- block graph (symmetric adjacency without the diagonal)
    std::vector<std::vector<int>> adj(nb_);
    {
      std::vector<int> stamp(nb_, -1);
      for (int J = 0; J < nb_; ++J) {
        for (int c = obase_[J]; c < obase_[J + 1]; ++c)
          for (int k = Cp[c]; k < Cp[c + 1]; ++k) {
            int i = Ci[k];
            if (i < 0 || i > c) throw std::invalid_argument("pattern is not upper triangular");
            int I = blk_of[i];
            if (I != J && stamp[I] != J) { stamp[I] = J; adj[J].push_back(I); adj[I].push_back(J); }
          }
      }
    }
    std::vector<int> w(nb_);
    for (int b = 0; b < nb_; ++b) w[b] = obase_[b + 1] - obase_[b];
    minimumDegree(adj, w);     // fills perm_ / iperm_
    symbolic(adj, w);          // fills the panel layout
    // ---- where every entry of the caller's value array lands in the panels
    amap_.resize(Cp[n]);
    for (int J = 0; J < nb_; ++J) {
      int pJ = iperm_[J];
      int lastI = -1; long long base = 0; int ld = 0; bool transposed = false;
      for (int c = obase_[J]; c < obase_[J + 1]; ++c) {
        int lj = c - obase_[J];
        for (int k = Cp[c]; k < Cp[c + 1]; ++k) {
          int i = Ci[k], I = blk_of[i], li = i - obase_[I];
          if (I == J) {                    // diagonal block: keep its lower triangle (r >= c)
            amap_[k] = valptr_[pJ] + lj + (long long)li * m_[pJ];
            continue;
          }
          if (I != lastI) {
            lastI = I;
            int pI = iperm_[I];
            int col = std::min(pI, pJ), row = std::max(pI, pJ);
            const int* rb = rowblk_.data() + colptr_[col];
            const int* re = rowblk_.data() + colptr_[col + 1];
            const int* it = std::lower_bound(rb, re, row);
            if (it == re || *it != row) throw std::logic_error("symbolic factor misses an entry of the matrix");
            ld = m_[col];
            base = valptr_[col] + rowoff_[colptr_[col] + (it - rb)];
            transposed = (pI < pJ);        // entry (i, j) of the upper triangle sits at (row j, column i) of the lower factor
          }
          amap_[k] = transposed ? base + lj + (long long)li * ld : base + li + (long long)lj * ld;
        }
      }
    }
    analyzed_ = true;
    factored_ = false;
  }

  // Numeric phase on values in the order of the analysed pattern.  Returns false if a pivot is not positive (or NaN).
  bool factorize(const double* Cx) {
    if (!analyzed_) throw std::logic_error("factorize before analyze");
    factored_ = false;
    std::fill(Lx_.begin(), Lx_.end(), 0.0);
    for (size_t k = 0; k < amap_.size(); ++k) Lx_[amap_[k]] = Cx[k];
    std::vector<int> head(nb_, -1), next(nb_, -1), where(nb_, 0);
    std::vector<long long> pos(nb_, 0);
    std::vector<double> tmp((size_t)max_m_ * max_d_);
    for (int j = 0; j < nb_; ++j) {
      const int dj = dim_[j], mj = m_[j];
      double* P = Lx_.data() + valptr_[j];
      for (long long p = colptr_[j]; p < colptr_[j + 1]; ++p) where[rowblk_[p]] = rowoff_[p];
      for (int k = head[j], knext; k != -1; k = knext) {
        knext = next[k];
        const long long p0 = pos[k], pe = colptr_[k + 1];
        const int dk = dim_[k], mk = m_[k], r0 = rowoff_[p0];
        const double* T = Lx_.data() + valptr_[k] + r0;      // rows of block column k from block row j downwards
        // P[rows of k below j, :] -= T * T[0:dj, :]^T, by runs of row blocks that are contiguous in both panels (in the
        // banded part a descendant's rows map onto the target's rows as two or three runs: band, border)
        for (long long q = p0; q < pe;) {
          const int src = rowoff_[q] - r0, dst = where[rowblk_[q]];
          int len = dim_[rowblk_[q]];
          long long q2 = q + 1;
          while (q2 < pe && where[rowblk_[q2]] == dst + len) { len += dim_[rowblk_[q2]]; ++q2; }
          panelUpdate(T, mk, src, len, dk, dj, P + dst, mj, tmp.data());
          q = q2;
        }
        pos[k] = p0 + 1;
        if (p0 + 1 < pe) { int r = rowblk_[p0 + 1]; next[k] = head[r]; head[r] = k; }
      }
      // panel factorisation: Cholesky of the dj x dj diagonal block and the triangular solve of the rows below it
      for (int c = 0; c < dj; ++c) {
        double* pc = P + (size_t)c * mj;
        for (int k = 0; k < c; ++k) {
          const double* pk = P + (size_t)k * mj;
          const double b = pk[c];
          for (int r = c; r < mj; ++r) pc[r] -= pk[r] * b;
        }
        const double d = pc[c];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double inv = 1.0 / std::sqrt(d);
        for (int r = c; r < mj; ++r) pc[r] *= inv;
      }
      if (colptr_[j + 1] - colptr_[j] > 1) {
        pos[j] = colptr_[j] + 1;
        int r = rowblk_[pos[j]];
        next[j] = head[r]; head[r] = j;
      }
    }
    factored_ = true;
    return true;
  }

  // x = A^-1 b with the last factorisation (b and x may alias)
  void solve(const double* b, double* x) const {
    if (!factored_) throw std::logic_error("solve before a successful factorize");
    std::vector<double> y(n_);
    for (int j = 0; j < nb_; ++j) {
      const double* s = b + obase_[perm_[j]];
      for (int l = 0; l < dim_[j]; ++l) y[pbase_[j] + l] = s[l];
    }
    for (int j = 0; j < nb_; ++j) {              // L y' = y
      const int dj = dim_[j], mj = m_[j];
      const double* P = Lx_.data() + valptr_[j];
      double* yj = y.data() + pbase_[j];
      for (int c = 0; c < dj; ++c) {
        const double* pc = P + (size_t)c * mj;
        yj[c] /= pc[c];
        const double v = yj[c];
        for (int r = c + 1; r < dj; ++r) yj[r] -= pc[r] * v;
      }
      // rows below the diagonal block, by runs of row blocks that are neighbours in the permuted order (contiguous in y)
      for (long long q = colptr_[j] + 1; q < colptr_[j + 1];) {
        const int ro = rowoff_[q], y0 = pbase_[rowblk_[q]];
        int len = dim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < colptr_[j + 1] && pbase_[rowblk_[q2]] == y0 + len) { len += dim_[rowblk_[q2]]; ++q2; }
        double* yr = y.data() + y0;
        for (int c = 0; c < dj; ++c) {
          const double* pc = P + ro + (size_t)c * mj;
          const double v = yj[c];
          for (int r = 0; r < len; ++r) yr[r] -= pc[r] * v;
        }
        q = q2;
      }
    }
    for (int j = nb_ - 1; j >= 0; --j) {         // L^T x' = y'
      const int dj = dim_[j], mj = m_[j];
      const double* P = Lx_.data() + valptr_[j];
      double* yj = y.data() + pbase_[j];
      for (long long q = colptr_[j] + 1; q < colptr_[j + 1];) {
        const int ro = rowoff_[q], y0 = pbase_[rowblk_[q]];
        int len = dim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < colptr_[j + 1] && pbase_[rowblk_[q2]] == y0 + len) { len += dim_[rowblk_[q2]]; ++q2; }
        const double* yr = y.data() + y0;
        for (int c = 0; c < dj; ++c) {
          const double* pc = P + ro + (size_t)c * mj;
          double s = 0.0;
          for (int r = 0; r < len; ++r) s += pc[r] * yr[r];
          yj[c] -= s;
        }
        q = q2;
      }
      for (int c = dj - 1; c >= 0; --c) {
        const double* pc = P + (size_t)c * mj;
        double s = yj[c];
        for (int r = c + 1; r < dj; ++r) s -= pc[r] * yj[r];
        yj[c] = s / pc[c];
      }
    }
    for (int j = 0; j < nb_; ++j) {
      double* d = x + obase_[perm_[j]];
      for (int l = 0; l < dim_[j]; ++l) d[l] = y[pbase_[j] + l];
    }
  }

  // Blocks (bi[k], bj[k]) of A^-1 (LinearSolverCholmod::solvePattern -> MarginalCovarianceCholesky in the reference,
  // linear_solver_cholmod.h:114-165): block k is written row major (dim(bi) x dim(bj)) at out + offsets[k].  One pair of
  // triangular solves per distinct scalar column asked for -- meant for the handful of calibration parameters
  // Calibrator.recoverCovariance reads (calibrator.py:236-249), not for the whole matrix.
  void covarianceBlocks(int n_pairs, const int* bi, const int* bj, const long long* offsets, double* out) const {
    if (!factored_) throw std::logic_error("covarianceBlocks before a successful factorize");
    std::vector<std::vector<int>> byCol(nb_);
    for (int k = 0; k < n_pairs; ++k) {
      if (bi[k] < 0 || bi[k] >= nb_ || bj[k] < 0 || bj[k] >= nb_) throw std::invalid_argument("block index out of range");
      byCol[bj[k]].push_back(k);
    }
    std::vector<double> rhs(n_), x(n_);
    for (int J = 0; J < nb_; ++J) {
      if (byCol[J].empty()) continue;
      const int dJ = obase_[J + 1] - obase_[J];
      for (int c = 0; c < dJ; ++c) {
        std::fill(rhs.begin(), rhs.end(), 0.0);
        rhs[obase_[J] + c] = 1.0;
        solve(rhs.data(), x.data());
        for (int k : byCol[J]) {
          const int I = bi[k], dI = obase_[I + 1] - obase_[I];
          for (int r = 0; r < dI; ++r) out[offsets[k] + (long long)r * dJ + c] = x[obase_[I] + r];
        }
      }
    }
  }

  const Stats& stats() const { return stats_; }
  const std::vector<int>& blockPermutation() const { return perm_; }   // position -> original block
  bool analyzed() const { return analyzed_; }

 private:
  // P (len x DJ, leading dimension ldp) -= T[src : src + len, :] (len x DK, leading dimension ld) * T[0:DJ, :]^T.  This
  // update is where the factorisation spends its flops; with the block dimensions known at compile time the DJ x DK
  // coefficients and the DK values of a row stay in registers and the loop over the rows vectorises (DK loads, DJ
  // load-subtract-stores per DK * DJ multiply-adds).
  template <int DK, int DJ>
  static void panelUpdateFixed(const double* __restrict__ T, int ld, int src, int len, double* __restrict__ P, int ldp) {
    double B[DJ][DK];
    for (int cj = 0; cj < DJ; ++cj)
      for (int ck = 0; ck < DK; ++ck) B[cj][ck] = T[cj + (size_t)ck * ld];
    const double* __restrict__ S = T + src;
#pragma GCC ivdep
    for (int r = 0; r < len; ++r) {
      double a[DK];
      for (int ck = 0; ck < DK; ++ck) a[ck] = S[r + (size_t)ck * ld];
      for (int cj = 0; cj < DJ; ++cj) {
        double s = a[0] * B[cj][0];
        for (int ck = 1; ck < DK; ++ck) s += a[ck] * B[cj][ck];
        P[r + (size_t)cj * ldp] -= s;
      }
    }
  }
  template <int DK>
  static bool panelUpdateDj(const double* T, int ld, int src, int len, int dj, double* P, int ldp) {
    switch (dj) {
      case 1: panelUpdateFixed<DK, 1>(T, ld, src, len, P, ldp); return true;
      case 2: panelUpdateFixed<DK, 2>(T, ld, src, len, P, ldp); return true;
      case 3: panelUpdateFixed<DK, 3>(T, ld, src, len, P, ldp); return true;
      case 4: panelUpdateFixed<DK, 4>(T, ld, src, len, P, ldp); return true;
      case 5: panelUpdateFixed<DK, 5>(T, ld, src, len, P, ldp); return true;
      case 6: panelUpdateFixed<DK, 6>(T, ld, src, len, P, ldp); return true;
      default: return false;
    }
  }
  static void panelUpdate(const double* T, int ld, int src, int len, int dk, int dj, double* P, int ldp, double* tmp) {
    bool done = false;
    switch (dk) {
      case 1: done = panelUpdateDj<1>(T, ld, src, len, dj, P, ldp); break;
      case 2: done = panelUpdateDj<2>(T, ld, src, len, dj, P, ldp); break;
      case 3: done = panelUpdateDj<3>(T, ld, src, len, dj, P, ldp); break;
      case 4: done = panelUpdateDj<4>(T, ld, src, len, dj, P, ldp); break;
      case 5: done = panelUpdateDj<5>(T, ld, src, len, dj, P, ldp); break;
      case 6: done = panelUpdateDj<6>(T, ld, src, len, dj, P, ldp); break;
      default: break;
    }
    if (done) return;
    for (int cj = 0; cj < dj; ++cj) {     // any other block dimension: column by column through tmp
      {
        const double b = T[cj];
        for (int r = 0; r < len; ++r) tmp[r] = T[src + r] * b;
      }
      for (int ck = 1; ck < dk; ++ck) {
        const double* a = T + (size_t)ck * ld;
        const double b = a[cj];
        for (int r = 0; r < len; ++r) tmp[r] += a[src + r] * b;
      }
      double* pd = P + (size_t)cj * ldp;
      for (int r = 0; r < len; ++r) pd[r] -= tmp[r];
    }
  }

  // Minimum (weighted external) degree on the quotient graph: eliminated blocks become elements whose member lists stand
  // for the cliques they created; degrees of the members of a new element are recomputed exactly.
  void minimumDegree(const std::vector<std::vector<int>>& adj, const std::vector<int>& w) {
    const int nb = nb_;
    enum { VAR = 0, ELEM = 1, DENSE = 2, ABSORBED = 3 };
    std::vector<char> status(nb, VAR);
    const double dense_limit = std::max(16.0, 10.0 * std::sqrt((double)nb));
    std::vector<int> dense;
    for (int v = 0; v < nb; ++v)
      if ((double)adj[v].size() > dense_limit) { status[v] = DENSE; dense.push_back(v); }
    stats_.n_dense_blocks = (int)dense.size();
    std::vector<std::vector<int>> vadj(nb), eadj(nb), emem(nb);
    std::vector<long long> deg(nb, 0);
    std::set<std::pair<long long, int>> pq;
    for (int v = 0; v < nb; ++v) {
      if (status[v] != VAR) continue;
      for (int u : adj[v]) if (status[u] == VAR) { vadj[v].push_back(u); deg[v] += w[u]; }
      pq.insert({deg[v], v});
    }
    std::vector<int> mark(nb, -1), mark2(nb, -1), Lp;
    int stamp = 0, stamp2 = 0;
    perm_.clear();
    perm_.reserve(nb);
    while (!pq.empty()) {
      const int p = pq.begin()->second;
      pq.erase(pq.begin());
      ++stamp;
      Lp.clear();
      mark[p] = stamp;
      for (int u : vadj[p]) if (status[u] == VAR && mark[u] != stamp) { mark[u] = stamp; Lp.push_back(u); }
      for (int e : eadj[p]) {
        if (status[e] != ELEM) continue;
        for (int u : emem[e]) if (status[u] == VAR && mark[u] != stamp) { mark[u] = stamp; Lp.push_back(u); }
        status[e] = ABSORBED;
        std::vector<int>().swap(emem[e]);
      }
      status[p] = ELEM;
      perm_.push_back(p);
      std::vector<int>().swap(vadj[p]);
      std::vector<int>().swap(eadj[p]);
      for (int u : Lp) {
        auto& va = vadj[u];
        va.erase(std::remove_if(va.begin(), va.end(), [&](int x) { return status[x] != VAR || mark[x] == stamp; }), va.end());
        auto& ea = eadj[u];
        ea.erase(std::remove_if(ea.begin(), ea.end(), [&](int e) { return status[e] != ELEM; }), ea.end());
        ea.push_back(p);
      }
      emem[p] = Lp;
      for (int u : Lp) {
        ++stamp2;
        mark2[u] = stamp2;
        long long d = 0;
        for (int x : vadj[u]) if (mark2[x] != stamp2) { mark2[x] = stamp2; d += w[x]; }
        for (int e : eadj[u])
          for (int x : emem[e]) if (status[x] == VAR && mark2[x] != stamp2) { mark2[x] = stamp2; d += w[x]; }
        if (d != deg[u]) { pq.erase({deg[u], u}); deg[u] = d; pq.insert({d, u}); }
      }
    }
    std::stable_sort(dense.begin(), dense.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    for (int v : dense) perm_.push_back(v);
    iperm_.assign(nb, 0);
    for (int i = 0; i < nb; ++i) iperm_[perm_[i]] = i;
  }

  // Block structure of the factor in the new order and the panel layout.
  void symbolic(const std::vector<std::vector<int>>& adj, const std::vector<int>& w) {
    const int nb = nb_;
    dim_.resize(nb); pbase_.resize(nb + 1);
    pbase_[0] = 0;
    for (int j = 0; j < nb; ++j) { dim_[j] = w[perm_[j]]; pbase_[j + 1] = pbase_[j] + dim_[j]; }
    std::vector<std::vector<int>> st(nb), children(nb);
    std::vector<int> mark(nb, -1);
    for (int j = 0; j < nb; ++j) {
      auto& s = st[j];
      mark[j] = j;
      for (int u : adj[perm_[j]]) { int i = iperm_[u]; if (i > j && mark[i] != j) { mark[i] = j; s.push_back(i); } }
      for (int c : children[j]) {
        for (int i : st[c]) if (mark[i] != j) { mark[i] = j; s.push_back(i); }
      }
      std::sort(s.begin(), s.end());
      if (!s.empty()) children[s.front()].push_back(j);
    }
    colptr_.assign(nb + 1, 0); valptr_.assign(nb + 1, 0); m_.assign(nb, 0);
    for (int j = 0; j < nb; ++j) colptr_[j + 1] = colptr_[j] + 1 + (long long)st[j].size();
    rowblk_.resize(colptr_[nb]); rowoff_.resize(colptr_[nb]);
    max_m_ = 0; max_d_ = 0;
    stats_.flops = 0;
    for (int j = 0; j < nb; ++j) {
      long long p = colptr_[j];
      int off = 0;
      rowblk_[p] = j; rowoff_[p] = 0; off += dim_[j]; ++p;
      for (int i : st[j]) { rowblk_[p] = i; rowoff_[p] = off; off += dim_[i]; ++p; }
      m_[j] = off;
      valptr_[j + 1] = valptr_[j] + (long long)off * dim_[j];
      max_m_ = std::max(max_m_, off); max_d_ = std::max(max_d_, dim_[j]);
      stats_.flops += (double)dim_[j] * off * off;      // ~ sum over the columns of (column count)^2
      std::vector<int>().swap(st[j]);
    }
    Lx_.assign(valptr_[nb], 0.0);
    stats_.n = n_; stats_.n_blocks = nb; stats_.nnz_factor = valptr_[nb];
  }

  int n_ = 0, nb_ = 0, max_m_ = 0, max_d_ = 0;
  bool analyzed_ = false, factored_ = false;
  std::vector<int> obase_, perm_, iperm_, dim_, pbase_, m_, rowblk_, rowoff_;
  std::vector<long long> colptr_, valptr_, amap_;
  std::vector<double> Lx_;
  Stats stats_;
};

}  // namespace multical_amd
