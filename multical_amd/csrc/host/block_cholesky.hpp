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
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

namespace multical_amd {

// threads of the host factorisation: MC_CHOLESKY_THREADS, else the hardware threads, at most 16 (a GPU node reports every
// CPU of the machine; one process per GPU should not take more than its share)
inline int hostCholeskyThreads() {
  if (const char* v = std::getenv("MC_CHOLESKY_THREADS")) return std::max(1, std::atoi(v));
  const unsigned hc = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(16u, hc ? hc : 1u));
}

class BlockSparseCholesky {
 public:
  struct Stats {
    int n = 0, n_blocks = 0, n_dense_blocks = 0;
    long long nnz_factor = 0;     // scalar entries stored in the panels
    double flops = 0;             // of one numeric factorisation
    int symbolic_passes = 0;      // 1 + the number of times the numerical pattern grew
    int n_parts = 0, n_top_blocks = 0;   // independent parts (threads) and the blocks left for the serial top
  };

  // Structure.  block_base has n_blocks + 1 entries (scalar column base of every block, last = n); NULL: every column is
  // its own block.  Cp / Ci are copied.  The symbolic work (ordering, factor structure) happens at the first factorize,
  // on the blocks that hold a non-zero value there: the assembled pattern is the reference's, which includes the
  // structurally present but numerically zero blocks of padded spline coefficients (more than half of the entries on
  // the calibration problems), and a factor of the numerical pattern costs a fraction of one of the full pattern.  A
  // block that becomes non-zero later (a time offset moved a live window over a knot) is added and the symbolic work
  // redone; blocks are never dropped again.
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
    blk_of_.assign(n, 0);
    for (int b = 0; b < nb_; ++b) {
      if (obase_[b + 1] <= obase_[b]) throw std::invalid_argument("empty block");
      for (int c = obase_[b]; c < obase_[b + 1]; ++c) blk_of_[c] = b;
    }
    Cp_.assign(Cp, Cp + n + 1);
    Ci_.assign(Ci, Ci + Cp[n]);
    for (int c = 0; c < n; ++c)
      for (int k = Cp_[c]; k < Cp_[c + 1]; ++k)
        if (Ci_[k] < 0 || Ci_[k] > c) throw std::invalid_argument("pattern is not upper triangular");
    kept_.assign(nb_, std::vector<int>());
    analyzed_ = true;
    symbolic_ = false;
    factored_ = false;
  }

  // Optional, between analyze and the first factorize: a coordinate per block along which the matrix is banded (the
  // calibration problems: the time of a spline coefficient; NaN for blocks without one).  With it the blocks are cut into
  // `threads` intervals of equal weight; the blocks with a neighbour in a later interval, the blocks without a coordinate
  // and the nearly dense ones form the top, everything else falls into independent parts that are factorised (and
  // solved) by one thread each.  The top is done by one thread from the parts' contributions.
  void setBlockCoordinates(const double* coord, int threads) {
    if (!analyzed_) throw std::logic_error("coordinates before analyze");
    coord_.assign(coord, coord + nb_);
    threads_ = std::max(1, std::min(threads, 64));
    symbolic_ = false;
  }

  // Numeric phase on values in the order of the analysed pattern.  Returns false if a pivot is not positive (or NaN).
  bool factorize(const double* Cx) {
    if (!analyzed_) throw std::logic_error("factorize before analyze");
    factored_ = false;
    const bool timing = std::getenv("MC_CHOLESKY_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    if (!symbolic_) { markKept(Cx); buildSymbolic(); }
    const auto t1 = now();
    if (!scatter(Cx)) {          // a value outside the kept blocks: grow the structure
      markKept(Cx);
      buildSymbolic();
      scatter(Cx);
    }
    const auto t2 = now();
    next_.assign(nb_, -1);
    pos_.assign(nb_, 0);
    if (part_start_.empty()) {
      Work ws(nb_, (size_t)max_m_ * max_d_, 0);
      for (int j = 0; j < nb_; ++j) {
        double* P = Lx_.data() + valptr_[j];
        applyUpdates(j, P, ws);
        if (!factorPanel(P, dim_[j], m_[j])) return false;
        link(j, ws.head);
      }
      factored_ = true;
      return true;
    }
    // the independent parts, one thread each: their own columns, then their contributions to the top's panels (into a
    // private copy: the top's columns receive updates from every part)
    const int P = (int)part_start_.size() - 1;
    const long long top0 = valptr_[top_start_];
    const size_t top_size = (size_t)(valptr_[nb_] - top0);
    std::vector<Work> work;
    work.reserve(P);
    for (int t = 0; t < P; ++t) work.emplace_back(nb_, (size_t)max_m_ * max_d_, top_size);
    std::atomic<bool> ok(true);
    auto run = [&](int t) {
      Work& wt = work[t];
      for (int j = part_start_[t]; j < part_start_[t + 1]; ++j) {
        double* Pj = Lx_.data() + valptr_[j];
        applyUpdates(j, Pj, wt);
        if (!factorPanel(Pj, dim_[j], m_[j])) { ok = false; return; }
        link(j, wt.head);
      }
      for (int j = top_start_; j < nb_; ++j)
        if (wt.head[j] != -1) { applyUpdates(j, wt.top.data() + (valptr_[j] - top0), wt); wt.touched_lo = std::min(wt.touched_lo, j); wt.touched_hi = std::max(wt.touched_hi, j); }
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < P; ++t) th.emplace_back(run, t);
      run(0);
      for (std::thread& x : th) x.join();
    }
    if (!ok) return false;
    const auto t3 = now();
    Work ws(nb_, (size_t)max_m_ * max_d_, 0);
    for (int j = top_start_; j < nb_; ++j) {
      double* Pj = Lx_.data() + valptr_[j];
      const size_t len = (size_t)m_[j] * dim_[j];
      for (int t = 0; t < P; ++t) {
        if (j < work[t].touched_lo || j > work[t].touched_hi) continue;
        const double* add = work[t].top.data() + (valptr_[j] - top0);
        for (size_t i = 0; i < len; ++i) Pj[i] += add[i];
      }
      applyUpdates(j, Pj, ws);
      if (!factorPanel(Pj, dim_[j], m_[j])) return false;
      link(j, ws.head);
    }
    if (timing)
      std::fprintf(stderr, "cholesky: symbolic %.1f ms, scatter %.1f ms, %d parts %.1f ms, top (%d blocks, %d columns) %.1f ms\n", ms(t0, t1),
                   ms(t1, t2), P, ms(t2, t3), nb_ - top_start_, n_ - pbase_[top_start_], ms(t3, now()));
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
    const int split = part_start_.empty() ? nb_ : top_start_;      // row blocks >= split belong to the top
    const int ysplit = pbase_[split];
    // L y' = y for block column j; rows in the top go to `top` (indexed from ysplit), which may be a private image
    auto forward = [&](int j, double* top) {
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
        const bool in_top = rowblk_[q] >= split;
        int len = dim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < colptr_[j + 1] && pbase_[rowblk_[q2]] == y0 + len && (rowblk_[q2] >= split) == in_top) { len += dim_[rowblk_[q2]]; ++q2; }
        double* yr = in_top ? top + (y0 - ysplit) : y.data() + y0;
        for (int c = 0; c < dj; ++c) {
          const double* pc = P + ro + (size_t)c * mj;
          const double v = yj[c];
          for (int r = 0; r < len; ++r) yr[r] -= pc[r] * v;
        }
        q = q2;
      }
    };
    auto backward = [&](int j) {                 // L^T x' = y'
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
    };
    if (part_start_.empty()) {
      for (int j = 0; j < nb_; ++j) forward(j, y.data() + ysplit);
      for (int j = nb_ - 1; j >= 0; --j) backward(j);
    } else {
      const int P = (int)part_start_.size() - 1;
      const size_t ntop = (size_t)(n_ - ysplit);
      std::vector<std::vector<double>> tops(P, std::vector<double>(ntop, 0.0));
      auto each_part = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int t = 1; t < P; ++t) th.emplace_back(fn, t);
        fn(0);
        for (std::thread& z : th) z.join();
      };
      each_part([&](int t) { for (int j = part_start_[t]; j < part_start_[t + 1]; ++j) forward(j, tops[t].data()); });
      for (int t = 0; t < P; ++t)
        for (size_t i = 0; i < ntop; ++i) y[ysplit + i] += tops[t][i];
      for (int j = top_start_; j < nb_; ++j) forward(j, y.data() + ysplit);
      for (int j = nb_ - 1; j >= top_start_; --j) backward(j);
      each_part([&](int t) { for (int j = part_start_[t + 1] - 1; j >= part_start_[t]; --j) backward(j); });
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

  // Both describe the last symbolic phase; asked for before the first factorize they describe the full pattern.
  const Stats& stats() { structureForQueries(); return stats_; }
  const std::vector<int>& blockPermutation() { structureForQueries(); return perm_; }   // position -> original block
  bool analyzed() const { return analyzed_; }

 private:
  // per-thread scratch of the numeric phase
  struct Work {
    std::vector<int> head, where;
    std::vector<double> tmp, top;
    int touched_lo, touched_hi;
    Work(int nb, size_t tmp_size, size_t top_size) : head(nb, -1), where(nb, 0), tmp(tmp_size), top(top_size, 0.0), touched_lo(nb), touched_hi(-1) {}
  };
  // Left-looking step for block column j: P (the column's panel, or a private image of it) -= the products of the block
  // columns queued in wk.head[j]; every one of them then moves on to the list of its next row block.
  void applyUpdates(int j, double* P, Work& wk) {
    const int dj = dim_[j], mj = m_[j];
    for (long long p = colptr_[j]; p < colptr_[j + 1]; ++p) wk.where[rowblk_[p]] = rowoff_[p];
    for (int k = wk.head[j], knext; k != -1; k = knext) {
      knext = next_[k];
      const long long p0 = pos_[k], pe = colptr_[k + 1];
      const int dk = dim_[k], mk = m_[k], r0 = rowoff_[p0];
      const double* T = Lx_.data() + valptr_[k] + r0;      // rows of block column k from block row j downwards
      // P[rows of k below j, :] -= T * T[0:dj, :]^T, by runs of row blocks that are contiguous in both panels (in the
      // banded part a descendant's rows map onto the target's rows as two or three runs: band, border)
      for (long long q = p0; q < pe;) {
        const int src = rowoff_[q] - r0, dst = wk.where[rowblk_[q]];
        int len = dim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < pe && wk.where[rowblk_[q2]] == dst + len) { len += dim_[rowblk_[q2]]; ++q2; }
        panelUpdate(T, mk, src, len, dk, dj, P + dst, mj, wk.tmp.data());
        q = q2;
      }
      pos_[k] = p0 + 1;
      if (p0 + 1 < pe) { const int r = rowblk_[p0 + 1]; next_[k] = wk.head[r]; wk.head[r] = k; }
    }
    wk.head[j] = -1;
  }
  // the finished block column j queues itself at its first row block below the diagonal
  void link(int j, std::vector<int>& head) {
    if (colptr_[j + 1] - colptr_[j] > 1) {
      pos_[j] = colptr_[j] + 1;
      const int r = rowblk_[pos_[j]];
      next_[j] = head[r]; head[r] = j;
    }
  }

  // panel factorisation: Cholesky of the dj x dj diagonal block and the triangular solve of the rows below it
  static bool factorPanel(double* P, int dj, int mj) {
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
    return true;
  }

  // Ordering.  Without coordinates: minimum degree.  With coordinates (a banded matrix and the position of every block
  // along the band; the calibration problems pass the END of a spline coefficient's support, so that sorting by it
  // eliminates a bias coefficient right after the last pose coefficient it is coupled to): the blocks in coordinate order,
  // which keeps the row structure of neighbouring columns nested -- the panel updates then run over long contiguous row
  // ranges -- cut into `threads_` independent parts plus a top (see setBlockCoordinates).
  void partition(const std::vector<std::vector<int>>& adj, const std::vector<int>& w) {
    part_start_.clear();
    top_start_ = 0;
    stats_.n_parts = 0; stats_.n_top_blocks = 0;
    if (coord_.empty()) { minimumDegree(adj, w); return; }
    const double dense_limit = denseLimit();
    std::vector<int> order, loose, dense;
    for (int v = 0; v < nb_; ++v) {
      if ((double)adj[v].size() > dense_limit) dense.push_back(v);
      else if (std::isnan(coord_[v])) loose.push_back(v);
      else order.push_back(v);
    }
    stats_.n_dense_blocks = (int)dense.size();
    auto byCoord = [&](int a2, int b2) { return coord_[a2] < coord_[b2] || (coord_[a2] == coord_[b2] && a2 < b2); };
    std::sort(order.begin(), order.end(), byCoord);
    std::stable_sort(dense.begin(), dense.end(), [&](int a2, int b2) { return adj[a2].size() < adj[b2].size(); });
    int P = threads_;
    if ((int)order.size() < 64 * P) P = 1;
    const int TOP = P;
    std::vector<int> stage(nb_, TOP);
    if (P > 1) {
      // weight ~ the work of eliminating the block: dim * (dims of its neighbourhood)^2
      std::vector<double> cost(order.size());
      double total = 0;
      for (size_t i = 0; i < order.size(); ++i) {
        double m = w[order[i]];
        for (int u : adj[order[i]]) m += w[u];
        cost[i] = w[order[i]] * m * m;
        total += cost[i];
      }
      double acc = 0;
      for (size_t i = 0; i < order.size(); ++i) {
        stage[order[i]] = std::min(P - 1, (int)(acc / total * P));
        acc += cost[i];
      }
      // The edges between two different parts have to lose an endpoint to the top.  Greedy vertex cover: the block with
      // the most uncovered crossing edges goes first (a bias-spline coefficient is coupled to every pose coefficient of
      // its support, so lifting it covers dozens of edges that would otherwise lift as many pose blocks).
      std::vector<int> cross(nb_, 0);
      std::set<std::pair<int, int>> byCount;     // (-count, block)
      for (int v : order) {
        for (int u : adj[v]) if (stage[u] != TOP && stage[u] != stage[v]) ++cross[v];
        if (cross[v]) byCount.insert({-cross[v], v});
      }
      while (!byCount.empty()) {
        const int v = byCount.begin()->second;
        byCount.erase(byCount.begin());
        const int sv = stage[v];
        stage[v] = TOP;
        cross[v] = 0;
        for (int u : adj[v]) {
          if (stage[u] == TOP || stage[u] == sv || cross[u] == 0) continue;
          byCount.erase({-cross[u], u});
          if (--cross[u] > 0) byCount.insert({-cross[u], u});
        }
      }
    } else {
      for (int v : order) stage[v] = 0;
    }
    perm_.clear();
    perm_.reserve(nb_);
    if (P > 1) part_start_.assign(P + 1, 0);
    for (int p2 = 0; p2 < P; ++p2) {
      for (int v : order) if (stage[v] == p2) perm_.push_back(v);
      if (P > 1) part_start_[p2 + 1] = (int)perm_.size();
    }
    top_start_ = P > 1 ? (int)perm_.size() : 0;
    if (P > 1) for (int v : order) if (stage[v] == TOP) perm_.push_back(v);     // the separators, in coordinate order
    for (int v : loose) perm_.push_back(v);
    for (int v : dense) perm_.push_back(v);
    iperm_.assign(nb_, 0);
    for (int i = 0; i < nb_; ++i) iperm_[perm_[i]] = i;
  }

  // the parts must be independent: every row block of a part's column lies in the same part or in the top
  void checkParts() {
    if (part_start_.empty()) return;
    const int P = (int)part_start_.size() - 1;
    for (int p2 = 0; p2 < P; ++p2)
      for (int j = part_start_[p2]; j < part_start_[p2 + 1]; ++j)
        for (long long q = colptr_[j] + 1; q < colptr_[j + 1]; ++q) {
          const int r = rowblk_[q];
          if (r < top_start_ && r >= part_start_[p2 + 1]) { part_start_.clear(); top_start_ = 0; return; }
        }
    stats_.n_parts = P;
    stats_.n_top_blocks = nb_ - top_start_;
  }

  void structureForQueries() {
    if (!analyzed_ || symbolic_) return;
    std::vector<int> stamp(nb_, -1);
    for (int J = 0; J < nb_; ++J)
      for (int c = obase_[J]; c < obase_[J + 1]; ++c)
        for (int k = Cp_[c]; k < Cp_[c + 1]; ++k) {
          const int I = blk_of_[Ci_[k]];
          if (I != J && stamp[I] != J) { stamp[I] = J; kept_[J].push_back(I); }
        }
    buildSymbolic();
  }

  // Blocks with a non-zero value join the kept set (kept_[J] = row blocks I < J of block column J, unsorted).
  void markKept(const double* Cx) {
    std::vector<int> stamp(nb_, -1);
    for (int J = 0; J < nb_; ++J) {
      std::vector<int>& kj = kept_[J];
      for (int I : kj) stamp[I] = J;
      for (int c = obase_[J]; c < obase_[J + 1]; ++c)
        for (int k = Cp_[c]; k < Cp_[c + 1]; ++k) {
          if (Cx[k] == 0.0) continue;
          const int I = blk_of_[Ci_[k]];
          if (I != J && stamp[I] != J) { stamp[I] = J; kj.push_back(I); }
        }
    }
  }

  // Ordering, factor structure and the map from the caller's value array into the panels, for the kept blocks.
  void buildSymbolic() {
    std::vector<std::vector<int>> adj(nb_);
    for (int J = 0; J < nb_; ++J)
      for (int I : kept_[J]) { adj[J].push_back(I); adj[I].push_back(J); }
    std::vector<int> w(nb_);
    for (int b = 0; b < nb_; ++b) w[b] = obase_[b + 1] - obase_[b];
    partition(adj, w);         // fills perm_ / iperm_ (and the independent parts, if the blocks have coordinates)
    symbolic(adj, w);          // fills the panel layout
    checkParts();
    amap_.assign(Cp_[n_], -1);
    std::vector<char> keep(nb_, 0);
    for (int J = 0; J < nb_; ++J) {
      for (int I : kept_[J]) keep[I] = 1;
      keep[J] = 1;
      const int pJ = iperm_[J];
      int lastI = -1; long long base = 0; int ld = 0; bool transposed = false;
      for (int c = obase_[J]; c < obase_[J + 1]; ++c) {
        const int lj = c - obase_[J];
        for (int k = Cp_[c]; k < Cp_[c + 1]; ++k) {
          const int i = Ci_[k], I = blk_of_[i], li = i - obase_[I];
          if (!keep[I]) continue;
          if (I == J) {                    // diagonal block: keep its lower triangle (r >= c)
            amap_[k] = (int)(valptr_[pJ] + lj + (long long)li * m_[pJ]);
            continue;
          }
          if (I != lastI) {
            lastI = I;
            const int pI = iperm_[I];
            const int col = std::min(pI, pJ), row = std::max(pI, pJ);
            const int* rb = rowblk_.data() + colptr_[col];
            const int* re = rowblk_.data() + colptr_[col + 1];
            const int* it = std::lower_bound(rb, re, row);
            if (it == re || *it != row) throw std::logic_error("symbolic factor misses an entry of the matrix");
            ld = m_[col];
            base = valptr_[col] + rowoff_[colptr_[col] + (it - rb)];
            transposed = (pI < pJ);        // entry (i, j) of the upper triangle sits at (row j, column i) of the lower factor
          }
          amap_[k] = (int)(transposed ? base + lj + (long long)li * ld : base + li + (long long)lj * ld);
        }
      }
      for (int I : kept_[J]) keep[I] = 0;
      keep[J] = 0;
    }
    symbolic_ = true;
    ++stats_.symbolic_passes;
  }

  // values -> panels; false if a non-zero value falls outside the kept blocks
  bool scatter(const double* Cx) {
    const size_t nnz = amap_.size(), nl = Lx_.size();
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads_, nnz / 200000 + 1));   // small systems: not worth the threads
    std::vector<char> inside(T, 1);
    auto zero = [&](int t) { std::fill(Lx_.begin() + nl * t / T, Lx_.begin() + nl * (t + 1) / T, 0.0); };
    auto fill = [&](int t) {
      bool in = true;
      for (size_t k = nnz * t / T, ke = nnz * (t + 1) / T; k < ke; ++k) {
        const int a = amap_[k];
        if (a >= 0) Lx_[a] = Cx[k];
        else if (Cx[k] != 0.0) in = false;
      }
      inside[t] = in ? 1 : 0;
    };
    auto each = [&](auto&& fn) {
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
      fn(0);
      for (std::thread& z : th) z.join();
    };
    each(zero);
    each(fill);
    for (int t = 0; t < T; ++t) if (!inside[t]) return false;
    return true;
  }

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

  double denseLimit() const { return std::max(16.0, 10.0 * std::sqrt((double)nb_)); }

  // Minimum (weighted external) degree on the quotient graph: eliminated blocks become elements whose member lists stand
  // for the cliques they created; degrees of the members of a new element are recomputed exactly.
  // `stage` (optional, one entry per block, values 0 .. n_stages - 1): the blocks of stage s are eliminated -- in
  // minimum-degree order among themselves -- before those of stage s + 1.  stage_end (optional) receives, per stage, the
  // number of blocks ordered when the stage was finished.
  void minimumDegree(const std::vector<std::vector<int>>& adj, const std::vector<int>& w, const std::vector<int>* stage = nullptr,
                     int n_stages = 1, std::vector<int>* stage_end = nullptr) {
    const int nb = nb_;
    enum { VAR = 0, ELEM = 1, DENSE = 2, ABSORBED = 3 };
    std::vector<char> status(nb, VAR);
    const double dense_limit = denseLimit();
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
    }
    std::vector<int> mark(nb, -1), mark2(nb, -1), Lp;
    int stamp = 0, stamp2 = 0;
    perm_.clear();
    perm_.reserve(nb);
    if (stage_end) stage_end->assign(n_stages, 0);
    for (int cur = 0; cur < n_stages; ++cur) {
    for (int v = 0; v < nb; ++v)
      if (status[v] == VAR && (!stage || (*stage)[v] == cur)) pq.insert({deg[v], v});
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
        if (d != deg[u]) {
          const bool queued = !stage || (*stage)[u] == cur;
          if (queued) pq.erase({deg[u], u});
          deg[u] = d;
          if (queued) pq.insert({d, u});
        }
      }
    }
    if (stage_end) (*stage_end)[cur] = (int)perm_.size();
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
    if (valptr_[nb] > (long long)0x7fffffff) throw std::overflow_error("factor exceeds 2^31 entries");
    Lx_.assign(valptr_[nb], 0.0);
    stats_.n = n_; stats_.n_blocks = nb; stats_.nnz_factor = valptr_[nb];
  }

  int n_ = 0, nb_ = 0, max_m_ = 0, max_d_ = 0;
  bool analyzed_ = false, symbolic_ = false, factored_ = false;
  std::vector<int> Cp_, Ci_, blk_of_;
  std::vector<double> coord_;
  std::vector<int> next_;
  std::vector<long long> pos_;
  int threads_ = 1, top_start_ = 0;
  std::vector<int> part_start_;
  std::vector<std::vector<int>> kept_;
  std::vector<int> obase_, perm_, iperm_, dim_, pbase_, m_, rowblk_, rowoff_;
  std::vector<long long> colptr_, valptr_;
  std::vector<int> amap_;   // value index -> panel entry (the panels hold fewer than 2^31 values, checked in symbolic)
  std::vector<double> Lx_;
  Stats stats_;
};

}  // namespace multical_amd
