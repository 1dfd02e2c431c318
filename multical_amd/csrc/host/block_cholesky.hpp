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
#include <memory>
#include <set>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif

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
    int n_panels = 0;             // block columns of the factor (neighbouring blocks with one structure share a panel)
    int n_parts = 0, n_top_blocks = 0;   // independent parts (threads) and the blocks left for the serial top
  };

  // Structure.  block_base has n_blocks + 1 entries (scalar column base of every block, last = n); NULL: every column is
  // its own block.  Cp / Ci are copied.  The symbolic work (ordering, factor structure) happens at the first factorize,
  // on the blocks that hold a non-zero value there: the assembled pattern is the reference's, which includes the
  // structurally present but numerically zero blocks of padded spline coefficients (more than half of the entries on
  // the calibration problems), and a factor of the numerical pattern costs a fraction of one of the full pattern.  A
  // block that becomes non-zero later (a time offset moved a live window over a knot) usually lies inside the fill of the
  // factor and has its place already; one that does not is added and the symbolic work redone.  Blocks are never dropped
  // again.
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
    if (!symbolic_) {
      markKept(Cx);
      if (timing) std::fprintf(stderr, "cholesky: non-zero blocks marked in %.1f ms\n", ms(t0, now()));
      buildSymbolic();
    }
    const auto t1 = now();
    if (!checkDropped(Cx)) {     // a value outside the kept blocks: grow the structure
      markKept(Cx);
      buildSymbolic();
    }
    const auto t2 = now();
    next_.assign(np_, -1);
    pos_.assign(np_, 0);
    if (part_start_.empty()) {
      Work ws(np_, nb_, 0);
      for (int j = 0; j < np_; ++j) {
        double* P = Lx_.data() + valptr_[j];
        preparePanel(j, Cx);
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
    const long long top0 = valptr_[top_start_panel_];
    const size_t top_size = (size_t)(valptr_[np_] - top0);
    // per-part scratch, kept between factorisations (first-touch page faults of 16 private top images cost more than
    // the factorisation of a part); each part clears its own image
    std::vector<Work>& work = work_;
    if ((int)work.size() != P || work[0].top.size() != top_size || (int)work[0].head.size() != np_) {
      work.clear();
      work.reserve(P);
      for (int t = 0; t < P; ++t) work.emplace_back(np_, nb_, top_size);
    }
    std::atomic<bool> ok(true);
    const auto t2b = now();
    std::vector<double> part_ms(P, 0.0);
    auto run = [&](int t) {
      const auto r0 = now();
      Work& wt = work[t];
      for (int j = top_start_panel_; j < np_; ++j)
        if (wt.hit[j]) { std::fill(wt.top.begin() + (valptr_[j] - top0), wt.top.begin() + (valptr_[j + 1] - top0), 0.0); wt.hit[j] = 0; }
      std::fill(wt.head.begin(), wt.head.end(), -1);
      for (int j = top_start_panel_ + t; j < np_; j += P) preparePanel(j, Cx);     // the top's panels: shared out
      for (int j = part_start_[t]; j < part_start_[t + 1]; ++j) {
        double* Pj = Lx_.data() + valptr_[j];
        preparePanel(j, Cx);
        applyUpdates(j, Pj, wt);
        if (!factorPanel(Pj, dim_[j], m_[j])) { ok = false; return; }
        link(j, wt.head);
        pushToTop(j, wt, top0);
      }
      part_ms[t] = ms(r0, now());
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < P; ++t) th.emplace_back(run, t);
      run(0);
      for (std::thread& x : th) x.join();
    }
    if (!ok) return false;
    const auto t3 = now();
    // the parts' images of the top's panels are summed by all threads (a streaming pass over up to P copies of the top:
    // on one thread it took as long as the top's factorisation itself), the panels shared out like their preparation
    {
      auto sum = [&](int t) {
        for (int j = top_start_panel_ + t; j < np_; j += P) {
          double* Pj = Lx_.data() + valptr_[j];
          const size_t len = (size_t)m_[j] * dim_[j];
          for (int u = 0; u < P; ++u) {
            if (!work[u].hit[j]) continue;
            const double* add = work[u].top.data() + (valptr_[j] - top0);
            for (size_t i = 0; i < len; ++i) Pj[i] += add[i];
          }
        }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < P; ++t) th.emplace_back(sum, t);
      sum(0);
      for (std::thread& x : th) x.join();
    }
    const auto t3b = now();
    Work ws(np_, nb_, 0);
    for (int j = top_start_panel_; j < np_; ++j) {
      double* Pj = Lx_.data() + valptr_[j];
      applyUpdates(j, Pj, ws);
      if (!factorPanel(Pj, dim_[j], m_[j])) return false;
      link(j, ws.head);
    }
    if (timing) {
      std::fprintf(stderr, "cholesky: set-up of the parts %.1f ms (top image %.1f MB per part); per part [ms]:", ms(t2, t2b), top_size * 8e-6);
      for (int t = 0; t < P; ++t) std::fprintf(stderr, " %.0f", part_ms[t]);
      std::fprintf(stderr, "\n");
    }
    if (timing)
      std::fprintf(stderr, "cholesky: %d blocks in %d panels, symbolic %.1f ms, value check %.1f ms, %d parts %.1f ms, images summed %.1f ms, top (%d panels, %d columns) %.1f ms\n", nb_, np_, ms(t0, t1),
                   ms(t1, t2), P, ms(t2, t3), ms(t3, t3b), np_ - top_start_panel_, n_ - pbase_[top_start_panel_], ms(t3b, now()));
    factored_ = true;
    return true;
  }

  // x = A^-1 b with the last factorisation (b and x may alias)
  void solve(const double* b, double* x) const {
    if (!factored_) throw std::logic_error("solve before a successful factorize");
    std::vector<double> y(n_);
    for (int J = 0; J < nb_; ++J) {
      const double* s = b + obase_[J];
      double* d = y.data() + ybase_[iperm_[J]];
      for (int l = 0, dl = obase_[J + 1] - obase_[J]; l < dl; ++l) d[l] = s[l];
    }
    const int split = part_start_.empty() ? np_ : top_start_panel_;      // row blocks >= split belong to the top
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
      for (long long q = colptr_[j] + (panel_start_[j + 1] - panel_start_[j]); q < colptr_[j + 1];) {
        const int ro = rowoff_[q], y0 = ybase_[rowblk_[q]];
        const bool in_top = pos_panel_[rowblk_[q]] >= split;
        int len = pdim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < colptr_[j + 1] && ybase_[rowblk_[q2]] == y0 + len && (pos_panel_[rowblk_[q2]] >= split) == in_top) { len += pdim_[rowblk_[q2]]; ++q2; }
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
      for (long long q = colptr_[j] + (panel_start_[j + 1] - panel_start_[j]); q < colptr_[j + 1];) {
        const int ro = rowoff_[q], y0 = ybase_[rowblk_[q]];
        int len = pdim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < colptr_[j + 1] && ybase_[rowblk_[q2]] == y0 + len) { len += pdim_[rowblk_[q2]]; ++q2; }
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
      for (int j = 0; j < np_; ++j) forward(j, y.data() + ysplit);
      for (int j = np_ - 1; j >= 0; --j) backward(j);
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
      for (int j = top_start_panel_; j < np_; ++j) forward(j, y.data() + ysplit);
      for (int j = np_ - 1; j >= top_start_panel_; --j) backward(j);
      each_part([&](int t) { for (int j = part_start_[t + 1] - 1; j >= part_start_[t]; --j) backward(j); });
    }
    for (int J = 0; J < nb_; ++J) {
      const double* s = y.data() + ybase_[iperm_[J]];
      double* d = x + obase_[J];
      for (int l = 0, dl = obase_[J + 1] - obase_[J]; l < dl; ++l) d[l] = s[l];
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
    std::vector<double> top;
    std::vector<char> hit;      // per panel: the part's image of it holds something (a part reaches the two separators next
                                // to it and the dense blocks, a fraction of the top)
    Work(int np, int nb, size_t top_size) : head(np, -1), where(nb, 0), top(top_size, 0.0), hit(np, 0) {}
  };
  // Left-looking step for panel j: P (the panel, or a private image of it) -= the products of the panels queued in
  // wk.head[j]; every one of them then moves on to the list of the panel of its next row block.  A queued panel k has
  // rows in some of j's member blocks (usually all of them); each run of those that is contiguous in j's columns is one
  // column tile of the update, applied to k's rows from that run downwards.
  void applyUpdates(int j, double* P, Work& wk) {
    for (long long p = colptr_[j]; p < colptr_[j + 1]; ++p) wk.where[rowblk_[p]] = rowoff_[p];
    for (int k = wk.head[j], knext; k != -1; k = knext) {
      knext = next_[k];
      const long long pe = colptr_[k + 1];
      const long long pj = updateFrom(k, pos_[k], j, P, wk.where.data());
      pos_[k] = pj;
      if (pj < pe) {
        const int r = pos_panel_[rowblk_[pj]];
        if (!(k < top_start_panel_ && r >= top_start_panel_)) { next_[k] = wk.head[r]; wk.head[r] = k; }   // a part's panel reaches the top by pushToTop
      }
    }
    wk.head[j] = -1;
  }
  // P (panel j or an image of it; where[position] = row offset in it) -= the product of panel k's rows from row-list
  // entry p0 on with its rows inside panel j, which start at p0.  Returns the entry of k's first row beyond panel j.
  long long updateFrom(int k, long long p0, int j, double* P, const int* where) const {
    const int mj = m_[j], dk = dim_[k], mk = m_[k];
    const long long pe = colptr_[k + 1];
    long long pj = p0;                                   // [p0, pj): k's rows inside panel j
    while (pj < pe && pos_panel_[rowblk_[pj]] == j) ++pj;
    for (long long qa = p0; qa < pj;) {
      const int c0 = where[rowblk_[qa]];
      int wd = pdim_[rowblk_[qa]];
      long long qb = qa + 1;
      while (qb < pj && where[rowblk_[qb]] == c0 + wd) { wd += pdim_[rowblk_[qb]]; ++qb; }
      const int r0 = rowoff_[qa];
      const double* T = Lx_.data() + valptr_[k] + r0;    // rows of panel k from this run downwards; its first wd rows are B
      double* Pc = P + (size_t)c0 * mj;
      // by runs of row blocks that are contiguous in both panels
      for (long long q = qa; q < pe;) {
        const int src = rowoff_[q] - r0, dst = where[rowblk_[q]];
        int len = pdim_[rowblk_[q]];
        long long q2 = q + 1;
        while (q2 < pe && where[rowblk_[q2]] == dst + len) { len += pdim_[rowblk_[q2]]; ++q2; }
        panelUpdate(T, mk, src, len, dk, wd, Pc + dst, mj);
        q = q2;
      }
      qa = qb;
    }
    return pj;
  }
  // A part's finished panel k adds its products to the part's image of the top's panels right away, while it is in
  // cache (left-looking, every top panel would stream the part's whole factor past it once more).
  void pushToTop(int k, Work& wk, long long top0) {
    const long long pe = colptr_[k + 1];
    long long q = pe;
    while (q > colptr_[k] && rowblk_[q - 1] >= top_start_) --q;      // rows are sorted: the top's are at the end
    while (q < pe) {
      const int j = pos_panel_[rowblk_[q]];
      wk.hit[j] = 1;
      q = updateFrom(k, q, j, wk.top.data() + (valptr_[j] - top0), top_where_.data() + (size_t)(j - top_start_panel_) * (nb_ - top_start_) - top_start_);
    }
  }
  // the finished panel j queues itself at the panel of its first row block below the diagonal block
  void link(int j, std::vector<int>& head) {
    const long long q = colptr_[j] + (panel_start_[j + 1] - panel_start_[j]);
    if (q < colptr_[j + 1]) {
      pos_[j] = q;
      const int r = pos_panel_[rowblk_[q]];
      if (j < top_start_panel_ && r >= top_start_panel_) return;
      next_[j] = head[r]; head[r] = j;
    }
  }

  // panel factorisation: Cholesky of the dj x dj diagonal block and the triangular solve of the rows below it, by column
  // tiles of 6: a tile first receives the products of the panel's earlier columns through the update kernel, then its
  // own columns are finished one by one
  static bool factorPanel(double* P, int dj, int mj) {
    for (int c0 = 0; c0 < dj; c0 += 6) {
      const int c1 = std::min(dj, c0 + 6);
      if (c0 > 0) panelUpdate(P + c0, mj, 0, mj - c0, c0, c1 - c0, P + c0 + (size_t)c0 * mj, mj);
      for (int c = c0; c < c1; ++c) {
        double* pc = P + (size_t)c * mj;
        for (int k = c0; k < c; ++k) {
          const double* pk = P + (size_t)k * mj;
          const double b = pk[c];
          for (int r = c; r < mj; ++r) pc[r] -= pk[r] * b;
        }
        const double d = pc[c];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double inv = 1.0 / std::sqrt(d);
        for (int r = c; r < mj; ++r) pc[r] *= inv;
      }
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
      // every part but the first carries the rows of the separator on its left through all of its columns, which about
      // doubles its work: the first part gets two shares
      double acc = 0;
      for (size_t i = 0; i < order.size(); ++i) {
        stage[order[i]] = std::min(P - 1, std::max(0, (int)(acc / total * (P + 1)) - 1));
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
        for (long long q = colptr_[j]; q < colptr_[j + 1]; ++q) {
          const int r = pos_panel_[rowblk_[q]];
          if (r < top_start_panel_ && r >= part_start_[p2 + 1]) { part_start_.clear(); top_start_ = 0; top_start_panel_ = 0; return; }
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

  // fn(t, T) on T threads (T = threads_ when the work is worth it, else 1)
  int threadsFor(size_t work) const { return (int)std::max<size_t>(1, std::min<size_t>((size_t)threads_, work / 200000 + 1)); }
  template <class F> void eachThread(size_t work, F&& fn) const {
    const int T = threadsFor(work);
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(fn, t, T);
    fn(0, T);
    for (std::thread& z : th) z.join();
  }
  // block columns [lo, hi) of thread t of T with about equal numbers of stored values
  std::pair<int, int> blockRange(int t, int T) const {
    auto cut = [&](int q) {
      if (q <= 0) return 0;
      if (q >= T) return nb_;
      const long long target = (long long)Cp_[n_] * q / T;
      const int c = (int)(std::lower_bound(Cp_.begin(), Cp_.end(), (int)target) - Cp_.begin());
      return c >= n_ ? nb_ : blk_of_[c];
    };
    return {cut(t), cut(t + 1)};
  }

  // Blocks with a non-zero value join the kept set (kept_[J] = row blocks I < J of block column J, unsorted).
  void markKept(const double* Cx) {
    eachThread((size_t)Cp_[n_], [&](int t, int T) {
      std::vector<int> stamp(nb_, -1);
      const std::pair<int, int> range = blockRange(t, T);
      for (int J = range.first; J < range.second; ++J) {
        std::vector<int>& kj = kept_[J];
        for (int I : kj) stamp[I] = J;
        for (int c = obase_[J]; c < obase_[J + 1]; ++c)
          for (int k = Cp_[c]; k < Cp_[c + 1]; ++k) {
            if (Cx[k] == 0.0) continue;
            const int I = blk_of_[Ci_[k]];
            if (I != J && stamp[I] != J) { stamp[I] = J; kj.push_back(I); }
          }
      }
    });
  }

  // emit(value index, panel, offset in the panel) for every stored value of block columns [Jlo, Jhi); panel = -1 for a value
  // without a place in the factor.  false if a kept block has no place (the symbolic structure would be wrong).
  template <class E> bool mapBlockColumns(int Jlo, int Jhi, E&& emit) const {
    bool ok = true;
    std::vector<char> keep(nb_, 0);
    for (int J = Jlo; J < Jhi; ++J) {
      for (int I : kept_[J]) keep[I] = 1;
      keep[J] = 1;
      const int gJ = panel_of_[J], oJ = panel_off_[J], pJ = iperm_[J];
      int lastI = -1, lastCol = -1; long long base = 0; int ld = 0, coff = 0; bool transposed = false, found = false;
      for (int c = obase_[J]; c < obase_[J + 1]; ++c) {
        const int lj = c - obase_[J];
        for (int k = Cp_[c]; k < Cp_[c + 1]; ++k) {
          const int i = Ci_[k], I = blk_of_[i];
          const int li = i - obase_[I];
          if (panel_of_[I] == gJ) {        // inside the panel's diagonal block: its lower triangle
            const int a = li + panel_off_[I], b = lj + oJ;
            emit(k, gJ, (int)(std::max(a, b) + (long long)std::min(a, b) * m_[gJ]));
            continue;
          }
          if (I != lastI) {
            lastI = I;
            const int pI = iperm_[I];
            transposed = (pI < pJ);        // entry (i, j) of the upper triangle sits at (row j, column i) of the lower factor
            const int colBlock = transposed ? I : J, rowPos = transposed ? pJ : pI;
            const int col = panel_of_[colBlock];
            const long long first = colptr_[col] + (panel_start_[col + 1] - panel_start_[col]);
            const int* rb = rowblk_.data() + first;
            const int* re = rowblk_.data() + colptr_[col + 1];
            const int* it = std::lower_bound(rb, re, rowPos);
            // A block of the assembled pattern that holds only zeros now, but whose place exists in the factor anyway (it
            // lies inside the fill), is mapped too: if it becomes non-zero later -- a time offset moved a live window over
            // a knot -- nothing has to be redone.  Only zero blocks outside the factor's structure are left out.
            found = it != re && *it == rowPos;
            if (found) {
              ld = m_[col];
              coff = panel_off_[colBlock];
              base = rowoff_[first + (it - rb)];
              lastCol = col;
            } else if (keep[I]) {
              ok = false;
            }
          }
          if (!found) { emit(k, -1, 0); continue; }
          emit(k, lastCol, (int)(transposed ? base + lj + (long long)(li + coff) * ld : base + li + (long long)(lj + coff) * ld));
        }
      }
      for (int I : kept_[J]) keep[I] = 0;
      keep[J] = 0;
    }
    return ok;
  }

  // Ordering, factor structure and the map from the caller's value array into the panels, for the kept blocks.
  void buildSymbolic() {
    std::vector<std::vector<int>> adj(nb_);
    for (int J = 0; J < nb_; ++J)
      for (int I : kept_[J]) { adj[J].push_back(I); adj[I].push_back(J); }
    std::vector<int> w(nb_);
    for (int b = 0; b < nb_; ++b) w[b] = obase_[b + 1] - obase_[b];
    work_.clear();
    const bool timing = std::getenv("MC_CHOLESKY_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto s0 = now();
    partition(adj, w);         // fills perm_ / iperm_ (and the independent parts, if the blocks have coordinates)
    const auto s1 = now();
    symbolic(adj, w);          // fills the panel layout
    checkParts();
    const auto s2 = now();
    // Value map: every stored value goes to an entry of a panel (sc_src_ -> sc_dst_, grouped by panel, ascending value
    // index inside a panel) or, a zero block outside the factor's structure, to dropped_.  Two passes over the pattern on
    // all threads, count then write; the arrays are first touched by the threads that fill them.
    const int T = threadsFor((size_t)Cp_[n_]);
    std::vector<std::vector<long long>> cnt(T, std::vector<long long>(np_ + 1, 0));   // per thread: entries per panel, dropped at [np_]
    std::atomic<bool> missing(false);
    eachThread((size_t)Cp_[n_], [&](int t, int) {
      std::vector<long long>& c = cnt[t];
      const std::pair<int, int> range = blockRange(t, T);
      if (!mapBlockColumns(range.first, range.second, [&](int, int panel, int) { ++c[panel >= 0 ? panel : np_]; })) missing = true;
    });
    if (missing) throw std::logic_error("symbolic factor misses an entry of the matrix");
    sc_ptr_.assign(np_ + 1, 0);
    long long run = 0, drop = 0;
    for (int g = 0; g < np_; ++g) {
      sc_ptr_[g] = run;
      for (int t = 0; t < T; ++t) { const long long c = cnt[t][g]; cnt[t][g] = run; run += c; }
    }
    sc_ptr_[np_] = run;
    for (int t = 0; t < T; ++t) { const long long c = cnt[t][np_]; cnt[t][np_] = drop; drop += c; }
    sc_src_.reset(new int[run]); sc_dst_.reset(new int[run]);
    dropped_.reset(new int[drop]); n_dropped_ = (size_t)drop;
    eachThread((size_t)Cp_[n_], [&](int t, int) {
      std::vector<long long>& at = cnt[t];
      const std::pair<int, int> range = blockRange(t, T);
      mapBlockColumns(range.first, range.second, [&](int k, int panel, int offset) {
        if (panel < 0) { dropped_[at[np_]++] = k; return; }
        const long long e = at[panel]++;
        sc_src_[e] = k; sc_dst_[e] = offset;
      });
    });
    if (timing) std::fprintf(stderr, "cholesky: symbolic phase: ordering %.1f ms, structure %.1f ms, value map %.1f ms\n", ms(s0, s1), ms(s1, s2), ms(s2, now()));
    symbolic_ = true;
    ++stats_.symbolic_passes;
  }

  // false if a non-zero value falls outside the kept blocks
  bool checkDropped(const double* Cx) const {
    const size_t nd = n_dropped_;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads_, nd / 200000 + 1));   // small systems: not worth the threads
    std::vector<char> inside(T, 1);
    auto scan = [&](int t) {
      bool in = true;
      for (size_t i = nd * t / T, ie = nd * (t + 1) / T; i < ie; ++i) if (Cx[dropped_[i]] != 0.0) in = false;
      inside[t] = in ? 1 : 0;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(scan, t);
    scan(0);
    for (std::thread& z : th) z.join();
    for (int t = 0; t < T; ++t) if (!inside[t]) return false;
    return true;
  }
  // values -> panel g, right before the panel is worked on (it then sits in the cache of the thread that factorises it)
  void preparePanel(int g, const double* Cx) {
    double* P = Lx_.data() + valptr_[g];
    std::fill(P, P + (valptr_[g + 1] - valptr_[g]), 0.0);
    for (long long e = sc_ptr_[g]; e < sc_ptr_[g + 1]; ++e) P[sc_dst_[e]] = Cx[sc_src_[e]];
  }

  // P (len x dj, leading dimension ldp) -= T[src : src + len, :] (len x dk, leading dimension ld) * T[0:dj, :]^T.  This
  // update is where the factorisation spends its flops.  It runs in column tiles of DJ <= 6 and row tiles of 4 RV rows
  // whose DJ x RV vector accumulators stay in registers over the whole dk loop: per column of T, RV row vectors and DJ
  // broadcasts feed DJ RV fused multiply-adds, and P is read and written once.
  enum { KC = 32 };   // columns of T per pass (the coefficient tile B lives on the stack)
  template <int DJ, int RV>
  static void updateTile(const double* __restrict__ T, int ld, int boff, int src, int len, int dk, double* __restrict__ P, int ldp) {
    alignas(32) double B[KC][DJ];
    for (int ck = 0; ck < dk; ++ck)
      for (int cj = 0; cj < DJ; ++cj) B[ck][cj] = T[boff + cj + (size_t)ck * ld];
    const double* __restrict__ S = T + src;
    int r = 0;
#if defined(__AVX2__) && defined(__FMA__)
    for (; r + 4 * RV <= len; r += 4 * RV) {
      __m256d acc[DJ][RV];
#pragma GCC unroll 6
      for (int cj = 0; cj < DJ; ++cj)
#pragma GCC unroll 4
        for (int v = 0; v < RV; ++v) acc[cj][v] = _mm256_loadu_pd(P + r + 4 * v + (size_t)cj * ldp);
      for (int ck = 0; ck < dk; ++ck) {
        __m256d a[RV];
#pragma GCC unroll 4
        for (int v = 0; v < RV; ++v) a[v] = _mm256_loadu_pd(S + r + 4 * v + (size_t)ck * ld);
#pragma GCC unroll 6
        for (int cj = 0; cj < DJ; ++cj) {
          const __m256d bb = _mm256_broadcast_sd(&B[ck][cj]);
#pragma GCC unroll 4
          for (int v = 0; v < RV; ++v) acc[cj][v] = _mm256_fnmadd_pd(a[v], bb, acc[cj][v]);
        }
      }
#pragma GCC unroll 6
      for (int cj = 0; cj < DJ; ++cj)
#pragma GCC unroll 4
        for (int v = 0; v < RV; ++v) _mm256_storeu_pd(P + r + 4 * v + (size_t)cj * ldp, acc[cj][v]);
    }
    for (; r + 4 <= len; r += 4) {
      __m256d acc[DJ];
#pragma GCC unroll 6
      for (int cj = 0; cj < DJ; ++cj) acc[cj] = _mm256_loadu_pd(P + r + (size_t)cj * ldp);
      for (int ck = 0; ck < dk; ++ck) {
        const __m256d a0 = _mm256_loadu_pd(S + r + (size_t)ck * ld);
#pragma GCC unroll 6
        for (int cj = 0; cj < DJ; ++cj) acc[cj] = _mm256_fnmadd_pd(a0, _mm256_broadcast_sd(&B[ck][cj]), acc[cj]);
      }
#pragma GCC unroll 6
      for (int cj = 0; cj < DJ; ++cj) _mm256_storeu_pd(P + r + (size_t)cj * ldp, acc[cj]);
    }
#endif
    for (; r < len; ++r) {
      double s[DJ];
      for (int cj = 0; cj < DJ; ++cj) s[cj] = 0.0;
      for (int ck = 0; ck < dk; ++ck) {
        const double a = S[r + (size_t)ck * ld];
        for (int cj = 0; cj < DJ; ++cj) s[cj] += a * B[ck][cj];
      }
      for (int cj = 0; cj < DJ; ++cj) P[r + (size_t)cj * ldp] -= s[cj];
    }
  }
#if defined(__AVX2__) && defined(__FMA__) && defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define MC_CHOLESKY_AVX512 1
  // The same tile with 512-bit vectors (hosts that have them; chosen once at run time): DJ columns x 8 RV rows in
  // registers, the tail rows under a mask.
  template <int DJ, int RV>
  __attribute__((target("avx512f"))) static void updateTile512(const double* __restrict__ T, int ld, int boff, int src, int len, int dk,
                                                               double* __restrict__ P, int ldp) {
    alignas(64) double B[KC][DJ];
    for (int ck = 0; ck < dk; ++ck)
      for (int cj = 0; cj < DJ; ++cj) B[ck][cj] = T[boff + cj + (size_t)ck * ld];
    const double* __restrict__ S = T + src;
    for (int r = 0; r < len; r += 8 * RV) {
      __mmask8 m[RV];
#pragma GCC unroll 4
      for (int v = 0; v < RV; ++v) {
        const int left = len - r - 8 * v;
        m[v] = left >= 8 ? (__mmask8)0xff : left > 0 ? (__mmask8)((1u << left) - 1u) : (__mmask8)0;
      }
      __m512d acc[DJ][RV];
#pragma GCC unroll 6
      for (int cj = 0; cj < DJ; ++cj)
#pragma GCC unroll 4
        for (int v = 0; v < RV; ++v) acc[cj][v] = _mm512_maskz_loadu_pd(m[v], P + r + 8 * v + (size_t)cj * ldp);
      for (int ck = 0; ck < dk; ++ck) {
        __m512d a[RV];
#pragma GCC unroll 4
        for (int v = 0; v < RV; ++v) a[v] = _mm512_maskz_loadu_pd(m[v], S + r + 8 * v + (size_t)ck * ld);
#pragma GCC unroll 6
        for (int cj = 0; cj < DJ; ++cj) {
          const __m512d bb = _mm512_set1_pd(B[ck][cj]);
#pragma GCC unroll 4
          for (int v = 0; v < RV; ++v) acc[cj][v] = _mm512_fnmadd_pd(a[v], bb, acc[cj][v]);
        }
      }
#pragma GCC unroll 6
      for (int cj = 0; cj < DJ; ++cj)
#pragma GCC unroll 4
        for (int v = 0; v < RV; ++v) _mm512_mask_storeu_pd(P + r + 8 * v + (size_t)cj * ldp, m[v], acc[cj][v]);
    }
  }
  static bool wide() {
    static const bool w = std::getenv("MC_CHOLESKY_NO_AVX512") == nullptr && __builtin_cpu_supports("avx512f");
    return w;
  }
#endif
  static void panelUpdate(const double* T, int ld, int src, int len, int dk, int dj, double* P, int ldp) {
#ifdef MC_CHOLESKY_AVX512
    if (wide()) {
      for (int k0 = 0; k0 < dk; k0 += KC) {
        const double* Tk = T + (size_t)k0 * ld;
        const int kc = std::min<int>(KC, dk - k0);
        for (int j0 = 0; j0 < dj;) {
          const int c = std::min(6, dj - j0);
          double* Pj = P + (size_t)j0 * ldp;
          switch (c) {
            case 6: updateTile512<6, 3>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
            case 5: updateTile512<5, 3>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
            case 4: updateTile512<4, 4>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
            case 3: updateTile512<3, 4>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
            case 2: updateTile512<2, 4>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
            default: updateTile512<1, 4>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
          }
          j0 += c;
        }
      }
      return;
    }
#endif
    for (int k0 = 0; k0 < dk; k0 += KC) {
      const double* Tk = T + (size_t)k0 * ld;
      const int kc = std::min<int>(KC, dk - k0);
      for (int j0 = 0; j0 < dj;) {
        const int c = std::min(6, dj - j0);
        double* Pj = P + (size_t)j0 * ldp;
        switch (c) {
          case 6: updateTile<6, 2>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
          case 5: updateTile<5, 2>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
          case 4: updateTile<4, 3>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
          case 3: updateTile<3, 3>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
          case 2: updateTile<2, 4>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
          default: updateTile<1, 4>(Tk, ld, j0, src, len, kc, Pj, ldp); break;
        }
        j0 += c;
      }
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

  // Structure of the factor of a graph given, per node j in elimination order, by its neighbours > j (up[j]; duplicates
  // allowed): st[j] = sorted rows below the diagonal of column j.
  static void eliminationStructure(const std::vector<std::vector<int>>& up, std::vector<std::vector<int>>& st) {
    const int nn = (int)up.size();
    st.assign(nn, std::vector<int>());
    std::vector<std::vector<int>> children(nn);
    std::vector<int> mark(nn, -1);
    for (int j = 0; j < nn; ++j) {
      auto& sj = st[j];
      mark[j] = j;
      for (int i : up[j]) if (mark[i] != j) { mark[i] = j; sj.push_back(i); }
      for (int c : children[j])
        for (int i : st[c]) if (mark[i] != j) { mark[i] = j; sj.push_back(i); }
      std::sort(sj.begin(), sj.end());
      if (!sj.empty()) children[sj.front()].push_back(j);
    }
  }

  // Block structure of the factor in the new order and the panel layout.  Neighbouring blocks of the order whose columns
  // have the same structure below them (a pose coefficient and the bias coefficients whose support ends with it) share
  // one panel: the updates then run on panels of up to MAX_PANEL columns instead of 3 or 6, which is what lets the
  // update kernel reuse the rows it loads.
  enum { MAX_PANEL = 24 };
  void symbolic(const std::vector<std::vector<int>>& adj, const std::vector<int>& w) {
    const int nb = nb_;
    std::vector<std::vector<int>> up(nb), st;
    for (int j = 0; j < nb; ++j)
      for (int u : adj[perm_[j]]) if (iperm_[u] > j) up[j].push_back(iperm_[u]);
    eliminationStructure(up, st);
    // panels: maximal runs of positions j, j + 1, ... with struct(j) = {j + 1} + struct(j + 1), inside one part
    std::vector<char> boundary(nb + 1, 0);
    for (int b : part_start_) boundary[b] = 1;
    boundary[top_start_] = 1;
    panel_start_.assign(1, 0);
    pos_panel_.assign(nb, 0);
    double relax = 0.0;
    {
      // scalar rows below the diagonal block of every position
      std::vector<long long> sz(nb, 0);
      for (int j = 0; j < nb; ++j)
        for (int i : st[j]) sz[j] += w[perm_[i]];
      // tuning aids: MC_CHOLESKY_RELAX = share of explicit zeros a panel may hold (0: only blocks with identical
      // structure share a panel), MC_CHOLESKY_MAX_PANEL = widest panel in columns (0: no panels)
      relax = std::getenv("MC_CHOLESKY_RELAX") ? std::atof(std::getenv("MC_CHOLESKY_RELAX")) : 0.05;
      const int maxp = std::getenv("MC_CHOLESKY_MAX_PANEL") ? std::atoi(std::getenv("MC_CHOLESKY_MAX_PANEL")) : MAX_PANEL;
      const bool off = maxp <= 0;
      int first = 0, width = w[perm_[0]];
      for (int j = 1; j < nb; ++j) {
        const int dj = w[perm_[j]];
        bool same = !off && !boundary[j] && !st[j - 1].empty() && st[j - 1].front() == j && width + dj <= maxp;
        if (same) {
          // explicit zeros if [first, j] become one panel: member p gets the rows of j's structure it does not have itself
          double zeros = 0, area = 0;
          int after = dj;
          for (int q = j - 1; q >= first; --q) {
            const int dq = w[perm_[q]];
            zeros += (double)std::max<long long>(0, sz[j] - (sz[q] - after)) * dq;
            after += dq;
          }
          area = (double)(width + dj) * (sz[j] + width + dj);
          same = zeros <= relax * area;
        }
        if (same) width += dj;
        else { panel_start_.push_back(j); width = dj; first = j; }
        pos_panel_[j] = (int)panel_start_.size() - 1;
      }
      panel_start_.push_back(nb);
    }
    const int np = np_ = (int)panel_start_.size() - 1;
    // parts and top in panel numbers
    for (int& b : part_start_) b = b < nb ? pos_panel_[b] : np;
    top_start_panel_ = (!part_start_.empty() && top_start_ < nb) ? pos_panel_[top_start_] : (part_start_.empty() ? 0 : np);
    dim_.assign(np, 0); pbase_.assign(np + 1, 0);
    panel_of_.assign(nb, 0); panel_off_.assign(nb, 0); pdim_.assign(nb, 0); ybase_.assign(nb, 0);
    for (int g = 0; g < np; ++g) {
      for (int j = panel_start_[g]; j < panel_start_[g + 1]; ++j) {
        panel_of_[perm_[j]] = g;
        panel_off_[perm_[j]] = dim_[g];
        pdim_[j] = w[perm_[j]];
        ybase_[j] = pbase_[g] + dim_[g];
        dim_[g] += pdim_[j];
      }
      pbase_[g + 1] = pbase_[g] + dim_[g];
    }
    // Members of a panel are eliminated together, so they must have one structure: every member gets the later members
    // and the union of the members' neighbours beyond the panel (nothing new for members that had one structure already).
    if (np < nb) {
      std::vector<int> mark(nb, -1);
      for (int g = 0; g < np; ++g) {
        const int p0 = panel_start_[g], p1 = panel_start_[g + 1];
        if (p1 - p0 < 2) continue;
        std::vector<int> U;
        for (int p = p0; p < p1; ++p)
          for (int i : up[p]) if (i >= p1 && mark[i] != g) { mark[i] = g; U.push_back(i); }
        for (int p = p0; p < p1; ++p) {
          up[p] = U;
          for (int q = p + 1; q < p1; ++q) up[p].push_back(q);
        }
      }
      eliminationStructure(up, st);
    }
    std::vector<std::vector<int>>().swap(up);
    // rows of a panel: its member blocks (the diagonal block), then the structure of its last member
    colptr_.assign(np + 1, 0); valptr_.assign(np + 1, 0); m_.assign(np, 0);
    for (int g = 0; g < np; ++g)
      colptr_[g + 1] = colptr_[g] + (panel_start_[g + 1] - panel_start_[g]) + (long long)st[panel_start_[g + 1] - 1].size();
    rowblk_.resize(colptr_[np]); rowoff_.resize(colptr_[np]);
    max_m_ = 0; max_d_ = 0;
    stats_.flops = 0;
    for (int g = 0; g < np; ++g) {
      long long p = colptr_[g];
      int off = 0;
      for (int j = panel_start_[g]; j < panel_start_[g + 1]; ++j) { rowblk_[p] = j; rowoff_[p] = off; off += pdim_[j]; ++p; }
      for (int i : st[panel_start_[g + 1] - 1]) { rowblk_[p] = i; rowoff_[p] = off; off += pdim_[i]; ++p; }
      m_[g] = off;
      valptr_[g + 1] = valptr_[g] + (long long)off * dim_[g];
      max_m_ = std::max(max_m_, off); max_d_ = std::max(max_d_, dim_[g]);
      stats_.flops += (double)dim_[g] * off * off;      // ~ sum over the columns of (column count)^2
    }
    // row offsets of the top's blocks in every top panel (pushToTop)
    top_where_.clear();
    if (!part_start_.empty()) {
      const int ntp = nb - top_start_;
      top_where_.assign((size_t)(np - top_start_panel_) * ntp, -1);
      for (int g = top_start_panel_; g < np; ++g)
        for (long long q = colptr_[g]; q < colptr_[g + 1]; ++q) top_where_[(size_t)(g - top_start_panel_) * ntp + (rowblk_[q] - top_start_)] = rowoff_[q];
    }
    if (valptr_[np] > (long long)0x7fffffff) throw std::overflow_error("factor exceeds 2^31 entries");
    Lx_.assign(valptr_[np], 0.0);
    stats_.n = n_; stats_.n_blocks = nb; stats_.n_panels = np; stats_.nnz_factor = valptr_[np];
  }

  int n_ = 0, nb_ = 0, np_ = 0, max_m_ = 0, max_d_ = 0;      // scalar columns, blocks, panels
  bool analyzed_ = false, symbolic_ = false, factored_ = false;
  std::vector<int> Cp_, Ci_, blk_of_;
  std::vector<double> coord_;
  std::vector<int> next_;
  std::vector<long long> pos_;
  int threads_ = 1, top_start_ = 0, top_start_panel_ = 0;    // top_start_: position in perm_; part_start_ and top_start_panel_: panels
  std::vector<int> panel_start_, pos_panel_, panel_of_, panel_off_, pdim_, ybase_, top_where_;   // panel -> first position; position -> panel; block -> panel, column offset inside it
  std::vector<int> part_start_;
  std::vector<std::vector<int>> kept_;
  std::vector<int> obase_, perm_, iperm_, dim_, pbase_, m_, rowblk_, rowoff_;
  std::vector<long long> colptr_, valptr_;
  std::vector<long long> sc_ptr_;           // per panel: its entries of the caller's value array ...
  std::unique_ptr<int[]> sc_src_, sc_dst_, dropped_;   // ... (value index, offset in the panel); value indices outside the kept blocks
  size_t n_dropped_ = 0;
  std::vector<double> Lx_;
  std::vector<Work> work_;
  Stats stats_;
};

}  // namespace multical_amd
