// schur_plan.hpp -- symbolic side of the factorisation's GPU stage (device-free; the kernels are in mc_schur.hip).
//
// The normal equations of a calibration problem are banded along time: the coefficients of the pose spline (by far the
// most columns) couple to their r neighbours on either side, to the few bias-spline coefficients whose support covers
// them and to the calibration parameters.  Cut the pose coefficients into P intervals by removing r consecutive blocks at
// every cut ("separators"): the interiors I_1 .. I_P are then mutually independent, and eliminating them,
//
//      A = | A_II  A_IB |      A_II = U^T U,   W = A_BI U^-1,   y = U^-T b_I
//          | A_BI  A_BB |      S    = A_BB - W W^T,              b_S = b_B - W y
//                              x_B  = S^-1 b_S (host, BlockSparseCholesky),  x_I = U^-1 (y - W^T x_B)
//
// is P independent banded factorisations + P dense products -- the part of sparse_block_matrix::LinearSolverCholmod::solve
// (linear_solver_cholmod.h:70-112) that has the flops and no dependency chain across the sequence.  The reduced system S
// (separators, bias coefficients, calibration parameters) goes to the host factorisation that already exists.
//
// The plan works on the NUMERICAL pattern, like the host factorisation does: the assembled pattern holds the structurally
// present, numerically zero blocks of padded spline-coefficient buffers, which would make the pose band three times as
// wide.  Block pairs that touch an interior block and were zero when the plan was made are listed (`check`): the device
// tests them before every elimination and asks for a new plan if one has become non-zero.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace multical_amd {

// Block view of a scalar upper-triangular CCS matrix in SparseBlockMatrix::fillCCS layout: within a block column the blocks
// top to bottom, the diagonal block last and upper triangular.  Value index of element (r of row block, c of column block),
// k = position of the row block in the column's list:  cp0[cb] + c h[cb] + c (c + 1) / 2 + rowOff[k] + r.
struct BlockPattern {
  int n = 0, nb = 0;
  std::vector<int> base;                    // nb + 1
  std::vector<int> colPtr, rowBlk, rowOff;  // per block column: its row blocks (ascending, diagonal last)
  std::vector<long long> cp0;               // first value of the block column
  std::vector<int> h;                       // scalar rows above the diagonal block
  int dim(int b) const { return base[b + 1] - base[b]; }
  int find(int rb, int cb) const {
    const int* lo = rowBlk.data() + colPtr[cb];
    const int* hi = rowBlk.data() + colPtr[cb + 1];
    const int* it = std::lower_bound(lo, hi, rb);
    return (it != hi && *it == rb) ? (int)(it - rowBlk.data()) : -1;
  }
  long long first(int k, int cb) const { return cp0[cb] + rowOff[k]; }   // element (0, 0) of the pair; then + c h + c(c+1)/2 + r
  long long nnz() const { return nb ? cp0[nb - 1] + (long long)dim(nb - 1) * h[nb - 1] + (long long)dim(nb - 1) * (dim(nb - 1) + 1) / 2 : 0; }

  void fromScalar(int n_, const int* Cp, const int* Ci, int nb_, const int* base_) {
    n = n_; nb = nb_;
    base.assign(base_, base_ + nb + 1);
    std::vector<int> blkOf(n);
    for (int b = 0; b < nb; ++b) for (int c = base[b]; c < base[b + 1]; ++c) blkOf[c] = b;
    colPtr.assign(nb + 1, 0); rowBlk.clear(); rowOff.clear(); cp0.assign(nb, 0); h.assign(nb, 0);
    for (int cb = 0; cb < nb; ++cb) {
      const int c0 = base[cb];
      cp0[cb] = Cp[c0];
      int k = Cp[c0];
      const int end = Cp[c0 + 1] - 1;   // the last entry of the block's first column is its diagonal element
      int off = 0;
      while (k < end) {
        const int b = blkOf[Ci[k]];
        if (Ci[k] != base[b] || b >= cb) throw std::invalid_argument("pattern is not in fillCCS block layout");
        rowBlk.push_back(b); rowOff.push_back(off);
        off += dim(b); k += dim(b);
      }
      if (k != end || Ci[end] != c0) throw std::invalid_argument("pattern is not in fillCCS block layout (diagonal)");
      h[cb] = off;
      rowBlk.push_back(cb); rowOff.push_back(off);
      colPtr[cb + 1] = (int)rowBlk.size();
      for (int c = 0; c < dim(cb); ++c)
        if (Cp[c0 + c + 1] - Cp[c0 + c] != off + c + 1) throw std::invalid_argument("pattern is not in fillCCS block layout (column height)");
    }
  }
  // from per-column row-block lists (each ascending, ending with the diagonal)
  void fromLists(const std::vector<int>& dims, const std::vector<std::vector<int>>& cols) {
    nb = (int)dims.size();
    base.assign(nb + 1, 0);
    for (int b = 0; b < nb; ++b) base[b + 1] = base[b] + dims[b];
    n = base[nb];
    colPtr.assign(nb + 1, 0); rowBlk.clear(); rowOff.clear(); cp0.assign(nb, 0); h.assign(nb, 0);
    long long v = 0;
    for (int cb = 0; cb < nb; ++cb) {
      cp0[cb] = v;
      int off = 0;
      for (int b : cols[cb]) { rowBlk.push_back(b); rowOff.push_back(off); if (b != cb) off += dims[b]; }
      h[cb] = off;
      colPtr[cb + 1] = (int)rowBlk.size();
      v += (long long)dims[cb] * off + (long long)dims[cb] * (dims[cb] + 1) / 2;
    }
  }
  void toScalar(std::vector<int>& Cp, std::vector<int>& Ci) const {
    Cp.assign(n + 1, 0); Ci.clear();
    for (int cb = 0; cb < nb; ++cb)
      for (int c = 0; c < dim(cb); ++c) {
        for (int k = colPtr[cb]; k < colPtr[cb + 1] - 1; ++k)
          for (int r = 0; r < dim(rowBlk[k]); ++r) Ci.push_back(base[rowBlk[k]] + r);
        for (int r = 0; r <= c; ++r) Ci.push_back(base[cb] + r);
        Cp[base[cb] + c + 1] = (int)Ci.size();
      }
  }
};

struct SchurPlan {
  enum { MAX_BW = 36,       // scalar width of the band window the elimination kernel keeps in LDS: d (r + 1)
         MAX_DIM = 16 };    // widest design-variable block the gather kernels take (a lane per column of a block pair)
  bool ok = false;
  std::string why;          // when !ok
  BlockPattern A;           // the assembled matrix
  BlockPattern R;           // the reduced system
  int d = 0, r = 0, bw = 0; // block dimension of the band spline, block half-bandwidth, scalar window d (r + 1)
  int band0 = 0, bandN = 0; // its blocks [band0, band0 + bandN)
  std::vector<int> rmap;    // block -> reduced block, -1 for interior blocks
  std::vector<int> rblocks; // reduced block -> block
  std::vector<int> rcol;    // reduced scalar column -> scalar column

  struct Part {
    int i0 = 0, m = 0;        // first interior block, number of interior blocks
    int nI = 0, nB = 0, ldW = 0;
    std::vector<int> B;       // boundary blocks, ascending
    std::vector<int> Boff;    // scalar offset of every boundary block (B.size() + 1)
    long long wOff = 0, bandOff = 0;   // offsets (doubles) of the part's W^T and band arrays in the device work area
  };
  std::vector<Part> parts;
  long long wTotal = 0, bandTotal = 0;
  int maxNB = 0, maxNI = 0;

  // flat device tables (see mc_schur.hip for the kernels that read them)
  struct CopyPair { long long src; long long dst; int hs, hd, dr, dc, diag, pad; };                      // R x R pairs of A -> S
  struct CheckPair { long long src; int hs, dr, dc, diag; };                                             // must stay zero
  struct BandPair { long long src; int hs, part, j, k, pad; };                                           // (I_{j-k}, I_j) -> band
  struct WPair { long long src; int hs, part, row0, col0, dr, dc, byRow; };                              // (interior, boundary) -> W^T
  struct Scatter { long long dst; int hd, pad; };                                                        // per part: nBlk x nBlk
  std::vector<CopyPair> copy;
  std::vector<CheckPair> check;
  std::vector<BandPair> band;
  std::vector<WPair> wpairs;
  std::vector<Scatter> scatter;       // concatenated; scatterOff[p] = first of part p
  std::vector<long long> scatterOff;
  std::vector<int> browBlk, browOff, browCol;   // per part and scalar boundary row: local block, offset in it, reduced scalar column
  std::vector<long long> browBase;              // first entry of part p in the three arrays above

  // nz: one flag per block pair of A (position in A.rowBlk): does the pair hold a non-zero value
  static std::vector<char> nonZeroPairs(const BlockPattern& A, const double* Cx) {
    std::vector<char> nz(A.rowBlk.size(), 0);
    for (int cb = 0; cb < A.nb; ++cb)
      for (int k = A.colPtr[cb]; k < A.colPtr[cb + 1]; ++k) {
        const int rb = A.rowBlk[k], dr = A.dim(rb), dc = A.dim(cb);
        bool any = rb == cb;   // the diagonal block always counts
        for (int c = 0; c < dc && !any; ++c) {
          const double* v = Cx + A.first(k, cb) + (long long)c * A.h[cb] + (c * (c + 1)) / 2;
          for (int rr = 0; rr < dr; ++rr) if (v[rr] != 0.0) { any = true; break; }
        }
        nz[k] = any;
      }
    return nz;
  }

  // spline_of[b]: spline of a coefficient block, -1 otherwise.  want_parts: interiors asked for (fewer if the band is short).
  void build(int n, const int* Cp, const int* Ci, int nb, const int* base, const int* spline_of, const double* Cx, int want_parts) {
    ok = false; why.clear();
    A.fromScalar(n, Cp, Ci, nb, base);
    for (int b = 0; b < nb; ++b) if (A.dim(b) > MAX_DIM) { why = "a design-variable block wider than 16 columns"; return; }
    // ---- the band spline: the one with the most blocks; they must be consecutive and of one dimension
    int best = -1, bestCount = 0;
    {
      std::vector<int> count;
      for (int b = 0; b < nb; ++b) if (spline_of[b] >= 0) { if ((int)count.size() <= spline_of[b]) count.resize(spline_of[b] + 1, 0); ++count[spline_of[b]]; }
      for (size_t s = 0; s < count.size(); ++s) if (count[s] > bestCount) { bestCount = count[s]; best = (int)s; }
    }
    if (best < 0) { why = "no spline"; return; }
    band0 = -1;
    for (int b = 0; b < nb; ++b) if (spline_of[b] == best) { band0 = b; break; }
    bandN = bestCount;
    d = A.dim(band0);
    for (int b = band0; b < band0 + bandN; ++b)
      if (b >= nb || spline_of[b] != best || A.dim(b) != d) { why = "the largest spline's blocks are not consecutive"; return; }
    const std::vector<char> nz = nonZeroPairs(A, Cx);
    auto inBand = [&](int b) { return b >= band0 && b < band0 + bandN; };
    r = 0;
    for (int cb = band0; cb < band0 + bandN; ++cb)
      for (int k = A.colPtr[cb]; k < A.colPtr[cb + 1]; ++k)
        if (nz[k] && inBand(A.rowBlk[k])) r = std::max(r, cb - A.rowBlk[k]);
    bw = d * (r + 1);
    if (bw > MAX_BW || r < 1) { why = "band too wide for the elimination kernel"; return; }
    // ---- interiors: P intervals with r separator blocks between neighbours, each at least 4 r blocks long
    int P = std::max(1, want_parts);
    while (P > 1 && (bandN - (P - 1) * r) / P < 4 * r) --P;
    if (bandN - (P - 1) * r < 4 * r) { why = "band too short"; return; }
    parts.assign(P, Part());
    std::vector<int> partOf(nb, -1);
    {
      const int interior = bandN - (P - 1) * r;
      int at = band0;
      for (int p = 0; p < P; ++p) {
        const int m = interior / P + (p < interior % P ? 1 : 0);
        parts[p].i0 = at; parts[p].m = m; parts[p].nI = m * d;
        for (int b = at; b < at + m; ++b) partOf[b] = p;
        at += m + r;
      }
    }
    // ---- boundaries: the blocks that hold a non-zero value against an interior block
    std::vector<std::vector<int>> Bset(P);
    for (int cb = 0; cb < nb; ++cb)
      for (int k = A.colPtr[cb]; k < A.colPtr[cb + 1]; ++k) {
        const int rb = A.rowBlk[k];
        const int pr = partOf[rb], pc = partOf[cb];
        if (!nz[k] || (pr < 0 && pc < 0) || (pr >= 0 && pr == pc)) continue;
        if (pr >= 0 && pc >= 0) { why = "two interiors are coupled"; return; }
        if (pr >= 0) Bset[pr].push_back(cb); else Bset[pc].push_back(rb);
      }
    maxNB = maxNI = 0; wTotal = bandTotal = 0;
    for (int p = 0; p < P; ++p) {
      Part& pt = parts[p];
      std::sort(Bset[p].begin(), Bset[p].end());
      Bset[p].erase(std::unique(Bset[p].begin(), Bset[p].end()), Bset[p].end());
      pt.B = Bset[p];
      pt.Boff.assign(pt.B.size() + 1, 0);
      for (size_t i = 0; i < pt.B.size(); ++i) pt.Boff[i + 1] = pt.Boff[i] + A.dim(pt.B[i]);
      pt.nB = pt.Boff.back();
      pt.ldW = ((pt.nB + 1 + 3) / 4) * 4;     // boundary rows + the right-hand side's row
      pt.wOff = wTotal; wTotal += (long long)pt.nI * pt.ldW;
      pt.bandOff = bandTotal; bandTotal += (long long)pt.nI * bw;
      maxNB = std::max(maxNB, pt.nB); maxNI = std::max(maxNI, pt.nI);
    }
    // ---- every stored pair that touches an interior block: inside the band (distance <= r) and against a block of the
    // boundary it is gathered whether it holds a value now or not -- a live window that moves by a segment brings values to
    // such pairs all the time, and they cost nothing here; any other pair must stay zero and is watched
    band.clear(); check.clear(); wpairs.clear();
    for (int cb = 0; cb < nb; ++cb)
      for (int k = A.colPtr[cb]; k < A.colPtr[cb + 1]; ++k) {
        const int rb = A.rowBlk[k];
        const int pr = partOf[rb], pc = partOf[cb];
        if (pr < 0 && pc < 0) continue;
        bool taken = false;
        if (pr >= 0 && pr == pc) {
          if (cb - rb <= r) { band.push_back(BandPair{A.first(k, cb), A.h[cb], pr, cb - parts[pr].i0, cb - rb, 0}); taken = true; }
        } else if ((pr < 0) != (pc < 0)) {
          const int p = pr >= 0 ? pr : pc, ib = pr >= 0 ? rb : cb, ob = pr >= 0 ? cb : rb;
          const Part& pt = parts[p];
          const auto it = std::lower_bound(pt.B.begin(), pt.B.end(), ob);
          if (it != pt.B.end() && *it == ob) {
            const int lb = (int)(it - pt.B.begin());
            // byRow = 1: the interior block is the ROW block of the stored pair (element (rr, cc): interior component rr, boundary component cc)
            wpairs.push_back(WPair{A.first(k, cb), A.h[cb], p, pt.Boff[lb], (ib - pt.i0) * d, A.dim(rb), A.dim(cb), pr >= 0 ? 1 : 0});
            taken = true;
          }
        }
        if (taken) continue;
        if (nz[k]) { why = "a non-zero block pair outside the planned structure"; return; }
        check.push_back(CheckPair{A.first(k, cb), A.h[cb], A.dim(rb), A.dim(cb), rb == cb ? 1 : 0});
      }
    // ---- reduced system: every block that is not interior, in the original order
    rmap.assign(nb, -1); rblocks.clear();
    for (int b = 0; b < nb; ++b) if (partOf[b] < 0) { rmap[b] = (int)rblocks.size(); rblocks.push_back(b); }
    const int nbR = (int)rblocks.size();
    std::vector<int> dimsR(nbR);
    for (int i = 0; i < nbR; ++i) dimsR[i] = A.dim(rblocks[i]);
    std::vector<std::vector<int>> colsR(nbR);
    for (int cb = 0; cb < nb; ++cb) {
      if (rmap[cb] < 0) continue;
      for (int k = A.colPtr[cb]; k < A.colPtr[cb + 1]; ++k) if (rmap[A.rowBlk[k]] >= 0) colsR[rmap[cb]].push_back(rmap[A.rowBlk[k]]);
    }
    for (int p = 0; p < P; ++p) {
      const std::vector<int>& B = parts[p].B;
      for (size_t j = 0; j < B.size(); ++j) for (size_t i = 0; i <= j; ++i) colsR[rmap[B[j]]].push_back(rmap[B[i]]);
    }
    for (int cb = 0; cb < nbR; ++cb) {
      std::vector<int>& c = colsR[cb];
      std::sort(c.begin(), c.end());
      c.erase(std::unique(c.begin(), c.end()), c.end());
      if (c.empty() || c.back() != cb) { why = "reduced pattern lost a diagonal block"; return; }
    }
    R.fromLists(dimsR, colsR);
    rcol.assign(R.n, 0);
    for (int i = 0; i < nbR; ++i) for (int c = 0; c < dimsR[i]; ++c) rcol[R.base[i] + c] = A.base[rblocks[i]] + c;
    // ---- copy list and scatter tables
    copy.clear();
    for (int cb = 0; cb < nb; ++cb) {
      if (rmap[cb] < 0) continue;
      for (int k = A.colPtr[cb]; k < A.colPtr[cb + 1]; ++k) {
        const int rb = A.rowBlk[k];
        if (rmap[rb] < 0) continue;
        const int kr = R.find(rmap[rb], rmap[cb]);
        copy.push_back(CopyPair{A.first(k, cb), R.first(kr, rmap[cb]), A.h[cb], R.h[rmap[cb]], A.dim(rb), A.dim(cb), rb == cb ? 1 : 0, 0});
      }
    }
    scatter.clear(); scatterOff.assign(P + 1, 0);
    browBlk.clear(); browOff.clear(); browCol.clear(); browBase.assign(P + 1, 0);
    for (int p = 0; p < P; ++p) {
      const Part& pt = parts[p];
      const int nblk = (int)pt.B.size();
      scatterOff[p] = (long long)scatter.size();
      for (int lb = 0; lb < nblk; ++lb)          // table[la * nblk + lb], la <= lb
        for (int la = 0; la < nblk; ++la) {
          Scatter s{-1, 0, 0};
          if (la <= lb) {
            const int cbR = rmap[pt.B[lb]], kr = R.find(rmap[pt.B[la]], cbR);
            s.dst = R.first(kr, cbR); s.hd = R.h[cbR];
          }
          scatter.push_back(s);
        }
      browBase[p] = (long long)browBlk.size();
      for (int lb = 0; lb < nblk; ++lb)
        for (int c = 0; c < A.dim(pt.B[lb]); ++c) { browBlk.push_back(lb); browOff.push_back(c); browCol.push_back(R.base[rmap[pt.B[lb]]] + c); }
    }
    scatterOff[P] = (long long)scatter.size();
    browBase[P] = (long long)browBlk.size();
    // the scatter table above is indexed [lb * nblk + la]: column block major
    ok = true;
  }

  // ------------------------------------------------------------------ reference executor (host, plain loops): what the
  // kernels compute, for the tests.  eliminate() fills S (values of R's pattern) and bS; back() completes x.
  struct HostWork { std::vector<double> band, Wt; };
  // band storage: Ab[c * bw + q] = element (row c - q, column c) of the interior matrix, q = 0 .. bw - 1 (upper triangle)
  bool eliminate(const double* Cx, const double* g, double shift, std::vector<double>& S, std::vector<double>& bS, HostWork& wk) const {
    S.assign((size_t)R.nnz(), 0.0);
    bS.assign(R.n, 0.0);
    for (const CopyPair& cp : copy)
      for (int c = 0; c < cp.dc; ++c)
        for (int rr = 0; rr < (cp.diag ? c + 1 : cp.dr); ++rr)
          S[cp.dst + (long long)c * cp.hd + (c * (c + 1)) / 2 + rr] = Cx[cp.src + (long long)c * cp.hs + (c * (c + 1)) / 2 + rr] + ((cp.diag && rr == c) ? shift : 0.0);
    for (int i = 0; i < R.n; ++i) bS[i] = g[rcol[i]];
    for (const CheckPair& ck : check)
      for (int c = 0; c < ck.dc; ++c)
        for (int rr = 0; rr < (ck.diag ? c + 1 : ck.dr); ++rr)
          if (Cx[ck.src + (long long)c * ck.hs + (c * (c + 1)) / 2 + rr] != 0.0) return false;   // the structure grew: new plan
    wk.band.assign((size_t)bandTotal, 0.0);
    wk.Wt.assign((size_t)wTotal, 0.0);
    for (const BandPair& bp : band) {
      const Part& pt = parts[bp.part];
      double* Ab = wk.band.data() + pt.bandOff;
      for (int c = 0; c < d; ++c)
        for (int rr = 0; rr < (bp.k == 0 ? c + 1 : d); ++rr) {
          const int col = bp.j * d + c, q = bp.k * d + c - rr;
          Ab[(long long)col * bw + q] = Cx[bp.src + (long long)c * bp.hs + (c * (c + 1)) / 2 + rr] + ((bp.k == 0 && rr == c) ? shift : 0.0);
        }
    }
    for (const WPair& wp : wpairs) {
      const Part& pt = parts[wp.part];
      double* Wt = wk.Wt.data() + pt.wOff;
      for (int c = 0; c < wp.dc; ++c)
        for (int rr = 0; rr < wp.dr; ++rr) {
          const double v = Cx[wp.src + (long long)c * wp.hs + (c * (c + 1)) / 2 + rr];
          const int icomp = wp.byRow ? rr : c, bcomp = wp.byRow ? c : rr;
          Wt[(long long)(wp.col0 + icomp) * pt.ldW + wp.row0 + bcomp] = v;
        }
    }
    for (size_t p = 0; p < parts.size(); ++p) {
      const Part& pt = parts[p];
      double* Ab = wk.band.data() + pt.bandOff;
      double* Wt = wk.Wt.data() + pt.wOff;
      const int nI = pt.nI, nW = pt.nB + 1, c00 = A.base[pt.i0];
      for (int c = 0; c < nI; ++c) Wt[(long long)c * pt.ldW + pt.nB] = g[c00 + c];
      // banded Cholesky A_II = U^T U, U in place (same band layout)
      for (int c = 0; c < nI; ++c) {
        for (int q = std::min(c, bw - 1); q >= 0; --q) {
          const int i = c - q;              // row
          double s = Ab[(long long)c * bw + q];
          for (int k = std::max(0, c - (bw - 1)); k < i; ++k) s -= Ab[(long long)i * bw + (i - k)] * Ab[(long long)c * bw + (c - k)];
          if (q > 0) Ab[(long long)c * bw + q] = s / Ab[(long long)i * bw];
          else { if (!(s > 0.0) || !std::isfinite(s)) return false; Ab[(long long)c * bw] = std::sqrt(s); }
        }
      }
      // rows of W: w U = a   (forward along the columns)
      for (int c = 0; c < nI; ++c) {
        double* wc = Wt + (long long)c * pt.ldW;
        for (int k = std::max(0, c - (bw - 1)); k < c; ++k) {
          const double u = Ab[(long long)c * bw + (c - k)];
          if (u == 0.0) continue;
          const double* wkk = Wt + (long long)k * pt.ldW;
          for (int rho = 0; rho < nW; ++rho) wc[rho] -= wkk[rho] * u;
        }
        const double inv = 1.0 / Ab[(long long)c * bw];
        for (int rho = 0; rho < nW; ++rho) wc[rho] *= inv;
      }
      // S -= W W^T, bS -= W y
      const int nblk = (int)pt.B.size();
      const Scatter* tab = scatter.data() + scatterOff[p];
      const int* bb = browBlk.data() + browBase[p];
      const int* bo = browOff.data() + browBase[p];
      const int* bc = browCol.data() + browBase[p];
      for (int b2 = 0; b2 < pt.nB; ++b2)
        for (int a2 = 0; a2 <= b2; ++a2) {
          double t = 0.0;
          for (int c = 0; c < nI; ++c) t += Wt[(long long)c * pt.ldW + a2] * Wt[(long long)c * pt.ldW + b2];
          const Scatter& s = tab[bb[b2] * nblk + bb[a2]];
          S[s.dst + (long long)bo[b2] * s.hd + (bo[b2] * (bo[b2] + 1)) / 2 + bo[a2]] -= t;
        }
      for (int a2 = 0; a2 < pt.nB; ++a2) {
        double t = 0.0;
        for (int c = 0; c < nI; ++c) t += Wt[(long long)c * pt.ldW + a2] * Wt[(long long)c * pt.ldW + pt.nB];
        bS[bc[a2]] -= t;
      }
    }
    return true;
  }
  // x (n, original numbering) from the reduced solution xR
  void back(const double* xR, const HostWork& wk, double* x) const {
    for (int i = 0; i < R.n; ++i) x[rcol[i]] = xR[i];
    for (size_t p = 0; p < parts.size(); ++p) {
      const Part& pt = parts[p];
      const double* Ab = wk.band.data() + pt.bandOff;
      const double* Wt = wk.Wt.data() + pt.wOff;
      const int* bc = browCol.data() + browBase[p];
      const int nI = pt.nI, c00 = A.base[pt.i0];
      std::vector<double> t(nI);
      for (int c = 0; c < nI; ++c) {
        double s = Wt[(long long)c * pt.ldW + pt.nB];
        for (int rho = 0; rho < pt.nB; ++rho) s -= Wt[(long long)c * pt.ldW + rho] * xR[bc[rho]];
        t[c] = s;
      }
      for (int c = nI - 1; c >= 0; --c) {      // U x = t
        double s = t[c];
        for (int k = c + 1; k < std::min(nI, c + bw); ++k) s -= Ab[(long long)k * bw + (k - c)] * x[c00 + k];
        x[c00 + c] = s / Ab[(long long)c * bw];
      }
    }
  }
};

}  // namespace multical_amd
