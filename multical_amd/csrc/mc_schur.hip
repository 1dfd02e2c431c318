// mc_schur.hip -- the factorisation's GPU stage: elimination of the pose-spline interiors on the device, the reduced system
// on the host (host/schur_plan.hpp has the algebra and the symbolic plan, host/block_cholesky.hpp the host factorisation).
//
// Stands in for the interior of sparse_block_matrix::LinearSolverCholmod::solve (linear_solver_cholmod.h:70-112) as
// BlockCholeskyLinearSystemSolver::solveSystem calls it (aslam_backend/src/BlockCholeskyLinearSystemSolver.cpp:76-98).
// The system never leaves the device whole: what crosses PCIe per solve is the reduced system (a few 10^6 values where the
// assembled matrix has 2 10^7) and the solution.
//
// Kernels, per solve:
//   k_schur_copy / _check / _band / _w / _rhs_rows   gather: S <- A_RR (+ shift), watch list, band arrays, W^T, right-hand sides
//   k_schur_factor   one workgroup per interior: banded Cholesky A_II = U^T U in a circular bw x bw LDS window (right-looking,
//                    one scalar column per step) and, after every d columns, the block of W = A_BI U^-1 they complete
//                    (lane = boundary row; the last bw - d columns of W in an LDS ring)
//   k_schur_syrk     S -= W W^T and b_S -= W y: 64 x 64 output tiles, 16-deep LDS panels, fp64 matrix cores (a 32 x 32 quadrant per wave), fp64 atomics into
//                    the reduced system's value array (neighbouring interiors share separators, all share the calibration blocks)
//   k_schur_back     x_I = U^-1 (y - W^T x_B): wave-per-column products, then the banded back-substitution by one wave with
//                    the last bw - 1 unknowns in its lanes
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/multical_hip.h"
#include "host/block_cholesky.hpp"
#include "host/schur_device.hpp"
#include "host/schur_plan.hpp"

namespace multical_amd {

namespace {

struct PartDev {
  int c00, nI, nB, ldW, nblk, pad;
  long long wOff, bandOff, scatterOff, browBase;
};

__device__ __forceinline__ double waveSum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ void addF64(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_schur_copy(const SchurPlan::CopyPair* __restrict__ cp, int n, const double* __restrict__ Cx,
                                                    double* __restrict__ S, double shift) {
  const int id = blockIdx.x * 256 + threadIdx.x, pair = id >> 4, c = id & 15;   // up to SchurPlan::MAX_DIM columns per block
  if (pair >= n) return;
  const SchurPlan::CopyPair p = cp[pair];
  if (c >= p.dc) return;
  const double* src = Cx + p.src + (long long)c * p.hs + (c * (c + 1)) / 2;
  double* dst = S + p.dst + (long long)c * p.hd + (c * (c + 1)) / 2;
  const int nr = p.diag ? c + 1 : p.dr;
  for (int r = 0; r < nr; ++r) dst[r] = src[r] + ((p.diag && r == c) ? shift : 0.0);
}
__global__ __launch_bounds__(256) void k_schur_check(const SchurPlan::CheckPair* __restrict__ ck, int n, const double* __restrict__ Cx,
                                                     int* __restrict__ flag) {
  const int id = blockIdx.x * 256 + threadIdx.x, pair = id >> 4, c = id & 15;   // up to SchurPlan::MAX_DIM columns per block
  if (pair >= n) return;
  const SchurPlan::CheckPair p = ck[pair];
  if (c >= p.dc) return;
  const double* src = Cx + p.src + (long long)c * p.hs + (c * (c + 1)) / 2;
  const int nr = p.diag ? c + 1 : p.dr;
  bool any = false;
  for (int r = 0; r < nr; ++r) any = any || src[r] != 0.0;
  if (any) atomicOr(flag + 1, 1);
}
__global__ __launch_bounds__(256) void k_schur_band(const SchurPlan::BandPair* __restrict__ bp, int n, const PartDev* __restrict__ parts,
                                                    const double* __restrict__ Cx, double* __restrict__ band, int d, int bw, double shift) {
  const int id = blockIdx.x * 256 + threadIdx.x, pair = id >> 4, c = id & 15;   // up to SchurPlan::MAX_DIM columns per block
  if (pair >= n || c >= d) return;
  const SchurPlan::BandPair p = bp[pair];
  const double* src = Cx + p.src + (long long)c * p.hs + (c * (c + 1)) / 2;
  double* Ab = band + parts[p.part].bandOff + (long long)(p.j * d + c) * bw;
  const int nr = p.k == 0 ? c + 1 : d;
  for (int r = 0; r < nr; ++r) Ab[p.k * d + c - r] = src[r] + ((p.k == 0 && r == c) ? shift : 0.0);
}
__global__ __launch_bounds__(256) void k_schur_w(const SchurPlan::WPair* __restrict__ wp, int n, const PartDev* __restrict__ parts,
                                                 const double* __restrict__ Cx, double* __restrict__ W) {
  const int id = blockIdx.x * 256 + threadIdx.x, pair = id >> 4, c = id & 15;   // up to SchurPlan::MAX_DIM columns per block
  if (pair >= n) return;
  const SchurPlan::WPair p = wp[pair];
  if (c >= p.dc) return;
  const PartDev pt = parts[p.part];
  const double* src = Cx + p.src + (long long)c * p.hs + (c * (c + 1)) / 2;
  double* Wt = W + pt.wOff;
  for (int r = 0; r < p.dr; ++r) {
    const int icomp = p.byRow ? r : c, bcomp = p.byRow ? c : r;
    Wt[(long long)(p.col0 + icomp) * pt.ldW + p.row0 + bcomp] = src[r];
  }
}
__global__ __launch_bounds__(256) void k_schur_rhs(const PartDev* __restrict__ parts, const int* __restrict__ rcol, int nR,
                                                   const double* __restrict__ g, double* __restrict__ W, double* __restrict__ bS) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.y == gridDim.y - 1) { if (c < nR) bS[c] = g[rcol[c]]; return; }   // the last row of the grid: the reduced right-hand side
  const PartDev pt = parts[blockIdx.y];
  if (c < pt.nI) W[pt.wOff + (long long)c * pt.ldW + pt.nB] = g[pt.c00 + c];
}

constexpr int FACTOR_THREADS = 512;
__global__ __launch_bounds__(FACTOR_THREADS) void k_schur_factor(const PartDev* __restrict__ parts, double* __restrict__ band, double* __restrict__ W,
                                                      int d, int bw, int* __restrict__ flag) {
  extern __shared__ double sm[];
  const PartDev pt = parts[blockIdx.x];
  double* Wn = sm;                // bw x bw window of the trailing matrix, circular: (row, col), row <= col, at [(row % bw) bw + col % bw]
  double* Ur = Wn + bw * bw;      // the last bw rows of U: Ur[(k % bw) bw + j] = U(k, k + j)
  double* ring = Ur + bw * bw;    // the last bw - d columns of W: column k at [(k % (bw - d)) ldW + row]
  double* Ab = band + pt.bandOff;
  double* Wt = W + pt.wOff;
  const int tid = threadIdx.x, nI = pt.nI, ldW = pt.ldW, nW = pt.nB + 1, rw = bw - d;
  for (int e = tid; e < bw * bw; e += FACTOR_THREADS) {
    const int rho = e / bw, kap = e - rho * bw;
    Wn[e] = (rho <= kap && kap < nI) ? Ab[(long long)kap * bw + (kap - rho)] : (rho == kap ? 1.0 : 0.0);
  }
  // the trailing update touches the pairs (i, jj), 1 <= i <= jj < bw, relative to the pivot: the same ones at every step, so
  // a lane's share is decoded once (integer division costs more than the update itself)
  constexpr int MAXP = (SchurPlan::MAX_BW - 1) * SchurPlan::MAX_BW / 2;   // 630 pairs at most
  constexpr int PER = (MAXP + FACTOR_THREADS - 1) / FACTOR_THREADS;
  int pi[PER], pj[PER];
  {
    const int npairs = (bw - 1) * bw / 2;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int e = tid + FACTOR_THREADS * t;
      pi[t] = pj[t] = 0;                 // 0: no pair
      if (e < npairs) {
        int jj = 1, rem = e;             // pairs numbered column by column: column jj holds i = 1 .. jj
        while (rem >= jj) { rem -= jj; ++jj; }
        pi[t] = 1 + rem; pj[t] = jj;
      }
    }
  }
  __syncthreads();
  int cs = 0;                            // c % bw
  // the column that enters the window at the end of step c is loaded one step ahead (a dependent global load per step would
  // otherwise sit between two barriers)
  double incoming = 0.0;
  if (tid < bw) { const int kap = bw, rho = 1 + tid; incoming = kap < nI ? Ab[(long long)kap * bw + (kap - rho)] : (rho == kap ? 1.0 : 0.0); }
  for (int j = 0; j * d < nI; ++j) {
    for (int q = 0; q < d; ++q) {
      const int c = j * d + q;
      double next_in = 0.0;
      if (tid < bw) {             // row c of U: the pivot's row of the window, scaled
        const int kap = c + 1 + bw, rho = c + 2 + tid;
        next_in = kap < nI ? Ab[(long long)kap * bw + (kap - rho)] : (rho == kap ? 1.0 : 0.0);
        const double piv = Wn[cs * bw + cs];
        double sq = 1.0;
        if (!(piv > 0.0) || !isfinite(piv)) { if (tid == 0) atomicOr(flag, 1); } else sq = sqrt(piv);
        const int col = c + tid;
        int cm = cs + tid; if (cm >= bw) cm -= bw;
        const double u = col >= nI ? 0.0 : (tid == 0 ? sq : Wn[cs * bw + cm] / sq);
        Ur[cs * bw + tid] = u;
        if (col < nI) Ab[(long long)col * bw + tid] = u;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < PER; ++t) {   // trailing update: (c + i, c + jj)
        if (pj[t]) {
          int ri = cs + pi[t]; if (ri >= bw) ri -= bw;
          int rj = cs + pj[t]; if (rj >= bw) rj -= bw;
          Wn[ri * bw + rj] -= Ur[cs * bw + pi[t]] * Ur[cs * bw + pj[t]];
        }
      }
      if (tid < bw) {             // column c + bw enters the slot row / column c leaves; its last row is its diagonal element
        int rm = cs + 1 + tid; if (rm >= bw) rm -= bw;
        Wn[rm * bw + cs] = incoming;
        incoming = next_in;
      }
      __syncthreads();
      if (++cs == bw) cs = 0;
    }
    // the block of W these d columns complete: w(c) = (a(c) - sum_{k < c} w(k) U(k, c)) / U(c, c), lane = row of W
    const int c0 = j * d, k0 = max(0, c0 - rw);
    int c0s = cs - d; if (c0s < 0) c0s += bw;          // c0 % bw  (cs is (c0 + d) % bw here)
    const int k0r = k0 % rw, k0b = k0 % bw;
    for (int rho = tid; rho < nW; rho += FACTOR_THREADS) {
      double acc[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q] = q < d ? Wt[(long long)(c0 + q) * ldW + rho] : 0.0;
      int kr = k0r, kb = k0b;
      if (d == 6 && bw == 36 && c0 >= 30) {
        // the common shape, unrolled: 30 earlier columns, their LDS reads issued together
#pragma unroll
        for (int kk = 0; kk < 30; ++kk) {
          const double wk = ring[kr * ldW + rho];
          const double* ur = Ur + kb * 36 + (30 - kk);
#pragma unroll
          for (int q = 0; q < 6; ++q) acc[q] -= wk * ur[q];
          if (++kr == 30) kr = 0;
          if (++kb == 36) kb = 0;
        }
      } else {
        for (int k = k0; k < c0; ++k) {
          const double wk = ring[kr * ldW + rho];
          const double* ur = Ur + kb * bw + (c0 - k);
#pragma unroll
          for (int q = 0; q < 6; ++q) if (q < d) acc[q] -= wk * ur[q];
          if (++kr == rw) kr = 0;
          if (++kb == bw) kb = 0;
        }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (q < d) {
#pragma unroll
          for (int q2 = 0; q2 < 6; ++q2) {
            if (q2 < q) { int s2 = c0s + q2; if (s2 >= bw) s2 -= bw; acc[q] -= acc[q2] * Ur[s2 * bw + (q - q2)]; }
          }
          int sq = c0s + q; if (sq >= bw) sq -= bw;
          acc[q] /= Ur[sq * bw];
        }
      }
      int cr = c0 % rw;
#pragma unroll
      for (int q = 0; q < 6; ++q)
        if (q < d) { Wt[(long long)(c0 + q) * ldW + rho] = acc[q]; ring[cr * ldW + rho] = acc[q]; if (++cr == rw) cr = 0; }
    }
    __syncthreads();
  }
}

// S -= W W^T (upper triangle, through the part's scatter table), bS -= W y.  grid: 8 ceil(parts / 8) tile pairs
__global__ __launch_bounds__(256) void k_schur_syrk(const PartDev* __restrict__ parts, const double* __restrict__ W,
                                                    const SchurPlan::Scatter* __restrict__ scatter, const int* __restrict__ browBlk,
                                                    const int* __restrict__ browOff, const int* __restrict__ browCol,
                                                    double* __restrict__ S, double* __restrict__ bS, int tilePairs, int nParts) {
  // workgroup n runs on XCD n % 8 (round-robin dispatch), and every tile pair of a part reads that part's whole W: the
  // pairs of one part are given to one XCD, so W comes from HBM once and from that XCD's L2 afterwards
  const int n = blockIdx.x, slot = n >> 3, part = (n & 7) + 8 * (slot / tilePairs);
  if (part >= nParts) return;
  const PartDev pt = parts[part];
  const int nW = pt.nB + 1, nT = (nW + 63) >> 6;
  int tb = 0, rem = slot % tilePairs;     // tile pair (ta <= tb), numbered column by column
  while (tb < nT && rem > tb) { rem -= tb + 1; ++tb; }
  if (tb >= nT) return;
  const int ta = rem, a0 = ta << 6, b0 = tb << 6;
  constexpr int KP = 16;     // depth of the LDS panels (32: slower)
  __shared__ double As[KP][65], Bs[KP][65];
  const double* Wt = W + pt.wOff;
  const int tid = threadIdx.x, ldW = pt.ldW;
  // fp64 matrix cores: each wave owns a 32 x 32 quadrant of the tile as 2 x 2 products of v_mfma_f64_16x16x4 (A[i][k] and
  // B[k][j] one double per lane, lane = 16 k + i; D: row = (lane >> 4) + 4 reg, column = lane & 15).  Same peak as the vector
  // unit on this part, but an operand pair feeds 2048 flops where the 4 x 4 register tile re-read 8 LDS values per 32.
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int lane = tid & 63, wv = tid >> 6, wr = (wv >> 1) * 32, wc = (wv & 1) * 32, l16 = lane & 15, lk = lane >> 4;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < pt.nI; k0 += KP) {
#pragma unroll
    for (int i = 0; i < KP / 4; ++i) {
      const int e = tid + 256 * i, kk = e >> 6, ai = e & 63, k = k0 + kk;
      As[kk][ai] = (k < pt.nI && a0 + ai < nW) ? Wt[(long long)k * ldW + a0 + ai] : 0.0;
      Bs[kk][ai] = (k < pt.nI && b0 + ai < nW) ? Wt[(long long)k * ldW + b0 + ai] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < KP; k4 += 4) {
      const double a0v = As[k4 + lk][wr + l16], a1v = As[k4 + lk][wr + 16 + l16];
      const double b0v = Bs[k4 + lk][wc + l16], b1v = Bs[k4 + lk][wc + 16 + l16];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v, b0v, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v, b1v, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b0v, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b1v, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  const SchurPlan::Scatter* tab = scatter + pt.scatterOff;
  const int* bb = browBlk + pt.browBase;
  const int* bo = browOff + pt.browBase;
  const int* bc = browCol + pt.browBase;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + wr + 16 * i + lk + 4 * r, b = b0 + wc + 16 * j + l16;
        if (a > b || b >= nW || a >= pt.nB) continue;
        const double v = acc[i][j][r];
        if (v == 0.0) continue;
        if (b == pt.nB) { addF64(bS + bc[a], -v); continue; }
        const SchurPlan::Scatter s = tab[bb[b] * pt.nblk + bb[a]];
        addF64(S + s.dst + (long long)bo[b] * s.hd + (bo[b] * (bo[b] + 1)) / 2 + bo[a], -v);
      }
}

__global__ __launch_bounds__(256) void k_schur_place(const int* __restrict__ rcol, int nR, const double* __restrict__ xR, double* __restrict__ x) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nR) x[rcol[i]] = xR[i];
}

__global__ __launch_bounds__(256) void k_schur_back(const PartDev* __restrict__ parts, const double* __restrict__ band, const double* __restrict__ W,
                                                    const int* __restrict__ browCol, const double* __restrict__ xR, double* __restrict__ x, int bw) {
  extern __shared__ double xB[];
  const PartDev pt = parts[blockIdx.x];
  const double* Ab = band + pt.bandOff;
  const double* Wt = W + pt.wOff;
  const int* bc = browCol + pt.browBase;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nI = pt.nI, nB = pt.nB, ldW = pt.ldW;
  for (int rho = tid; rho < nB; rho += 256) xB[rho] = xR[bc[rho]];
  __syncthreads();
  for (int c = wave; c < nI; c += 4) {     // t(c) = y(c) - sum_rho W(rho, c) x_B(rho)
    const double* wc = Wt + (long long)c * ldW;
    double s = 0.0;
    for (int rho = lane; rho < nB; rho += 64) s += wc[rho] * xB[rho];
    s = waveSum(s);
    if (lane == 0) x[pt.c00 + c] = wc[nB] - s;
  }
  __threadfence();
  __syncthreads();
  if (wave != 0) return;
  double xk = 0.0;                         // lane l holds x(c + 1 + l)
  // the loads of step c - 1 (its row of U, its diagonal, its t) are issued before the reduction of step c: they do not depend on it
  auto rowU = [&](int c) { const int k = c + 1 + lane; return (c >= 0 && lane < bw - 1 && k < nI) ? Ab[(long long)k * bw + (k - c)] : 0.0; };
  double u = rowU(nI - 1), dg = Ab[(long long)(nI - 1) * bw], t = x[pt.c00 + nI - 1];
  for (int c = nI - 1; c >= 0; --c) {
    const double un = rowU(c - 1);
    const double dn = c > 0 ? Ab[(long long)(c - 1) * bw] : 1.0;
    const double tn = c > 0 ? x[pt.c00 + c - 1] : 0.0;
    const double s = waveSum(u * xk);
    const double xc = (t - s) / dg;
    const double up = __shfl_up(xk, 1);
    xk = lane == 0 ? xc : up;
    if (lane == 0) x[pt.c00 + c] = xc;
    u = un; dg = dn; t = tn;
  }
}

template <class T> struct DBuf {
  T* p = nullptr; size_t n = 0;
  ~DBuf() { if (p) (void)hipFree(p); }
  void alloc(size_t count) {
    if (p) { (void)hipFree(p); p = nullptr; }
    n = count;
    if (count && hipMalloc((void**)&p, sizeof(T) * count) != hipSuccess) throw std::runtime_error("hipMalloc failed in the factorisation's GPU stage");
  }
  void upload(const std::vector<T>& v) {
    alloc(v.size());
    if (!v.empty() && hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy failed");
  }
};
inline void hipDo(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
double nowSeconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int schurPartsFor(int band_blocks) {
  if (const char* v = std::getenv("MC_SCHUR_PARTS")) return std::max(1, std::atoi(v));
  return std::max(1, std::min(512, band_blocks / 90));
}

struct SchurSolver::Impl {
  SchurPlan plan;
  BlockSparseCholesky chol;
  Info info;
  std::string why;
  bool planned = false;
  int device = 0;
  DBuf<SchurPlan::CopyPair> copy;
  DBuf<SchurPlan::CheckPair> check;
  DBuf<SchurPlan::BandPair> band;
  DBuf<SchurPlan::WPair> wpairs;
  DBuf<SchurPlan::Scatter> scatter;
  DBuf<int> browBlk, browOff, browCol, rcol, flag;
  DBuf<PartDev> parts;
  DBuf<double> S, bS, W, Ab, xR, x;
  double* hS = nullptr;      // pinned: values of the reduced system, then its right-hand side
  double* hx = nullptr;      // pinned: reduced solution
  int maxTilePairs = 0, factorLds = 0;
  ~Impl() { if (hS) (void)hipHostFree(hS); if (hx) (void)hipHostFree(hx); }
};

SchurSolver::SchurSolver() : p_(new Impl()) {}
SchurSolver::~SchurSolver() {}
const SchurSolver::Info& SchurSolver::info() const { return p_->info; }
const std::string& SchurSolver::why() const { return p_->why; }
bool SchurSolver::planned() const { return p_->planned; }

bool SchurSolver::plan(int device, int n, const int* Cp, const int* Ci, int n_blocks, const int* block_base, const int* spline_of,
                       const double* when, const double* Cx_host, int parts, int host_threads) {
  Impl& m = *p_;
  m.planned = false;
  m.device = device;
  const double t0 = nowSeconds();
  m.plan.build(n, Cp, Ci, n_blocks, block_base, spline_of, Cx_host, parts);
  if (!m.plan.ok) { m.why = m.plan.why; return false; }
  const SchurPlan& pl = m.plan;
  int maxLd = 0;
  for (const SchurPlan::Part& pt : pl.parts) maxLd = std::max(maxLd, pt.ldW);
  m.factorLds = (int)sizeof(double) * (2 * pl.bw * pl.bw + (pl.bw - pl.d) * maxLd);
  if (m.factorLds > 150 * 1024) { m.why = "an interior's boundary is too large for the elimination kernel's LDS ring"; return false; }
  if (pl.d > 6) { m.why = "block dimension above 6"; return false; }
  hipDo(hipSetDevice(device), "hipSetDevice");
  m.copy.upload(pl.copy); m.check.upload(pl.check); m.band.upload(pl.band); m.wpairs.upload(pl.wpairs); m.scatter.upload(pl.scatter);
  m.browBlk.upload(pl.browBlk); m.browOff.upload(pl.browOff); m.browCol.upload(pl.browCol); m.rcol.upload(pl.rcol);
  std::vector<PartDev> pd(pl.parts.size());
  m.maxTilePairs = 0;
  for (size_t p = 0; p < pl.parts.size(); ++p) {
    const SchurPlan::Part& pt = pl.parts[p];
    pd[p] = PartDev{pl.A.base[pt.i0], pt.nI, pt.nB, pt.ldW, (int)pt.B.size(), 0, pt.wOff, pt.bandOff, pl.scatterOff[p], pl.browBase[p]};
    const int nT = (pt.nB + 1 + 63) / 64;
    m.maxTilePairs = std::max(m.maxTilePairs, nT * (nT + 1) / 2);
  }
  m.parts.upload(pd);
  m.flag.alloc(2);
  const size_t nnzR = (size_t)pl.R.nnz();
  m.S.alloc(nnzR); m.bS.alloc(pl.R.n); m.W.alloc((size_t)pl.wTotal); m.Ab.alloc((size_t)pl.bandTotal); m.xR.alloc(pl.R.n); m.x.alloc(n);
  if (m.hS) { (void)hipHostFree(m.hS); m.hS = nullptr; }
  if (m.hx) { (void)hipHostFree(m.hx); m.hx = nullptr; }
  hipDo(hipHostMalloc((void**)&m.hS, sizeof(double) * (nnzR + pl.R.n), hipHostMallocDefault), "hipHostMalloc(reduced system)");
  hipDo(hipHostMalloc((void**)&m.hx, sizeof(double) * (size_t)std::max(n, pl.R.n), hipHostMallocDefault), "hipHostMalloc(solution)");
  if (m.factorLds > 64 * 1024)
    hipDo(hipFuncSetAttribute((const void*)k_schur_factor, hipFuncAttributeMaxDynamicSharedMemorySize, m.factorLds), "hipFuncSetAttribute");
  // the reduced system on the host
  std::vector<int> CpR, CiR;
  pl.R.toScalar(CpR, CiR);
  m.chol = BlockSparseCholesky();
  m.chol.analyze(pl.R.n, CpR.data(), CiR.data(), pl.R.nb, pl.R.base.data());
  if (when) {
    std::vector<double> whenR(pl.R.nb);
    for (int i = 0; i < pl.R.nb; ++i) whenR[i] = when[pl.rblocks[i]];
    // the reduced system is a sixth of the assembled one: MC_SCHUR_HOST_THREADS overrides the host factorisation's thread
    // count for it (more parts mean a larger serial top)
    if (const char* v = std::getenv("MC_SCHUR_HOST_THREADS")) host_threads = std::max(1, std::atoi(v));
    m.chol.setBlockCoordinates(whenR.data(), host_threads);
  }
  Info& in = m.info;
  in = Info();
  in.parts = (int)pl.parts.size(); in.r = pl.r; in.d = pl.d; in.n = n; in.reduced_n = pl.R.n;
  in.nnz = pl.A.nnz(); in.reduced_nnz = pl.R.nnz(); in.watched_pairs = (int)pl.check.size();
  in.max_boundary = pl.maxNB; in.max_interior = pl.maxNI;
  in.work_bytes = 8.0 * ((double)pl.wTotal + (double)pl.bandTotal + (double)nnzR);
  in.plan_seconds = nowSeconds() - t0;
  m.planned = true;
  return true;
}

int SchurSolver::solve(const double* d_Cx, const double* d_g, double shift, double* x_host, void* stream) {
  Impl& m = *p_;
  if (!m.planned) throw std::logic_error("SchurSolver::solve before plan");
  const SchurPlan& pl = m.plan;
  hipStream_t st = (hipStream_t)stream;
  hipDo(hipSetDevice(m.device), "hipSetDevice");
  const int P = (int)pl.parts.size();
  const size_t nnzR = (size_t)pl.R.nnz();
  double t0 = nowSeconds();
  auto lap = [&](int slot) {
    hipDo(hipGetLastError(), "kernel launch (factorisation, GPU stage)");
    hipDo(hipStreamSynchronize(st), "hipStreamSynchronize");
    const double t = nowSeconds();
    m.info.seconds[slot] = t - t0; t0 = t;
  };
  for (double& v : m.info.seconds) v = 0.0;
  auto blocks8 = [](size_t pairs) { return dim3((unsigned)((pairs * 16 + 255) / 256)); };
  hipDo(hipMemsetAsync(m.flag.p, 0, 2 * sizeof(int), st), "hipMemsetAsync");
  hipDo(hipMemsetAsync(m.S.p, 0, sizeof(double) * nnzR, st), "hipMemsetAsync");
  hipDo(hipMemsetAsync(m.W.p, 0, sizeof(double) * (size_t)pl.wTotal, st), "hipMemsetAsync");
  hipDo(hipMemsetAsync(m.Ab.p, 0, sizeof(double) * (size_t)pl.bandTotal, st), "hipMemsetAsync");
  if (!pl.check.empty()) hipLaunchKernelGGL(k_schur_check, blocks8(pl.check.size()), dim3(256), 0, st, m.check.p, (int)pl.check.size(), d_Cx, m.flag.p);
  if (!pl.copy.empty()) hipLaunchKernelGGL(k_schur_copy, blocks8(pl.copy.size()), dim3(256), 0, st, m.copy.p, (int)pl.copy.size(), d_Cx, m.S.p, shift);
  if (!pl.band.empty()) hipLaunchKernelGGL(k_schur_band, blocks8(pl.band.size()), dim3(256), 0, st, m.band.p, (int)pl.band.size(), m.parts.p, d_Cx, m.Ab.p, pl.d, pl.bw, shift);
  if (!pl.wpairs.empty()) hipLaunchKernelGGL(k_schur_w, blocks8(pl.wpairs.size()), dim3(256), 0, st, m.wpairs.p, (int)pl.wpairs.size(), m.parts.p, d_Cx, m.W.p);
  {
    const int most = std::max(pl.maxNI, pl.R.n);
    hipLaunchKernelGGL(k_schur_rhs, dim3((most + 255) / 256, P + 1), dim3(256), 0, st, m.parts.p, m.rcol.p, pl.R.n, d_g, m.W.p, m.bS.p);
  }
  lap(0);
  hipLaunchKernelGGL(k_schur_factor, dim3(P), dim3(FACTOR_THREADS), m.factorLds, st, m.parts.p, m.Ab.p, m.W.p, pl.d, pl.bw, m.flag.p);
  lap(1);
  hipLaunchKernelGGL(k_schur_syrk, dim3(8 * ((P + 7) / 8) * m.maxTilePairs), dim3(256), 0, st, m.parts.p, m.W.p, m.scatter.p, m.browBlk.p, m.browOff.p, m.browCol.p, m.S.p, m.bS.p, m.maxTilePairs, P);
  lap(2);
  int flag[2] = {0, 0};
  hipDo(hipMemcpyAsync(flag, m.flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st), "flag read-back");
  hipDo(hipMemcpyAsync(m.hS, m.S.p, sizeof(double) * nnzR, hipMemcpyDeviceToHost, st), "reduced system read-back");
  hipDo(hipMemcpyAsync(m.hS + nnzR, m.bS.p, sizeof(double) * (size_t)pl.R.n, hipMemcpyDeviceToHost, st), "reduced right-hand side read-back");
  lap(3);
  if (flag[1]) return 2;
  if (flag[0]) return 1;
  if (!m.chol.factorize(m.hS)) return 1;
  m.chol.solve(m.hS + nnzR, m.hx);
  m.info.reduced_factor_nnz = m.chol.stats().nnz_factor;
  m.info.reduced_factor_flops = m.chol.stats().flops;
  for (int i = 0; i < pl.R.n; ++i) if (!std::isfinite(m.hx[i])) return 1;
  { const double t = nowSeconds(); m.info.seconds[4] = t - t0; t0 = t; }
  hipDo(hipMemcpyAsync(m.xR.p, m.hx, sizeof(double) * (size_t)pl.R.n, hipMemcpyHostToDevice, st), "reduced solution upload");
  hipLaunchKernelGGL(k_schur_place, dim3((pl.R.n + 255) / 256), dim3(256), 0, st, m.rcol.p, pl.R.n, m.xR.p, m.x.p);
  hipLaunchKernelGGL(k_schur_back, dim3(P), dim3(256), sizeof(double) * (size_t)std::max(1, pl.maxNB), st, m.parts.p, m.Ab.p, m.W.p, m.browCol.p, m.xR.p, m.x.p, pl.bw);
  hipDo(hipMemcpyAsync(m.hx, m.x.p, sizeof(double) * (size_t)pl.A.n, hipMemcpyDeviceToHost, st), "solution read-back");
  lap(5);
  std::memcpy(x_host, m.hx, sizeof(double) * (size_t)pl.A.n);
  return 0;
}

}  // namespace multical_amd

// ------------------------------------------------------------------------------------------------ C entry points
using multical_amd::SchurSolver;
struct mc_schur {
  SchurSolver solver;
  std::string err;
};
extern "C" {

mc_schur* mc_schur_create(mc_context* ctx, const double* d_Cx, int parts) {
  static thread_local std::string create_err;
  mc_schur* h = nullptr;
  try {
    int rows = 0, n = 0, nb = 0, terms = 0, nnz = 0;
    if (!ctx || mc_dims(ctx, &rows, &n, &nb, &terms, &nnz) < 0) return nullptr;
    std::vector<int> Cp(n + 1), Ci(nnz), base(nb + 1), spl(nb);
    std::vector<double> when(nb), Cx(nnz);
    if (mc_get_pattern(ctx, Cp.data(), Ci.data()) < 0 || mc_get_block_bases(ctx, base.data()) < 0 || mc_get_block_times(ctx, when.data()) < 0 ||
        mc_get_block_splines(ctx, spl.data()) < 0) return nullptr;
    // the device that holds the system (the context's): the plan's tables and work arrays go there
    int device = 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_Cx) == hipSuccess) device = attr.device;
    else { (void)hipGetLastError(); if (hipGetDevice(&device) != hipSuccess) return nullptr; }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    if (hipMemcpy(Cx.data(), d_Cx, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    h = new mc_schur();
    if (parts <= 0) {
      int band = 0;
      std::vector<int> count;
      for (int b = 0; b < nb; ++b) if (spl[b] >= 0) { if ((int)count.size() <= spl[b]) count.resize(spl[b] + 1, 0); band = std::max(band, ++count[spl[b]]); }
      parts = multical_amd::schurPartsFor(band);
    }
    if (!h->solver.plan(device, n, Cp.data(), Ci.data(), nb, base.data(), spl.data(), when.data(), Cx.data(), parts, multical_amd::hostCholeskyThreads()))
      h->err = h->solver.why();
    return h;
  } catch (const std::exception& e) {
    if (h) { h->err = e.what(); return h; }
    return nullptr;
  }
}
void mc_schur_destroy(mc_schur* h) { delete h; }
const char* mc_schur_last_error(const mc_schur* h) { return h ? h->err.c_str() : "mc_schur_create failed"; }
int mc_schur_planned(const mc_schur* h) { return h && h->solver.planned() ? 1 : 0; }
int mc_schur_solve(mc_schur* h, const double* d_Cx, const double* d_rhs, double diag_shift, double* x_host, void* stream) {
  if (!h || !h->solver.planned()) return -1;
  try { return h->solver.solve(d_Cx, d_rhs, diag_shift, x_host, stream); }
  catch (const std::exception& e) { h->err = e.what(); return -1; }
}
int mc_schur_info(const mc_schur* h, int* counts, double* seconds) {
  if (!h || !h->solver.planned()) return -1;
  const SchurSolver::Info& in = h->solver.info();
  if (counts) {
    counts[0] = in.parts; counts[1] = in.r; counts[2] = in.d; counts[3] = in.n; counts[4] = in.reduced_n;
    counts[5] = (int)std::min<long long>(in.reduced_nnz, 0x7fffffff); counts[6] = in.watched_pairs; counts[7] = in.max_boundary;
    counts[8] = in.max_interior; counts[9] = (int)std::min<long long>(in.reduced_factor_nnz, 0x7fffffff);
  }
  if (seconds) { for (int i = 0; i < 6; ++i) seconds[i] = in.seconds[i]; seconds[6] = in.plan_seconds; seconds[7] = in.work_bytes; seconds[8] = in.reduced_factor_flops; }
  return 0;
}

}  // extern "C"
