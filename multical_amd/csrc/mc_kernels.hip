// mc_kernels.hip -- gfx950 kernels for LinearSystemSolver::evaluateError and
// BlockCholeskyLinearSystemSolver::buildSystem (aslam_backend/src/LinearSystemSolver.cpp:12-92,
// BlockCholeskyLinearSystemSolver.cpp:58-72) plus the state update, and the C ABI of include/multical_hip.h.
//
// Execution shape (DESIGN.md section 4 has the numbers).  One launch per term family and pass:
//   F1 reprojection   k_obs_common (lane per observation: what its corners share, and the 6 x 19 matrix of the chain rule
//                     below the camera) -> k_reproj_eval / k_reproj_build: workgroup per image, wave per observation,
//                     lane per corner; build reduces the 7 x 7 Gram of [a, a x p_c | residual] on the fp64 matrix cores
//                     and expands it once per observation
//   F2 static views   k_static_common -> k_static: workgroup per run of observations of one camera
//   F3/F4/F5 streams  k_stream: workgroup per chunk of consecutive time-sorted terms; rows in an LDS tile, every Hessian
//                     entry of the chunk's window gathered out of LDS by one thread
//   F6, F7            k_small, k_motion
// Each distinct Hessian entry of a work item costs ONE fp64 atomic, lanes walking down a CCS column so the adds of a wave
// instruction are contiguous; entries between global design variables (written by every work item) go to replicas that
// k_rep_reduce folds in.  The value array the atomics land in *is* the fillCCS(upper) array -- there is no block matrix
// to convert afterwards.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <memory>
#include <string>
#include <type_traits>
#include "../../include/multical_hip.h"
#include "mc_describe.hpp"

namespace mc {

#define HIP_CHECK(expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

enum { STATUS_TIME_OUT_OF_SPLINE = 1, STATUS_BUFFER_EXCEEDED = 2, STATUS_INTERNAL = 4 };

struct DevCommon {
  const double* x;
  const int* fp_data;
  const int* fp_off;
  const int* col_h;
  const int* ofp_data;   // other-block footprints of the chunk kernels
  const int* ofp_off;
  const int* pat_colptr; // block pattern (CSR by block column): row blocks, scalar row offsets, first value of the column
  const int* pat_rows;
  const int* pat_roff;
  const int* pat_cp0;
  const int* chunk_tab;  // per-chunk slot tables (Problem::chunk_tab)
  // replicas of the contended global x global entries (Problem::gcb; null / 0 when disabled)
  const int* gcb;
  double* rep;           // REP_COPIES x rep_stride: dense upper triangle over the compact global columns
  double* rep_rhs;       // REP_COPIES x rep_cols
  int rep_stride, rep_cols;
  double* Cx;
  double* rhs;
  double* e;        // may be null
  double* chi2;     // per term raw chi^2 (global term numbering)
  double* Jpart;    // one partial cost per wave
  int* status;
  int useM;
  int build;
  unsigned long long* phase;   // optional (mc_phase_cycles): per-phase cycle sums written by wave 0 of each workgroup
};
// phase timing marks: compiled in only with -DMC_PHASE_TIMING (a separate diagnostic build, scripts/build_timing_lib.sh):
// even a disabled mark costs registers in kernels that already sit at the occupancy limit
// Diagnostic sub-modes of a build pass (2: everything but the atomics, 3: stop before the gather / flush) exist in the
// same diagnostic build only; in the product MC_MODE is the constant 1 and the branches below fold away.
#ifdef MC_PHASE_TIMING
#define MC_MODE(dc) ((dc).build)
#else
#define MC_MODE(dc) 1
#endif
#ifdef MC_EXP_NOATOM
#undef MC_MODE
#define MC_MODE(dc) 2
#endif
#ifdef MC_PHASE_TIMING
#define MC_MARK(slot)                                                                      \
  do {                                                                                     \
    if (dc.phase && threadIdx.x == 0) {                                                    \
      unsigned long long _t = clock64();                                              \
      atomicAdd(dc.phase + (slot), _t - _tmark);                                           \
      _tmark = _t;                                                                         \
    }                                                                                      \
  } while (0)
#define MC_MARK_BEGIN() unsigned long long _tmark = (dc.phase && threadIdx.x == 0) ? clock64() : 0ull
#else
#define MC_MARK(slot) do { } while (0)
#define MC_MARK_BEGIN() do { } while (0)
#endif

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
// orders a wave's own LDS traffic: its lanes wrote, now any of its lanes may read (no workgroup barrier involved)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// replica of this workgroup for the contended entries (gi <= gj compact global columns)
__device__ __forceinline__ double* rep_entry(const DevCommon& dc, int gi, int gj) {
  return dc.rep + (size_t)(blockIdx.x & (Problem::REP_COPIES - 1)) * dc.rep_stride + (gj * (gj + 1)) / 2 + gi;
}
__device__ __forceinline__ double* rep_rhs_entry(const DevCommon& dc, int g) {
  return dc.rep_rhs + (size_t)(blockIdx.x & (Problem::REP_COPIES - 1)) * dc.rep_cols + g;
}

// decoded footprint
struct FpView {
  int nb, C, coef_lo, nbuf;
  const int* rolecol;
  const int* blocks;   // nb x (block index, local col start, dim, global column base)
  const int* slots;    // nb(nb+1)/2
  const int* colblk;   // C
  const int* h;        // nb: col_h of each block
};
// d points at the footprint table (global memory, or a copy staged in LDS)
__device__ __forceinline__ FpView fp_view_at(const int* d) {
  FpView v;
  v.nb = d[0]; v.C = d[1]; v.coef_lo = d[2]; v.nbuf = d[3];
  v.rolecol = d + 4;
  v.blocks = d + FP_HDR;
  v.slots = v.blocks + 4 * v.nb;
  v.colblk = v.slots + (v.nb * (v.nb + 1)) / 2;
  v.h = v.colblk + v.C;
  return v;
}
__device__ __forceinline__ FpView fp_view(const DevCommon& dc, int fp) { return fp_view_at(dc.fp_data + dc.fp_off[fp]); }
// value index of Hessian entry (local columns ca <= cb) in the CCS array:
// slot(ia, ib) + rb * h(ib) + rb (rb + 1) / 2 + ra     (see Problem::finalize for the layout)
__device__ __forceinline__ int entry_addr(const FpView& f, const int*, int ca, int cb) {
  int ia = f.colblk[ca], ib = f.colblk[cb];
  int ra = ca - f.blocks[4 * ia + 1], rb = cb - f.blocks[4 * ib + 1];
  return f.slots[(ib * (ib + 1)) / 2 + ia] + rb * f.h[ib] + (rb * (rb + 1)) / 2 + ra;
}
__device__ __forceinline__ int global_col(const FpView& f, int ca) {
  int ia = f.colblk[ca];
  return f.blocks[4 * ia + 3] + (ca - f.blocks[4 * ia + 1]);
}

// ---------------------------------------------------------------------------------------------
// F1: one wave per image observation.  Every corner of the image shares the spline sample, so the
// per-corner work is a 2 x 19 "core" row pair; the wave reduces the 20 x 20 core Gram matrix (column 19 is
// the weighted residual) and expands it to the 6 live coefficient blocks with the basis weights.
// F1 pre-pass: one LANE per image observation computes what all its corners share (spline sample at t + d, the
// transformation chain prefixes).  Done inside k_reproj this would be repeated by every lane of the workgroup -- as
// many flops as the corners themselves.
// fp_data / fp_off / status (optional, the split build path): the observation's live support is checked against the buffer
// frozen at construction here, once, and a failing observation is marked not ok for the kernels behind
// what the owner tiles of k_reproj_assemble need of an observation besides G and the chain matrix, in one 64-byte line
struct F1Lite { double w[S6]; int seg, ok; double pad; };
__global__ __launch_bounds__(64) void k_obs_common(const ReprojBatchView* __restrict__ views, const double* __restrict__ x,
                                                   const GroupRec* __restrict__ groups, int n, F1Common* __restrict__ out,
                                                   F1Chain* __restrict__ chain, const int* __restrict__ fp_data = nullptr,
                                                   const int* __restrict__ fp_off = nullptr, int* __restrict__ status = nullptr,
                                                   F1Lite* __restrict__ lite = nullptr) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const GroupRec gr = groups[g];
  F1Common cm;
  f1_common(views[gr.batch], x, gr.aux, cm);
  if (status) {
    if (!cm.ok) atomicOr(status, STATUS_TIME_OUT_OF_SPLINE);
    else {
      // live support must lie inside the buffer frozen at construction (BSplineExpressions.cpp:447,503)
      const int* f = fp_data + fp_off[gr.fp];
      const int shift = cm.seg - f[2];
      if (f[4 + F1_ROLE_COEF] >= 0 && (shift < 0 || shift + S6 > f[3])) { atomicOr(status, STATUS_BUFFER_EXCEEDED); cm.ok = 0; }
    }
  }
  out[g] = cm;
  if (lite) {
    F1Lite l;
#pragma unroll
    for (int i = 0; i < S6; ++i) l.w[i] = cm.w[i];
    l.seg = cm.seg; l.ok = cm.ok; l.pad = 0.0;
    if (cm.seg < 0) {   // time outside the spline: keep the list ordered for k_f1_ranges (before the start / past the end)
      const ReprojBatchView& b = views[gr.batch];
      const double t = b.obs_time[gr.aux] + (b.toff.off >= 0 ? x[b.toff.off] : 0.0);
      l.seg = t < b.sp.t_min ? -(1 << 30) : (1 << 30);
    }
    lite[g] = l;
  }
  if (chain && cm.ok) f1_chain_matrix(views[gr.batch], cm, chain[g]);   // build passes only
}

// consecutive observations (groups) of one camera taken at the same time: one image seeing several boards
struct ImageRec { int first, count; };

// Expansion of a 20 x 20 core Gram (Msh, upper triangle, idx = b(b+1)/2 + a; column 19 = residual) to Hessian entries.
// "Live" columns: 0..ncoef-1 are the 6 live coefficient blocks (core columns 0..5 fanned out with the basis weights),
// then the global columns glc[] / core columns gk[], ordered so that the n_nb columns that do NOT belong to the board
// pose come first.  board_part = false: all pairs among the first M = ncoef + n_nb live columns (once per image);
// board_part = true: every pair with a board column (once per observation).
__device__ __forceinline__ void f1_live(int p, int ncoef, int rc0, int coef_shift, const double* w, const int* glc, const int* gk,
                                        int& lc, int& k, double& wt) {
  if (p < ncoef) { int i = p / 6; k = p - 6 * i; lc = rc0 + 6 * (coef_shift + i) + k; wt = w[i]; }
  else { lc = glc[p - ncoef]; k = gk[p - ncoef]; wt = 1.0; }
}
__device__ __forceinline__ void f1_flush(const FpView& f, const DevCommon& dc, const double* Msh, const double* w, int rc0,
                                         int coef_shift, const int* glc, const int* gk, const int* ggc, int n_nb, int nglob,
                                         bool board_part, int tid, int nthr = 256) {
#ifdef MC_F1_NOFLUSH
  if (dc.useM >= 0) return;   // experiment: what the kernel costs without the expansion of M to Hessian entries
#endif
  const int ncoef = rc0 >= 0 ? 36 : 0;
  const int nlive = ncoef + nglob, M = ncoef + n_nb;
  const int npair = board_part ? (nlive - M) * nlive : M * (M + 1) / 2;
  for (int idx = tid; idx < npair; idx += nthr) {
    int p, q;
    if (board_part) { q = M + idx / nlive; p = idx - (q - M) * nlive; if (p > q) continue; }
    else {
      q = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
      while ((q + 1) * (q + 2) / 2 <= idx) ++q;
      while (q * (q + 1) / 2 > idx) --q;
      p = idx - q * (q + 1) / 2;
    }
    int lcp, kp_, lcq, kq_;
    double wp, wq;
    f1_live(p, ncoef, rc0, coef_shift, w, glc, gk, lcp, kp_, wp);
    f1_live(q, ncoef, rc0, coef_shift, w, glc, gk, lcq, kq_, wq);
    int ka = min(kp_, kq_), kb = max(kp_, kq_);
    double v = wp * wq * Msh[kb * (kb + 1) / 2 + ka];
    if (p >= ncoef) {   // then q >= ncoef too: both columns belong to global design variables, entries every image writes
      const int gp = ggc[p - ncoef], gq = ggc[q - ncoef];
      if (gp >= 0 && gq >= 0) {
        if (MC_MODE(dc) == 1 || v == 1.2345e-300) atomic_add_f64(rep_entry(dc, min(gp, gq), max(gp, gq)), v);
        continue;
      }
    }
    int ca = min(lcp, lcq), cb = max(lcp, lcq);
    int addr = entry_addr(f, dc.col_h, ca, cb);
    if (MC_MODE(dc) == 1 || v == 1.2345e-300) atomic_add_f64(dc.Cx + addr, v);
  }
  const int p0 = board_part ? M : 0, p1 = board_part ? nlive : M;
  for (int p = p0 + tid; p < p1; p += nthr) {
    int lcp, kp_;
    double wp;
    f1_live(p, ncoef, rc0, coef_shift, w, glc, gk, lcp, kp_, wp);
    double v = -wp * Msh[19 * 20 / 2 + kp_];
    const int gp = p >= ncoef ? ggc[p - ncoef] : -1;
    if (MC_MODE(dc) == 1 || v == 1.2345e-300) atomic_add_f64(gp >= 0 ? rep_rhs_entry(dc, gp) : dc.rhs + global_col(f, lcp), v);
  }
}

// evaluateError for F1: one 256-thread workgroup per IMAGE, one WAVE per observation (the four waves take the image's
// observations in turn, no workgroup barrier), lane = corner.  The per-observation record is staged in the wave's LDS slot
// and read back by broadcast; the kernel streams y / p / sinv, projects and stores chi^2 (and e).
template <bool EQUI, bool EXPW>
__global__ __launch_bounds__(256) void k_reproj_eval(const ReprojBatchView* __restrict__ views, DevCommon dc,
                                                     const GroupRec* __restrict__ groups, const ImageRec* __restrict__ images,
                                                     const F1Common* __restrict__ common, int jpart_base) {
  constexpr int NCM = (int)(sizeof(F1Common) / 8);
  static_assert(sizeof(F1Common) % 8 == 0, "F1Common is copied as doubles");
  __shared__ double red[4];
  __shared__ double cm_all[4][NCM];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const F1Common& cm = *reinterpret_cast<const F1Common*>(cm_all[wv]);
  const ImageRec im = images[blockIdx.x];
  double Jacc = 0.0;
  for (int go = wv; go < im.count; go += 4) {
    const int gi = im.first + go;
    const GroupRec gr = groups[gi];
    const ReprojBatchView b = views[gr.batch];
    wave_sync();   // the previous observation is done with the record
    {
      const double* src = reinterpret_cast<const double*>(common + gi);
      for (int i = lane; i < NCM; i += 64) cm_all[wv][i] = src[i];
    }
    wave_sync();
    if (!cm.ok) { if (lane == 0) atomicOr(dc.status, STATUS_TIME_OUT_OF_SPLINE); continue; }
    {
      // live support must lie inside the buffer frozen at construction (BSplineExpressions.cpp:447,503)
      const FpView f = fp_view_at(dc.fp_data + dc.fp_off[gr.fp]);
      const int shift = cm.seg - f.coef_lo;
      if (f.rolecol[F1_ROLE_COEF] >= 0 && (shift < 0 || shift + S6 > f.nbuf)) { if (lane == 0) atomicOr(dc.status, STATUS_BUFFER_EXCEEDED); continue; }
    }
    for (int k = lane; k < gr.term_count; k += 64) {
      const int term = gr.term_first + k;
      double pc[3], Jc[6], e[2], sw;
      TermOut out;
      f1_project<EQUI, EXPW>(b, dc.x, cm, term, dc.useM != 0, false, pc, Jc, e, sw, out);
      Jacc += out.w * out.chi2;
      dc.chi2[b.term_base_global + term] = out.chi2;
      if (dc.e) { dc.e[b.row_base + 2 * term] = -e[0]; dc.e[b.row_base + 2 * term + 1] = -e[1]; }
    }
  }
  Jacc = wave_sum(Jacc);
  if (lane == 0) red[wv] = Jacc;
  __syncthreads();
  if (tid == 0) dc.Jpart[jpart_base + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// buildSystem for F1: one WAVE per observation (the four waves of a workgroup take the observations of one
// image in turn), no workgroup barrier inside the observation loop.  A lane projects one corner per pass and parks its two
// 7-vectors [a, a x p_c | residual] in the wave's private z tile; the wave reduces their 7 x 7 Gram on the fp64 matrix
// cores, applies the observation's chain matrix (M = Kt^T G Kt, 20 x 20), flushes the board rows / columns itself and adds
// the rest into the image's accumulator in LDS; after the single barrier the whole workgroup flushes the image-level part.
// (The previous shape -- a workgroup per observation, ~9 barriers per observation with half the lanes idle -- was 10 % slower.)
#ifndef MC_RB_OCC
#define MC_RB_OCC 3
#endif
// EQUI / EXPW (both F1 kernels): whether any camera of the launch has the equidistant distortion (atan) / any batch the Blake-Zisserman
// weight (exp).  Without them the kernel fits its register budget; with them the compiler parks their polynomial
// constants in scratch (188 bytes per lane, written by every thread: 460 MB of HBM writes per launch on configs[3]).
template <bool EQUI, bool EXPW>
__global__ __launch_bounds__(256, MC_RB_OCC) void k_reproj_build(const ReprojBatchView* __restrict__ views, DevCommon dc,
                                                         const GroupRec* __restrict__ groups, const ImageRec* __restrict__ images,
                                                         const F1Common* __restrict__ common, const F1Chain* __restrict__ chain,
                                                         int jpart_base) {
  enum { ZW = 2 * 64 + 4,      // 132: column stride of a wave's z tile (2 * 132 mod 64 = 8: the 7 columns x 4 rows of an MFMA step hit distinct banks)
         FPW = 512,            // footprint ints staged per wave (longer ones are read from global memory)
         O_STAGE = 0, O_G = 128, O_T = 192, O_M = 336 };   // after the MFMA the z tile is reused: accumulators, G, T, M
  constexpr int NCM = (int)(sizeof(F1Common) / 8), NCK = 6 * F1_CORE;
  static_assert(O_M + 210 <= 7 * ZW && NCM <= 128 && NCK <= 128, "wave tile layout");
  __shared__ double zs_all[4][7 * ZW];
  __shared__ double cm_all[4][NCM];
  __shared__ double Ks_all[4][NCK];
  __shared__ int fpl_all[4][FPW];
  __shared__ int gl_all[4][3][13];
  __shared__ double Mimg[210];
  __shared__ double red[4];
  __shared__ int last_info[6];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double* zs = zs_all[wv];
  const F1Common& cm = *reinterpret_cast<const F1Common*>(cm_all[wv]);
  double* Ks = Ks_all[wv];
  int* fpl = fpl_all[wv];
  int* glc = gl_all[wv][0]; int* gk = gl_all[wv][1]; int* ggc = gl_all[wv][2];
  const ImageRec im = images[blockIdx.x];
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  for (int i = tid; i < 210; i += 256) Mimg[i] = 0.0;
  if (tid < 6) last_info[tid] = 0;
  __syncthreads();
  MC_MARK_BEGIN();
  double Jacc = 0.0;
  for (int go = wv; go < im.count; go += 4) {
    const int gi = im.first + go;
    const GroupRec gr = groups[gi];
    const ReprojBatchView b = views[gr.batch];
    const int fp_begin = dc.fp_off[gr.fp], fp_len = dc.fp_off[gr.fp + 1] - fp_begin;
    const bool fp_staged = fp_len <= FPW;
    // record, chain matrix and footprint of the observation -> the wave's LDS (uniform data, read back by broadcast)
    {
      const double* src = reinterpret_cast<const double*>(common + gi);
      for (int i = lane; i < NCM; i += 64) cm_all[wv][i] = src[i];
      for (int i = lane; i < NCK; i += 64) Ks[i] = chain[gi].K[i];
      if (fp_staged) for (int i = lane; i < fp_len; i += 64) fpl[i] = dc.fp_data[fp_begin + i];
    }
    wave_sync();
    const bool last = go == im.count - 1;
    if (!cm.ok) { if (lane == 0) atomicOr(dc.status, STATUS_TIME_OUT_OF_SPLINE); continue; }
    FpView f = fp_staged ? fp_view_at(fpl) : fp_view_at(dc.fp_data + fp_begin);
    const int rc0 = f.rolecol[F1_ROLE_COEF];
    int coef_shift = 0;
    if (rc0 >= 0) {
      // live support must lie inside the buffer frozen at construction (BSplineExpressions.cpp:447,503)
      coef_shift = cm.seg - f.coef_lo;
      if (coef_shift < 0 || coef_shift + S6 > f.nbuf) { if (lane == 0) atomicOr(dc.status, STATUS_BUFFER_EXCEEDED); continue; }
    }
    // global live columns, non-board first: q_c_b (core 6..8), t_c_b (9..11), time offset (18), then q_p_w (12..14), t_p_w (15..17)
    const int a_qcb = f.rolecol[F1_ROLE_QCB] >= 0, a_tcb = f.rolecol[F1_ROLE_TCB] >= 0, a_toff = f.rolecol[F1_ROLE_TOFF] >= 0;
    const int a_qpw = f.rolecol[F1_ROLE_QPW] >= 0, a_tpw = f.rolecol[F1_ROLE_TPW] >= 0;
    const int n_nb = 3 * a_qcb + 3 * a_tcb + a_toff;
    const int n_glob = n_nb + 3 * a_qpw + 3 * a_tpw;
    if (lane < 13) {
      int role, d, pos, kcore;
      if (lane < 3) { role = F1_ROLE_QCB; d = lane; pos = 0; kcore = 6 + d; }
      else if (lane < 6) { role = F1_ROLE_TCB; d = lane - 3; pos = 3 * a_qcb; kcore = 9 + d; }
      else if (lane == 6) { role = F1_ROLE_TOFF; d = 0; pos = 3 * a_qcb + 3 * a_tcb; kcore = 18; }
      else if (lane < 10) { role = F1_ROLE_QPW; d = lane - 7; pos = n_nb; kcore = 12 + d; }
      else { role = F1_ROLE_TPW; d = lane - 10; pos = n_nb + 3 * a_qpw; kcore = 15 + d; }
      const int rc = f.rolecol[role];
      if (rc >= 0) {
        glc[pos + d] = rc + d; gk[pos + d] = kcore;
        const int g = dc.rep ? dc.gcb[f.blocks[4 * f.colblk[rc]]] : -1;   // compact global column (contended-entry replicas)
        ggc[pos + d] = g >= 0 ? g + d : -1;
      }
    }
    f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const int c16 = lane & 15, kr = lane >> 4;
    MC_MARK(0);   // record / chain / footprint staged, live-column tables
    for (int base = 0; base < gr.term_count; base += 64) {
      const int k = base + lane;
      const int nrows = 2 * min(64, gr.term_count - base);
      if (k < gr.term_count) {
        const int term = gr.term_first + k;
        double pc[3], Jc[6], e[2], sw;
        TermOut out;
        f1_project<EQUI, EXPW>(b, dc.x, cm, term, dc.useM != 0, true, pc, Jc, e, sw, out);
        Jacc += out.w * out.chi2;
        dc.chi2[b.term_base_global + term] = out.chi2;
        if (dc.e) { dc.e[b.row_base + 2 * term] = -e[0]; dc.e[b.row_base + 2 * term + 1] = -e[1]; }
        // A0 = -J_cam through HomogeneousExpressionNodeMultiply: [a, a x p_c]; both residual rows of the corner side by side
        const double a0[3] = {-sw * Jc[0], -sw * Jc[1], -sw * Jc[2]}, a1[3] = {-sw * Jc[3], -sw * Jc[4], -sw * Jc[5]};
        double c0[3], c1[3];
        v3_cross(a0, pc, c0);
        v3_cross(a1, pc, c1);
        double2* zp = reinterpret_cast<double2*>(zs + 2 * lane);   // ZW is even: 16-byte aligned
        constexpr int Z2 = ZW / 2;
        zp[0 * Z2] = make_double2(a0[0], a1[0]); zp[1 * Z2] = make_double2(a0[1], a1[1]); zp[2 * Z2] = make_double2(a0[2], a1[2]);
        zp[3 * Z2] = make_double2(c0[0], c1[0]); zp[4 * Z2] = make_double2(c0[1], c1[1]); zp[5 * Z2] = make_double2(c0[2], c1[2]);
        zp[6 * Z2] = make_double2(e[0], e[1]);
      }
      wave_sync();
      MC_MARK(1);   // projection + z rows
      // 7 x 7 Gram Z^T Z: v_mfma_f64_16x16x4_f64 with A[i][k] = Z[k][i], B[k][j] = Z[k][j], one double per lane
      // (lane = 16 (k % 4) + column); two accumulators so consecutive steps do not wait on each other
      const int nsteps = (nrows + 3) >> 2;
      for (int st = 0; st < nsteps; st += 2) {
        const int r0 = 4 * st + kr, r1 = r0 + 4;
        const double v0 = (c16 < 7 && r0 < nrows) ? zs[c16 * ZW + r0] : 0.0;
        const double v1 = (c16 < 7 && r1 < nrows) ? zs[c16 * ZW + r1] : 0.0;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc1, 0, 0, 0);
      }
      wave_sync();   // the next pass overwrites the tile
      MC_MARK(3);   // MFMA Gram
    }
    // accumulators -> G (7 x 7) -> T = G Kt (7 x 20) -> M = Kt^T T (20 x 20 upper), Kt = [K 0; 0 1]; all inside the wave's tile.
    // C/D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * reg: rows 0..6 are regs 0 and 1.
    zs[O_STAGE + lane] = acc0[0] + acc1[0];
    zs[O_STAGE + 64 + lane] = acc0[1] + acc1[1];
    wave_sync();
    if (lane < 49) {
      const int i = lane / 7, j = lane - 7 * i;
      zs[O_G + lane] = zs[O_STAGE + (i >> 2) * 64 + (i & 3) * 16 + j];
    }
    wave_sync();
    for (int idx = lane; idx < 140; idx += 64) {
      const int i = idx / 20, c = idx - 20 * i;
      double t = zs[O_G + i * 7 + 6];
      if (c < F1_CORE) {
        t = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) t += zs[O_G + i * 7 + j] * Ks[j * F1_CORE + c];
      }
      zs[O_T + idx] = t;
    }
    wave_sync();
    for (int idx = lane; idx < 210; idx += 64) {
      int bb = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
      while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
      while (bb * (bb + 1) / 2 > idx) --bb;
      const int ea = idx - bb * (bb + 1) / 2, eb = bb;
      double m = zs[O_T + 6 * 20 + eb];
      if (ea < F1_CORE) {
        m = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) m += Ks[i * F1_CORE + ea] * zs[O_T + i * 20 + eb];
      }
      zs[O_M + idx] = m;
      const bool board_entry = (ea >= 12 && ea <= 17) || (eb >= 12 && eb <= 17);
      if (!board_entry) atomicAdd(&Mimg[idx], m);
    }
    wave_sync();
    MC_MARK(4);   // G -> T -> M
    // board rows / columns of this observation
    f1_flush(f, dc, zs + O_M, cm.w, rc0, coef_shift, glc, gk, ggc, n_nb, n_glob, true, lane, 64);
    MC_MARK(5);   // board flush
    if (last && lane == 0) {
      last_info[0] = 1; last_info[1] = rc0; last_info[2] = coef_shift; last_info[3] = n_nb; last_info[4] = fp_staged ? 1 : 0;
      last_info[5] = fp_begin;
    }
    wave_sync();   // the flush is done with the tile before the next observation overwrites it
  }
  Jacc = wave_sum(Jacc);
  if (lane == 0) red[wv] = Jacc;
  __syncthreads();
  if (tid == 0) dc.Jpart[jpart_base + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (!last_info[0]) return;
  // image-level part: everything that does not involve a board pose, through the last observation's tables, which its
  // wave left in place (same camera, same time: same coefficient buffer, extrinsics and time-offset blocks)
  const int lw = (im.count - 1) & 3;
  FpView f = last_info[4] ? fp_view_at(fpl_all[lw]) : fp_view_at(dc.fp_data + last_info[5]);
  const F1Common& cml = *reinterpret_cast<const F1Common*>(cm_all[lw]);
  MC_MARK(2);   // waiting for the other waves
  f1_flush(f, dc, Mimg, cml.w, last_info[1], last_info[2], gl_all[lw][0], gl_all[lw][1], gl_all[lw][2], last_info[3], last_info[3], false, tid, 256);
  MC_MARK(6);   // image flush
}

// buildSystem for F1, second shape: the CORNERS of an image are shared out, not its observations.  An aprilgrid
// observation has up to 144 corners, i.e. 2 - 3 passes of 64 lanes with the last one nearly empty, and an image has 1 - 6
// observations for 4 waves: with a wave per observation the critical path was 3 passes plus a chain of dependent loads
// (group -> footprint -> record) per observation.  Here
//   1. all 256 threads stage the records and chain matrices of the image's observations at once (rounds of OB);
//   2. the corners of all observations form one list; a wave takes 64 consecutive ones per pass (lane -> observation by
//      the prefix table, its record read from LDS per lane) and reduces the 7 x 7 Gram of every observation present in the
//      pass on the fp64 matrix cores, rows of other observations masked to zero, into the observation's G in LDS;
//   3. after one barrier a wave per observation forms M = Kt^T G Kt, adds the image-level part to Mimg and flushes the
//      board rows / columns, exactly as k_reproj_build does;
//   4. image-level flush by the whole workgroup.
#ifndef MC_F1_OB
#define MC_F1_OB 4
#endif
// 39.3 KB of LDS: four workgroups per CU where the registers allow it (109 VGPRs without atan / exp)
template <bool EQUI, bool EXPW>
__global__ __launch_bounds__(256, (EQUI || EXPW) ? MC_RB_OCC : 4) void k_reproj_build_img(const ReprojBatchView* __restrict__ views, DevCommon dc,
                                                         const GroupRec* __restrict__ groups, const ImageRec* __restrict__ images,
                                                         const F1Common* __restrict__ common, const F1Chain* __restrict__ chain,
                                                         int jpart_base) {
  enum { ZW = 2 * 64 + 4, FPW = 512, O_T = 0, O_M = 140, O_FP = 352,   // phase 3 reuses the wave's z tile: T, M, the staged footprint
         OB = MC_F1_OB };
  constexpr int NCM = (int)(sizeof(F1Common) / 8), NCK = 6 * F1_CORE;
  static_assert(O_M + 210 <= O_FP && O_FP + FPW / 2 <= 7 * ZW, "wave tile layout");
  __shared__ double zs_all[4][7 * ZW];
  __shared__ double cm_s[OB][NCM];
  __shared__ double K_s[OB][NCK];
  __shared__ double Gs[OB][49];
  __shared__ int gl_all[4][3][13];
  __shared__ double Mimg[210];
  __shared__ double red[4];
  __shared__ int last_info[7];
  __shared__ int pre[OB + 1], gfirst[OB], gcount[OB], gfp[OB], ook[OB];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double* zs = zs_all[wv];
  int* fpl = reinterpret_cast<int*>(zs + O_FP);
  int* glc = gl_all[wv][0]; int* gk = gl_all[wv][1]; int* ggc = gl_all[wv][2];
  const ImageRec im = images[blockIdx.x];
  const ReprojBatchView b = views[groups[im.first].batch];   // an image is one camera's: one batch
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  for (int i = tid; i < 210; i += 256) Mimg[i] = 0.0;
  if (tid < 7) last_info[tid] = 0;
  double Jacc = 0.0;
  const int c16 = lane & 15, kr = lane >> 4;
  for (int ob0 = 0; ob0 < im.count; ob0 += OB) {
    const int nob = min((int)OB, im.count - ob0);
    __syncthreads();   // Mimg zeroed / the previous round is done with the staged records
    // ---- 1. records, chain matrices, group descriptors
    for (int i = tid; i < nob * NCM; i += 256) {
      const int o = i / NCM, j = i - o * NCM;
      cm_s[o][j] = reinterpret_cast<const double*>(common + im.first + ob0 + o)[j];
    }
    for (int i = tid; i < nob * NCK; i += 256) {
      const int o = i / NCK, j = i - o * NCK;
      K_s[o][j] = chain[im.first + ob0 + o].K[j];
    }
    for (int i = tid; i < nob * 49; i += 256) (&Gs[0][0])[i] = 0.0;
    if (tid < nob) {
      const GroupRec gr = groups[im.first + ob0 + tid];
      gfirst[tid] = gr.term_first; gcount[tid] = gr.term_count; gfp[tid] = gr.fp;
    }
    __syncthreads();
    if (tid < nob) {
      const F1Common& cmo = *reinterpret_cast<const F1Common*>(cm_s[tid]);
      int ok = 1;
      if (!cmo.ok) { atomicOr(dc.status, STATUS_TIME_OUT_OF_SPLINE); ok = 0; }
      else {
        // live support must lie inside the buffer frozen at construction (BSplineExpressions.cpp:447,503)
        const FpView f = fp_view_at(dc.fp_data + dc.fp_off[gfp[tid]]);
        const int shift = cmo.seg - f.coef_lo;
        if (f.rolecol[F1_ROLE_COEF] >= 0 && (shift < 0 || shift + S6 > f.nbuf)) { atomicOr(dc.status, STATUS_BUFFER_EXCEEDED); ok = 0; }
      }
      ook[tid] = ok;
    }
    wave_sync();   // the lanes above are all in wave 0
    if (tid == 0) {
      int run = 0;
      for (int o = 0; o < nob; ++o) { pre[o] = run; run += ook[o] ? gcount[o] : 0; }
      pre[nob] = run;
    }
    __syncthreads();
    // ---- 2. corner passes
    const int ctot = pre[nob];
    for (int base = 64 * wv; base < ctot; base += 256) {
      const int c = base + lane;
      const int nrows = 2 * min(64, ctot - base);
      if (c < ctot) {
        int o = 0;
        while (o + 1 < nob && c >= pre[o + 1]) ++o;
        const int term = gfirst[o] + (c - pre[o]);
        const F1Common& cmo = *reinterpret_cast<const F1Common*>(cm_s[o]);
        double pc[3], Jc[6], e[2], sw;
        TermOut out;
        f1_project<EQUI, EXPW>(b, dc.x, cmo, term, dc.useM != 0, true, pc, Jc, e, sw, out);
        Jacc += out.w * out.chi2;
        dc.chi2[b.term_base_global + term] = out.chi2;
        if (dc.e) { dc.e[b.row_base + 2 * term] = -e[0]; dc.e[b.row_base + 2 * term + 1] = -e[1]; }
        const double a0[3] = {-sw * Jc[0], -sw * Jc[1], -sw * Jc[2]}, a1[3] = {-sw * Jc[3], -sw * Jc[4], -sw * Jc[5]};
        double c0[3], c1[3];
        v3_cross(a0, pc, c0);
        v3_cross(a1, pc, c1);
        double2* zp = reinterpret_cast<double2*>(zs + 2 * lane);
        constexpr int Z2 = ZW / 2;
        zp[0 * Z2] = make_double2(a0[0], a1[0]); zp[1 * Z2] = make_double2(a0[1], a1[1]); zp[2 * Z2] = make_double2(a0[2], a1[2]);
        zp[3 * Z2] = make_double2(c0[0], c1[0]); zp[4 * Z2] = make_double2(c0[1], c1[1]); zp[5 * Z2] = make_double2(c0[2], c1[2]);
        zp[6 * Z2] = make_double2(e[0], e[1]);
      }
      wave_sync();
      // 7 x 7 Gram per observation present in the pass (rows [ra, rb) of the tile), see k_reproj_build for the operand layout
      int o_lo = 0;
      while (o_lo + 1 < nob && base >= pre[o_lo + 1]) ++o_lo;
      for (int oo = o_lo; oo < nob && pre[oo] < base + 64; ++oo) {
        const int ra = 2 * max(pre[oo] - base, 0), rb = min(2 * (pre[oo + 1] - base), nrows);
        if (rb <= ra) continue;
        f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int st = ra >> 2; 4 * st < rb; st += 2) {
          const int r0 = 4 * st + kr, r1 = r0 + 4;
          const double v0 = (c16 < 7 && r0 >= ra && r0 < rb) ? zs[c16 * ZW + r0] : 0.0;
          const double v1 = (c16 < 7 && r1 >= ra && r1 < rb) ? zs[c16 * ZW + r1] : 0.0;
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc1, 0, 0, 0);
        }
        // C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
        if (c16 < 7) {
          atomicAdd(&Gs[oo][kr * 7 + c16], acc0[0] + acc1[0]);
          if (kr + 4 < 7) atomicAdd(&Gs[oo][(kr + 4) * 7 + c16], acc0[1] + acc1[1]);
        }
      }
      wave_sync();   // the next pass overwrites the tile
    }
    __syncthreads();   // every observation's G is complete
    // ---- 3. a wave per observation: G -> T = G Kt -> M = Kt^T T (Kt = [K 0; 0 1]), image part to Mimg, board part flushed
    for (int o = wv; o < nob; o += 4) {
      if (!ook[o]) continue;
      const int fp_begin = dc.fp_off[gfp[o]], fp_len = dc.fp_off[gfp[o] + 1] - fp_begin;
      const bool fp_staged = fp_len <= FPW;
      if (fp_staged) for (int i = lane; i < fp_len; i += 64) fpl[i] = dc.fp_data[fp_begin + i];
      wave_sync();
      const F1Common& cm = *reinterpret_cast<const F1Common*>(cm_s[o]);
      const double* Ks = K_s[o];
      const double* G = Gs[o];
      FpView f = fp_staged ? fp_view_at(fpl) : fp_view_at(dc.fp_data + fp_begin);
      const int rc0 = f.rolecol[F1_ROLE_COEF];
      const int coef_shift = rc0 >= 0 ? cm.seg - f.coef_lo : 0;
      const int a_qcb = f.rolecol[F1_ROLE_QCB] >= 0, a_tcb = f.rolecol[F1_ROLE_TCB] >= 0, a_toff = f.rolecol[F1_ROLE_TOFF] >= 0;
      const int a_qpw = f.rolecol[F1_ROLE_QPW] >= 0, a_tpw = f.rolecol[F1_ROLE_TPW] >= 0;
      const int n_nb = 3 * a_qcb + 3 * a_tcb + a_toff;
      const int n_glob = n_nb + 3 * a_qpw + 3 * a_tpw;
      if (lane < 13) {
        int role, d, pos, kcore;
        if (lane < 3) { role = F1_ROLE_QCB; d = lane; pos = 0; kcore = 6 + d; }
        else if (lane < 6) { role = F1_ROLE_TCB; d = lane - 3; pos = 3 * a_qcb; kcore = 9 + d; }
        else if (lane == 6) { role = F1_ROLE_TOFF; d = 0; pos = 3 * a_qcb + 3 * a_tcb; kcore = 18; }
        else if (lane < 10) { role = F1_ROLE_QPW; d = lane - 7; pos = n_nb; kcore = 12 + d; }
        else { role = F1_ROLE_TPW; d = lane - 10; pos = n_nb + 3 * a_qpw; kcore = 15 + d; }
        const int rc = f.rolecol[role];
        if (rc >= 0) {
          glc[pos + d] = rc + d; gk[pos + d] = kcore;
          const int g = dc.rep ? dc.gcb[f.blocks[4 * f.colblk[rc]]] : -1;
          ggc[pos + d] = g >= 0 ? g + d : -1;
        }
      }
      for (int idx = lane; idx < 140; idx += 64) {
        const int i = idx / 20, c = idx - 20 * i;
        double t = G[i * 7 + 6];
        if (c < F1_CORE) {
          t = 0.0;
#pragma unroll
          for (int j = 0; j < 6; ++j) t += G[i * 7 + j] * Ks[j * F1_CORE + c];
        }
        zs[O_T + idx] = t;
      }
      wave_sync();
      for (int idx = lane; idx < 210; idx += 64) {
        int bb = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
        while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
        while (bb * (bb + 1) / 2 > idx) --bb;
        const int ea = idx - bb * (bb + 1) / 2, eb = bb;
        double m = zs[O_T + 6 * 20 + eb];
        if (ea < F1_CORE) {
          m = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) m += Ks[i * F1_CORE + ea] * zs[O_T + i * 20 + eb];
        }
        zs[O_M + idx] = m;
        const bool board_entry = (ea >= 12 && ea <= 17) || (eb >= 12 && eb <= 17);
        if (!board_entry) atomicAdd(&Mimg[idx], m);
      }
      wave_sync();
      f1_flush(f, dc, zs + O_M, cm.w, rc0, coef_shift, glc, gk, ggc, n_nb, n_glob, true, lane, 64);
      if (ob0 + o == im.count - 1 && lane == 0) {
        last_info[0] = 1; last_info[1] = rc0; last_info[2] = coef_shift; last_info[3] = n_nb; last_info[4] = fp_staged ? 1 : 0;
        last_info[5] = fp_begin; last_info[6] = o;
      }
      wave_sync();   // the flush is done with the tile and the tables before the wave's next observation
    }
  }
  Jacc = wave_sum(Jacc);
  if (lane == 0) red[wv] = Jacc;
  __syncthreads();
  if (tid == 0) dc.Jpart[jpart_base + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (!last_info[0]) return;
  // image-level part through the last observation's tables, which its wave left in place (same camera, same time: same
  // coefficient buffer, extrinsics and time-offset blocks)
  const int lo = last_info[6], lw = lo & 3;
  FpView f = last_info[4] ? fp_view_at(reinterpret_cast<const int*>(zs_all[lw] + O_FP)) : fp_view_at(dc.fp_data + last_info[5]);
  const F1Common& cml = *reinterpret_cast<const F1Common*>(cm_s[lo]);
  f1_flush(f, dc, Mimg, cml.w, last_info[1], last_info[2], gl_all[lw][0], gl_all[lw][1], gl_all[lw][2], last_info[3], last_info[3], false, tid, 256);
}

// ---------------------------------------------------------------------------------------------
// buildSystem for F1, third shape (round 3): evaluation split from assembly, assembly by OWNER.
//   k_reproj_gram      lane = corner over the whole launch, no image structure: a wave takes 64 consecutive corners of one
//                      camera (they may belong to several observations), projects them and reduces the 7 x 7 Gram of
//                      [a, a x p_c | residual] of every observation present on the fp64 matrix cores, rows of the other
//                      observations masked to zero, into the observation's G in global memory (49 doubles, a handful of
//                      contiguous fp64 atomics per wave).  Nothing else leaves the kernel: 30 KB of LDS, no barrier in the
//                      loop, every SIMD busy -- the per-image workgroup of the second shape spent most of its time in
//                      dependent loads and barriers with 1.6 corners per lane.
//   k_f1_ranges        per camera: first observation whose spline segment (at the CURRENT time offset) reaches a tile
//                      boundary; the observations of a camera are time sorted, so a binary search per boundary.
//   k_reproj_assemble  a workgroup OWNS K consecutive pose-coefficient blocks: the band columns of those blocks and their
//                      rows in every global design variable's columns.  It pulls the observations whose live window
//                      touches its blocks (segments c0 - 5 .. c0 + K - 1; an observation is expanded by up to two
//                      owners, the expansion is ~3 k multiply-adds), forms M = Kt^T G Kt per observation, adds
//                      w_i w_j M into LDS accumulators and leaves with PLAIN read-modify-write stores: no atomic touches
//                      the coefficient band or the coefficient x global entries.  Entries between two global design
//                      variables (every observation writes them) still go to the replicas, from the observation's
//                      primary owner only.  Runs after the other families' kernels have joined (its stores are not atomic).
struct F1Pass { int batch, term_first, n_terms, obs_first, n_obs; };
// static per observation: its global live columns in core order (q_c_b, t_c_b, q_p_w, t_p_w, time offset; inactive ones
// left out): footprint-local column, core column, compact global column of the replicas (-1: none), accumulator column
struct F1ObsTab { int n_glob, has_coef, pad0, pad1; int glc[13], gk[13], ggc[13], acol[13]; };
struct F1Asm {
  int K, n_tiles, ncol, nGB, n_batches;
  int pose_block0, pose_col0;
  const int* band;    // [n_tiles][K * 6]  value offset of block pair (c0 + k - d, c0 + k), -1 if absent
  const int* hcol;    // [n_tiles][K]      col_h of the owned blocks
  const int* arrow;   // [n_tiles][nGB * K] >= 0: (pose block rows, global block column), < -1: -2 - slot with the pose block as column
  const int* col_gb;  // [ncol] global block (index into the launch's list) of an accumulator column, and its component
  const int* col_sc;
  const int* gb_h;    // [nGB] col_h of the global block
  int* ranges;        // [n_batches][2][n_tiles + 1]: first observation with segment >= j K - 5, and >= j K
  const int* gbase;   // [n_batches + 1] first group of every batch in the launch's group array
  const F1ObsTab* otab;     // per group
  const F1Lite* lite;       // per group, rewritten by k_obs_common
};

template <bool EQUI, bool EXPW>
__global__ __launch_bounds__(256, (EQUI || EXPW) ? MC_RB_OCC : 4) void k_reproj_gram(const ReprojBatchView* __restrict__ views, DevCommon dc,
                                                      const GroupRec* __restrict__ groups, const F1Pass* __restrict__ passes, int n_passes,
                                                      const F1Common* __restrict__ common, double* __restrict__ Gbuf, int jpart_base) {
  enum { ZW = 2 * 64 + 4 };
  __shared__ double zs_all[4][7 * ZW];
  __shared__ double red[4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double* zs = zs_all[wv];
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  const int pi = blockIdx.x * 4 + wv;
  double Jacc = 0.0;
  if (pi < n_passes) {
    const F1Pass pr = passes[pi];
    const ReprojBatchView b = views[pr.batch];
    const int term = pr.term_first + lane;
    bool live = lane < pr.n_terms;
    int gi = pr.obs_first;
    if (live) {
      int o = 0;
      while (o + 1 < pr.n_obs && term >= groups[pr.obs_first + o + 1].term_first) ++o;
      gi = pr.obs_first + o;
    }
    const F1Common& cm = common[gi];
    if (live && !cm.ok) live = false;   // k_obs_common has raised the status
    double2* zp = reinterpret_cast<double2*>(zs + 2 * lane);
    constexpr int Z2 = ZW / 2;
    if (live) {
      double pc[3], Jc[6], e[2], sw;
      TermOut out;
      f1_project<EQUI, EXPW>(b, dc.x, cm, term, dc.useM != 0, true, pc, Jc, e, sw, out);
      Jacc = out.w * out.chi2;
      dc.chi2[b.term_base_global + term] = out.chi2;
      if (dc.e) { dc.e[b.row_base + 2 * term] = -e[0]; dc.e[b.row_base + 2 * term + 1] = -e[1]; }
      const double a0[3] = {-sw * Jc[0], -sw * Jc[1], -sw * Jc[2]}, a1[3] = {-sw * Jc[3], -sw * Jc[4], -sw * Jc[5]};
      double c0[3], c1[3];
      v3_cross(a0, pc, c0);
      v3_cross(a1, pc, c1);
      zp[0 * Z2] = make_double2(a0[0], a1[0]); zp[1 * Z2] = make_double2(a0[1], a1[1]); zp[2 * Z2] = make_double2(a0[2], a1[2]);
      zp[3 * Z2] = make_double2(c0[0], c1[0]); zp[4 * Z2] = make_double2(c0[1], c1[1]); zp[5 * Z2] = make_double2(c0[2], c1[2]);
      zp[6 * Z2] = make_double2(e[0], e[1]);
    } else {
#pragma unroll
      for (int c = 0; c < 7; ++c) zp[c * Z2] = make_double2(0.0, 0.0);
    }
    wave_sync();
    // 7 x 7 Gram per observation present in the pass (rows [ra, rb) of the tile); operand layout as in k_reproj_build
    const int c16 = lane & 15, kr = lane >> 4;
    for (int oo = 0; oo < pr.n_obs; ++oo) {
      const GroupRec g = groups[pr.obs_first + oo];
      const int la = max(g.term_first - pr.term_first, 0), lb = min(g.term_first + g.term_count - pr.term_first, pr.n_terms);
      const int ra = 2 * la, rb = 2 * lb;
      if (rb <= ra) continue;
      f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      for (int st = ra >> 2; 4 * st < rb; st += 2) {
        const int r0 = 4 * st + kr, r1 = r0 + 4;
        const double v0 = (c16 < 7 && r0 >= ra && r0 < rb) ? zs[c16 * ZW + r0] : 0.0;
        const double v1 = (c16 < 7 && r1 >= ra && r1 < rb) ? zs[c16 * ZW + r1] : 0.0;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc1, 0, 0, 0);
      }
      // C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
      double* G = Gbuf + (size_t)(pr.obs_first + oo) * 49;
      if (c16 < 7) {
        atomic_add_f64(G + kr * 7 + c16, acc0[0] + acc1[0]);
        if (kr + 4 < 7) atomic_add_f64(G + (kr + 4) * 7 + c16, acc0[1] + acc1[1]);
      }
    }
  }
  Jacc = wave_sum(Jacc);
  if (lane == 0) red[wv] = Jacc;
  __syncthreads();
  if (tid == 0) dc.Jpart[jpart_base + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Expansion behind k_reproj_gram, atomic form: a WAVE per image (the observations of one camera at one time), no workgroup
// barrier, 7.6 KB of LDS per wave -- five workgroups per CU, so the dependent loads of an observation's record (G, chain
// matrix, footprint) of one wave hide behind the others'.  Per observation M = Kt^T G Kt; the part that does not involve the
// board pose is summed over the image's observations in the wave's LDS and flushed once per image, the board rows /
// columns per observation -- the same entries, the same fp64 atomics and replicas as k_reproj_build_img's phases 3 and 4.
__global__ __launch_bounds__(256) void k_reproj_expand(DevCommon dc, const GroupRec* __restrict__ groups, const ImageRec* __restrict__ images,
                                                       int n_images, const F1Lite* __restrict__ lite, const F1Chain* __restrict__ chain,
                                                       const double* __restrict__ Gbuf) {
  enum { FPW = 384, O_G = 0, O_K = 52, O_T = 168, O_M = 308, O_MI = 520, O_W = 732, O_FP = 738, O_GL = O_FP + FPW / 2, WS = O_GL + 20 };
  __shared__ double ws_all[4][WS];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int imi = blockIdx.x * 4 + wv;
  if (imi >= n_images) return;   // no workgroup barrier below
  double* ws = ws_all[wv];
  int* fpl = reinterpret_cast<int*>(ws + O_FP);
  int* glc = reinterpret_cast<int*>(ws + O_GL); int* gk = glc + 13; int* ggc = gk + 13;
  const ImageRec im = images[imi];
  for (int i = lane; i < 210; i += 64) ws[O_MI + i] = 0.0;
  int l_rc0 = -1, l_shift = 0, l_nnb = 0, l_fp_begin = 0;
  bool any = false, l_staged = false;
  for (int o = 0; o < im.count; ++o) {
    const int gi = im.first + o;
    const F1Lite lt = lite[gi];
    if (!lt.ok) continue;                      // k_obs_common has raised the status
    const GroupRec gr = groups[gi];
    const int fp_begin = dc.fp_off[gr.fp], fp_len = dc.fp_off[gr.fp + 1] - fp_begin;
    const bool fp_staged = fp_len <= FPW;
    wave_sync();                               // the previous observation's flush is done with the scratch
    if (lane < 49) ws[O_G + lane] = Gbuf[(size_t)gi * 49 + lane];
    for (int i = lane; i < 6 * F1_CORE; i += 64) ws[O_K + i] = chain[gi].K[i];
    if (lane < S6) ws[O_W + lane] = lt.w[lane];
    if (fp_staged) for (int i = lane; i < fp_len; i += 64) fpl[i] = dc.fp_data[fp_begin + i];
    wave_sync();
    FpView f = fp_staged ? fp_view_at(fpl) : fp_view_at(dc.fp_data + fp_begin);
    const int rc0 = f.rolecol[F1_ROLE_COEF];
    const int coef_shift = rc0 >= 0 ? lt.seg - f.coef_lo : 0;
    // global live columns, non-board first: q_c_b (core 6..8), t_c_b (9..11), time offset (18), then q_p_w (12..14), t_p_w (15..17)
    const int a_qcb = f.rolecol[F1_ROLE_QCB] >= 0, a_tcb = f.rolecol[F1_ROLE_TCB] >= 0, a_toff = f.rolecol[F1_ROLE_TOFF] >= 0;
    const int a_qpw = f.rolecol[F1_ROLE_QPW] >= 0, a_tpw = f.rolecol[F1_ROLE_TPW] >= 0;
    const int n_nb = 3 * a_qcb + 3 * a_tcb + a_toff;
    const int n_glob = n_nb + 3 * a_qpw + 3 * a_tpw;
    if (lane < 13) {
      int role, d, pos, kcore;
      if (lane < 3) { role = F1_ROLE_QCB; d = lane; pos = 0; kcore = 6 + d; }
      else if (lane < 6) { role = F1_ROLE_TCB; d = lane - 3; pos = 3 * a_qcb; kcore = 9 + d; }
      else if (lane == 6) { role = F1_ROLE_TOFF; d = 0; pos = 3 * a_qcb + 3 * a_tcb; kcore = 18; }
      else if (lane < 10) { role = F1_ROLE_QPW; d = lane - 7; pos = n_nb; kcore = 12 + d; }
      else { role = F1_ROLE_TPW; d = lane - 10; pos = n_nb + 3 * a_qpw; kcore = 15 + d; }
      const int rc = f.rolecol[role];
      if (rc >= 0) {
        glc[pos + d] = rc + d; gk[pos + d] = kcore;
        const int g = dc.rep ? dc.gcb[f.blocks[4 * f.colblk[rc]]] : -1;
        ggc[pos + d] = g >= 0 ? g + d : -1;
      }
    }
    const double* G = ws + O_G;
    const double* Ks = ws + O_K;
    for (int idx = lane; idx < 140; idx += 64) {
      const int i = idx / 20, c = idx - 20 * i;
      double v = G[i * 7 + 6];
      if (c < F1_CORE) {
        v = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) v += G[i * 7 + j] * Ks[j * F1_CORE + c];
      }
      ws[O_T + idx] = v;
    }
    wave_sync();
    for (int idx = lane; idx < 210; idx += 64) {
      int bb = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
      while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
      while (bb * (bb + 1) / 2 > idx) --bb;
      const int ea = idx - bb * (bb + 1) / 2, eb = bb;
      double m = ws[O_T + 6 * 20 + eb];
      if (ea < F1_CORE) {
        m = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) m += Ks[i * F1_CORE + ea] * ws[O_T + i * 20 + eb];
      }
      ws[O_M + idx] = m;
      const bool board_entry = (ea >= 12 && ea <= 17) || (eb >= 12 && eb <= 17);
      if (!board_entry) ws[O_MI + idx] += m;     // the wave owns the image: plain add
    }
    wave_sync();
    // board rows / columns of this observation
    f1_flush(f, dc, ws + O_M, ws + O_W, rc0, coef_shift, glc, gk, ggc, n_nb, n_glob, true, lane, 64);
    any = true; l_rc0 = rc0; l_shift = coef_shift; l_nnb = n_nb; l_staged = fp_staged; l_fp_begin = fp_begin;
  }
  if (!any) return;
  wave_sync();
  // image-level part through the last observation's tables, still in place (same camera, same time: same coefficient
  // buffer, extrinsics and time-offset blocks)
  FpView f = l_staged ? fp_view_at(fpl) : fp_view_at(dc.fp_data + l_fp_begin);
  f1_flush(f, dc, ws + O_MI, ws + O_W, l_rc0, l_shift, glc, gk, ggc, l_nnb, l_nnb, false, lane, 64);
}

// lane = observation g of the launch: every tile boundary (segment j K - 5 for the first kind, j K for the second) that
// lies between the segment of the observation before it in its camera's time-sorted list and its own gets this
// observation's position as "first observation at or past the boundary"; the last observation of a camera also closes
// the boundaries past its own segment.
__global__ __launch_bounds__(256) void k_f1_ranges(const GroupRec* __restrict__ groups, int n, F1Asm A) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int cb = groups[g].batch;
  const int g0 = A.gbase[cb], g1 = A.gbase[cb + 1];
  const int pos = g - g0, per = A.n_tiles + 1;
  const long long seg = A.lite[g].seg;
  const long long prev = g > g0 ? (long long)A.lite[g - 1].seg : -(1LL << 40);
  int* r0 = A.ranges + (size_t)(cb * 2 + 0) * per;
  int* r1 = A.ranges + (size_t)(cb * 2 + 1) * per;
  // boundaries b with prev < b <= seg: first kind b = j K - 5, second kind b = j K
  auto ceil_div = [](long long a, long long k) { return a >= 0 ? (a + k - 1) / k : -((-a) / k); };
  {
    long long j_lo = ceil_div(prev + 1 + 5, A.K), j_hi = (seg + 5 >= 0) ? (seg + 5) / A.K : -1;
    if (j_lo < 0) j_lo = 0;
    if (j_hi > A.n_tiles) j_hi = A.n_tiles;
    for (long long j = j_lo; j <= j_hi; ++j) r0[j] = pos;
  }
  {
    long long j_lo = ceil_div(prev + 1, A.K), j_hi = seg >= 0 ? seg / A.K : -1;
    if (j_lo < 0) j_lo = 0;
    if (j_hi > A.n_tiles) j_hi = A.n_tiles;
    for (long long j = j_lo; j <= j_hi; ++j) r1[j] = pos;
  }
  if (g == g1 - 1) {   // boundaries past the last observation
    long long j0 = seg + 5 >= 0 ? (seg + 5) / A.K + 1 : 0, j1 = seg >= 0 ? seg / A.K + 1 : 0;
    for (long long j = j0 < 0 ? 0 : j0; j <= A.n_tiles; ++j) r0[j] = pos + 1;
    for (long long j = j1 < 0 ? 0 : j1; j <= A.n_tiles; ++j) r1[j] = pos + 1;
  }
}

#ifndef MC_ASM_SKIP
#define MC_ASM_SKIP 0   // diagnostic builds: bit mask of phases left out (results wrong, timing meaningful)
#endif
__global__ __launch_bounds__(256) void k_reproj_assemble(DevCommon dc, const GroupRec* __restrict__ groups,
                                                         const F1Chain* __restrict__ chain, const double* __restrict__ Gbuf, F1Asm A) {
  extern __shared__ double sm[];
  // per-wave scratch (doubles): G 49, K 6 x 19, T 7 x 20, M 210, w 6, then the observation's table (54 ints)
  enum { O_G = 0, O_K = 52, O_T = 168, O_M = 308, O_W = 518, O_TAB = 524, WS = 552 };
  static_assert(sizeof(F1ObsTab) == 56 * sizeof(int) && O_TAB + 28 <= WS, "wave scratch layout");
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int K = A.K, R = 6 * K, ncol = A.ncol, nGB = A.nGB;
  double* bandAcc = sm;                    // [R columns][36 rows]: column (k, b), row (5 - d) * 6 + a
  double* arrowAcc = bandAcc + R * 36;     // [ncol][R]
  double* rhsAcc = arrowAcc + R * ncol;    // [R]
  double* ws = rhsAcc + R + wv * WS;
  int* it = reinterpret_cast<int*>(rhsAcc + R + 4 * WS);
  int* band_slot = it;           it += R;          // K * 6
  int* hcol = it;                it += K;
  int* arrow_slot = it;          it += nGB * K;
  int* cnt = it;                 it += A.n_batches + 1;
  int* first = it;
  const int t = blockIdx.x, c0 = t * K;
  for (int i = tid; i < R * 36 + R * ncol + R; i += 256) sm[i] = 0.0;
  for (int i = tid; i < R; i += 256) band_slot[i] = A.band[(size_t)t * R + i];
  for (int i = tid; i < K; i += 256) hcol[i] = A.hcol[(size_t)t * K + i];
  for (int i = tid; i < nGB * K; i += 256) arrow_slot[i] = A.arrow[(size_t)t * nGB * K + i];
  if (tid < A.n_batches) {
    const int per = A.n_tiles + 1;
    const int lo = A.ranges[(tid * 2 + 0) * per + t], hi = A.ranges[(tid * 2 + 1) * per + t + 1];
    first[tid] = lo;
    cnt[tid + 1] = max(hi - lo, 0);
  }
  __syncthreads();
  if (tid == 0) { cnt[0] = 0; for (int i = 0; i < A.n_batches; ++i) cnt[i + 1] += cnt[i]; }
  __syncthreads();
  const int total = cnt[A.n_batches];
  const F1ObsTab& tab = *reinterpret_cast<const F1ObsTab*>(ws + O_TAB);
  const double* w = ws + O_W;
  for (int item = wv; item < total; item += 4) {
    int cb = 0;
    while (item >= cnt[cb + 1]) ++cb;
    const int gi = A.gbase[cb] + first[cb] + (item - cnt[cb]);
    wave_sync();                               // the previous observation is done with the wave's scratch
    // everything the observation needs, in one round of independent loads
    if (lane < 49) ws[O_G + lane] = Gbuf[(size_t)gi * 49 + lane];
    for (int i = lane; i < 6 * F1_CORE; i += 64) ws[O_K + i] = chain[gi].K[i];
    if (lane < 28) ws[O_TAB + lane] = reinterpret_cast<const double*>(A.otab + gi)[lane];
    const F1Lite lt = A.lite[gi];
    if (lane < S6) ws[O_W + lane] = lt.w[lane];
    if (!lt.ok) continue;                      // k_obs_common has raised the status
    const int ls = lt.seg - (c0 - 5);          // local segment: live blocks ls .. ls + 5, owned blocks 5 .. 5 + K - 1
    if (ls < 0 || ls >= K + 5) continue;       // cannot happen: the ranges come from the same segments
    wave_sync();
    if (MC_ASM_SKIP & 16) continue;
    const double* G = ws + O_G;
    const double* Ks = ws + O_K;
    for (int idx = lane; idx < 140; idx += 64) {
      const int i = idx / 20, c = idx - 20 * i;
      double v = G[i * 7 + 6];
      if (c < F1_CORE) {
        v = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) v += G[i * 7 + j] * Ks[j * F1_CORE + c];
      }
      ws[O_T + idx] = v;
    }
    wave_sync();
    for (int idx = lane; idx < 210; idx += 64) {
      int bb = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
      while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
      while (bb * (bb + 1) / 2 > idx) --bb;
      const int ea = idx - bb * (bb + 1) / 2, eb = bb;
      double m = ws[O_T + 6 * 20 + eb];
      if (ea < F1_CORE) {
        m = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) m += Ks[i * F1_CORE + ea] * ws[O_T + i * 20 + eb];
      }
      ws[O_M + idx] = m;
    }
    wave_sync();
    const double* M = ws + O_M;               // upper triangle of the 20 x 20 core Gram: (a <= b) at b (b + 1) / 2 + a
    const int n_glob = tab.n_glob;
    if (tab.has_coef && !(MC_ASM_SKIP & 1)) {
      // band: block pair (i <= j) of the live coefficients, entry (a, b) = w_i w_j M(a, b); owner of block column ls + j
      for (int idx = lane; idx < 21 * 36; idx += 64) {
        const int p = idx / 36, e = idx - 36 * p, a = e / 6, bq = e - 6 * a;
        const int j = p >= 15 ? 5 : p >= 10 ? 4 : p >= 6 ? 3 : p >= 3 ? 2 : p >= 1 ? 1 : 0;
        const int i = p - (j * (j + 1)) / 2, d = j - i;
        const int k = ls + j - 5;
        if (k < 0 || k >= K || (d == 0 && a > bq)) continue;
        const int lo = min(a, bq), hi = max(a, bq);
        const double v = w[i] * w[j] * M[(hi * (hi + 1)) / 2 + lo];
        if (v != 0.0) atomicAdd(&bandAcc[(k * 6 + bq) * 36 + (5 - d) * 6 + a], v);
      }
      // coefficient rows x global columns, and the coefficient part of rhs (core column 19 = residual)
      for (int idx = lane; idx < ((MC_ASM_SKIP & 2) ? 0 : 36 * (n_glob + 1)); idx += 64) {
        const int gq = idx / 36, e = idx - 36 * gq, i = e / 6, a = e - 6 * i;
        const int k = ls + i - 5;
        if (k < 0 || k >= K) continue;
        const int kc = gq < n_glob ? tab.gk[gq] : 19;
        const double v = w[i] * M[(kc * (kc + 1)) / 2 + a];
        if (v == 0.0) continue;
        if (gq < n_glob) atomicAdd(&arrowAcc[tab.acol[gq] * R + k * 6 + a], v);
        else atomicAdd(&rhsAcc[k * 6 + a], -v);
      }
    }
    // entries between two global design variables and their part of rhs: once per observation, from its primary owner
    if (ls >= 5 && !(MC_ASM_SKIP & 4)) {
      const FpView f = fp_view_at(dc.fp_data + dc.fp_off[groups[gi].fp]);   // read only where a pair has no replica
      f1_flush(f, dc, M, w, -1, 0, tab.glc, tab.gk, tab.ggc, n_glob, n_glob, false, lane, 64);
    }
  }
  __syncthreads();
  if (MC_ASM_SKIP & 8) return;
  // ---- the owner's stores: plain read-modify-write, one pass over what it accumulated
  for (int idx = tid; idx < R * 36; idx += 256) {
    const double v = bandAcc[idx];
    if (v == 0.0) continue;
    const int col = idx / 36, e = idx - 36 * col, k = col / 6, bq = col - 6 * k, d = 5 - e / 6, a = e % 6;
    const int slot = band_slot[k * 6 + d];
    if (slot < 0) continue;
    dc.Cx[slot + bq * hcol[k] + (bq * (bq + 1)) / 2 + a] += v;
  }
  for (int idx = tid; idx < ncol * R; idx += 256) {
    const double v = arrowAcc[idx];
    if (v == 0.0) continue;
    const int gc = idx / R, row = idx - gc * R, k = row / 6, a = row - 6 * k;
    const int gb = A.col_gb[gc], sc = A.col_sc[gc];
    const int enc = arrow_slot[gb * K + k];
    if (enc >= 0) dc.Cx[enc + sc * A.gb_h[gb] + (sc * (sc + 1)) / 2 + a] += v;                  // column = global design variable
    else if (enc < -1) dc.Cx[(-2 - enc) + sc + a * hcol[k] + (a * (a + 1)) / 2] += v;          // column = coefficient
  }
  for (int idx = tid; idx < R; idx += 256) {
    const double v = rhsAcc[idx];
    if (v != 0.0) dc.rhs[A.pose_col0 + c0 * 6 + idx] += v;
  }
}

__device__ __forceinline__ int b_pose_first_block(const LidarBatchView& b) { return b.sp.first_block; }
__device__ __forceinline__ int b_pose_first_col(const LidarBatchView& b) { return b.sp.first_col; }
__device__ __forceinline__ int b_pose_first_block(const ImuBatchView& b) { return b.pose.first_block; }
__device__ __forceinline__ int b_pose_first_col(const ImuBatchView& b) { return b.pose.first_col; }

// ---------------------------------------------------------------------------------------------
// Stream kernel (F3 / F4 / F5): one 256-thread workgroup per *chunk* of consecutive time-sorted terms.
//   phase 0  value-array offsets of every block pair the chunk can touch, looked up once in the block pattern -> LDS
//   phase 1  thread = term: closed-form residual + Jacobian into a *compact* LDS row
//            [ 6 live pose-coefficient blocks (36) | the chunk's other columns (G) | residual ], plus beta0 = position
//            of the live window inside the chunk's coefficient range
//   phase 2  lane = 6 x 6 tile of the window's coefficient band (one block pair), then 6 x 4 tile of coefficient x other
//            columns: it walks the run of terms whose live window covers the tile (terms are time sorted, so that is
//            a contiguous run) and keeps the tile in registers -- 12 / 10 LDS values per 36 / 24 multiply-adds, where
//            one lane per entry loads 2 per multiply-add and is bound by LDS bandwidth.  The tiles are turned through
//            a small per-wave LDS strip so that a wave's fp64 atomics go to consecutive addresses (ONE atomic per
//            entry and chunk).  The few other x other entries stay one per lane.
//            The fp64 matrix cores were tried for this Gram (16 x 16 window tiles, operands gathered from the compact
//            rows): v_mfma_f64_16x16x4 has the same peak as the vector unit on this part and the band fills only half
//            of each tile, so it ran 2-3x slower than the vector gathers.
// Compared with one group per footprint this cuts the atomics by the number of terms that share a window.
struct StreamDims { int maxT, maxNb, maxNob, maxG, ld, maxTab;
                    int ldc, nbp;      // compact-bias layout (IMU): row stride, doubles per term behind the rows
                    int ldp; };        // ... with compact pose columns as well (calibrated IMU model): row stride
template <class View> struct StreamBatch {
  View v;
  const int* order;      // time-sorted term order of the batch
  const int* term_lo;    // per term: first coefficient of the buffer frozen at construction, and its length
  const int* term_nbuf;
};

__device__ __forceinline__ int slot_lookup(const DevCommon& dc, int rb, int cb) {
  int lo = dc.pat_colptr[cb], end = dc.pat_colptr[cb + 1], hi = end;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (dc.pat_rows[mid] < rb) lo = mid + 1; else hi = mid;
  }
  return (lo < end && dc.pat_rows[lo] == rb) ? dc.pat_cp0[cb] + dc.pat_roff[lo] : -1;
}

#ifndef MC_GATHER_UNROLL
#define MC_GATHER_UNROLL 1
#endif
#ifndef MC_STREAM_OCC
#define MC_STREAM_OCC 1
#endif
#ifndef MC_LIDAR_OCC
#define MC_LIDAR_OCC 3   // compact rows: 47 KB of LDS per workgroup, 3 per CU; 168 VGPRs cost 196 B of scratch per lane and still win (0.468 -> 0.427 ms)
#endif
template <class View> struct StreamTraits;
template <> struct StreamTraits<LidarBatchView> {
  // CO = columns the coefficient part of a row takes in the LDS tile: the lidar row is rank one in the coefficients
  // (w_i * B), stored as [B 6 | w 6]
  enum { D = 1, CO = 12, OCC = MC_LIDAR_OCC };
  static __device__ __forceinline__ int biasColumns(const LidarBatchView&, const int*, int& n) { n = 0; return 0; }
  template <bool CP = false>
  static __device__ __forceinline__ void coef6(const double* row, const double*, int o, double (&v)[6]) {
    const double2* q = reinterpret_cast<const double2*>(row);
    const double2 b01 = q[0], b23 = q[1], b45 = q[2];
    const double w = row[6 + o];
    v[0] = w * b01.x; v[1] = w * b01.y; v[2] = w * b23.x; v[3] = w * b23.y; v[4] = w * b45.x; v[5] = w * b45.y;
  }
  // the two 6-vectors of a band tile's row product, x = row[block o], y = row[block o2]: x y^T = (w_o w_o2) B B^T, so B is
  // read once and only x carries the weights (7 multiplies instead of 12, 5 LDS reads instead of 8)
  template <bool CP = false>
  static __device__ __forceinline__ void pair6(const double* row, const double*, int o, int o2, double (&x)[6], double (&y)[6]) {
    const double2* q = reinterpret_cast<const double2*>(row);
    const double2 b01 = q[0], b23 = q[1], b45 = q[2];
    const double w = row[6 + o] * row[6 + o2];
    y[0] = b01.x; y[1] = b01.y; y[2] = b23.x; y[3] = b23.y; y[4] = b45.x; y[5] = b45.y;
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = w * y[i];
  }
  // returns beta0 (>= 0) or -1 after setting *st
  template <bool CB = false, bool CP = false>
  static __device__ __forceinline__ int eval(const LidarBatchView& b, const double* x, int u, bool useM, bool wantJ,
                                             const int* of, int pose_lo, int pose_nb, int term_lo, int term_nbuf,
                                             double* J, int ld, double* e, TermOut& out, int* st, int = 0) {
    int rc[MC_MAX_ROLES];
#pragma unroll
    for (int r = 0; r < MC_MAX_ROLES; ++r) rc[r] = of[4 + r] >= 0 ? CO + of[4 + r] : -1;
    int beta0 = 0;
    rc[F5_ROLE_COEF] = -1;
    if (pose_nb > 0) {
      double t = b.t[u] + (b.toff.off >= 0 ? x[b.toff.off] : 0.0);
      if (!(t >= b.sp.t_min && t <= b.sp.t_max)) { *st = STATUS_TIME_OUT_OF_SPLINE; return -1; }
      int seg = spline_segment(b.sp, t);
      int shift = seg - term_lo;  // live support must stay inside the buffer frozen at construction (BSplineExpressions.cpp:447)
      if (shift < 0 || shift + S6 > term_nbuf) { *st = STATUS_BUFFER_EXCEEDED; return -1; }
      beta0 = seg - pose_lo;
      rc[F5_ROLE_COEF] = 0;
    }
    if (!f5_term<true>(b, x, u, useM, wantJ, rc, J, ld, e, out)) { *st = STATUS_TIME_OUT_OF_SPLINE; return -1; }
    return beta0;
  }
};
// The two IMU kinds (calibrated: imu_term; scale-misalignment / size-effect: imu_sm_term) share the row layout, the
// live-window bookkeeping and the compact bias columns; they differ in the term function only.
template <class View> struct ImuTermFn;
template <> struct ImuTermFn<ImuBatchView> {
  template <bool CB, bool CP> static __device__ __forceinline__ bool run(const ImuBatchView& b, const double* x, int u, bool useM, bool wantJ,
                                                                         const int* rc, double* J, int ld, double* e, TermOut& out, int wofs) {
    return imu_term<CB, CP>(b, x, u, useM, wantJ, rc, J, ld, e, out, wofs);
  }
};
template <> struct ImuTermFn<ImuSmBatchView> {
  template <bool CB, bool CP> static __device__ __forceinline__ bool run(const ImuSmBatchView& b, const double* x, int u, bool useM, bool wantJ,
                                                                         const int* rc, double* J, int ld, double* e, TermOut& out, int) {
    static_assert(!CP, "compact pose columns: calibrated model only");
    return imu_sm_term<CB>(b, x, u, useM, wantJ, rc, J, ld, e, out);
  }
};
template <class View> struct ImuStreamTraits {
  enum { D = 3, CO = 36, OCC = MC_STREAM_OCC };
  // first local "other" column and number of columns of the bias (spline window or static point); n = 0: no shortcut
  static __device__ __forceinline__ int biasColumns(const View& b, const int* of, int& n) {
    const int c0 = of[4 + IMU_ROLE_BIAS];
    n = (c0 >= 0 && b.S == nullptr) ? (b.bias.order > 0 ? 3 * of[3] : 3) : 0;
    return c0;
  }
  // CP: the row holds [P 3 | Q0 3 | Q1 3 | Q2 3], wts = the term's basis weights [w 6 | wd 6 | wdd 6] (imu_term<CB, CP>)
  template <bool CP = false>
  static __device__ __forceinline__ void coef6(const double* row, const double* wts, int o, double (&v)[6]) {
    if constexpr (CP) {
      const double2* q = reinterpret_cast<const double2*>(row);          // 16-byte aligned: ld and the term stride are even
      const double2 a = q[0], b2 = q[1], c = q[2], d = q[3], e2 = q[4], f = q[5];
      const double w0 = wts[o], w1 = wts[6 + o], w2 = wts[12 + o];
      v[0] = w2 * a.x; v[1] = w2 * a.y; v[2] = w2 * b2.x;
      v[3] = w0 * b2.y + w1 * d.x + w2 * e2.y;
      v[4] = w0 * c.x + w1 * d.y + w2 * f.x;
      v[5] = w0 * c.y + w1 * e2.x + w2 * f.y;
    } else {
      const double2* q = reinterpret_cast<const double2*>(row + 6 * o);   // 16-byte aligned: ld and 6 o are even
      const double2 a = q[0], b2 = q[1], c = q[2];
      v[0] = a.x; v[1] = a.y; v[2] = b2.x; v[3] = b2.y; v[4] = c.x; v[5] = c.y;
    }
  }
  template <bool CP = false>
  static __device__ __forceinline__ void pair6(const double* row, const double* wts, int o, int o2, double (&x)[6], double (&y)[6]) {
    if constexpr (CP) {
      const double2* q = reinterpret_cast<const double2*>(row);
      const double2 a = q[0], b2 = q[1], c = q[2], d = q[3], e2 = q[4], f = q[5];
      {
        const double w0 = wts[o], w1 = wts[6 + o], w2 = wts[12 + o];
        x[0] = w2 * a.x; x[1] = w2 * a.y; x[2] = w2 * b2.x;
        x[3] = w0 * b2.y + w1 * d.x + w2 * e2.y; x[4] = w0 * c.x + w1 * d.y + w2 * f.x; x[5] = w0 * c.y + w1 * e2.x + w2 * f.y;
      }
      {
        const double w0 = wts[o2], w1 = wts[6 + o2], w2 = wts[12 + o2];
        y[0] = w2 * a.x; y[1] = w2 * a.y; y[2] = w2 * b2.x;
        y[3] = w0 * b2.y + w1 * d.x + w2 * e2.y; y[4] = w0 * c.x + w1 * d.y + w2 * f.x; y[5] = w0 * c.y + w1 * e2.x + w2 * f.y;
      }
    } else {
      coef6<false>(row, wts, o, x);
      coef6<false>(row, wts, o2, y);
    }
  }
  // CB: compact bias columns -- the bias columns are not stored in the rows (the columns behind them move down by their
  // number) and the role column of the bias is the index of the first live coefficient in the term's bias-weight array
  template <bool CB = false, bool CP = false>
  static __device__ __forceinline__ int eval(const View& b, const double* x, int u, bool useM, bool wantJ,
                                             const int* of, int pose_lo, int pose_nb, int, int,
                                             double* J, int ld, double* e, TermOut& out, int* st, int wofs = 0) {
    int rc[MC_MAX_ROLES];
    const int bias_0 = of[4 + IMU_ROLE_BIAS], bias_n = bias_0 >= 0 ? 3 * (b.bias.order > 0 ? of[3] : 1) : 0;
    constexpr int co = CP ? 12 : (int)CO;
#pragma unroll
    for (int r = 0; r < MC_MAX_ROLES; ++r) {
      const int c = of[4 + r];
      rc[r] = c >= 0 ? co + ((CB && bias_0 >= 0 && c > bias_0) ? c - bias_n : c) : -1;
    }
    const double t = b.t[u];
    if (!(t >= b.pose.t_min && t <= b.pose.t_max)) { *st = STATUS_TIME_OUT_OF_SPLINE; return -1; }
    int beta0 = 0;
    rc[IMU_ROLE_POSE] = -1;
    if (pose_nb > 0) { beta0 = spline_segment(b.pose, t) - pose_lo; rc[IMU_ROLE_POSE] = 0; }
    if (CB && rc[IMU_ROLE_BIAS] >= 0) rc[IMU_ROLE_BIAS] = 0;
    if (b.bias.order > 0 && rc[IMU_ROLE_BIAS] >= 0) {
      if (!(t >= b.bias.t_min && t <= b.bias.t_max)) { *st = STATUS_TIME_OUT_OF_SPLINE; return -1; }
      rc[IMU_ROLE_BIAS] += (CB ? 1 : 3) * (spline_segment(b.bias, t) - of[2]);
    }
    if (!ImuTermFn<View>::template run<CB, CP>(b, x, u, useM, wantJ, rc, J, ld, e, out, wofs)) { *st = STATUS_TIME_OUT_OF_SPLINE; return -1; }
    return beta0;
  }
};
template <> struct StreamTraits<ImuBatchView> : ImuStreamTraits<ImuBatchView> {};
template <> struct StreamTraits<ImuSmBatchView> : ImuStreamTraits<ImuSmBatchView> {};

template <class View, bool BUILD, bool CB = false>
__global__ __launch_bounds__(256, BUILD ? StreamTraits<View>::OCC : 1) void k_stream(const StreamBatch<View>* __restrict__ batches, DevCommon dc,
                                                const ChunkRec* __restrict__ chunks, StreamDims dm, int jpart_base) {
  extern __shared__ double smem[];
  // CP: compact pose columns on top of the compact bias columns (calibrated IMU model; imu_term<CB, CP>)
  constexpr bool CP = CB && std::is_same<View, ImuBatchView>::value;
  constexpr int D = StreamTraits<View>::D, CO = CP ? 12 : (int)StreamTraits<View>::CO;
  const int tid = threadIdx.x;
  const ChunkRec ch = chunks[blockIdx.x];
  const View b = batches[ch.batch].v;
  const int* __restrict__ order = batches[ch.batch].order;
  const int* __restrict__ term_lo = batches[ch.batch].term_lo;
  const int* __restrict__ term_nbuf = batches[ch.batch].term_nbuf;
  const int* of = dc.ofp_data + dc.ofp_off[ch.ofp];
  const int nob = of[0], G = of[1];
  const int* oblk = of + OFP_HDR;   // nob x (block index, dim, local column start, global column base)
  // CB (IMU kinds without a general sqrtInvR): the 3 x bias_n bias entries of a term are nbc values (w_i on the diagonal of
  // a 3 x 3 block, the same in the three rows); they are kept once per term behind its rows and the rows hold the other
  // Gs = G - bias_n columns only: 152 instead of 216 doubles per term on the calibration problems
  const int nb = ch.pose_nb, T = ch.term_count, ld = CP ? dm.ldp : CB ? dm.ldc : dm.ld;
  const int wofs = D * ld + (CB ? dm.nbp : 0);  // CP: the term's basis weights (3 x 6) sit behind its bias weights
  const int TS = wofs + (CP ? 18 : 0);          // doubles per term
  int cb_0 = -1, cb_n = 0;                     // first bias column among the footprint's others, their number
  if constexpr (CB) { cb_0 = of[4 + IMU_ROLE_BIAS]; cb_n = cb_0 >= 0 ? 3 * (b.bias.order > 0 ? of[3] : 1) : 0; }
  const int Gs = G - cb_n;                     // other columns stored in a row
  const int pb = nb > 0 ? b_pose_first_block(b) + ch.pose_lo : 0;
  double* rows = smem;
  int* ip = (int*)(rows + (size_t)dm.maxT * TS);
  int* beta0 = ip; ip += dm.maxT;
  int* rlo = ip; ip += dm.maxNb + 1;
  int* rhi = ip; ip += dm.maxNb + 1;
  // the chunk's slot table (Problem::chunk_tab), staged by one coalesced read
  int* tab = ip;
  const int* band_slot = tab;
  const int* hcol = band_slot + nb * 6;
  const int* arrow_slot = hcol + nb;
  const int* oo_slot = arrow_slot + nob * nb;
  const int* oh = oo_slot + nob * nob;
  const int* ocolblk = oh + nob;
  const int* oinfo = ocolblk + G;   // nob x (block index, local column start, global column base), staged with the table
  int* atab = tab + dm.maxTab;      // (G, nb) coefficient x other addresses, filled below
  __shared__ double red[4];
  constexpr bool build = BUILD;
  if (build) {
    for (int i = tid; i < T * TS; i += 256) rows[i] = 0.0;
    for (int i = tid; i <= nb; i += 256) { rlo[i] = 0x7fffffff; rhi[i] = 0; }
    for (int i = tid; i < ch.tab_len; i += 256) tab[i] = dc.chunk_tab[ch.tab_off + i];
    __syncthreads();
  }
  MC_MARK_BEGIN();
  const int pslot = D == 1 ? 8 : 16;
  // ---- phase 1: one term per thread
  double Jacc = 0.0;
  if (tid < T) {
    const int u = order[ch.term_first + tid];
    double e[D];
    TermOut out;
    int st = 0;
    double* Jrow = rows + (size_t)tid * TS;
    int b0 = StreamTraits<View>::template eval<CB, CP>(b, dc.x, u, dc.useM != 0, build, of, ch.pose_lo, nb, term_lo[u], term_nbuf[u], Jrow, ld, e, out, &st, CB ? dm.nbp : 0);
    if (b0 >= 0) {
      Jacc = out.w * out.chi2;
      dc.chi2[b.term_base_global + u] = out.chi2;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        if (dc.e) dc.e[b.row_base + D * u + r] = -e[r];
        if (build) Jrow[r * ld + CO + Gs] = e[r];
      }
      if (build) {
        beta0[tid] = b0;
        if (nb > 0) { atomicMin(&rlo[b0], tid); atomicMax(&rhi[b0], tid + 1); }
      }
    } else {
      atomicOr(dc.status, st);
      if (build) beta0[tid] = -1000;   // covers nothing
    }
  }
  MC_MARK(pslot + 0);   // term evaluation (wave 0's terms)
  Jacc = wave_sum(Jacc);
  if ((tid & 63) == 0) red[tid >> 6] = Jacc;
  __syncthreads();
  MC_MARK(pslot + 1);   // waiting for the other waves
  if (tid == 0) {
    double Js = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) Js += red[w];   // evaluate launches use fewer waves per chunk
    dc.Jpart[jpart_base + blockIdx.x] = Js;
  }
  if (!build) return;
  // Terms are time sorted and share one time offset, so beta0 is non-decreasing over the chunk and the terms whose
  // first live block is >= s are a suffix [fge[s], T) (a failed term inside has an all-zero row).  A violation would
  // silently drop products, so it is checked and reported.
  if (tid > 0 && tid < T && beta0[tid] >= 0 && beta0[tid - 1] >= 0 && beta0[tid] < beta0[tid - 1]) atomicOr(dc.status, STATUS_INTERNAL);
  if (MC_MODE(dc) == 3) return;
  int* fge = rhi;
  if (tid <= nb) {
    int m = T;
    for (int s = tid; s <= nb; ++s) m = min(m, rlo[s]);
    fge[tid] = m;
  }
  __syncthreads();
  MC_MARK(pslot + 2);   // run table
  // ---- phase 2: the chunk's Gram, register blocked.  A lane owns a 6 x 6 (band) or 6 x 4 (coefficient x other) tile
  // of the window's Hessian and walks the run of terms whose live window covers it: 12 (10) LDS values feed 36 (24)
  // multiply-adds.  The tiles are then turned through a per-wave LDS strip so that the fp64 adds of a wave go to
  // consecutive addresses: a lane's 6 values of one Hessian column are consecutive in the CCS array, and so are those
  // of the next lane (the rows of a column are ordered by block), hence element e of the strip is address base + e.
  __shared__ double wstrip[4][32 * 6];
  const int lane = tid & 63, wave = tid >> 6;
  double* strip = wstrip[wave];
  const int rstride = TS;
  const int smax = max(nb - S6, 0);   // last possible first live block
  // turn6: every lane hands in 6 values; emit(p, v) receives element e = 64 p + lane (p = 0..5) of the wave's 384 values
  // (element 6 l + i = value i of lane l).  Half a wave at a time through the 32 x 6 strip; LDS accesses of one wave
  // execute in program order, the barriers only pin the compiler's schedule.
  auto turn6 = [&](const double (&v)[6], auto&& emit) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if ((lane >> 5) == h) {
#pragma unroll
        for (int i = 0; i < 6; ++i) strip[(lane & 31) * 6 + i] = v[i];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int p = 0; p < 3; ++p) emit(3 * h + p, strip[64 * p + lane]);
      __builtin_amdgcn_wave_barrier();
    }
  };
  // address table of the coefficient x other entries: atab[g nb + bi] = value index of (row 0 of block bi, other column
  // g) when the other block is the column, -2 - (index of (other column's component, row 0 of block bi)) when the
  // coefficient block is the column, -1 when the pair is not in the pattern
  for (int i = tid; i < G * nb; i += 256) {
    const int g = i / nb, bi = i - g * nb;
    const int o = ocolblk[g], sc = g - oinfo[3 * o + 1];
    const int slot = arrow_slot[o * nb + bi];
    int a = -1;
    if (slot >= 0) a = oinfo[3 * o] < pb + bi ? -2 - (slot + sc) : slot + sc * oh[o] + (sc * (sc + 1)) / 2;
    atab[i] = a;
  }
  // (i) coefficient band: task = (block column bj, block row bi = bj - d), ordered down the CCS column (d = 5 .. 0)
  // Only the pairs with bi >= 0 are numbered (block columns 0..4 have 1..5 of them): a chunk of 13 coefficient blocks has
  // 63 tasks -- one wave -- where 6 nb = 78 put 14 lanes into a second wave that walked the rows for them alone.
  auto band_task = [](int t, int& bj, int& d) {
    if (t < 15) { bj = t >= 10 ? 4 : t >= 6 ? 3 : t >= 3 ? 2 : t >= 1 ? 1 : 0; d = bj - (t - (bj * (bj + 1)) / 2); }
    else { bj = 5 + (t - 15) / 6; d = 5 - (t - 15 - 6 * (bj - 5)); }
  };
  const int nband = nb >= 5 ? 6 * nb - 15 : (nb * (nb + 1)) / 2;
  for (int base = 0; base < nband; base += 256) {
    const int t = base + tid;
    int bj, d;
    band_task(t, bj, d);
    const int bi = bj - d;
    double acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = 0.0;
    if (t < nband && bi >= 0 && band_slot[bj * 6 + d] >= 0) {
      // by segment: the terms whose live window starts at block sg are the run [fge[sg], fge[sg + 1]) (a failed term inside
      // has an all-zero row), so the position o = bi - sg of the tile in the row is known without reading beta0 and the
      // LDS reads of consecutive terms do not depend on one another
      for (int sg = max(0, bj - 5); sg <= bi; ++sg) {
        const int o = bi - sg, k1 = fge[sg + 1];
        const double* px = rows + (size_t)fge[sg] * rstride;
#pragma unroll MC_GATHER_UNROLL
        for (int k = fge[sg]; k < k1; ++k, px += rstride) {
#pragma unroll
          for (int r = 0; r < D; ++r) {
            double xv[6], yv[6];
            StreamTraits<View>::template pair6<CP>(px + r * ld, px + wofs, o, o + d, xv, yv);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int j = 0; j < 6; ++j) acc[i][j] += xv[i] * yv[j];
          }
        }
      }
    }
    if (MC_MODE(dc) != 1) { if (acc[0][0] + acc[5][5] == 1.2345e-300) dc.rhs[0] = 1.0; continue; }
    // element e of the wave belongs to task tb + e / 6, row e % 6; its address less the column part is the same for
    // the 6 columns
    const int tb = base + 64 * wave;
    int eaddr[6], eh[6], era[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int e = 64 * p + lane, t2 = tb + e / 6, ra = e % 6;
      int bj2, d2;
      band_task(t2, bj2, d2);
      int slot = -1;
      if (t2 < nband) slot = band_slot[bj2 * 6 + d2];
      eaddr[p] = slot >= 0 ? slot + ra : -1;
      eh[p] = t2 < nband ? hcol[bj2] : 0;
      era[p] = d2 == 0 ? ra : -1;   // diagonal block: rows below the diagonal are not stored
    }
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      const double v6[6] = {acc[0][cc], acc[1][cc], acc[2][cc], acc[3][cc], acc[4][cc], acc[5][cc]};
      turn6(v6, [&](int p, double v) {
        if (v != 0.0 && eaddr[p] >= 0 && era[p] <= cc) atomic_add_f64(dc.Cx + eaddr[p] + cc * eh[p] + (cc * (cc + 1)) / 2, v);
      });
    }
  }
  MC_MARK(pslot + 3);   // band
  __syncthreads();      // atab
  // (ii) coefficient x other columns (and the coefficient part of rhs: other column G = residual).
  // task = (tile of 4 other columns, coefficient block bi), consecutive lanes = consecutive blocks
  constexpr int GT = 4;
  const int ngt = (Gs + 1 + GT - 1) / GT;        // over the stored columns
  const int narrow = nb > 0 ? ngt * nb : 0;
  for (int base = 0; base < narrow; base += 256) {
    const int t = base + tid;
    const int gt = t / max(nb, 1), bi = t - gt * nb;
    double acc[6][GT];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < GT; ++j) acc[i][j] = 0.0;
    if (t < narrow) {
      for (int sg = max(0, bi - 5); sg <= min(bi, smax); ++sg) {
        const int o = bi - sg, k1 = fge[sg + 1];
        const double* px = rows + (size_t)fge[sg] * rstride;
#pragma unroll MC_GATHER_UNROLL
        for (int k = fge[sg]; k < k1; ++k, px += rstride) {
#pragma unroll
        for (int r = 0; r < D; ++r) {
          double xv[6];
          StreamTraits<View>::template coef6<CP>(px + r * ld, px + wofs, o, xv);
          const double2* qy = reinterpret_cast<const double2*>(px + r * ld + CO + GT * gt);   // columns past G: zeroed padding
          const double2 y01 = qy[0], y23 = qy[1];
          const double yv[GT] = {y01.x, y01.y, y23.x, y23.y};
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < GT; ++j) acc[i][j] += xv[i] * yv[j];
        }
        }
      }
    }
    if (MC_MODE(dc) != 1) { if (acc[0][0] + acc[5][GT - 1] == 1.2345e-300) dc.rhs[0] = 1.0; continue; }
    const int tb = base + 64 * wave;
    int eg[6], ebi[6], era[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int e = 64 * p + lane, t2 = tb + e / 6;
      const int gt2 = t2 / max(nb, 1);
      era[p] = e % 6; ebi[p] = t2 - gt2 * nb; eg[p] = t2 < narrow ? GT * gt2 : Gs + 1;
    }
    const int rhs0 = b_pose_first_col(b) + 6 * ch.pose_lo;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      const double v6[6] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j], acc[4][j], acc[5][j]};
      turn6(v6, [&](int p, double v) {
        const int gs = eg[p] + j;                  // stored column
        if (v == 0.0 || gs > Gs) return;
        if (gs == Gs) { atomic_add_f64(dc.rhs + rhs0 + 6 * ebi[p] + era[p], -v); return; }
        const int g = (CB && cb_0 >= 0 && gs >= cb_0) ? gs + cb_n : gs;   // its number among the footprint's columns
        const int a = atab[g * nb + ebi[p]];
        if (a >= 0) atomic_add_f64(dc.Cx + a + era[p], v);                                                                  // column = other
        else if (a < -1) atomic_add_f64(dc.Cx + (-2 - a) + era[p] * hcol[ebi[p]] + (era[p] * (era[p] + 1)) / 2, v);         // column = coefficient
      });
    }
  }
  // (ii b) compact bias columns: task = (bias coefficient ci, coefficient block bi), a 6 x 3 tile whose column c collects
  // residual row c only: 18 multiply-adds per term where the plain layout spends 72 on a 6 x 4 tile that is two thirds zeros
  if constexpr (CB) {
    const int nbc = cb_n / 3, nbias = nb > 0 ? nbc * nb : 0;
    for (int base = 0; base < nbias; base += 256) {
      const int t = base + tid;
      const int ci = t / max(nb, 1), bi = t - ci * nb;
      double acc[6][3];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = 0.0;
      if (t < nbias) {
        for (int sg = max(0, bi - 5); sg <= min(bi, smax); ++sg) {
          const int o = bi - sg, k1 = fge[sg + 1];
          const double* px = rows + (size_t)fge[sg] * rstride;
#pragma unroll MC_GATHER_UNROLL
          for (int k = fge[sg]; k < k1; ++k, px += rstride) {
            const double w = px[D * ld + ci];
#pragma unroll
            for (int r = 0; r < D; ++r) {
              double xv[6];
              StreamTraits<View>::template coef6<CP>(px + r * ld, px + wofs, o, xv);
#pragma unroll
              for (int i = 0; i < 6; ++i) acc[i][r] += xv[i] * w;
            }
          }
        }
      }
      if (MC_MODE(dc) != 1) { if (acc[0][0] + acc[5][2] == 1.2345e-300) dc.rhs[0] = 1.0; continue; }
      const int tb = base + 64 * wave;
      int ecol[6], ebi[6], era[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const int e = 64 * p + lane, t2 = tb + e / 6;
        const int ci2 = t2 / max(nb, 1);
        era[p] = e % 6; ebi[p] = t2 - ci2 * nb; ecol[p] = t2 < nbias ? cb_0 + 3 * ci2 : -1;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v6[6] = {acc[0][c], acc[1][c], acc[2][c], acc[3][c], acc[4][c], acc[5][c]};
        turn6(v6, [&](int p, double v) {
          if (v == 0.0 || ecol[p] < 0) return;
          const int a = atab[(ecol[p] + c) * nb + ebi[p]];
          if (a >= 0) atomic_add_f64(dc.Cx + a + era[p], v);                                                                  // column = bias
          else if (a < -1) atomic_add_f64(dc.Cx + (-2 - a) + era[p] * hcol[ebi[p]] + (era[p] * (era[p] + 1)) / 2, v);         // column = coefficient
        });
      }
    }
  }
  MC_MARK(pslot + 4);   // coefficient x other
  // (iii) other x other (upper triangle) and the other part of rhs
  int bias_n_ = 0;
  auto oo_flush = [&](int g1, int g2, double acc) {
    if (g1 > g2 || g1 >= G || g2 > G || acc == 0.0) return;
    const int o1 = ocolblk[g1], s1 = g1 - oinfo[3 * o1 + 1];
    // entries between two global design variables (extrinsics, time offset, target poses, gravity ...) are written by every
    // chunk of the stream: they go to this workgroup's replica (folded into Cx / rhs by k_rep_reduce)
    const int gb1 = dc.rep ? dc.gcb[oinfo[3 * o1]] : -1;
    if (g2 == G) { atomic_add_f64(gb1 >= 0 ? rep_rhs_entry(dc, gb1 + s1) : dc.rhs + oinfo[3 * o1 + 2] + s1, -acc); return; }
    const int o2 = ocolblk[g2], s2 = g2 - oinfo[3 * o2 + 1];
    if (gb1 >= 0) {
      const int gb2 = dc.gcb[oinfo[3 * o2]];
      if (gb2 >= 0) { atomic_add_f64(rep_entry(dc, min(gb1 + s1, gb2 + s2), max(gb1 + s1, gb2 + s2)), acc); return; }
    }
    if (o1 == o2) {
      int slot = oo_slot[o1 * nob + o1];
      if (slot >= 0) atomic_add_f64(dc.Cx + slot + s2 * oh[o1] + (s2 * (s2 + 1)) / 2 + s1, acc);
    } else if (oinfo[3 * o1] < oinfo[3 * o2]) {
      int slot = oo_slot[o1 * nob + o2];
      if (slot >= 0) atomic_add_f64(dc.Cx + slot + s2 * oh[o2] + (s2 * (s2 + 1)) / 2 + s1, acc);
    } else {
      int slot = oo_slot[o2 * nob + o1];
      if (slot >= 0) atomic_add_f64(dc.Cx + slot + s1 * oh[o1] + (s1 * (s1 + 1)) / 2 + s2, acc);
    }
  };
  const int G1 = G + 1;
  if constexpr (CB) {
    // place of column g (of the footprint's numbering; G = residual) in a term: inside row r, or in the bias weights
    auto place = [&](int g, int r) {
      if ((unsigned)(g - cb_0) < (unsigned)cb_n) return D * ld + (g - cb_0) / 3;
      return r * ld + CO + (g >= cb_0 + cb_n && cb_0 >= 0 ? g - cb_n : g);
    };
    for (int idx = tid; idx < G1 * G1; idx += 256) {
      const int g2 = idx / G1, g1 = idx - g2 * G1;
      if (g1 > g2 || g1 == G) continue;
      const int c1 = (unsigned)(g1 - cb_0) < (unsigned)cb_n ? (g1 - cb_0) % 3 : -1;
      const int c2 = (unsigned)(g2 - cb_0) < (unsigned)cb_n ? (g2 - cb_0) % 3 : -1;
      if (c1 >= 0 && c2 >= 0 && c1 != c2) continue;
      const int only = c1 >= 0 ? c1 : c2;
      double acc = 0.0;
      if (only >= 0) {
        const double* t1 = rows + place(g1, only);
        const double* t2 = rows + place(g2, only);
#pragma unroll 8
        for (int k = 0; k < T; ++k, t1 += rstride, t2 += rstride) acc += *t1 * *t2;
      } else {
        const int p1 = place(g1, 0), p2 = place(g2, 0);
        const double* trm = rows;
#pragma unroll 4
        for (int k = 0; k < T; ++k, trm += rstride) {
#pragma unroll
          for (int r = 0; r < D; ++r) acc += trm[r * ld + p1] * trm[r * ld + p2];
        }
      }
      if (MC_MODE(dc) != 1) { if (acc == 1.2345e-300) dc.rhs[0] = 1.0; continue; }
      oo_flush(g1, g2, acc);
    }
    MC_MARK(pslot + 5);   // other x other
    return;
  }
  const int bias_0 = StreamTraits<View>::biasColumns(b, of, bias_n_);
  const int bias_n = bias_n_;
  for (int idx = tid; idx < G1 * G1; idx += 256) {
    const int g2 = idx / G1, g1 = idx - g2 * G1;
    if (g1 > g2 || g1 == G) continue;
    // IMU rows: residual row r holds the bias columns of component r only (the bias Jacobian is w_i I), so a product with
    // a bias column runs over every third row, and two bias columns of different components never meet.  Not with a
    // general sqrtInvR, which mixes the rows.
    int only = -1;
    if (D == 3 && bias_n > 0) {
      const int c1 = (unsigned)(g1 - bias_0) < (unsigned)bias_n ? (g1 - bias_0) % 3 : -1;
      const int c2 = (unsigned)(g2 - bias_0) < (unsigned)bias_n ? (g2 - bias_0) % 3 : -1;
      if (c1 >= 0 && c2 >= 0 && c1 != c2) continue;
      only = c1 >= 0 ? c1 : c2;
    }
    double acc = 0.0;
    if (only >= 0) {
      const double* row = rows + CO + only * ld;
#pragma unroll 8
      for (int k = 0; k < T; ++k, row += D * ld) acc += row[g1] * row[g2];
    } else {
      const double* row = rows + CO;
#pragma unroll 8
      for (int k = 0; k < T * D; ++k, row += ld) acc += row[g1] * row[g2];   // unrolled: independent LDS reads in flight
    }
    if (MC_MODE(dc) != 1) { if (acc == 1.2345e-300) dc.rhs[0] = 1.0; continue; }
    oo_flush(g1, g2, acc);
  }
  MC_MARK(pslot + 5);   // other x other
}

// ---------------------------------------------------------------------------------------------
// F2 (ReprojectionError with camera design variables over a static chain: the camera-calibration problems).
// One 256-thread workgroup per RUN of consecutive observations (one camera's view of the target) of one camera.
//   build: thread = residual ROW (two threads per corner); the weighted row goes to LDS in a fixed "core" column order
//          [q_t_c 3 | t_t_c 3 | baselines 6 n_chain | projection | distortion | residual]  (<= 41 columns),
//          the core Gram R^T R is formed on the fp64 matrix cores (16 x 16 tiles, K-steps split over the 4 waves,
//          per-wave partial tiles summed by LDS atomics), and is then split in two:
//            - entries that involve the target pose of this observation are flushed at once,
//            - everything else (baselines x intrinsics ... of this camera) is shared by ALL observations of the camera:
//              it is summed in registers over the run and flushed once per run -- without that every observation would
//              queue its atomics on the same few hundred addresses.
//   evaluate: thread = corner.
// Pre-pass, one LANE per observation: target pose, baseline chain and intrinsics the corners of the observation share
// (k_static reads the record back from LDS by broadcast instead of every lane recomputing ~130 registers' worth).
__global__ __launch_bounds__(64) void k_static_common(const StaticReprojBatchView* __restrict__ views, const double* __restrict__ x,
                                                      const GroupRec* __restrict__ groups, int n, F2Common* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const GroupRec gr = groups[g];
  F2Common cm;
  f2_common(views[gr.batch], x, gr.aux, cm);
  out[g] = cm;
}
enum { F2_MAXCORE = 6 + 3 + 6 * F2_MAX_CHAIN + 6 + 4 + 1, F2_NE = (F2_MAXCORE * (F2_MAXCORE + 1) / 2 + 255) / 256 };
template <bool BUILD>
__global__ __launch_bounds__(256, BUILD ? 2 : 1) void k_static(const StaticReprojBatchView* __restrict__ views, DevCommon dc,
                                                const GroupRec* __restrict__ groups, const ImageRec* __restrict__ runs,
                                                const F2Common* __restrict__ common, int ld, int jpart_base) {
  extern __shared__ double rows[];   // build: 256 x ld
  __shared__ F2Common cm;
  static_assert(sizeof(F2Common) % 8 == 0 && sizeof(F2Common) / 8 <= 256, "F2Common is copied as doubles, one per thread");
  __shared__ double gram[6 * 256];   // upper tile pairs, [tile][reg][lane] = the MFMA C/D layout
  __shared__ int crole[F2_MAXCORE], cdim[F2_MAXCORE], lcol[F2_MAXCORE];
  __shared__ double red[4];
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const ImageRec run = runs[blockIdx.x];
  const StaticReprojBatchView b = views[groups[run.first].batch];
  const int nch = b.n_chain, pd = b.cam.proj_dim, nd = b.cam.dist_off >= 0 ? 4 : 0;
  // batches with landmark design variables carry 3 more columns per row: the landmark of the row's TERM.  The core Gram's
  // entries over them mix different landmarks and are never flushed (their role is inactive in the observation's
  // footprint); the landmark's rows of the system are added per term below, through the term's own footprint
  const int L = b.lm_off ? 3 : 0, npo = 6 + L;
  const int ncore = npo + 6 * nch + pd + nd + 1, ecol = ncore - 1;
  const int nt = (ncore + 15) >> 4, ntile = nt * (nt + 1) / 2;
  const int E = ncore * (ncore + 1) / 2;
  int ea[F2_NE], eb[F2_NE];
  double shared_acc[F2_NE];
  if (BUILD) {
    if (tid < ecol) {
      int role, d;
      if (tid < 3) { role = F2_ROLE_QTC; d = tid; }
      else if (tid < 6) { role = F2_ROLE_TTC; d = tid - 3; }
      else if (tid < npo) { role = F2_ROLE_LM; d = tid - 6; }
      else if (tid < npo + 6 * nch) { const int j = (tid - npo) / 6, k = (tid - npo) - 6 * j; role = F2_ROLE_CHAIN + 2 * j + (k >= 3 ? 1 : 0); d = k >= 3 ? k - 3 : k; }
      else if (tid < npo + 6 * nch + pd) { role = F2_ROLE_PROJ; d = tid - (npo + 6 * nch); }
      else { role = F2_ROLE_DIST; d = tid - (npo + 6 * nch + pd); }
      crole[tid] = role; cdim[tid] = d;
    }
#pragma unroll
    for (int j = 0; j < F2_NE; ++j) {   // entry idx -> (a <= b) of the upper triangle, idx = b(b+1)/2 + a
      const int idx = tid + 256 * j;
      int bb = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
      while ((bb + 1) * (bb + 2) / 2 <= idx) ++bb;
      while (bb * (bb + 1) / 2 > idx) --bb;
      ea[j] = idx - bb * (bb + 1) / 2; eb[j] = bb;
      shared_acc[j] = 0.0;
    }
  }
  constexpr int CPP = BUILD ? 128 : 256;   // corners per pass
  const int r = BUILD ? (tid & 1) : 0;
  const int lane = tid & 63, c16 = lane & 15, kr = lane >> 4;
  double Jacc = 0.0;
  int last_fp = -1;
  for (int go = 0; go < run.count; ++go) {
    const GroupRec gr = groups[run.first + go];
    FpView f = fp_view(dc, gr.fp);
    int rc[MC_MAX_ROLES];
#pragma unroll
    for (int q = 0; q < MC_MAX_ROLES; ++q) rc[q] = -1;
    __syncthreads();   // the previous observation is done with cm / lcol / gram (and crole / cdim are published)
    if (tid < (int)(sizeof(F2Common) / 8)) reinterpret_cast<double*>(&cm)[tid] = reinterpret_cast<const double*>(common + run.first + go)[tid];
    if (BUILD) {
      if (tid < ecol) { const int c0 = f.rolecol[crole[tid]]; lcol[tid] = c0 >= 0 ? c0 + cdim[tid] : -1; }
      for (int i = tid; i < ntile * 256; i += 256) gram[i] = 0.0;
      if (f.rolecol[F2_ROLE_QTC] >= 0) rc[F2_ROLE_QTC] = 0;
      if (f.rolecol[F2_ROLE_TTC] >= 0) rc[F2_ROLE_TTC] = 3;
      if (L) rc[F2_ROLE_LM] = 6;
#pragma unroll
      for (int j = 0; j < F2_MAX_CHAIN; ++j) {
        if (j < nch && f.rolecol[F2_ROLE_CHAIN + 2 * j] >= 0) rc[F2_ROLE_CHAIN + 2 * j] = npo + 6 * j;
        if (j < nch && f.rolecol[F2_ROLE_CHAIN + 2 * j + 1] >= 0) rc[F2_ROLE_CHAIN + 2 * j + 1] = npo + 3 + 6 * j;
      }
      if (f.rolecol[F2_ROLE_PROJ] >= 0) rc[F2_ROLE_PROJ] = npo + 6 * nch;
      if (nd && f.rolecol[F2_ROLE_DIST] >= 0) rc[F2_ROLE_DIST] = npo + 6 * nch + pd;
    }
    __syncthreads();
    f64x4 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < gr.term_count; base += CPP) {
      const int k = base + (BUILD ? (tid >> 1) : tid);
      const int nrows = 2 * min(CPP, gr.term_count - base);
      if (k < gr.term_count) {
        const int term = gr.term_first + k;
        double e[2];
        TermOut out;
        double* row = rows + tid * ld;
        // the inactive roles' columns keep whatever the tile held: their Gram entries are never flushed
        f2_term_cm(b, cm, dc.x, term, dc.useM != 0, BUILD, rc, BUILD ? row - r * ld : nullptr, ld, e, out, BUILD ? (1 << r) : 3);
        if (r == 0) { Jacc += out.w * out.chi2; dc.chi2[b.term_base_global + term] = out.chi2; }
        if (dc.e) {
          if (BUILD) dc.e[b.row_base + 2 * term + r] = -e[r];
          else { dc.e[b.row_base + 2 * term] = -e[0]; dc.e[b.row_base + 2 * term + 1] = -e[1]; }
        }
        if (BUILD) row[ecol] = e[r];
      }
      if (BUILD) {
        __syncthreads();
        const int nsteps = (nrows + 3) >> 2;
        for (int st = tid >> 6; st < nsteps; st += 4) {
          const int rr = 4 * st + kr;
          double v0 = 0.0, v1 = 0.0, v2 = 0.0;
          if (rr < nrows) {
            const double* rp = rows + rr * ld + c16;
            if (c16 < ncore) v0 = rp[0];
            if (16 + c16 < ncore) v1 = rp[16];
            if (32 + c16 < ncore) v2 = rp[32];
          }
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc[0], 0, 0, 0);
          if (nt > 1) {
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc[2], 0, 0, 0);
          }
          if (nt > 2) {
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v2, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v2, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(v2, v2, acc[5], 0, 0, 0);
          }
        }
        if (L && f.rolecol[F2_ROLE_LM] < 0) {
          // landmark x target pose of this observation: task = (term of the pass, landmark component a, pose column j < 6); the
          // two block pairs' places in the value array come precomputed per term (Batch::lm_slots).  These entries belong to
          // one (observation, landmark) pair.  a runs fastest: three consecutive addresses when the landmark block is the row
          // block.  Everything else of the landmark's rows (x camera-level columns, x itself, rhs) is common to all
          // observations of the landmark: k_static_lm, landmark-major
          const int npass = nrows >> 1;
          // consecutive lanes should meet consecutive addresses: with the landmark as the column block a column holds the 6
          // pose rows of the observation (j fastest); with the landmark as the row block the rows of a column are the
          // landmarks, those of neighbouring corners next to one another (a fastest, then the term, j slowest)
          const int4* slots = reinterpret_cast<const int4*>(b.lm_slots) + gr.term_first + base;
          const bool lm_is_col = (slots[0].z & 1) != 0 || (slots[0].z < 0 && (slots[0].w & 1) != 0);
          for (int it = tid; it < npass * 18; it += 256) {
            int a, j, kk;
            if (lm_is_col) { j = it % 6; a = (it / 6) % 3; kk = it / 18; }
            else { a = it % 3; kk = (it / 3) % npass; j = it / (3 * npass); }
            const int4 sl = slots[kk];
            const int slot = j < 3 ? sl.x : sl.y, ho = j < 3 ? sl.z : sl.w, jc = j < 3 ? j : j - 3;
            if (slot < 0) continue;   // constant / inactive point, inactive pose
            const double* r0 = rows + (2 * kk) * ld;
            const double m = r0[6 + a] * r0[j] + r0[ld + 6 + a] * r0[ld + j];
            const int h = ho >> 1;
            const int addr = (ho & 1) ? slot + a * h + (a * (a + 1)) / 2 + jc     // landmark block = column block
                                      : slot + jc * h + (jc * (jc + 1)) / 2 + a;  // target-pose block = column block
            atomic_add_f64(dc.Cx + addr, m);
          }
        }
        __syncthreads();   // the row tile is rewritten by the next pass
      }
    }
    if (BUILD) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
        if (t < ntile) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) atomicAdd(&gram[(t * 4 + rg) * 64 + lane], acc[t][rg]);
        }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < F2_NE; ++j) {
        const int idx = tid + 256 * j;
        if (idx >= E || ea[j] == ecol) continue;
        const int a = ea[j], bq = eb[j];
        const int ta = a >> 4, tb = bq >> 4, i = a & 15, jj = bq & 15;
        const double m = gram[((tb * (tb + 1) / 2 + ta) * 4 + (i >> 2)) * 64 + (i & 3) * 16 + jj];
        if (a >= npo) { shared_acc[j] += m; continue; }
        const int la = lcol[a];
        if (la < 0) continue;
        if (bq == ecol) { atomic_add_f64(dc.rhs + global_col(f, la), -m); continue; }
        const int lb = lcol[bq];
        if (lb < 0) continue;
        atomic_add_f64(dc.Cx + entry_addr(f, dc.col_h, min(la, lb), max(la, lb)), m);
      }
      last_fp = gr.fp;
    }
  }
  Jacc = wave_sum(Jacc);
  if (lane == 0) red[tid >> 6] = Jacc;
  __syncthreads();
  if (tid == 0) dc.Jpart[jpart_base + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (!BUILD || last_fp < 0) return;
  // camera-level part, through the last observation's footprint (same camera: same baseline / intrinsics blocks)
  FpView f = fp_view(dc, last_fp);
#pragma unroll
  for (int j = 0; j < F2_NE; ++j) {
    const int idx = tid + 256 * j;
    if (idx >= E || ea[j] == ecol || ea[j] < npo) continue;
    const int la = lcol[ea[j]];
    if (la < 0) continue;
    const double m = shared_acc[j];
    if (eb[j] == ecol) { atomic_add_f64(dc.rhs + global_col(f, la), -m); continue; }
    const int lb = lcol[eb[j]];
    if (lb < 0) continue;
    atomic_add_f64(dc.Cx + entry_addr(f, dc.col_h, min(la, lb), max(la, lb)), m);
  }
}

// F2 with landmark design variables, second pass, landmark-major: one 256-thread workgroup per run of <= 1024 terms of ONE
// landmark in one camera (Batch::lm_groups; the terms come from different observations, so each reads its observation's
// record from the pre-pass array instead of an LDS copy).  Same rows, same MFMA Gram as k_static; flushed are the landmark's
// rows only: x camera-level columns (baselines, intrinsics), x itself, and the landmark part of rhs -- one atomic per
// entry and work item where the observation-major pass would issue one per term on the same few addresses.
__global__ __launch_bounds__(256, 2) void k_static_lm(const StaticReprojBatchView* __restrict__ views, DevCommon dc,
                                                      const GroupRec* __restrict__ lm_groups, const F2Common* __restrict__ common, int ld) {
  extern __shared__ double rows[];   // 256 x ld
  __shared__ double gram[6 * 256];   // upper tile pairs, [tile][reg][lane] = the MFMA C/D layout
  __shared__ int lcol[F2_MAXCORE];
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const GroupRec gr = lm_groups[blockIdx.x];
  const StaticReprojBatchView b = views[gr.batch];
  const FpView f = fp_view(dc, gr.fp);
  const int nch = b.n_chain, pd = b.cam.proj_dim, nd = b.cam.dist_off >= 0 ? 4 : 0;
  const int npo = 9, ncore = npo + 6 * nch + pd + nd + 1, ecol = ncore - 1;
  const int nt = (ncore + 15) >> 4, ntile = nt * (nt + 1) / 2;
  int rc[MC_MAX_ROLES];
#pragma unroll
  for (int q = 0; q < MC_MAX_ROLES; ++q) rc[q] = -1;
  rc[F2_ROLE_LM] = 6;   // the target pose columns 0..5 are not written here: they stay zero and are not flushed
#pragma unroll
  for (int j = 0; j < F2_MAX_CHAIN; ++j) {
    if (j < nch && f.rolecol[F2_ROLE_CHAIN + 2 * j] >= 0) rc[F2_ROLE_CHAIN + 2 * j] = npo + 6 * j;
    if (j < nch && f.rolecol[F2_ROLE_CHAIN + 2 * j + 1] >= 0) rc[F2_ROLE_CHAIN + 2 * j + 1] = npo + 3 + 6 * j;
  }
  if (f.rolecol[F2_ROLE_PROJ] >= 0) rc[F2_ROLE_PROJ] = npo + 6 * nch;
  if (nd && f.rolecol[F2_ROLE_DIST] >= 0) rc[F2_ROLE_DIST] = npo + 6 * nch + pd;
  if (tid < ecol) {
    int role, d;
    if (tid < 6) { role = -1; d = 0; }
    else if (tid < npo) { role = F2_ROLE_LM; d = tid - 6; }
    else if (tid < npo + 6 * nch) { const int j = (tid - npo) / 6, k = (tid - npo) - 6 * j; role = F2_ROLE_CHAIN + 2 * j + (k >= 3 ? 1 : 0); d = k >= 3 ? k - 3 : k; }
    else if (tid < npo + 6 * nch + pd) { role = F2_ROLE_PROJ; d = tid - (npo + 6 * nch); }
    else { role = F2_ROLE_DIST; d = tid - (npo + 6 * nch + pd); }
    const int c0 = role >= 0 ? f.rolecol[role] : -1;
    lcol[tid] = c0 >= 0 ? c0 + d : -1;
  }
  for (int i = tid; i < 256 * ld; i += 256) rows[i] = 0.0;   // columns of inactive roles are never written
  for (int i = tid; i < ntile * 256; i += 256) gram[i] = 0.0;
  __syncthreads();
  const int r = tid & 1, lane = tid & 63, c16 = lane & 15, kr = lane >> 4;
  f64x4 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int base = 0; base < gr.term_count; base += 128) {
    const int k = base + (tid >> 1);
    const int nrows = 2 * min(128, gr.term_count - base);
    double* row = rows + tid * ld;
    if (k < gr.term_count) {
      const int term = b.lm_terms[gr.term_first + k];
      const F2Common* cm = common + b.lm_obs_common[b.term_obs[term]];
      double e[2];
      TermOut out;
      f2_term_cm(b, *cm, dc.x, term, dc.useM != 0, true, rc, row - r * ld, ld, e, out, 1 << r);
      row[ecol] = e[r];
    }
    __syncthreads();
    const int nsteps = (nrows + 3) >> 2;
    for (int st = tid >> 6; st < nsteps; st += 4) {
      const int rr = 4 * st + kr;
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      if (rr < nrows) {
        const double* rp = rows + rr * ld + c16;
        if (c16 < ncore) v0 = rp[0];
        if (16 + c16 < ncore) v1 = rp[16];
        if (32 + c16 < ncore) v2 = rp[32];
      }
      // the landmark columns 6..8 lie in column tile 0: only the tile pairs (0, *) are needed
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc[0], 0, 0, 0);
      if (nt > 1) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v1, acc[1], 0, 0, 0);
      if (nt > 2) acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v2, acc[3], 0, 0, 0);
    }
    __syncthreads();   // the row tile is rewritten by the next pass
  }
#pragma unroll
  for (int t = 0; t < 6; ++t)
    if (t < ntile && (t == 0 || t == 1 || t == 3)) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) atomicAdd(&gram[(t * 4 + rg) * 64 + lane], acc[t][rg]);
    }
  __syncthreads();
  // the landmark's rows: (a in 6..8) x (bq in a..ecol)
  for (int it = tid; it < 3 * (ncore - 6); it += 256) {
    const int a = 6 + it % 3, bq = 6 + it / 3;
    if (bq < a) continue;
    const int ta = a >> 4, tb = bq >> 4, i = a & 15, jj = bq & 15;
    const double m = gram[((tb * (tb + 1) / 2 + ta) * 4 + (i >> 2)) * 64 + (i & 3) * 16 + jj];
    const int la = lcol[a];
    if (la < 0) continue;
    if (bq == ecol) { atomic_add_f64(dc.rhs + global_col(f, la), -m); continue; }
    const int lb = lcol[bq];
    if (lb < 0) continue;
    atomic_add_f64(dc.Cx + entry_addr(f, dc.col_h, min(la, lb), max(la, lb)), m);
  }
}

// ---------------------------------------------------------------------------------------------
// F6 (initialisation problems: GyroscopeError over constants, ErrorTermTransformation).  Every term of a batch writes
// the same few global blocks (<= 12 columns).  One 256-thread workgroup per group of <= 64 terms: thread = term in the
// Jacobian phase (rows into an LDS tile laid out by the footprint, last column = residual), thread = Hessian entry in
// the reduction, one atomic per entry and group.  These problems have a few thousand terms; the shape is chosen for
// simplicity, not for the last microsecond.
template <class View> struct SmallTraits;
template <> struct SmallTraits<GyroPriorBatchView> {
  enum { D = 3 };
  static __device__ __forceinline__ bool eval(const GyroPriorBatchView& b, const double* x, int k, bool useM, bool wantJ,
                                              const int* rolecol, double* J, int ld, double* e, TermOut& out) {
    return gyro_prior_term(b, x, k, useM, wantJ, rolecol, J, ld, e, out);
  }
};
template <> struct SmallTraits<TransformBatchView> {
  enum { D = 6 };
  static __device__ __forceinline__ bool eval(const TransformBatchView& b, const double* x, int k, bool useM, bool wantJ,
                                              const int* rolecol, double* J, int ld, double* e, TermOut& out) {
    return transform_term(b, x, k, useM, wantJ, rolecol, J, ld, e, out);
  }
};
enum { SMALL_LD = 13 };   // <= 12 Jacobian columns + residual
template <class View, bool BUILD>
__global__ __launch_bounds__(256) void k_small(const View* __restrict__ views, DevCommon dc, const GroupRec* __restrict__ groups,
                                               int jpart_base) {
  constexpr int D = SmallTraits<View>::D;
  __shared__ double tile[BUILD ? GLOBAL_GROUP_TERMS * D * SMALL_LD : 1];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const GroupRec gr = groups[blockIdx.x];
  const View b = views[gr.batch];
  FpView f = fp_view(dc, gr.fp);
  const int C = f.C, R = gr.term_count * D;
  if (BUILD) {
    for (int i = tid; i < R * SMALL_LD; i += 256) tile[i] = 0.0;
    __syncthreads();
  }
  double Jacc = 0.0;
  if (tid < gr.term_count) {
    const int term = gr.term_first + tid;
    double e[D];
    TermOut out;
    double* Jrow = tile + (size_t)(tid * D) * SMALL_LD;
    if (SmallTraits<View>::eval(b, dc.x, term, dc.useM != 0, BUILD, f.rolecol, Jrow, SMALL_LD, e, out)) {
      Jacc = out.w * out.chi2;
      dc.chi2[b.term_base_global + term] = out.chi2;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        if (dc.e) dc.e[b.row_base + D * term + r] = -e[r];
        if (BUILD) Jrow[r * SMALL_LD + C] = e[r];
      }
    } else {
      atomicOr(dc.status, STATUS_INTERNAL);
    }
  }
  Jacc = wave_sum(Jacc);
  if ((tid & 63) == 0) red[tid >> 6] = Jacc;
  __syncthreads();
  if (tid == 0) dc.Jpart[jpart_base + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (!BUILD || MC_MODE(dc) == 3) return;
  const int C1 = C + 1;
  for (int idx = tid; idx < C1 * C1; idx += 256) {
    const int cb = idx / C1, ca = idx - cb * C1;
    if (ca > cb || ca == C) continue;
    double s = 0.0;
    for (int r = 0; r < R; ++r) s += tile[r * SMALL_LD + ca] * tile[r * SMALL_LD + cb];
    if (MC_MODE(dc) != 1) { if (s == 1.2345e-300) dc.rhs[0] = 1.0; continue; }
    if (cb == C) atomic_add_f64(dc.rhs + global_col(f, ca), -s);
    else atomic_add_f64(dc.Cx + entry_addr(f, dc.col_h, ca, cb), s);
  }
}

// term evaluation for the J^T kernels: rc[] holds the element index of every role inside the term's CSC column
__device__ __forceinline__ bool jt_eval(const LidarBatchView& b, const double* x, int u, bool useM, int* rc, const int*, int term_lo,
                                        int nbuf, double* J, int ld, int* st) {
  if (rc[F5_ROLE_COEF] >= 0) {
    double t = b.t[u] + (b.toff.off >= 0 ? x[b.toff.off] : 0.0);
    if (!(t >= b.sp.t_min && t <= b.sp.t_max)) { *st = STATUS_TIME_OUT_OF_SPLINE; return false; }
    int shift = spline_segment(b.sp, t) - term_lo;
    if (shift < 0 || shift + S6 > nbuf) { *st = STATUS_BUFFER_EXCEEDED; return false; }
    rc[F5_ROLE_COEF] += 6 * shift;
  }
  double e[1];
  TermOut out;
  if (!f5_term(b, x, u, useM, true, rc, J, ld, e, out)) { *st = STATUS_TIME_OUT_OF_SPLINE; return false; }
  return true;
}
__device__ __forceinline__ bool jt_eval(const ImuBatchView& b, const double* x, int u, bool useM, int* rc, const int*, int, int,
                                        double* J, int ld, int* st) {
  double e[3];
  TermOut out;
  if (!imu_term(b, x, u, useM, true, rc, J, ld, e, out)) { *st = STATUS_TIME_OUT_OF_SPLINE; return false; }
  return true;
}
__device__ __forceinline__ bool jt_eval(const ImuSmBatchView& b, const double* x, int u, bool useM, int* rc, const int*, int, int,
                                        double* J, int ld, int* st) {
  double e[3];
  TermOut out;
  if (!imu_sm_term(b, x, u, useM, true, rc, J, ld, e, out)) { *st = STATUS_TIME_OUT_OF_SPLINE; return false; }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Second backend: J^T in compressed-column form for SparseCholeskyLinearSystemSolver
// (AB/src/SparseCholeskyLinearSystemSolver.cpp:38-45; CompressedColumnJacobianTransposeBuilder.hpp:94-104).  The same
// closed-form rows as buildSystem, but written out instead of multiplied: thread = term, the term functions write
// straight into the term's CSC columns (row stride = elements per column), positioned by block-index order.
struct JtView {
  const long long* start;   // per global term: first value index
  const int* epc;           // per global term: elements per column
  const int* ofp_jt;        // per other-block footprint: role layout (Problem::ofp_jt)
  double* values;
};
__global__ __launch_bounds__(64) void k_jt_reproj(const ReprojBatchView* __restrict__ views, DevCommon dc, const GroupRec* __restrict__ groups,
                                                  const F1Common* __restrict__ common, JtView jt) {
  const GroupRec gr = groups[blockIdx.x];
  const ReprojBatchView b = views[gr.batch];
  const F1Common cm = common[blockIdx.x];
  if (!cm.ok) { if (threadIdx.x == 0) atomicOr(dc.status, STATUS_TIME_OUT_OF_SPLINE); return; }
  FpView f = fp_view(dc, gr.fp);
  int rc[MC_MAX_ROLES];
#pragma unroll
  for (int r = 0; r < MC_MAX_ROLES; ++r) rc[r] = f.rolecol[r];
  if (rc[F1_ROLE_COEF] >= 0) {
    int shift = cm.seg - f.coef_lo;
    if (shift < 0 || shift + S6 > f.nbuf) { if (threadIdx.x == 0) atomicOr(dc.status, STATUS_BUFFER_EXCEEDED); return; }
    rc[F1_ROLE_COEF] += 6 * shift;
  }
  for (int k = threadIdx.x; k < gr.term_count; k += 64) {
    const int term = gr.term_first + k, gterm = b.term_base_global + term;
    double core[2 * F1_CORE], e[2];
    TermOut out;
    f1_term(b, dc.x, cm, term, dc.useM != 0, true, core, F1_CORE, e, out);
    double* J = jt.values + jt.start[gterm];
    const int epc = jt.epc[gterm];
    f1_expand_row(core, cm.w, rc, J);
    f1_expand_row(core + F1_CORE, cm.w, rc, J + epc);
  }
}
__global__ __launch_bounds__(64) void k_jt_static(const StaticReprojBatchView* __restrict__ views, DevCommon dc,
                                                  const GroupRec* __restrict__ groups, JtView jt) {
  const GroupRec gr = groups[blockIdx.x];
  const StaticReprojBatchView b = views[gr.batch];
  FpView f = fp_view(dc, gr.fp);
  for (int k = threadIdx.x; k < gr.term_count; k += 64) {
    const int term = gr.term_first + k, gterm = b.term_base_global + term;
    double e[2];
    TermOut out;
    f2_term(b, dc.x, term, dc.useM != 0, true, f.rolecol, jt.values + jt.start[gterm], jt.epc[gterm], e, out);
  }
}
template <class View>
__global__ __launch_bounds__(64) void k_jt_small(const View* __restrict__ views, DevCommon dc, const GroupRec* __restrict__ groups, JtView jt) {
  const GroupRec gr = groups[blockIdx.x];
  const View b = views[gr.batch];
  FpView f = fp_view(dc, gr.fp);
  for (int k = threadIdx.x; k < gr.term_count; k += 64) {
    const int term = gr.term_first + k, gterm = b.term_base_global + term;
    double e[SmallTraits<View>::D];
    TermOut out;
    SmallTraits<View>::eval(b, dc.x, term, dc.useM != 0, true, f.rolecol, jt.values + jt.start[gterm], jt.epc[gterm], e, out);
  }
}
template <class View>
__global__ __launch_bounds__(128) void k_jt_stream(const StreamBatch<View>* __restrict__ batches, DevCommon dc,
                                                   const ChunkRec* __restrict__ chunks, JtView jt) {
  const ChunkRec ch = chunks[blockIdx.x];
  const View b = batches[ch.batch].v;
  const int* __restrict__ order = batches[ch.batch].order;
  const int* lay = jt.ofp_jt + (size_t)ch.ofp * (3 * MC_MAX_ROLES + 1);
  const int* of = dc.ofp_data + dc.ofp_off[ch.ofp];
  for (int k = threadIdx.x; k < ch.term_count; k += 128) {
    const int u = order[ch.term_first + k], gterm = b.term_base_global + u;
    const int nbuf = ch.pose_nb > 0 ? batches[ch.batch].term_nbuf[u] : 0;
    int rc[MC_MAX_ROLES];
#pragma unroll
    for (int r = 0; r < MC_MAX_ROLES; ++r) rc[r] = lay[r] >= 0 ? lay[r] + lay[MC_MAX_ROLES + r] * 6 * nbuf : -1;
    if (ch.pose_nb == 0) rc[0] = -1;
    int st = 0;
    const bool ok = jt_eval(b, dc.x, u, dc.useM != 0, rc, of, batches[ch.batch].term_lo[u], nbuf, jt.values + jt.start[gterm], jt.epc[gterm], &st);
    if (!ok) atomicOr(dc.status, st);
  }
}

// ---------------------------------------------------------------------------------------------
// BSplineMotionError (BSplineMotionError.hpp:62-77, 93-166): one lane per (spline coefficient i, band offset d) -- the
// block Q(i, i - d) and its mirror image Q(i + d, i)^T.  (One lane per coefficient walked 6 blocks x 36 dependent atomics:
// 63 us for 12 000 coefficients on 47 workgroups.)
struct MotionView {
  const double* Qband;   // n_coeff x S x (D x D), Q(i, i - d) row-major
  const int* slot;       // n_coeff x S: value offset of block (i-d, i) or -1
  int n_coeff, S, D, param_off, first_block, first_col;
  int i_lo, i_hi;        // this rank's coefficients (Problem::motionRange; all of them without shards)
  int row;               // row of this term in e
  int term;              // global term index
};
__global__ __launch_bounds__(256) void k_motion(const MotionView* __restrict__ views, DevCommon dc, int jpart_base) {
  const MotionView m = views[blockIdx.y];
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int D = m.D, S = m.S;
  const int i = m.i_lo + id / S, d = id % S;
  double part = 0.0;
  if (i < m.i_hi) {
    const double* c = dc.x + m.param_off;
    double qc[6] = {0, 0, 0, 0, 0, 0};
    const int j = i - d, jj = i + d;
    if (j >= 0) {  // Q(i, j)
      const double* Q = m.Qband + ((size_t)i * S + d) * D * D;
      for (int r = 0; r < D; ++r) for (int cc = 0; cc < D; ++cc) qc[r] += Q[r * D + cc] * c[j * D + cc];
    }
    if (d > 0 && jj < m.n_coeff) {  // Q(i, jj) = Q(jj, i)^T
      const double* Q = m.Qband + ((size_t)jj * S + d) * D * D;
      for (int r = 0; r < D; ++r) for (int cc = 0; cc < D; ++cc) qc[r] += Q[cc * D + r] * c[jj * D + cc];
    }
    for (int r = 0; r < D; ++r) part += c[i * D + r] * qc[r];
    if (dc.build && m.first_block >= 0) {
      for (int r = 0; r < D; ++r) if (qc[r] != 0.0) atomic_add_f64(dc.rhs + m.first_col + i * D + r, -qc[r]);
      if (j >= 0) {
        const int h = dc.col_h[m.first_block + i];
        const double* Q = m.Qband + ((size_t)i * S + d) * D * D;
        const int base = m.slot[i * S + d];
        for (int cc = 0; cc < D; ++cc)
          for (int r = 0; r < (d == 0 ? cc + 1 : D); ++r)
            atomic_add_f64(dc.Cx + base + cc * h + (cc * (cc + 1)) / 2 + r, Q[r * D + cc]);
      }
    }
  }
  // block reduction of c^T Q c
  __shared__ double red[4];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = red[0] + red[1] + red[2] + red[3];
    dc.Jpart[jpart_base + blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
  if (id == 0) { if (dc.e) dc.e[m.row] = 0.0; }
}
// chi^2 of a motion term = sum of its partials (raw squared error, BSplineMotionError.hpp:76)
__global__ void k_motion_chi2(const MotionView* __restrict__ views, const double* Jpart, int base, int n, double* chi2) {
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += Jpart[base + blockIdx.x * n + i];
    chi2[views[blockIdx.x].term] = s;
  }
}

// deterministic final reduction of the per-wave partial costs
__global__ __launch_bounds__(256) void k_sum(const double* __restrict__ part, int n, double* out) {
  __shared__ double red[256];
  double s = 0.0;
  // block b sums the slice [b * per, (b+1) * per) in a fixed order -> out[b]; a second launch with one block finishes
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  for (int i = lo + threadIdx.x; i < hi; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// multi-GPU combination: gather / scatter of the value entries that more than one rank writes
// folds the replicas of the contended entries into Cx / rhs (runs after every build kernel has finished)
__global__ __launch_bounds__(256) void k_rep_reduce(const int* __restrict__ idx, const int* __restrict__ addr, int n,
                                                    const double* __restrict__ rep, int stride, double* __restrict__ Cx,
                                                    const int* __restrict__ rhs_col, int ncols, const double* __restrict__ rep_rhs,
                                                    double* __restrict__ rhs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int k = idx[i];
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < Problem::REP_COPIES; ++r) s += rep[(size_t)r * stride + k];
    Cx[addr[i]] += s;
  } else if (i < n + ncols) {
    const int g = i - n;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < Problem::REP_COPIES; ++r) s += rep_rhs[(size_t)r * ncols + g];
    rhs[rhs_col[g]] += s;
  }
}
__global__ __launch_bounds__(256) void k_pack(const int* __restrict__ idx, int n, const double* __restrict__ src, double* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_unpack(const int* __restrict__ idx, int n, const double* __restrict__ src, double* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}

// LiDAR.findPointsOnTarget (kalibr_sensor_calibration/sensors_and_targets.py:199-235): the front-end's per-point Python
// loop -- transformMeasurementsToWorldFrame (:199-219: p_w = T_w_b(t + d) * inverse(T_l_b) * p for every reading whose
// time lies strictly inside the spline) followed by _onPlane (:221-227: p_w into every target's frame, strict box test
// with `threshold` of slack along the plane normal).  One lane per reading; bit p of the mask = inside target p's box,
// bit 31 = time inside the spline (the readings the reference keeps).
struct AssocPlanes { int n; int q_off[30], t_off[30]; double lo[30][3], hi[30][3]; };
__global__ __launch_bounds__(256) void k_lidar_associate(SplineView sp, const double* __restrict__ x, int toff_off, int q_lb_off,
                                                         int t_lb_off, AssocPlanes pl, const double* __restrict__ pts, int n,
                                                         unsigned* __restrict__ mask, double* __restrict__ world) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double t = pts[4 * (size_t)i + 3] + (toff_off >= 0 ? x[toff_off] : 0.0);
  unsigned m = 0;
  double pw[3] = {0.0, 0.0, 0.0};
  if (t > sp.t_min && t < sp.t_max) {
    PoseSample ps;
    if (pose_sample<1>(sp, x, t, ps)) {
      m = 0x80000000u;
      Rigid Tlb;
      load_rigid(x, q_lb_off, t_lb_off, Tlb);
      const double p[3] = {pts[4 * (size_t)i], pts[4 * (size_t)i + 1], pts[4 * (size_t)i + 2]};
      const double d[3] = {p[0] - Tlb.r[0], p[1] - Tlb.r[1], p[2] - Tlb.r[2]};
      double pb[3];
      m3_tmulv(Tlb.C, d, pb);             // inverse(T_l_b) * p
      m3_mulv(ps.C, pb, pw);
      pw[0] += ps.c[0]; pw[1] += ps.c[1]; pw[2] += ps.c[2];
      for (int k = 0; k < pl.n; ++k) {
        Rigid Tpw;
        load_rigid(x, pl.q_off[k], pl.t_off[k], Tpw);
        double q[3];
        m3_mulv(Tpw.C, pw, q);
        bool in = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) { const double v = q[a] + Tpw.r[a]; in = in && v > pl.lo[k][a] && v < pl.hi[k][a]; }
        if (in) m |= 1u << k;
      }
    }
  }
  mask[i] = m;
  if (world) { world[3 * (size_t)i] = pw[0]; world[3 * (size_t)i + 1] = pw[1]; world[3 * (size_t)i + 2] = pw[2]; }
}

// Optimizer2::applyStateUpdate (Optimizer2.cpp:292-309) + every DesignVariable::updateImplementation
struct UpdRec { int kind, dim, off, col; double scaling; int mask; };
__global__ __launch_bounds__(256) void k_update(const UpdRec* __restrict__ recs, int n, const double* __restrict__ dx, double* x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  UpdRec u = recs[i];
  const double* d = dx + u.col;
  double* p = x + u.off;
  if (u.kind == DV_ROTATION_QUATERNION || u.kind == DV_HOMOGENEOUS_POINT) {   // both: sm::kinematics::updateQuat (HomogeneousPoint.cpp:26-32)
    double dq[3] = {d[0] * u.scaling, d[1] * u.scaling, d[2] * u.scaling};
    double q[4] = {p[0], p[1], p[2], p[3]}, o[4];
    update_quat(q, dq, o);
    p[0] = o[0]; p[1] = o[1]; p[2] = o[2]; p[3] = o[3];
  } else if (u.kind == DV_EUCLIDEAN_DIRECTION) {
    // C <- C * R_ypr(0, dp0, dp1)^T   (ABE/src/EuclideanDirection.cpp:33-38)
    double R[9], C[9], O[9];
    ypr0_to_r(d[0] * u.scaling, d[1] * u.scaling, R);
    for (int k = 0; k < 9; ++k) C[k] = p[k];
    m3_mult(C, R, O);
    for (int k = 0; k < 9; ++k) p[k] = O[k];
  } else if (u.kind == DV_MATRIX_BASIC) {
    // A += unvec(B d): minimal parameter k is the k-th flagged entry in row-major order (ABE/src/MatrixBasic.cpp:17-25, 40-46)
    int k = 0;
    for (int idx = 0; idx < 9; ++idx)
      if ((u.mask >> idx) & 1) p[idx] += d[k++] * u.scaling;
  } else {
    for (int k = 0; k < u.dim; ++k) p[k] += d[k] * u.scaling;
  }
}

// ---------------------------------------------------------------------------------------------
template <class T> struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void upload(const std::vector<T>& h) {
    release();
    n = h.size();
    if (n == 0) return;
    HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
    HIP_CHECK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (n == 0) return;
    HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
    HIP_CHECK(hipMemset(p, 0, n * sizeof(T)));
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
  ~DevBuf() { release(); }
};

// Per-batch device arrays (the SoA term records).
struct DevBatch {
  DevBuf<double> t, y, p, sinv, S, meas, range, Qband;
  DevBuf<int> obs_first, obs_q_off, obs_t_off, term_q_off, term_t_off, term_obs, slot, order, term_lo, term_nbuf, lm_terms, lm_obs_common, lm_slots;
};

// One kernel launch per term family: the batches (sensor streams) of a family share the launch through a device array
// of their views; work items (groups / chunks) carry the index of their batch.
enum { LAUNCH_REPROJ = 0, LAUNCH_STATIC = 1, LAUNCH_IMU = 2, LAUNCH_LIDAR = 3, LAUNCH_MOTION = 4, LAUNCH_IMU_SM = 5,
       LAUNCH_GYRO_PRIOR = 6, LAUNCH_TRANSFORMATION = 7, N_LAUNCH = 8 };
struct LaunchUnit {
  int n_batches = 0, n_terms = 0, n_items = 0, jpart_base = 0, n_jpart = 0, lds_bytes = 0;
  double flops_dense = 0, flops_live = 0, rec_bytes = 0;
  StreamDims dims{0, 0, 0, 0, 0, 0};
  int motion_blocks = 0;
  DevBuf<GroupRec> groups;
  DevBuf<GroupRec> lm_groups;   // F2 with landmark design variables: work items of the landmark-major pass (k_static_lm)
  int n_lm_groups = 0;
  DevBuf<GroupRec> jt_groups;   // F2 with landmark design variables: the per-term groups the J^T kernel walks (groups = whole observations)
  int n_jt_items = 0;
  DevBuf<ChunkRec> chunks;
  DevBuf<ReprojBatchView> rviews;
  DevBuf<StaticReprojBatchView> sviews;
  DevBuf<StreamBatch<ImuBatchView>> iviews;
  DevBuf<StreamBatch<LidarBatchView>> lviews;
  DevBuf<StreamBatch<ImuSmBatchView>> xviews;
  DevBuf<MotionView> mviews;
  DevBuf<GyroPriorBatchView> gviews;
  DevBuf<TransformBatchView> tviews;
  DevBuf<F2Common> common2;  // F2: per observation, rewritten by k_static_common before every k_static
  DevBuf<F1Common> common;   // per image observation, rewritten by k_obs_common before every k_reproj
  DevBuf<F1Chain> chain;     // per image observation: matrix of the chain below the camera (build passes)
  bool reproj_equi = false, reproj_expw = false;   // any equidistant camera / any Blake-Zisserman batch in the launch
  bool compact_bias = true;                        // IMU launches: no batch with a general sqrtInvR (k_stream<.., CB = true>)
  int max_gs = 0, max_nbc = 0;                     // IMU launches: stored other columns / bias coefficients per chunk, at most
  DevBuf<ImageRec> images;   // F1: runs of observations of one camera at one time; F2: runs of observations of one camera
  int n_images = 0;
  // F1, split build path (k_reproj_gram / k_f1_ranges / k_reproj_assemble); f1_split = false keeps k_reproj_build_img
  bool f1_split = false;
  bool f1_owner = false;     // assembly by owner tiles (k_reproj_assemble, MC_F1_ASSEMBLE=1) instead of k_reproj_expand
  int n_passes = 0, f1_lds = 0;
  DevBuf<F1Pass> passes;
  DevBuf<double> Gbuf;       // per observation: 7 x 7 Gram of [a, a x p_c | residual]
  DevBuf<F1ObsTab> otab;
  DevBuf<F1Lite> lite;
  DevBuf<int> f1_band, f1_hcol, f1_arrow, f1_col_gb, f1_col_sc, f1_gb_h, f1_ranges, f1_gbase;
  F1Asm f1asm{};
  int static_ld = 0;         // F2: row stride of the LDS tile
};

}  // namespace mc

using namespace mc;

// ---------------------------------------------------------------------------------------------
// Scratch of the allocation-free entry point (mc_build_system_device_noalloc): the replicas of the contended
// global x global entries live in this file-scope device array instead of a buffer the context allocated.  Every GPU the
// library runs on holds its own copy (4.05 MiB) until the process ends.  In host code the array's NAME is only a stand-in:
// the device address is looked up once per device (mc_create, with that device current) and kept in g_scratch_ptr[];
// host code below uses that pointer, never the name.
enum { NOALLOC_MAX_COLS = 256,
       NOALLOC_SCRATCH_DOUBLES = Problem::REP_COPIES * (NOALLOC_MAX_COLS * (NOALLOC_MAX_COLS + 1) / 2 + NOALLOC_MAX_COLS) };
static __device__ __align__(256) double scratch[NOALLOC_SCRATCH_DOUBLES];
enum { MAX_DEVICES = 64 };
static double* g_scratch_ptr[MAX_DEVICES];   // per device, set by scratchOfDevice()
// device must be the current device
static double* scratchOfDevice(int device) {
  if (device < 0 || device >= MAX_DEVICES) throw std::runtime_error("device index beyond the scratch table");
  if (!g_scratch_ptr[device]) {
    double* p = nullptr;
    HIP_CHECK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    if (!p) throw std::runtime_error("hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[device] = p;
  }
  return g_scratch_ptr[device];
}

// One launch clears everything a pass starts from zero: pieces of at most CLEAR_PIECE doubles of the caller's e / Cx / rhs and
// of the context's cost partials and replicas (five hipMemsetAsync calls before: 5 us of launch each on a 1.3 ms build).  The
// table is laid out [e][cost partials][Cx, rhs][replicas] so that every kind of pass takes one contiguous slice of it.
struct ClearPiece { long long off; int len; int buf; };
struct ClearBufs { double *e, *jpart, *cx, *rhs, *rep; };
enum { CLEAR_PIECE = 8192 };
__global__ __launch_bounds__(256) void k_clear(const ClearPiece* __restrict__ pieces, ClearBufs b) {
  const ClearPiece c = pieces[blockIdx.x];
  double* p = (c.buf == 0 ? b.e : c.buf == 1 ? b.jpart : c.buf == 2 ? b.cx : c.buf == 3 ? b.rhs : b.rep) + c.off;
  int n = c.len;
  if ((reinterpret_cast<uintptr_t>(p) & 8) != 0) {
    if (threadIdx.x == 0) p[0] = 0.0;
    ++p; --n;
  }
  double2* p2 = reinterpret_cast<double2*>(p);
  for (int i = threadIdx.x; i < (n >> 1); i += 256) p2[i] = make_double2(0.0, 0.0);
  if ((n & 1) && threadIdx.x == 0) p[n - 1] = 0.0;
}
struct ClearTable {
  DevBuf<ClearPiece> pieces;
  int n_e = 0, n_j = 0, n_sys = 0, n_rep = 0;
};

struct mc_context {
  Problem P;
  std::string err;
  int device = 0;
  DevBuf<double> x, x_prev, knots, basis, Cx, rhs, e, chi2, Jpart, Jpart2, J, dx;
  DevBuf<int> fp_data, fp_off, col_h, status, ofp_data, ofp_off, pat_colptr, pat_rows, pat_roff, pat_cp0, chunk_tab, shared_idx, shared_rhs;
  DevBuf<int> gcb, rep_idx, rep_addr, rep_rhs_col;
  DevBuf<double> rep;   // REP_COPIES x (triangle + rep_cols): matrix replicas, then rhs replicas
  int rep_stride = 0;
  DevBuf<UpdRec> upd;
  DevBuf<long long> jt_start;
  DevBuf<int> jt_epc, ofp_jt;
  DevBuf<double> jt_values;
  DevBuf<unsigned long long> phase;
  bool phase_on = false;
  std::vector<std::unique_ptr<DevBatch>> db;
  LaunchUnit lu[N_LAUNCH];
  int n_jpart = 0;
  ClearTable clear_all, clear_written;   // everything / with shards, only what this rank's launches write
  bool written_only = false;             // mc_set_clear_mode
  bool have_prev = false;
  // The term families only meet in the fp64 atomics on Cx / rhs, so their launches are independent: each family goes to
  // its own HIP stream (forked from / joined into the caller's stream with events) and the dispatcher fills the CUs one
  // latency-bound kernel leaves idle with another's workgroups.  MC_SERIAL_LAUNCH=1 keeps everything on the caller's stream.
  enum { N_SIDE = 3 };
  hipStream_t side[N_SIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[N_SIDE] = {nullptr, nullptr, nullptr};
  int side_state = 0;   // 0: not created yet, 1: ready, -1: disabled
  // Stream contract.  The *_device entry points enqueue on the caller's stream and return; everything else (apply / revert,
  // status, parameter and chi^2 read-back) runs on the null stream, and torch's pool streams do not synchronise with it.  The
  // end of every launch sequence is marked with this event and the null-stream entry points wait for it first, so an update can
  // never overwrite x while build kernels still read it and a status read never precedes the kernels that set it.
  hipEvent_t ev_done = nullptr;
  bool done_recorded = false;
  void waitForLaunches() { if (done_recorded) HIP_CHECK(hipEventSynchronize(ev_done)); }
  // End of a launch sequence.  Under stream capture nothing runs yet, and an event recorded into the capture cannot be
  // waited for from the host (hipEventSynchronize rejects it): the graph's owner orders its replays against the
  // null-stream entry points itself, so no event is recorded then and an earlier one stays valid.
  void markDone(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return;
    HIP_CHECK(hipEventRecord(ev_done, st));
    done_recorded = true;
  }
  ~mc_context() {
    for (int i = 0; i < N_SIDE; ++i) { if (side[i]) (void)hipStreamDestroy(side[i]); if (ev_join[i]) (void)hipEventDestroy(ev_join[i]); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_done) (void)hipEventDestroy(ev_done);
  }
  bool sideStreams() {
    if (side_state == 0) {
      const char* s = getenv("MC_SERIAL_LAUNCH");
      if (s && s[0] == '1') { side_state = -1; return false; }
      // With shards the step also holds a collective whose library brings streams of its own, and the process has
      // GPU_MAX_HW_QUEUES = 4 hardware queues.  The overlap is worth 2 % and the N > 1 runs cannot be rehearsed with RCCL on a
      // one-GPU box, so sharded contexts launch serially unless MC_SERIAL_LAUNCH=0 asks for the side streams.
      if (P.world > 1 && !(s && s[0] == '0')) { side_state = -1; return false; }
      for (int i = 0; i < N_SIDE; ++i) {
        HIP_CHECK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
      }
      HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
      side_state = 1;
    }
    return side_state == 1;
  }

  void uploadStructure();
  void setupF1Split(const std::vector<GroupRec>& groups);
  void launchF1Assemble(const DevCommon& dc, hipStream_t st) {
    LaunchUnit& L = lu[LAUNCH_REPROJ];
    hipLaunchKernelGGL(k_reproj_assemble, dim3(L.f1asm.n_tiles), dim3(256), L.f1_lds, st, dc, L.groups.p, L.chain.p, L.Gbuf.p, L.f1asm);
    HIP_CHECK(hipGetLastError());
  }
  // rep_buf: replica buffer to use instead of the context's own (the allocation-free entry point passes the per-device
  // scratch array); serial: keep every launch on the caller's stream (no side streams are created or used)
  void launchAll(int build, int useM, double* dCx, double* dRhs, double* dJ, double* dE, hipStream_t st,
                 std::vector<hipEvent_t>* ev = nullptr, double* rep_buf = nullptr, bool serial = false);
  void launchJt(int useM, double* dValues, hipStream_t st);
};

static thread_local std::string g_create_error;

static int launchOf(int kind) {
  switch (kind) {
    case KIND_REPROJ: return LAUNCH_REPROJ;
    case KIND_STATIC_REPROJ: return LAUNCH_STATIC;
    case KIND_ACCEL: case KIND_GYRO: return LAUNCH_IMU;
    case KIND_ACCEL_SM: case KIND_GYRO_SM: return LAUNCH_IMU_SM;
    case KIND_LIDAR: return LAUNCH_LIDAR;
    case KIND_GYRO_PRIOR: return LAUNCH_GYRO_PRIOR;
    case KIND_TRANSFORMATION: return LAUNCH_TRANSFORMATION;
    default: return LAUNCH_MOTION;
  }
}

void batchCounts(const Problem& P, const Batch& b, double& fd, double& fl, double& rec) {
  // buildSystem flops with every structurally present block (dense: what the reference computes, padded
  // zero blocks included), with the live blocks only, and the compulsory record bytes
  static const double fjac[N_KINDS] = {3500.0, 3500.0, 6000.0, 3000.0, 3000.0, 0.0, 9000.0, 9000.0, 100.0, 600.0};
  static const double recb[N_KINDS] = {72.0, 72.0, 52.0, 52.0, 64.0, 0.0, 52.0, 52.0, 56.0, 56.0};
  fd = b.flops_dense; fl = b.flops_live;
  for (const GroupRec& g : b.groups) {
    double s1 = 0, s2 = 0, l1 = 0, l2 = 0;
    for (int blk : P.fp_pattern[g.fp]) { double d = P.block_dim[blk]; s1 += d; s2 += d * d; }
    for (auto& w : P.fp_written[g.fp]) {
      int n = w[2];
      if (w[0] == 0 && b.kind == KIND_REPROJ) n = std::min(n, 6);  // live support
      for (int k = 0; k < n; ++k) { double d = P.block_dim[w[1] + k]; l1 += d; l2 += d * d; }
    }
    double D = b.dim;
    fd += g.term_count * (2.0 * D * 0.5 * (s1 * s1 + s2) + 2.0 * D * s1 + fjac[b.kind]);
    fl += g.term_count * (2.0 * D * 0.5 * (l1 * l1 + l2) + 2.0 * D * l1 + fjac[b.kind]);
  }
  if (b.kind == KIND_MOTION) {
    const SplineRec& sp = P.splines[b.spline];
    fd = fl = 2.0 * sp.n_coeff * (2 * sp.order - 1) * sp.dim * sp.dim;
  }
  rec = recb[b.kind] * b.n_terms;
}

// Tables of the split F1 build path (see k_reproj_assemble): corner passes, the global design variables the observations
// touch and their accumulator columns, and per owner tile the value offsets of its band blocks and of its rows in the global
// columns.  Leaves f1_split = false (k_reproj_build_img stays) when the problem does not have the shape the path assumes.
void mc_context::setupF1Split(const std::vector<GroupRec>& groups) {
  LaunchUnit& L = lu[LAUNCH_REPROJ];
  L.f1_split = false;
  {   // default while the assembly kernel is being tuned: off (MC_F1_SPLIT=1 turns it on)
    const char* e = std::getenv("MC_F1_SPLIT");
    if (!e || e[0] != '1') return;
  }
  if (groups.empty() || P.world > 1) return;   // shards keep the per-image kernel (its writers are what finalize() marked)
  // one active pose spline behind every batch, observations of a batch in time order, corners contiguous
  std::vector<const Batch*> rb;
  for (const Batch& b : P.batches) if (b.kind == KIND_REPROJ) rb.push_back(&b);
  const int spline = rb[0]->spline;
  for (const Batch* b : rb) if (b->spline != spline) return;
  const SplineRec& sp = P.splines[spline];
  const int block0 = P.dvs[sp.first_dv].block;
  if (block0 < 0) return;
  std::vector<int> gbase(rb.size() + 1, 0);
  {
    size_t g = 0;
    for (size_t k = 0; k < rb.size(); ++k) {
      gbase[k] = (int)g;
      double last = -INFINITY;
      int next_term = -1;
      while (g < groups.size() && groups[g].batch == (int)k) {
        const double tg = rb[k]->t[groups[g].aux];
        if (tg < last) return;
        last = tg;
        if (next_term >= 0 && groups[g].term_first != next_term) return;
        next_term = groups[g].term_first + groups[g].term_count;
        ++g;
      }
    }
    gbase[rb.size()] = (int)g;
    if (g != groups.size()) return;
  }
  // corner passes: 64 consecutive corners of one batch
  std::vector<F1Pass> passes;
  for (size_t k = 0; k < rb.size(); ++k) {
    int g = gbase[k];
    const int gend = gbase[k + 1];
    if (g == gend) continue;
    int term = groups[g].term_first;
    const int term_end = groups[gend - 1].term_first + groups[gend - 1].term_count;
    while (term < term_end) {
      while (groups[g].term_first + groups[g].term_count <= term) ++g;
      const int n = std::min(64, term_end - term);
      int g2 = g;
      while (g2 + 1 < gend && groups[g2 + 1].term_first < term + n) ++g2;
      passes.push_back(F1Pass{(int)k, term, n, g, g2 - g + 1});
      term += n;
    }
  }
  // global design variables of the observations -> accumulator columns
  std::map<int, int> gb_index;      // block -> position in the list
  std::vector<int> gb_block, gb_col0;
  int ncol = 0;
  auto gbOf = [&](int dv) {
    if (dv < 0 || P.dvs[dv].block < 0) return -1;
    const int blk = P.dvs[dv].block;
    auto it = gb_index.find(blk);
    if (it != gb_index.end()) return it->second;
    const int id = (int)gb_block.size();
    gb_index.emplace(blk, id);
    gb_block.push_back(blk);
    gb_col0.push_back(ncol);
    ncol += P.block_dim[blk];
    return id;
  };
  const bool with_rep = rep_stride > 0;
  std::vector<F1ObsTab> otab(groups.size());
  for (size_t g = 0; g < groups.size(); ++g) {
    const Batch& b = *rb[groups[g].batch];
    const int u = groups[g].aux;
    const int dv[5] = {b.dv[0], b.dv[1], b.obs_dv_q[u], b.obs_dv_t[u], b.dv[2]};
    static const int want[5] = {3, 3, 3, 3, 1}, role[5] = {F1_ROLE_QCB, F1_ROLE_TCB, F1_ROLE_QPW, F1_ROLE_TPW, F1_ROLE_TOFF}, core[5] = {6, 9, 12, 15, 18};
    F1ObsTab& tb = otab[g];
    std::memset(&tb, 0, sizeof tb);
    const int* fp = &P.fp_data[P.fp_off[groups[g].fp]];
    tb.has_coef = fp[4 + F1_ROLE_COEF] >= 0 ? 1 : 0;
    for (int r = 0; r < 5; ++r) {
      const int id = gbOf(dv[r]);
      if (id < 0) continue;
      if (P.block_dim[gb_block[id]] != want[r] || fp[4 + role[r]] < 0) return;
      for (int d = 0; d < want[r]; ++d) {
        const int at = tb.n_glob++;
        tb.glc[at] = fp[4 + role[r]] + d;
        tb.gk[at] = core[r] + d;
        tb.ggc[at] = (with_rep && P.gcb[gb_block[id]] >= 0) ? P.gcb[gb_block[id]] + d : -1;
        tb.acol[at] = gb_col0[id] + d;
      }
    }
  }
  const int nGB = (int)gb_block.size();
  std::vector<int> col_gb(std::max(ncol, 1), 0), col_sc(std::max(ncol, 1), 0), gb_h(std::max(nGB, 1), 0);
  for (int i = 0; i < nGB; ++i) {
    gb_h[i] = P.col_h[gb_block[i]];
    for (int c = 0; c < P.block_dim[gb_block[i]]; ++c) { col_gb[gb_col0[i] + c] = i; col_sc[gb_col0[i] + c] = c; }
  }
  // tile size: the largest whose accumulators leave three workgroups per CU
  int K = 0, lds = 0;
  const int forced = std::getenv("MC_F1_TILE") ? atoi(std::getenv("MC_F1_TILE")) : 0;
  for (int cand : {16, 8, 4, 2}) {
    if (forced > 0 && cand != forced) continue;
    const int R = 6 * cand;
    const size_t doubles = (size_t)R * 36 + (size_t)R * ncol + R + 4 * 552;
    const size_t ints = (size_t)R + cand + (size_t)nGB * cand + 2 * rb.size() + 1;
    const size_t bytes = doubles * sizeof(double) + ints * sizeof(int) + 16;
    if (bytes <= 53 * 1024 || (forced > 0 && bytes <= 150 * 1024)) { K = cand; lds = (int)bytes; break; }
  }
  if (K == 0) return;
  const int n_tiles = (sp.n_coeff + K - 1) / K;
  auto lookup = [&](int rb_, int cb_) {
    const std::vector<int>& rows = P.col_rows[cb_];
    auto it = std::lower_bound(rows.begin(), rows.end(), rb_);
    return (it != rows.end() && *it == rb_) ? P.Cp[P.block_col[cb_]] + P.col_roff[cb_][it - rows.begin()] : -1;
  };
  std::vector<int> band((size_t)n_tiles * K * 6, -1), hcol((size_t)n_tiles * K, 0), arrow((size_t)n_tiles * std::max(nGB, 1) * K, -1);
  for (int t = 0; t < n_tiles; ++t)
    for (int k = 0; k < K; ++k) {
      const int c = t * K + k;
      if (c >= sp.n_coeff) continue;
      const int pb = block0 + c;
      hcol[(size_t)t * K + k] = P.col_h[pb];
      for (int d = 0; d < 6 && c - d >= 0; ++d) band[((size_t)t * K + k) * 6 + d] = lookup(pb - d, pb);
      for (int i = 0; i < nGB; ++i) {
        const int gblk = gb_block[i];
        int enc = -1;
        if (gblk > pb) enc = lookup(pb, gblk);
        else { const int s = lookup(gblk, pb); enc = s >= 0 ? -2 - s : -1; }
        arrow[((size_t)t * nGB + i) * K + k] = enc;
      }
    }
  L.passes.upload(passes);
  L.n_passes = (int)passes.size();
  L.Gbuf.alloc(groups.size() * 49);
  L.otab.upload(otab);
  L.lite.alloc(groups.size());
  L.f1_band.upload(band); L.f1_hcol.upload(hcol); L.f1_arrow.upload(arrow);
  L.f1_col_gb.upload(col_gb); L.f1_col_sc.upload(col_sc); L.f1_gb_h.upload(gb_h);
  L.f1_ranges.alloc(rb.size() * 2 * (size_t)(n_tiles + 1));
  L.f1_gbase.upload(gbase);
  F1Asm& A = L.f1asm;
  A.K = K; A.n_tiles = n_tiles; A.ncol = ncol; A.nGB = nGB; A.n_batches = (int)rb.size();
  A.pose_block0 = block0; A.pose_col0 = P.dvs[sp.first_dv].col;
  A.band = L.f1_band.p; A.hcol = L.f1_hcol.p; A.arrow = L.f1_arrow.p;
  A.col_gb = L.f1_col_gb.p; A.col_sc = L.f1_col_sc.p; A.gb_h = L.f1_gb_h.p;
  A.ranges = L.f1_ranges.p; A.gbase = L.f1_gbase.p; A.otab = L.otab.p; A.lite = L.lite.p;
  L.f1_lds = lds;
  if (lds > 64 * 1024)
    HIP_CHECK(hipFuncSetAttribute((const void*)k_reproj_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  L.f1_owner = std::getenv("MC_F1_ASSEMBLE") != nullptr;
  L.f1_split = true;
}

void mc_context::uploadStructure() {
  HIP_CHECK(hipSetDevice(device));
  if (!ev_done) HIP_CHECK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
  x.upload(P.x);
  x_prev.alloc(P.x.size());
  knots.upload(P.knots);
  basis.upload(P.basis);
  fp_data.upload(P.fp_data);
  fp_off.upload(P.fp_off);
  col_h.upload(P.col_h);
  ofp_data.upload(P.ofp_data); ofp_off.upload(P.ofp_off);
  rep_stride = 0;
  if (P.rep_cols > 0 && std::getenv("MC_NO_REPLICAS") == nullptr) {
    gcb.upload(P.gcb); rep_idx.upload(P.rep_idx); rep_addr.upload(P.rep_addr); rep_rhs_col.upload(P.rep_rhs_col);
    rep_stride = P.rep_cols * (P.rep_cols + 1) / 2;
    rep.alloc((size_t)Problem::REP_COPIES * (rep_stride + P.rep_cols));
  }
  pat_colptr.upload(P.pat_colptr); pat_rows.upload(P.pat_rows); pat_roff.upload(P.pat_roff); pat_cp0.upload(P.pat_cp0);
  chunk_tab.upload(P.chunk_tab);
  shared_idx.upload(P.shared_idx);
  shared_rhs.upload(P.shared_rhs);
  status.alloc(1);
  Cx.alloc(P.nnz); rhs.alloc(P.j_cols); e.alloc(P.j_rows); chi2.alloc(P.n_terms); J.alloc(1); dx.alloc(P.j_cols);
  std::vector<UpdRec> ur;
  for (int dv : P.update_dvs) { const DvRec& d = P.dvs[dv]; ur.push_back(UpdRec{d.kind, d.dim, d.off, d.col, d.scaling, d.mask}); }
  upd.upload(ur);
  db.clear();
  std::vector<GroupRec> groups[N_LAUNCH];
  std::vector<ChunkRec> chunks[N_LAUNCH];
  std::vector<ReprojBatchView> rviews;
  std::vector<StaticReprojBatchView> sviews;
  std::vector<StreamBatch<ImuBatchView>> iviews;
  std::vector<StreamBatch<LidarBatchView>> lviews;
  std::vector<StreamBatch<ImuSmBatchView>> xviews;
  std::vector<MotionView> mviews;
  std::vector<GyroPriorBatchView> gviews;
  std::vector<TransformBatchView> tviews;
  std::vector<ImageRec> images, static_runs;
  // F2: observations per workgroup -- enough workgroups to fill the chip first, then longer runs (fewer camera-level flushes)
  int n_static_obs = 0;
  for (const Batch& b : P.batches)
    if (b.kind == KIND_STATIC_REPROJ)
      n_static_obs += (int)(b.obs_groups.empty() || std::getenv("MC_LM_PER_TERM") ? b.my_groups.size() : b.my_obs_groups.size());
  std::vector<GroupRec> jt_groups, lm_groups;
  bool jt_groups_differ = false;
  const int static_run = std::min(8, std::max(1, n_static_obs / 1024));
  for (int l = 0; l < N_LAUNCH; ++l) lu[l] = LaunchUnit();
  int term_base = 0;
  for (Batch& b : P.batches) {
    std::unique_ptr<DevBatch> d(new DevBatch());
    auto offs = [&](const std::vector<int>& ids) { std::vector<int> o(ids.size()); for (size_t i = 0; i < ids.size(); ++i) o[i] = P.dvs[ids[i]].off; return o; };
    LaunchUnit& L = lu[launchOf(b.kind)];
    const int li = launchOf(b.kind);
    const int view_index = L.n_batches++;
    L.n_terms += b.n_terms;
    {
      double fd, fl, rec;
      batchCounts(P, b, fd, fl, rec);
      const double share = 1.0 / P.world;   // this rank's part
      L.flops_dense += fd * share; L.flops_live += fl * share; L.rec_bytes += rec * share;
    }
    const bool stream = kindIsStream(b.kind);
    if (stream) {
      for (ChunkRec c : b.my_chunks) { c.batch = view_index; chunks[li].push_back(c); }
      d->order.upload(b.order); d->term_lo.upload(b.term_lo); d->term_nbuf.upload(b.term_nbuf);
      StreamDims& dm = L.dims;
      dm.maxT = std::max(dm.maxT, b.max_terms); dm.maxNb = std::max(dm.maxNb, b.max_nb);
      dm.maxNob = std::max(dm.maxNob, b.max_nob); dm.maxG = std::max(dm.maxG, b.max_g);
      dm.maxTab = std::max(dm.maxTab, b.max_tab);
      if (li == LAUNCH_IMU || li == LAUNCH_IMU_SM) {
        if (!b.S.empty()) L.compact_bias = false;
        for (const ChunkRec& c : b.my_chunks) {
          const int* of = &P.ofp_data[P.ofp_off[c.ofp]];
          const int bias_0 = of[4 + IMU_ROLE_BIAS], nbc = bias_0 >= 0 ? (b.bias_spline >= 0 ? of[3] : 1) : 0;
          L.max_gs = std::max(L.max_gs, of[1] - 3 * nbc);
          L.max_nbc = std::max(L.max_nbc, nbc);
        }
      }
    } else if (b.kind != KIND_MOTION) {
      double last_time = 0;
      bool first_of_batch = true;
      // landmark batches: the evaluate / build kernel takes whole observations, the J^T kernel the per-term groups
      // (MC_LM_PER_TERM=1: the round-2 shape, one group -- one workgroup pass -- per term over an active landmark; for comparison)
      const bool lm_batch = b.kind == KIND_STATIC_REPROJ && !b.obs_groups.empty() && std::getenv("MC_LM_PER_TERM") == nullptr;
      if (b.kind == KIND_STATIC_REPROJ) {
        jt_groups_differ = jt_groups_differ || lm_batch;
        for (GroupRec g : b.my_groups) if (g.term_count) { g.batch = view_index; jt_groups.push_back(g); }
      }
      if (lm_batch) {
        // where the pre-pass record of an observation sits in the launch's group list, for the landmark-major pass
        std::vector<int> at((int)b.obs_first.size() - 1, -1);
        int pos = (int)groups[li].size();
        for (const GroupRec& g : b.my_obs_groups) if (g.term_count) at[g.aux] = pos++;
        d->lm_obs_common.upload(at);
        d->lm_terms.upload(b.lm_terms);
        for (GroupRec g : b.lm_groups) { g.batch = view_index; lm_groups.push_back(g); }
      }
      for (GroupRec g : (lm_batch ? b.my_obs_groups : b.my_groups)) {
        if (g.term_count == 0) continue;
        g.batch = view_index;
        if (b.kind == KIND_STATIC_REPROJ) {
          if (first_of_batch || static_runs.back().count >= static_run) static_runs.push_back(ImageRec{(int)groups[li].size(), 0});
          static_runs.back().count++;
          first_of_batch = false;
        }
        if (b.kind == KIND_REPROJ) {
          // observations of one camera at one time form an image (they share everything but the board pose)
          double tg = b.t[g.aux];
          if (first_of_batch || tg != last_time || images.back().count >= 16) images.push_back(ImageRec{(int)groups[li].size(), 0});
          images.back().count++;
          last_time = tg; first_of_batch = false;
        }
        groups[li].push_back(g);
      }
      if (b.kind == KIND_STATIC_REPROJ) {
        if ((int)b.chain_q.size() > F2_MAX_CHAIN) throw std::runtime_error("camera chain longer than F2_MAX_CHAIN");
        const CameraRec& c = P.cams[b.cam];
        const int ncore = 6 + (b.term_dv_lm.empty() ? 0 : 3) + 6 * (int)b.chain_q.size() + P.dvs[c.dv_proj].dim + (c.dv_dist >= 0 ? 4 : 0) + 1;
        L.static_ld = std::max(L.static_ld, ncore | 1);   // odd stride: conflict-free row writes and column reads
        L.lds_bytes = 256 * L.static_ld * (int)sizeof(double);
      }
    }
    d->sinv.upload(b.sinv);
    if (!b.S.empty()) d->S.upload(b.S);
    if (b.kind == KIND_REPROJ) {
      d->t.upload(b.t); d->y.upload(b.y); d->p.upload(b.p);
      d->obs_first.upload(b.obs_first); d->obs_q_off.upload(offs(b.obs_dv_q)); d->obs_t_off.upload(offs(b.obs_dv_t));
      ReprojBatchView v;
      v.cam = P.cameraView(b.cam);
      v.sp = P.splineView(b.spline, knots.p, basis.p);
      v.q_cb = P.dvRef(b.dv[0]); v.t_cb = P.dvRef(b.dv[1]); v.toff = P.dvRef(b.dv[2]);
      v.coef_scale = 1.0; v.qpw_scale = 1.0; v.tpw_scale = 1.0;
      if (!b.obs_dv_q.empty()) { v.qpw_scale = P.dvs[b.obs_dv_q[0]].scaling; v.tpw_scale = P.dvs[b.obs_dv_t[0]].scaling; }
      v.obs_time = d->t.p; v.obs_first = d->obs_first.p; v.obs_qpw_off = d->obs_q_off.p; v.obs_tpw_off = d->obs_t_off.p;
      v.y = d->y.p; v.p = d->p.p; v.sinv = d->sinv.p; v.S = b.S.empty() ? nullptr : d->S.p;
      v.mest = b.mest; v.mest_a = b.mest_a; v.n_obs = (int)b.obs_first.size() - 1; v.n_terms = b.n_terms;
      v.row_base = b.row_base; v.term_base_global = term_base;
      rviews.push_back(v);
    } else if (b.kind == KIND_LIDAR) {
      d->t.upload(b.t); d->p.upload(b.p); d->range.upload(b.range);
      d->term_q_off.upload(offs(b.term_dv_q)); d->term_t_off.upload(offs(b.term_dv_t));
      StreamBatch<LidarBatchView> sb;
      LidarBatchView& v = sb.v;
      v.sp = P.splineView(b.spline, knots.p, basis.p);
      v.q_lb = P.dvRef(b.dv[0]); v.t_lb = P.dvRef(b.dv[1]); v.toff = P.dvRef(b.dv[2]);
      v.coef_scale = 1.0; v.qpw_scale = 1.0; v.tpw_scale = 1.0;
      if (!b.term_dv_q.empty()) { v.qpw_scale = P.dvs[b.term_dv_q[0]].scaling; v.tpw_scale = P.dvs[b.term_dv_t[0]].scaling; }
      v.t = d->t.p; v.dir = d->p.p; v.range = d->range.p; v.sinv = d->sinv.p; v.qpw_off = d->term_q_off.p; v.tpw_off = d->term_t_off.p;
      v.mest = b.mest; v.mest_a = b.mest_a; v.n_terms = b.n_terms; v.row_base = b.row_base; v.term_base_global = term_base;
      sb.order = d->order.p; sb.term_lo = d->term_lo.p; sb.term_nbuf = d->term_nbuf.p;
      lviews.push_back(sb);
    } else if (b.kind == KIND_ACCEL_SM || b.kind == KIND_GYRO_SM) {
      d->t.upload(b.t); d->meas.upload(b.meas);
      StreamBatch<ImuSmBatchView> sb;
      ImuSmBatchView& v = sb.v;
      v.pose = P.splineView(b.spline, knots.p, basis.p);
      v.bias = P.splineView(b.bias_spline, knots.p, basis.p);
      v.bias_dv = P.dvRef(b.dv[0]); v.g = P.dvRef(b.dv[1]); v.q_ib = P.dvRef(b.dv[2]); v.r_b = P.dvRef(b.dv[3]);
      v.g_is_direction = P.dvs[b.dv[1]].kind == DV_EUCLIDEAN_DIRECTION ? 1 : 0;
      v.pose_scale = 1.0; v.bias_scale = b.bias_spline >= 0 ? 1.0 : v.bias_dv.scale;
      v.t = d->t.p; v.meas = d->meas.p; v.sinv = d->sinv.p; v.S = b.S.empty() ? nullptr : d->S.p;
      v.mest = b.mest; v.mest_a = b.mest_a; v.is_accel = b.kind == KIND_ACCEL_SM ? 1 : 0;
      v.n_terms = b.n_terms; v.row_base = b.row_base; v.term_base_global = term_base;
      v.M = P.dvRef(b.dv[4]); v.M_mask = P.dvs[b.dv[4]].mask;
      v.q_gi = P.dvRef(b.dv[5]); v.Ma = P.dvRef(b.dv[6]); v.Ma_mask = b.dv[6] >= 0 ? P.dvs[b.dv[6]].mask : 0;
      v.size_effect = b.size_effect;
      for (int j = 0; j < 3; ++j) { v.arm[j] = P.dvRef(b.dv[7 + j]); v.I_off[j] = b.dv[10 + j] >= 0 ? P.dvs[b.dv[10 + j]].off : -1; }
      sb.order = d->order.p; sb.term_lo = d->term_lo.p; sb.term_nbuf = d->term_nbuf.p;
      xviews.push_back(sb);
    } else if (b.kind == KIND_ACCEL || b.kind == KIND_GYRO) {
      d->t.upload(b.t); d->meas.upload(b.meas);
      StreamBatch<ImuBatchView> sb;
      ImuBatchView& v = sb.v;
      v.pose = P.splineView(b.spline, knots.p, basis.p);
      v.bias = P.splineView(b.bias_spline, knots.p, basis.p);
      v.bias_dv = P.dvRef(b.dv[0]); v.g = P.dvRef(b.dv[1]); v.q_ib = P.dvRef(b.dv[2]); v.r_b = P.dvRef(b.dv[3]);
      v.g_is_direction = (b.dv[1] >= 0 && P.dvs[b.dv[1]].kind == DV_EUCLIDEAN_DIRECTION) ? 1 : 0;
      v.pose_scale = 1.0; v.bias_scale = b.bias_spline >= 0 ? 1.0 : v.bias_dv.scale;
      v.t = d->t.p; v.meas = d->meas.p; v.sinv = d->sinv.p; v.S = b.S.empty() ? nullptr : d->S.p;
      v.mest = b.mest; v.mest_a = b.mest_a; v.is_accel = b.kind == KIND_ACCEL ? 1 : 0;
      v.n_terms = b.n_terms; v.row_base = b.row_base; v.term_base_global = term_base;
      sb.order = d->order.p; sb.term_lo = d->term_lo.p; sb.term_nbuf = d->term_nbuf.p;
      iviews.push_back(sb);
    } else if (b.kind == KIND_GYRO_PRIOR) {
      d->p.upload(b.p); d->meas.upload(b.meas);
      GyroPriorBatchView v;
      v.q = P.dvRef(b.dv[0]); v.bias = P.dvRef(b.dv[1]);
      v.w_pred = d->p.p; v.meas = d->meas.p; v.sinv = d->sinv.p; v.S = b.S.empty() ? nullptr : d->S.p;
      v.mest = b.mest; v.mest_a = b.mest_a; v.n_terms = b.n_terms; v.row_base = b.row_base; v.term_base_global = term_base;
      gviews.push_back(v);
    } else if (b.kind == KIND_TRANSFORMATION) {
      d->y.upload(b.y);
      TransformBatchView v;
      v.qa = P.dvRef(b.dv[0]); v.ta = P.dvRef(b.dv[1]); v.qb = P.dvRef(b.dv[2]); v.tb = P.dvRef(b.dv[3]);
      v.prior_inv = d->y.p; v.sw_t = std::sqrt(b.w_t); v.sw_r = std::sqrt(b.w_r);
      v.mest = b.mest; v.mest_a = b.mest_a; v.n_terms = b.n_terms; v.row_base = b.row_base; v.term_base_global = term_base;
      tviews.push_back(v);
    } else if (b.kind == KIND_STATIC_REPROJ) {
      d->y.upload(b.y); d->p.upload(b.p);
      d->obs_first.upload(b.obs_first); d->obs_q_off.upload(offs(b.obs_dv_q)); d->obs_t_off.upload(offs(b.obs_dv_t));
      d->term_obs.upload(b.term_obs);
      StaticReprojBatchView v;
      v.cam = P.cameraView(b.cam);
      v.n_chain = (int)b.chain_q.size();
      for (int j = 0; j < v.n_chain; ++j) { v.chain_q[j] = P.dvRef(b.chain_q[j]); v.chain_t[j] = P.dvRef(b.chain_t[j]); }
      for (int j = v.n_chain; j < F2_MAX_CHAIN; ++j) { v.chain_q[j] = P.dvRef(-1); v.chain_t[j] = P.dvRef(-1); }
      v.qtc_scale = 1.0; v.ttc_scale = 1.0;
      if (!b.obs_dv_q.empty()) { v.qtc_scale = P.dvs[b.obs_dv_q[0]].scaling; v.ttc_scale = P.dvs[b.obs_dv_t[0]].scaling; }
      v.proj_scale = P.dvs[P.cams[b.cam].dv_proj].scaling;
      v.dist_scale = P.cams[b.cam].dv_dist >= 0 ? P.dvs[P.cams[b.cam].dv_dist].scaling : 1.0;
      v.obs_first = d->obs_first.p; v.obs_qtc_off = d->obs_q_off.p; v.obs_ttc_off = d->obs_t_off.p; v.term_obs = d->term_obs.p;
      v.y = d->y.p; v.p = d->p.p; v.sinv = d->sinv.p; v.S = b.S.empty() ? nullptr : d->S.p;
      if (!b.term_dv_lm.empty()) {   // landmark design variables: per term the parameter offset of the point, or -1
        std::vector<int> lo(b.term_dv_lm.size(), -1);
        for (size_t k = 0; k < lo.size(); ++k)
          if (b.term_dv_lm[k] >= 0) { lo[k] = P.dvs[b.term_dv_lm[k]].off; v.lm_scale = P.dvs[b.term_dv_lm[k]].scaling; }
        d->term_q_off.upload(lo);
        v.lm_off = d->term_q_off.p;
        d->lm_slots.upload(b.lm_slots);
        v.lm_slots = d->lm_slots.p;
        v.lm_terms = d->lm_terms.p; v.lm_obs_common = d->lm_obs_common.p;
      }
      v.mest = b.mest; v.mest_a = b.mest_a; v.n_obs = (int)b.obs_first.size() - 1; v.n_terms = b.n_terms;
      v.row_base = b.row_base; v.term_base_global = term_base;
      sviews.push_back(v);
    } else {  // KIND_MOTION: one whole-spline term, its sums split over the ranks by coefficient
      const SplineRec& s = P.splines[b.spline];
      d->Qband.upload(b.Qband);
      int b0 = P.dvs[s.first_dv].block;
      std::vector<int> slot((size_t)s.n_coeff * s.order, -1);
      if (b0 >= 0)
        for (int i = 0; i < s.n_coeff; ++i)
          for (int dd = 0; dd < s.order && i - dd >= 0; ++dd) {
            const std::vector<int>& rows = P.col_rows[b0 + i];
            size_t k = std::lower_bound(rows.begin(), rows.end(), b0 + i - dd) - rows.begin();
            slot[(size_t)i * s.order + dd] = P.Cp[P.block_col[b0 + i]] + P.col_roff[b0 + i][k];
          }
      d->slot.upload(slot);
      MotionView v;
      v.Qband = d->Qband.p; v.slot = d->slot.p; v.n_coeff = s.n_coeff; v.S = s.order; v.D = s.dim;
      v.param_off = s.param_off; v.first_block = b0; v.first_col = b0 >= 0 ? P.dvs[s.first_dv].col : -1;
      v.row = b.row_base; v.term = term_base;
      P.motionRange(s, P.rank, v.i_lo, v.i_hi);
      mviews.push_back(v);
      L.motion_blocks = std::max(L.motion_blocks, ((v.i_hi - v.i_lo) * s.order + 255) / 256);   // a lane per (coefficient, band offset)
    }
    term_base += b.n_terms;
    db.push_back(std::move(d));
  }
  // launch list
  n_jpart = 0;
  for (int l = 0; l < N_LAUNCH; ++l) {
    LaunchUnit& L = lu[l];
    L.jpart_base = n_jpart;
    if (l == LAUNCH_MOTION) {
      L.n_items = (int)mviews.size();
      L.motion_blocks = std::max(L.motion_blocks, 1);
      L.n_jpart = L.n_items * L.motion_blocks;
    } else if (l == LAUNCH_IMU || l == LAUNCH_LIDAR || l == LAUNCH_IMU_SM) {
      L.chunks.upload(chunks[l]);
      L.n_items = (int)chunks[l].size();
      L.n_jpart = L.n_items;
      StreamDims& dm = L.dims;
      // coefficient columns (36 live ones; lidar: 12, rank-one form) + (other columns + residual) padded to whole 4-column tiles
      dm.ld = (l == LAUNCH_LIDAR ? 12 : 36) + ((dm.maxG + 1 + 3) / 4) * 4;
      const int dim = l == LAUNCH_LIDAR ? 1 : 3;
      const bool cb = l != LAUNCH_LIDAR && L.compact_bias && std::getenv("MC_IMU_PLAIN_ROWS") == nullptr;
      L.compact_bias = cb;
      dm.ldc = 36 + ((L.max_gs + 1 + 3) / 4) * 4;
      dm.ldp = 12 + ((L.max_gs + 1 + 3) / 4) * 4;
      dm.nbp = (L.max_nbc + 1) & ~1;
      const bool cp = cb && l == LAUNCH_IMU;   // k_stream<ImuBatchView, true, true>: compact pose columns too
      const size_t term_doubles = cp ? (size_t)dim * dm.ldp + dm.nbp + 18 : cb ? (size_t)dim * dm.ldc + dm.nbp : (size_t)dim * dm.ld;
      size_t ints = (size_t)dm.maxT + 2 * (dm.maxNb + 1) + dm.maxTab + (size_t)dm.maxG * dm.maxNb + 4;
      L.lds_bytes = (int)((size_t)dm.maxT * term_doubles * sizeof(double) + ints * sizeof(int) + 16);
      if (L.lds_bytes > 150 * 1024) throw std::runtime_error("chunk tile exceeds the LDS budget");
      if (L.lds_bytes > 64 * 1024) {
        if (l == LAUNCH_LIDAR)
          HIP_CHECK(hipFuncSetAttribute((const void*)k_stream<LidarBatchView, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes));
        else if (l == LAUNCH_IMU_SM)
          HIP_CHECK(hipFuncSetAttribute(cb ? (const void*)k_stream<ImuSmBatchView, true, true> : (const void*)k_stream<ImuSmBatchView, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes));
        else
          HIP_CHECK(hipFuncSetAttribute(cb ? (const void*)k_stream<ImuBatchView, true, true> : (const void*)k_stream<ImuBatchView, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes));
      }
    } else {
      L.groups.upload(groups[l]);
      L.n_items = (int)groups[l].size();
      L.n_jpart = L.n_items;
      if (l == LAUNCH_REPROJ) {
        L.common.alloc(groups[l].size());
        L.chain.alloc(groups[l].size());
        L.images.upload(images);
        L.n_images = (int)images.size();
        L.n_jpart = L.n_images;
        setupF1Split(groups[l]);
        if (L.f1_split) L.n_jpart = std::max(L.n_jpart, (L.n_passes + 3) / 4);
      } else if (l == LAUNCH_STATIC) {
        L.common2.alloc(groups[l].size());
        if (jt_groups_differ) { L.jt_groups.upload(jt_groups); L.n_jt_items = (int)jt_groups.size(); }
        if (!lm_groups.empty()) { L.lm_groups.upload(lm_groups); L.n_lm_groups = (int)lm_groups.size(); }
        L.images.upload(static_runs);
        L.n_images = (int)static_runs.size();
        L.n_jpart = L.n_images;
        if (L.lds_bytes > 64 * 1024) {
          HIP_CHECK(hipFuncSetAttribute((const void*)k_static<true>, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes));
          HIP_CHECK(hipFuncSetAttribute((const void*)k_static_lm, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes));
        }
      }
    }
    n_jpart += L.n_jpart;
  }
  for (const ReprojBatchView& v : rviews) {
    if (v.cam.dist == MC_DIST_EQUI) lu[LAUNCH_REPROJ].reproj_equi = true;
    if (v.mest == MC_MEST_BLAKE_ZISSERMAN) lu[LAUNCH_REPROJ].reproj_expw = true;
  }
  lu[LAUNCH_REPROJ].rviews.upload(rviews);
  lu[LAUNCH_STATIC].sviews.upload(sviews);
  lu[LAUNCH_IMU].iviews.upload(iviews);
  lu[LAUNCH_LIDAR].lviews.upload(lviews);
  lu[LAUNCH_IMU_SM].xviews.upload(xviews);
  lu[LAUNCH_MOTION].mviews.upload(mviews);
  lu[LAUNCH_GYRO_PRIOR].gviews.upload(gviews);
  lu[LAUNCH_TRANSFORMATION].tviews.upload(tviews);
  Jpart.alloc(std::max(1, n_jpart));
  Jpart2.alloc(256);
  auto table = [&](ClearTable& T, bool written) {
    std::vector<ClearPiece> v;
    auto add = [&](int buf, long long off, long long n) {
      for (long long o = 0; o < n; o += CLEAR_PIECE) v.push_back(ClearPiece{off + o, (int)std::min<long long>(CLEAR_PIECE, n - o), buf});
    };
    // e: without shards every row is written by its term
    if (P.world > 1) {
      if (written) for (const auto& r : P.written_e) add(0, r.first, r.second);
      else add(0, 0, P.j_rows);
    }
    T.n_e = (int)v.size();
    add(1, 0, std::max(1, n_jpart));
    T.n_j = (int)v.size() - T.n_e;
    if (written) for (const auto& r : P.written_cx) add(2, r.first, r.second);
    else add(2, 0, P.nnz);
    add(3, 0, P.j_cols);
    T.n_sys = (int)v.size() - T.n_e - T.n_j;
    if (rep_stride > 0) add(4, 0, (long long)Problem::REP_COPIES * (rep_stride + P.rep_cols));
    T.n_rep = (int)v.size() - T.n_e - T.n_j - T.n_sys;
    T.pieces.upload(v);
  };
  table(clear_all, false);
  if (P.world > 1) table(clear_written, true);
}

void mc_context::launchAll(int build, int useM, double* dCx, double* dRhs, double* dJ, double* dE, hipStream_t st,
                           std::vector<hipEvent_t>* ev, double* rep_buf, bool serial) {
  DevCommon dc;
  dc.x = x.p; dc.fp_data = fp_data.p; dc.fp_off = fp_off.p; dc.col_h = col_h.p;
  dc.ofp_data = ofp_data.p; dc.ofp_off = ofp_off.p;
  dc.pat_colptr = pat_colptr.p; dc.pat_rows = pat_rows.p; dc.pat_roff = pat_roff.p; dc.pat_cp0 = pat_cp0.p;
  dc.chunk_tab = chunk_tab.p;
  dc.Cx = dCx; dc.rhs = dRhs; dc.e = dE; dc.chi2 = chi2.p; dc.Jpart = Jpart.p; dc.status = status.p;
  dc.useM = useM; dc.build = build;
  dc.phase = phase_on ? phase.p : nullptr;
  const bool use_rep = build && rep_stride > 0;
  dc.gcb = use_rep ? gcb.p : nullptr;
  double* const rep_base = rep_buf ? rep_buf : rep.p;
  dc.rep = use_rep ? rep_base : nullptr;
  dc.rep_rhs = use_rep ? rep_base + (size_t)Problem::REP_COPIES * rep_stride : nullptr;
  dc.rep_stride = rep_stride; dc.rep_cols = P.rep_cols;
  {
    // Cx, rhs and the replicas for a build; e with shards (the other ranks' rows stay zero); the cost partials always (the
    // evaluate and the build kernels of a family may use different numbers of workgroups)
    const ClearTable& T = (written_only && P.world > 1) ? clear_written : clear_all;
    const int first = dE ? 0 : T.n_e;
    const int last = T.n_e + T.n_j + (build ? T.n_sys + (use_rep ? T.n_rep : 0) : 0);
    ClearBufs cb{dE, Jpart.p, dCx, dRhs, rep_base};
    if (last > first) hipLaunchKernelGGL(k_clear, dim3(last - first), dim3(256), 0, st, T.pieces.p + first, cb);
  }
  // per-launch timing (ev) needs the launches back to back on one stream
  static const int side_of[N_LAUNCH] = {-1, -1, 0, 1, 2, 0, -1, -1};   // reprojection kernels stay on the caller's stream
  const hipStream_t caller = st;
  bool used[N_SIDE] = {false, false, false};
  const bool fork = !ev && !serial && sideStreams();
  if (fork) HIP_CHECK(hipEventRecord(ev_fork, caller));
  // a failure between fork and join must not leave the side streams running unjoined behind the caller's back
  auto joinSides = [&]() {
    for (int s = 0; s < N_SIDE; ++s)
      if (used[s]) { HIP_CHECK(hipEventRecord(ev_join[s], side[s])); HIP_CHECK(hipStreamWaitEvent(caller, ev_join[s], 0)); used[s] = false; }
  };
  try {
  for (int l = 0; l < N_LAUNCH; ++l) {
    LaunchUnit& L = lu[l];
    if (ev) HIP_CHECK(hipEventRecord((*ev)[l], caller));
    if (L.n_items == 0) continue;
    st = caller;
    if (fork && side_of[l] >= 0) {
      const int s = side_of[l];
      st = side[s];
      if (!used[s]) { HIP_CHECK(hipStreamWaitEvent(st, ev_fork, 0)); used[s] = true; }
    }
    switch (l) {
      case LAUNCH_REPROJ:
        if (build && L.f1_split) {
          // evaluation (lane = corner, 7 x 7 Grams per observation) now, assembly by owner tiles once the other families'
          // kernels have joined: its stores are plain read-modify-writes
          hipLaunchKernelGGL(k_obs_common, dim3((L.n_items + 63) / 64), dim3(64), 0, st, L.rviews.p, x.p, L.groups.p, L.n_items, L.common.p,
                             L.chain.p, fp_data.p, fp_off.p, status.p, L.lite.p);
          HIP_CHECK(hipMemsetAsync(L.Gbuf.p, 0, sizeof(double) * L.Gbuf.n, st));
          if (L.f1_owner) hipLaunchKernelGGL(k_f1_ranges, dim3((L.n_items + 255) / 256), dim3(256), 0, st, L.groups.p, L.n_items, L.f1asm);
          auto kg = L.reproj_equi ? (L.reproj_expw ? k_reproj_gram<true, true> : k_reproj_gram<true, false>)
                                  : (L.reproj_expw ? k_reproj_gram<false, true> : k_reproj_gram<false, false>);
          hipLaunchKernelGGL(kg, dim3((L.n_passes + 3) / 4), dim3(256), 0, st, L.rviews.p, dc, L.groups.p, L.passes.p, L.n_passes, L.common.p,
                             L.Gbuf.p, L.jpart_base);
          if (!L.f1_owner) {
            // expansion with atomics: may overlap the other families' kernels
            hipLaunchKernelGGL(k_reproj_expand, dim3((L.n_images + 3) / 4), dim3(256), 0, st, dc, L.groups.p, L.images.p, L.n_images, L.lite.p,
                               L.chain.p, L.Gbuf.p);
          } else if (!fork) launchF1Assemble(dc, st);   // serial launches: nothing else is in flight
          break;
        }
        hipLaunchKernelGGL(k_obs_common, dim3((L.n_items + 63) / 64), dim3(64), 0, st, L.rviews.p, x.p, L.groups.p, L.n_items, L.common.p,
                           build ? L.chain.p : nullptr);
        if (build) {
          // MC_F1_WAVE_PER_OBSERVATION=1: the first shape of the kernel (kept for comparison)
          static const bool per_obs = std::getenv("MC_F1_WAVE_PER_OBSERVATION") != nullptr;
          auto kb = per_obs ? (L.reproj_equi ? (L.reproj_expw ? k_reproj_build<true, true> : k_reproj_build<true, false>)
                                             : (L.reproj_expw ? k_reproj_build<false, true> : k_reproj_build<false, false>))
                            : (L.reproj_equi ? (L.reproj_expw ? k_reproj_build_img<true, true> : k_reproj_build_img<true, false>)
                                             : (L.reproj_expw ? k_reproj_build_img<false, true> : k_reproj_build_img<false, false>));
          hipLaunchKernelGGL(kb, dim3(L.n_images), dim3(256), 0, st, L.rviews.p, dc, L.groups.p, L.images.p, L.common.p, L.chain.p, L.jpart_base);
        }
        else {
          auto ke = L.reproj_equi ? (L.reproj_expw ? k_reproj_eval<true, true> : k_reproj_eval<true, false>)
                                  : (L.reproj_expw ? k_reproj_eval<false, true> : k_reproj_eval<false, false>);
          hipLaunchKernelGGL(ke, dim3(L.n_images), dim3(256), 0, st, L.rviews.p, dc, L.groups.p, L.images.p, L.common.p, L.jpart_base);
        }
        break;
      case LAUNCH_STATIC:
        hipLaunchKernelGGL(k_static_common, dim3((L.n_items + 63) / 64), dim3(64), 0, st, L.sviews.p, x.p, L.groups.p, L.n_items, L.common2.p);
        if (build) {
          hipLaunchKernelGGL(k_static<true>, dim3(L.n_images), dim3(256), L.lds_bytes, st, L.sviews.p, dc, L.groups.p, L.images.p, L.common2.p, L.static_ld, L.jpart_base);
          if (L.n_lm_groups) hipLaunchKernelGGL(k_static_lm, dim3(L.n_lm_groups), dim3(256), L.lds_bytes, st, L.sviews.p, dc, L.lm_groups.p, L.common2.p, L.static_ld);
        }
        else hipLaunchKernelGGL(k_static<false>, dim3(L.n_images), dim3(256), 0, st, L.sviews.p, dc, L.groups.p, L.images.p, L.common2.p, L.static_ld, L.jpart_base);
        break;
      case LAUNCH_IMU:
        if (build && L.compact_bias) hipLaunchKernelGGL((k_stream<ImuBatchView, true, true>), dim3(L.n_items), dim3(256), L.lds_bytes, st, L.iviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        else if (build) hipLaunchKernelGGL((k_stream<ImuBatchView, true>), dim3(L.n_items), dim3(256), L.lds_bytes, st, L.iviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        else hipLaunchKernelGGL((k_stream<ImuBatchView, false>), dim3(L.n_items), dim3(std::min(256, (L.dims.maxT + 63) & ~63)), 0, st, L.iviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        break;
      case LAUNCH_LIDAR:
        if (build) hipLaunchKernelGGL((k_stream<LidarBatchView, true>), dim3(L.n_items), dim3(256), L.lds_bytes, st, L.lviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        else hipLaunchKernelGGL((k_stream<LidarBatchView, false>), dim3(L.n_items), dim3(std::min(256, (L.dims.maxT + 63) & ~63)), 0, st, L.lviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        break;
      case LAUNCH_IMU_SM:
        if (build && L.compact_bias) hipLaunchKernelGGL((k_stream<ImuSmBatchView, true, true>), dim3(L.n_items), dim3(256), L.lds_bytes, st, L.xviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        else if (build) hipLaunchKernelGGL((k_stream<ImuSmBatchView, true>), dim3(L.n_items), dim3(256), L.lds_bytes, st, L.xviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        else hipLaunchKernelGGL((k_stream<ImuSmBatchView, false>), dim3(L.n_items), dim3(std::min(256, (L.dims.maxT + 63) & ~63)), 0, st, L.xviews.p, dc, L.chunks.p, L.dims, L.jpart_base);
        break;
      case LAUNCH_GYRO_PRIOR:
        if (build) hipLaunchKernelGGL((k_small<GyroPriorBatchView, true>), dim3(L.n_items), dim3(256), 0, st, L.gviews.p, dc, L.groups.p, L.jpart_base);
        else hipLaunchKernelGGL((k_small<GyroPriorBatchView, false>), dim3(L.n_items), dim3(256), 0, st, L.gviews.p, dc, L.groups.p, L.jpart_base);
        break;
      case LAUNCH_TRANSFORMATION:
        if (build) hipLaunchKernelGGL((k_small<TransformBatchView, true>), dim3(L.n_items), dim3(256), 0, st, L.tviews.p, dc, L.groups.p, L.jpart_base);
        else hipLaunchKernelGGL((k_small<TransformBatchView, false>), dim3(L.n_items), dim3(256), 0, st, L.tviews.p, dc, L.groups.p, L.jpart_base);
        break;
      case LAUNCH_MOTION:
        hipLaunchKernelGGL(k_motion, dim3(L.motion_blocks, L.n_items), dim3(256), 0, st, L.mviews.p, dc, L.jpart_base);
        hipLaunchKernelGGL(k_motion_chi2, dim3(L.n_items), dim3(64), 0, st, L.mviews.p, Jpart.p, L.jpart_base, L.motion_blocks, chi2.p);
        break;
    }
    HIP_CHECK(hipGetLastError());
  }
  } catch (...) {
    try { joinSides(); } catch (...) {}
    throw;
  }
  st = caller;
  joinSides();
  if (fork && build && lu[LAUNCH_REPROJ].f1_split && lu[LAUNCH_REPROJ].f1_owner && lu[LAUNCH_REPROJ].n_items) launchF1Assemble(dc, st);
  if (ev) HIP_CHECK(hipEventRecord((*ev)[N_LAUNCH], st));
  if (use_rep) {
    const int n = (int)P.rep_idx.size();
    hipLaunchKernelGGL(k_rep_reduce, dim3((n + P.rep_cols + 255) / 256), dim3(256), 0, st, rep_idx.p, rep_addr.p, n, dc.rep, rep_stride, dCx,
                       rep_rhs_col.p, P.rep_cols, dc.rep_rhs, dRhs);
    HIP_CHECK(hipGetLastError());
  }
  if (!build || dJ) {
    double* out = dJ ? dJ : J.p;
    int nblk = std::min(256, std::max(1, n_jpart / 1024));
    if (nblk > 1) {
      hipLaunchKernelGGL(k_sum, dim3(nblk), dim3(256), 0, st, Jpart.p, n_jpart, Jpart2.p);
      hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, Jpart2.p, nblk, out);
    } else {
      hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, Jpart.p, n_jpart, out);
    }
    HIP_CHECK(hipGetLastError());
  }
  markDone(st);
}

void mc_context::launchJt(int useM, double* dValues, hipStream_t st) {
  if (!P.jt_supported) throw std::runtime_error("This is currently unsupported");   // what BSplineMotionError::evaluateJacobians throws
  if (P.world > 1) throw std::logic_error("the J^T backend is single-GPU");
  if (!jt_start.p) { jt_start.upload(P.jt_start); jt_epc.upload(P.jt_epc); ofp_jt.upload(P.ofp_jt); }
  DevCommon dc;
  std::memset(&dc, 0, sizeof dc);
  dc.x = x.p; dc.fp_data = fp_data.p; dc.fp_off = fp_off.p; dc.col_h = col_h.p;
  dc.ofp_data = ofp_data.p; dc.ofp_off = ofp_off.p; dc.status = status.p; dc.useM = useM; dc.build = 1;
  JtView jt{jt_start.p, jt_epc.p, ofp_jt.p, dValues};
  HIP_CHECK(hipMemsetAsync(dValues, 0, sizeof(double) * (size_t)P.jt_nnz, st));
  if (lu[LAUNCH_REPROJ].n_items) {
    LaunchUnit& L = lu[LAUNCH_REPROJ];
    hipLaunchKernelGGL(k_obs_common, dim3((L.n_items + 63) / 64), dim3(64), 0, st, L.rviews.p, x.p, L.groups.p, L.n_items, L.common.p,
                       (F1Chain*)nullptr);
    hipLaunchKernelGGL(k_jt_reproj, dim3(L.n_items), dim3(64), 0, st, L.rviews.p, dc, L.groups.p, L.common.p, jt);
  }
  if (lu[LAUNCH_STATIC].n_items) {
    LaunchUnit& L = lu[LAUNCH_STATIC];
    if (L.n_jt_items) hipLaunchKernelGGL(k_jt_static, dim3(L.n_jt_items), dim3(64), 0, st, L.sviews.p, dc, L.jt_groups.p, jt);
    else hipLaunchKernelGGL(k_jt_static, dim3(L.n_items), dim3(64), 0, st, L.sviews.p, dc, L.groups.p, jt);
  }
  if (lu[LAUNCH_GYRO_PRIOR].n_items) {
    LaunchUnit& L = lu[LAUNCH_GYRO_PRIOR];
    hipLaunchKernelGGL(k_jt_small<GyroPriorBatchView>, dim3(L.n_items), dim3(64), 0, st, L.gviews.p, dc, L.groups.p, jt);
  }
  if (lu[LAUNCH_TRANSFORMATION].n_items) {
    LaunchUnit& L = lu[LAUNCH_TRANSFORMATION];
    hipLaunchKernelGGL(k_jt_small<TransformBatchView>, dim3(L.n_items), dim3(64), 0, st, L.tviews.p, dc, L.groups.p, jt);
  }
  if (lu[LAUNCH_IMU].n_items) {
    LaunchUnit& L = lu[LAUNCH_IMU];
    hipLaunchKernelGGL(k_jt_stream<ImuBatchView>, dim3(L.n_items), dim3(128), 0, st, L.iviews.p, dc, L.chunks.p, jt);
  }
  if (lu[LAUNCH_IMU_SM].n_items) {
    LaunchUnit& L = lu[LAUNCH_IMU_SM];
    hipLaunchKernelGGL(k_jt_stream<ImuSmBatchView>, dim3(L.n_items), dim3(128), 0, st, L.xviews.p, dc, L.chunks.p, jt);
  }
  if (lu[LAUNCH_LIDAR].n_items) {
    LaunchUnit& L = lu[LAUNCH_LIDAR];
    hipLaunchKernelGGL(k_jt_stream<LidarBatchView>, dim3(L.n_items), dim3(128), 0, st, L.lviews.p, dc, L.chunks.p, jt);
  }
  HIP_CHECK(hipGetLastError());
  markDone(st);
}

#define MC_TRY try {
#define MC_CATCH(ctx, rv) } catch (const std::exception& ex) { (ctx)->err = ex.what(); return rv; }
#define MC_NEED_FINAL(ctx) if (!(ctx)->P.finalized) throw std::logic_error("call mc_finalize first")

extern "C" {

mc_context* mc_create(int device) {
  try {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) throw std::runtime_error("no HIP device available: the multical_amd backend has no CPU fallback");
    std::unique_ptr<mc_context> c(new mc_context());
    if (device < 0) HIP_CHECK(hipGetDevice(&c->device));
    else { if (device >= n) throw std::runtime_error("device index out of range"); c->device = device; }
    HIP_CHECK(hipSetDevice(c->device));
    (void)scratchOfDevice(c->device);   // once per device: address of this GPU's copy of the scratch array
    return c.release();
  } catch (const std::exception& ex) { g_create_error = ex.what(); return nullptr; }
}
void mc_destroy(mc_context* ctx) { delete ctx; }
const char* mc_last_error(const mc_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mc_add_design_variable(mc_context* ctx, int kind, const double* params, int n_params, int active, double scaling) {
  MC_TRY requireOpen(ctx->P);
  if (kind == MC_DV_SPLINE_COEFF) throw std::invalid_argument("spline coefficients are added by mc_add_spline");
  return ctx->P.addDv(kind, params, n_params, active, scaling);
  MC_CATCH(ctx, -1)
}
int mc_add_spline(mc_context* ctx, int order, int dim, const double* knots, int n_knots, const double* coeffs, int n_coeff, int active) {
  MC_TRY requireOpen(ctx->P);
  return ctx->P.addSpline(order, dim, knots, n_knots, coeffs, n_coeff, active);
  MC_CATCH(ctx, -1)
}
int mc_spline_first_dv(const mc_context* ctx, int spline_id) {
  if (spline_id < 0 || spline_id >= (int)ctx->P.splines.size()) return -1;
  return ctx->P.splines[spline_id].first_dv;
}
int mc_add_camera(mc_context* ctx, int projection_model, int distortion_model, int dv_projection, int dv_distortion) {
  MC_TRY requireOpen(ctx->P);
  return ctx->P.addCamera(projection_model, distortion_model, dv_projection, dv_distortion);
  MC_CATCH(ctx, -1)
}
int mc_add_reprojection_batch(mc_context* ctx, int camera, int pose_spline, int dv_q_c_b, int dv_t_c_b, int dv_time_offset,
                              int n_obs, const double* obs_time, const int* obs_dv_q_p_w, const int* obs_dv_t_p_w,
                              const int* obs_first, const double* y, const double* p_target, const double* sqrt_inv_r,
                              double time_padding, int mest_kind, double mest_param) {
  MC_TRY describeReprojection(ctx->P, camera, pose_spline, dv_q_c_b, dv_t_c_b, dv_time_offset, n_obs, obs_time, obs_dv_q_p_w,
                              obs_dv_t_p_w, obs_first, y, p_target, sqrt_inv_r, time_padding, mest_kind, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_gyroscope_prior_batch(mc_context* ctx, int dv_q, int dv_bias, int n, const double* w_pred, const double* meas,
                                 const double* sinv, int mest, double mest_param) {
  MC_TRY describeGyroPrior(ctx->P, dv_q, dv_bias, n, w_pred, meas, sinv, mest, mest_param); return 0; MC_CATCH(ctx, -1)
}
int mc_add_transformation_batch(mc_context* ctx, int dv_q_a, int dv_t_a, int dv_q_b, int dv_t_b, int n, const double* prior_q,
                                const double* prior_t, double weight_rotation, double weight_translation, int mest,
                                double mest_param) {
  MC_TRY describeTransformation(ctx->P, dv_q_a, dv_t_a, dv_q_b, dv_t_b, n, prior_q, prior_t, weight_rotation, weight_translation,
                                mest, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_static_reprojection_batch(mc_context* ctx, int camera, int n_chain, const int* chain_dv_q, const int* chain_dv_t,
                                     int n_obs, const int* obs_dv_q_t_c, const int* obs_dv_t_t_c, const int* obs_first,
                                     const double* y, const double* p_target, const double* sqrt_inv_r, int mest_kind,
                                     double mest_param) {
  MC_TRY describeStaticReprojection(ctx->P, camera, n_chain, chain_dv_q, chain_dv_t, n_obs, obs_dv_q_t_c, obs_dv_t_t_c,
                                    obs_first, y, p_target, sqrt_inv_r, mest_kind, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_accelerometer_batch(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_gravity, int dv_q_i_b,
                               int dv_r_b, int n, const double* t, const double* meas, const double* sqrt_inv_r,
                               int mest_kind, double mest_param) {
  MC_TRY describeImu(ctx->P, true, pose_spline, bias_spline, dv_bias, dv_gravity, dv_q_i_b, dv_r_b, n, t, meas, sqrt_inv_r,
                     mest_kind, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_gyroscope_batch(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_q_i_b, int n, const double* t,
                           const double* meas, const double* sqrt_inv_r, int mest_kind, double mest_param) {
  MC_TRY describeImu(ctx->P, false, pose_spline, bias_spline, dv_bias, -1, dv_q_i_b, -1, n, t, meas, sqrt_inv_r, mest_kind,
                     mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_matrix_design_variable(mc_context* ctx, const double* A, const int* update_pattern, int active, double scaling) {
  MC_TRY requireOpen(ctx->P);
  return ctx->P.addMatrixDv(A, update_pattern, active, scaling);
  MC_CATCH(ctx, -1)
}
int mc_add_accelerometer_batch_sm(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_gravity, int dv_q_i_b,
                                  int dv_r_b, int dv_M_accel, const int* size_effect, int n, const double* t, const double* meas,
                                  const double* sqrt_inv_r, int mest_kind, double mest_param) {
  MC_TRY describeImuSm(ctx->P, true, pose_spline, bias_spline, dv_bias, dv_gravity, dv_q_i_b, dv_r_b, dv_M_accel, -1, -1, size_effect, n,
                       t, meas, sqrt_inv_r, mest_kind, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_gyroscope_batch_sm(mc_context* ctx, int pose_spline, int bias_spline, int dv_bias, int dv_q_i_b, int dv_q_gyro_i,
                              int dv_M_gyro, int dv_M_accel_gyro, int dv_gravity, int dv_r_b, int n, const double* t,
                              const double* meas, const double* sqrt_inv_r, int mest_kind, double mest_param) {
  MC_TRY describeImuSm(ctx->P, false, pose_spline, bias_spline, dv_bias, dv_gravity, dv_q_i_b, dv_r_b, dv_M_gyro, dv_q_gyro_i,
                       dv_M_accel_gyro, nullptr, n, t, meas, sqrt_inv_r, mest_kind, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_lidar_batch(mc_context* ctx, int pose_spline, int dv_time_offset, int dv_q_l_b, int dv_t_l_b, int n, const double* t,
                       const double* dir, const double* range, const double* sqrt_inv_r, const int* dv_q_p_w,
                       const int* dv_t_p_w, double time_padding, int mest_kind, double mest_param) {
  MC_TRY describeLidar(ctx->P, pose_spline, dv_time_offset, dv_q_l_b, dv_t_l_b, n, t, dir, range, sqrt_inv_r, dv_q_p_w, dv_t_p_w,
                       time_padding, mest_kind, mest_param);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_add_motion_term(mc_context* ctx, int spline, const double* W, int error_term_order) {
  MC_TRY describeMotion(ctx->P, spline, W, error_term_order);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_set_last_batch_sqrt_inv_r(mc_context* ctx, int dim, int n, const double* S) {
  MC_TRY describeSqrtInvR(ctx->P, dim, n, S);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_set_last_batch_landmarks(mc_context* ctx, int n, const int* dv_landmark) {
  MC_TRY describeLandmarks(ctx->P, n, dv_landmark);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_set_shard(mc_context* ctx, int rank, int world) {
  MC_TRY requireOpen(ctx->P);
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("need 0 <= rank < world");
  ctx->P.rank = rank; ctx->P.world = world;
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_set_clear_mode(mc_context* ctx, int written_only) {
  if (!ctx) return -1;
  ctx->written_only = written_only != 0;
  return 0;
}
int mc_shard_info(const mc_context* ctx, int* rank, int* world) {
  if (!ctx) return -1;
  if (rank) *rank = ctx->P.rank;
  if (world) *world = ctx->P.world;
  return 0;
}
int mc_finalize(mc_context* ctx) {
  MC_TRY ctx->P.finalize();
  ctx->uploadStructure();
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_dims(const mc_context* ctx, int* j_rows, int* j_cols, int* n_blocks, int* n_terms, int* nnz) {
  if (!ctx->P.finalized) return -1;
  if (j_rows) *j_rows = ctx->P.j_rows;
  if (j_cols) *j_cols = ctx->P.j_cols;
  if (n_blocks) *n_blocks = ctx->P.n_blocks;
  if (n_terms) *n_terms = ctx->P.n_terms;
  if (nnz) *nnz = ctx->P.nnz;
  return 0;
}
int mc_dv_block(const mc_context* ctx, int dv, int* block_index, int* column_base) {
  if (!ctx->P.finalized || dv < 0 || dv >= (int)ctx->P.dvs.size()) return -1;
  *block_index = ctx->P.dvs[dv].block; *column_base = ctx->P.dvs[dv].col;
  return 0;
}
int mc_get_block_bases(const mc_context* ctx, int* base) {
  if (!ctx->P.finalized || !base) return -1;
  int c = 0;
  for (int b = 0; b < ctx->P.n_blocks; ++b) { base[b] = c; c += ctx->P.block_dim[b]; }
  base[ctx->P.n_blocks] = c;
  return 0;
}
int mc_get_block_times(const mc_context* ctx, double* t) {
  if (!ctx->P.finalized || !t) return -1;
  const Problem& P = ctx->P;
  for (int b = 0; b < P.n_blocks; ++b) t[b] = std::nan("");
  for (const SplineRec& s : P.splines)
    for (int i = 0; i < s.n_coeff; ++i) {
      const int b = P.dvs[s.first_dv + i].block;
      if (b >= 0) t[b] = P.knots[s.knots_off + i + s.order];   // end of the coefficient's support
    }
  return 0;
}
int mc_get_block_splines(const mc_context* ctx, int* spline) {
  if (!ctx->P.finalized || !spline) return -1;
  const Problem& P = ctx->P;
  for (int b = 0; b < P.n_blocks; ++b) spline[b] = -1;
  for (size_t s = 0; s < P.splines.size(); ++s)
    for (int i = 0; i < P.splines[s].n_coeff; ++i) {
      const int b = P.dvs[P.splines[s].first_dv + i].block;
      if (b >= 0) spline[b] = (int)s;
    }
  return 0;
}
int mc_get_pattern(const mc_context* ctx, int* Cp, int* Ci) {
  if (!ctx->P.finalized) return -1;
  if (Cp) std::memcpy(Cp, ctx->P.Cp.data(), sizeof(int) * ctx->P.Cp.size());
  if (Ci) std::memcpy(Ci, ctx->P.Ci.data(), sizeof(int) * ctx->P.Ci.size());
  return 0;
}

int mc_evaluate_error_device(mc_context* ctx, int use_mestimator, double* d_J, double* d_e, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->launchAll(false, use_mestimator, nullptr, nullptr, d_J, d_e, (hipStream_t)stream);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_build_system_device(mc_context* ctx, int use_mestimator, double* d_Cx, double* d_rhs, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->launchAll(true, use_mestimator, d_Cx, d_rhs, nullptr, nullptr, (hipStream_t)stream);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_evaluate_error_and_build_system_device(mc_context* ctx, int use_mestimator, double* d_J, double* d_e, double* d_Cx,
                                              double* d_rhs, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  if (!d_J || !d_Cx || !d_rhs) throw std::invalid_argument("d_J, d_Cx and d_rhs are required");
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->launchAll(true, use_mestimator, d_Cx, d_rhs, d_J, d_e, (hipStream_t)stream);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_build_system_device_noalloc(mc_context* ctx, int use_mestimator, double* d_Cx, double* d_rhs, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  if (!d_Cx || !d_rhs) throw std::invalid_argument("d_Cx and d_rhs are required");
  if (ctx->rep_stride > 0 && ctx->P.rep_cols > NOALLOC_MAX_COLS)
    throw std::length_error("mc_build_system_device_noalloc: the problem has " + std::to_string(ctx->P.rep_cols) +
                            " scalar columns of global design variables; the static scratch array holds at most " +
                            std::to_string((int)NOALLOC_MAX_COLS) + " (use mc_build_system_device)");
  HIP_CHECK(hipSetDevice(ctx->device));
  // no hipMalloc / hipFree / stream or event creation from here on: the replicas live in this device's scratch array
  // (cleared by the call: it keeps the previous call's contents), every launch stays on the caller's stream
  ctx->launchAll(true, use_mestimator, d_Cx, d_rhs, nullptr, nullptr, (hipStream_t)stream, nullptr, scratchOfDevice(ctx->device), true);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_evaluate_error(mc_context* ctx, int use_mestimator, double* J_out, double* e_out) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->launchAll(false, use_mestimator, nullptr, nullptr, ctx->J.p, e_out ? ctx->e.p : nullptr, 0);
  HIP_CHECK(hipStreamSynchronize(0));
  if (J_out) HIP_CHECK(hipMemcpy(J_out, ctx->J.p, sizeof(double), hipMemcpyDeviceToHost));
  if (e_out) HIP_CHECK(hipMemcpy(e_out, ctx->e.p, sizeof(double) * ctx->P.j_rows, hipMemcpyDeviceToHost));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_build_system(mc_context* ctx, int use_mestimator, double* Cx_out, double* rhs_out) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->launchAll(true, use_mestimator, ctx->Cx.p, ctx->rhs.p, nullptr, nullptr, 0);
  HIP_CHECK(hipStreamSynchronize(0));
  if (Cx_out) HIP_CHECK(hipMemcpy(Cx_out, ctx->Cx.p, sizeof(double) * ctx->P.nnz, hipMemcpyDeviceToHost));
  if (rhs_out) HIP_CHECK(hipMemcpy(rhs_out, ctx->rhs.p, sizeof(double) * ctx->P.j_cols, hipMemcpyDeviceToHost));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_jt_nnz(const mc_context* ctx, long long* nnz) {
  if (!ctx->P.finalized) return -1;
  if (!ctx->P.jt_supported) return -2;
  *nnz = ctx->P.jt_nnz;
  return 0;
}
int mc_get_jt_pattern(mc_context* ctx, int* col_ptr, int* row_ind) {
  MC_TRY MC_NEED_FINAL(ctx);
  std::vector<int> cp, ri;
  ctx->P.jtPattern(cp, ri);
  std::memcpy(col_ptr, cp.data(), sizeof(int) * cp.size());
  std::memcpy(row_ind, ri.data(), sizeof(int) * ri.size());
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_build_jacobian_transpose_device(mc_context* ctx, int use_mestimator, double* d_values, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->launchJt(use_mestimator, d_values, (hipStream_t)stream);
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_build_jacobian_transpose(mc_context* ctx, int use_mestimator, double* values_out) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!ctx->P.jt_supported) throw std::runtime_error("This is currently unsupported");
  if (ctx->jt_values.n != (size_t)ctx->P.jt_nnz) ctx->jt_values.alloc((size_t)ctx->P.jt_nnz);
  ctx->launchJt(use_mestimator, ctx->jt_values.p, 0);
  HIP_CHECK(hipStreamSynchronize(0));
  HIP_CHECK(hipMemcpy(values_out, ctx->jt_values.p, sizeof(double) * (size_t)ctx->P.jt_nnz, hipMemcpyDeviceToHost));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_shared_count(const mc_context* ctx) { return ctx->P.finalized ? (int)ctx->P.shared_idx.size() : -1; }
int mc_get_shared_indices(const mc_context* ctx, int* out) {
  if (!ctx->P.finalized) return -1;
  std::memcpy(out, ctx->P.shared_idx.data(), sizeof(int) * ctx->P.shared_idx.size());
  return 0;
}
int mc_pack_shared(mc_context* ctx, const double* d_Cx, double* d_buf, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  int n = (int)ctx->P.shared_idx.size();
  if (n == 0) return 0;
  HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->shared_idx.p, n, d_Cx, d_buf);
  HIP_CHECK(hipGetLastError());
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_unpack_shared(mc_context* ctx, const double* d_buf, double* d_Cx, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  int n = (int)ctx->P.shared_idx.size();
  if (n == 0) return 0;
  HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_unpack, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->shared_idx.p, n, d_buf, d_Cx);
  HIP_CHECK(hipGetLastError());
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_shared_rhs_count(const mc_context* ctx) { return ctx->P.finalized ? (int)ctx->P.shared_rhs.size() : -1; }
int mc_get_shared_rhs_indices(const mc_context* ctx, int* out) {
  if (!ctx->P.finalized) return -1;
  std::memcpy(out, ctx->P.shared_rhs.data(), sizeof(int) * ctx->P.shared_rhs.size());
  return 0;
}
int mc_pack_shared_rhs(mc_context* ctx, const double* d_rhs, double* d_buf, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  int n = (int)ctx->P.shared_rhs.size();
  if (n == 0) return 0;
  HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->shared_rhs.p, n, d_rhs, d_buf);
  HIP_CHECK(hipGetLastError());
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_unpack_shared_rhs(mc_context* ctx, const double* d_buf, double* d_rhs, void* stream) {
  MC_TRY MC_NEED_FINAL(ctx);
  int n = (int)ctx->P.shared_rhs.size();
  if (n == 0) return 0;
  HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_unpack, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->shared_rhs.p, n, d_buf, d_rhs);
  HIP_CHECK(hipGetLastError());
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_num_launches(const mc_context*) { return N_LAUNCH; }
int mc_launch_info(const mc_context* ctx, int launch, int* n_batches, int* n_terms, int* n_workgroups) {
  if (!ctx->P.finalized || launch < 0 || launch >= N_LAUNCH) return -1;
  const LaunchUnit& L = ctx->lu[launch];
  if (n_batches) *n_batches = L.n_batches;
  if (n_terms) *n_terms = L.n_terms;
  if (n_workgroups) *n_workgroups = launch == LAUNCH_MOTION ? L.n_items * L.motion_blocks : ((launch == LAUNCH_REPROJ || launch == LAUNCH_STATIC) ? L.n_images : L.n_items);
  return 0;
}
int mc_launch_counts(const mc_context* ctx, int launch, double* flops_dense, double* flops_live, double* record_bytes) {
  if (!ctx->P.finalized || launch < 0 || launch >= N_LAUNCH) return -1;
  const LaunchUnit& L = ctx->lu[launch];
  if (flops_dense) *flops_dense = L.flops_dense;
  if (flops_live) *flops_live = L.flops_live;
  if (record_bytes) *record_bytes = L.rec_bytes;
  return 0;
}
int mc_build_system_timed(mc_context* ctx, int use_mestimator, int build, float* ms_per_launch) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  struct Events {   // destroyed on every path out
    std::vector<hipEvent_t> ev;
    explicit Events(size_t n) : ev(n, nullptr) { for (auto& e : ev) HIP_CHECK(hipEventCreate(&e)); }
    ~Events() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
  } evs(N_LAUNCH + 1);
  ctx->launchAll(build, use_mestimator, ctx->Cx.p, ctx->rhs.p, build ? nullptr : ctx->J.p, nullptr, 0, &evs.ev);
  HIP_CHECK(hipStreamSynchronize(0));
  for (int i = 0; i < N_LAUNCH; ++i) HIP_CHECK(hipEventElapsedTime(&ms_per_launch[i], evs.ev[i], evs.ev[i + 1]));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_phase_cycles(mc_context* ctx, int use_mestimator, int build, unsigned long long* out24) {
  MC_TRY MC_NEED_FINAL(ctx);
#ifndef MC_PHASE_TIMING
  throw std::logic_error("this build has no phase marks; compile with -DMC_PHASE_TIMING (scripts/build_timing_lib.sh)");
#endif
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!ctx->phase.p) ctx->phase.alloc(24);
  HIP_CHECK(hipMemset(ctx->phase.p, 0, 24 * sizeof(unsigned long long)));
  ctx->phase_on = true;
  try { ctx->launchAll(build, use_mestimator, ctx->Cx.p, ctx->rhs.p, build ? nullptr : ctx->J.p, nullptr, 0); }
  catch (...) { ctx->phase_on = false; throw; }
  ctx->phase_on = false;
  HIP_CHECK(hipStreamSynchronize(0));
  HIP_CHECK(hipMemcpy(out24, ctx->phase.p, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_get_term_chi2(mc_context* ctx, double* chi2_out) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpy(chi2_out, ctx->chi2.p, sizeof(double) * ctx->P.n_terms, hipMemcpyDeviceToHost));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_apply_update(mc_context* ctx, const double* dx) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->waitForLaunches();   // kernels enqueued on the caller's stream may still read x
  HIP_CHECK(hipMemcpyAsync(ctx->x_prev.p, ctx->x.p, sizeof(double) * ctx->P.x.size(), hipMemcpyDeviceToDevice, 0));
  HIP_CHECK(hipMemcpyAsync(ctx->dx.p, dx, sizeof(double) * ctx->P.j_cols, hipMemcpyHostToDevice, 0));
  int n = (int)ctx->upd.n;
  hipLaunchKernelGGL(k_update, dim3((n + 255) / 256), dim3(256), 0, 0, ctx->upd.p, n, ctx->dx.p, ctx->x.p);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(0));
  ctx->have_prev = true;
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_revert_update(mc_context* ctx) {
  MC_TRY MC_NEED_FINAL(ctx);
  if (!ctx->have_prev) throw std::logic_error("no update to revert");
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->waitForLaunches();
  HIP_CHECK(hipMemcpy(ctx->x.p, ctx->x_prev.p, sizeof(double) * ctx->P.x.size(), hipMemcpyDeviceToDevice));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_get_dv_params(mc_context* ctx, int dv, double* out) {
  MC_TRY
  if (dv < 0 || dv >= (int)ctx->P.dvs.size()) throw std::invalid_argument("design variable id out of range");
  const DvRec& d = ctx->P.dvs[dv];
  if (ctx->P.finalized) {
    HIP_CHECK(hipSetDevice(ctx->device));
    ctx->waitForLaunches();
    HIP_CHECK(hipMemcpy(out, ctx->x.p + d.off, sizeof(double) * d.nparam, hipMemcpyDeviceToHost));
  } else std::memcpy(out, &ctx->P.x[d.off], sizeof(double) * d.nparam);
  return d.nparam;
  MC_CATCH(ctx, -1)
}
int mc_lidar_find_points_on_targets(mc_context* ctx, int pose_spline, int dv_time_offset, int dv_q_l_b, int dv_t_l_b, int n_planes,
                                    const int* dv_q_p_w, const int* dv_t_p_w, const double* plane_min, const double* plane_max,
                                    double threshold, int n, const double* points, unsigned* mask_out, double* world_out) {
  MC_TRY MC_NEED_FINAL(ctx);
  Problem& P = ctx->P;
  P.checkSpline(pose_spline, 6);
  P.checkDv(dv_q_l_b, DV_ROTATION_QUATERNION); P.checkDv(dv_t_l_b, DV_EUCLIDEAN_POINT);
  if (dv_time_offset >= 0) P.checkDv(dv_time_offset, DV_SCALAR);
  if (n_planes < 0 || n_planes > 30) throw std::invalid_argument("at most 30 targets");
  if (n < 0 || !mask_out || (n > 0 && !points)) throw std::invalid_argument("bad point / mask arguments");
  AssocPlanes pl;
  std::memset(&pl, 0, sizeof pl);
  pl.n = n_planes;
  for (int k = 0; k < n_planes; ++k) {
    P.checkDv(dv_q_p_w[k], DV_ROTATION_QUATERNION); P.checkDv(dv_t_p_w[k], DV_EUCLIDEAN_POINT);
    pl.q_off[k] = P.dvs[dv_q_p_w[k]].off; pl.t_off[k] = P.dvs[dv_t_p_w[k]].off;
    for (int a = 0; a < 3; ++a) {
      pl.lo[k][a] = plane_min[3 * k + a] - (a == 2 ? threshold : 0.0);
      pl.hi[k][a] = plane_max[3 * k + a] + (a == 2 ? threshold : 0.0);
    }
  }
  if (n == 0) return 0;
  HIP_CHECK(hipSetDevice(ctx->device));
  DevBuf<double> dpts, dworld;
  DevBuf<unsigned> dmask;
  {
    std::vector<double> hp(points, points + 4 * (size_t)n);
    dpts.upload(hp);
  }
  dmask.alloc(n);
  if (world_out) dworld.alloc(3 * (size_t)n);
  hipLaunchKernelGGL(k_lidar_associate, dim3((n + 255) / 256), dim3(256), 0, 0, P.splineView(pose_spline, ctx->knots.p, ctx->basis.p),
                     ctx->x.p, dv_time_offset >= 0 ? P.dvs[dv_time_offset].off : -1, P.dvs[dv_q_l_b].off, P.dvs[dv_t_l_b].off, pl,
                     dpts.p, n, dmask.p, world_out ? dworld.p : nullptr);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpy(mask_out, dmask.p, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
  if (world_out) HIP_CHECK(hipMemcpy(world_out, dworld.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
  return 0;
  MC_CATCH(ctx, -1)
}
int mc_device_status(mc_context* ctx) {
  MC_TRY MC_NEED_FINAL(ctx);
  HIP_CHECK(hipSetDevice(ctx->device));
  ctx->waitForLaunches();   // the flag is set by kernels that may still be running on the caller's stream
  int s = 0;
  HIP_CHECK(hipMemcpy(&s, ctx->status.p, sizeof(int), hipMemcpyDeviceToHost));
  if (s) HIP_CHECK(hipMemset(ctx->status.p, 0, sizeof(int)));
  if (s & STATUS_BUFFER_EXCEEDED) ctx->err = "Spline Coefficient Buffer Exceeded. Set larger buffer margins!";
  else if (s & STATUS_TIME_OUT_OF_SPLINE) ctx->err = "The time is out of range of the spline";
  else if (s & STATUS_INTERNAL) ctx->err = "internal error: chunk terms are not ordered by spline segment";
  return s;
  MC_CATCH(ctx, -1)
}
int mc_device_info(mc_context* ctx, char* name, int name_len, int* n_cu, double* hbm_gib) {
  MC_TRY
  hipDeviceProp_t p;
  HIP_CHECK(hipGetDeviceProperties(&p, ctx->device));
  if (name && name_len > 0) { std::snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (hbm_gib) *hbm_gib = (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0);
  return 0;
  MC_CATCH(ctx, -1)
}

}  // extern "C"
