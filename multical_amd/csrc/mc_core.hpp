// mc_core.hpp -- host-side problem description and structure analysis (no HIP in this file).
//
// What the reference does lazily and per call -- Optimizer2::initialize (block indices, column bases,
// row bases; aslam_backend/src/Optimizer2.cpp:95-151), first-touch allocation of Hessian blocks inside
// SparseBlockMatrix::block(r, c, true) (sparse_block_matrix.hpp:121-143) and the symbolic half of fillCCS
// (:704-729) -- is done here once per optimize(): the term batches are cut into *groups* of terms that
// touch the same Hessian blocks, each distinct block list (*footprint*) gets a table of value offsets
// into the final CCS array, and the scalar pattern (Cp, Ci) is emitted in exactly fillCCS order.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "mc_terms.h"

namespace mc {

enum { DV_EUCLIDEAN_POINT = 0, DV_ROTATION_QUATERNION = 1, DV_SCALAR = 2, DV_EUCLIDEAN_DIRECTION = 3,
       DV_SPLINE_COEFF = 4, DV_PROJECTION = 5, DV_DISTORTION = 6, DV_MATRIX_BASIC = 7, DV_HOMOGENEOUS_POINT = 8 };
enum { KIND_REPROJ = 0, KIND_STATIC_REPROJ = 1, KIND_ACCEL = 2, KIND_GYRO = 3, KIND_LIDAR = 4, KIND_MOTION = 5,
       KIND_ACCEL_SM = 6, KIND_GYRO_SM = 7, KIND_GYRO_PRIOR = 8, KIND_TRANSFORMATION = 9, N_KINDS = 10 };
// terms over a handful of global design variables (the initialisation problems): every term of a batch writes the same blocks
inline bool kindIsGlobal(int k) { return k == KIND_GYRO_PRIOR || k == KIND_TRANSFORMATION; }
enum { GLOBAL_GROUP_TERMS = 64 };   // terms per workgroup of the small-problem kernel (64 x 6 rows x 13 columns of LDS)
inline bool kindIsImu(int k) { return k == KIND_ACCEL || k == KIND_GYRO || k == KIND_ACCEL_SM || k == KIND_GYRO_SM; }
inline bool kindIsAccel(int k) { return k == KIND_ACCEL || k == KIND_ACCEL_SM; }
inline bool kindIsStream(int k) { return k == KIND_LIDAR || kindIsImu(k); }

struct DvRec {
  int kind, dim, off, nparam, active;
  double scaling;
  int block = -1, col = -1;
  int spline = -1;
  int mask = 0;   // MatrixBasic: update pattern, bit 3 * row + col
};
struct SplineRec {
  int order, dim, n_knots, n_coeff, n_seg, first_dv, param_off, knots_off, basis_off, active;
  double t_min, t_max;
};
struct CameraRec { int model, dist, dv_proj, dv_dist; };

// A group = consecutive terms of one batch that write the same Hessian blocks.
struct GroupRec {
  int term_first, term_count;
  int fp;    // footprint id
  int aux;   // kind specific (F1: observation index)
  int batch; // index into the launch's view array (set when the launch list is built)
};

// A chunk = consecutive (time-sorted) terms of one stream batch (lidar, accelerometer, gyroscope) handled by one
// workgroup: their live pose-coefficient windows lie inside the coefficient range [pose_lo, pose_lo + pose_nb) and
// they share the list of other blocks `ofp` (extrinsics, time offset, board pose, bias-coefficient window...).
struct ChunkRec {
  int term_first, term_count;  // positions in the batch's time-sorted `order`
  int pose_lo, pose_nb;        // coefficient range laid out for the chunk (0 blocks if the pose spline is inactive)
  int ofp;                     // "other blocks" footprint id
  int batch;                   // index into the launch's view array (set when the launch list is built)
  int tab_off, tab_len;        // this chunk's slot table in Problem::chunk_tab (built for this rank's chunks)
};

struct Batch {
  int kind;
  int n_terms = 0, dim = 0, row_base = 0, term_base = 0;
  int mest = 0; double mest_a = 0;
  // shared ids
  int cam = -1, spline = -1, bias_spline = -1;
  // kind specific global dvs.  IMU kinds: 0 static bias, 1 gravity, 2 q_i_b, 3 r_b; scale-misalignment models add
  // 4 M (accel: M_accel, gyro: M_gyro), 5 q_gyro_i, 6 M_accel_gyro, 7..9 rx_i ry_i rz_i, 10..12 Ix Iy Iz
  int dv[14] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
  int size_effect = 0;
  double pad = 0;
  double w_t = 1.0, w_r = 1.0;   // ErrorTermTransformation: translation / rotation weights (diagonal invR)
  // per-term / per-observation host arrays
  std::vector<double> t, y, p, sinv, meas, range;
  std::vector<double> S;   // optional general sqrtInvR, n_terms x dim x dim row major (mc_set_last_batch_sqrt_inv_r)
  std::vector<int> obs_first, obs_dv_q, obs_dv_t, term_dv_q, term_dv_t, term_obs;
  std::vector<int> term_dv_lm;   // static reprojection: per term the HomogeneousPoint landmark, or -1 (constant target point)
  std::vector<int> chain_q, chain_t;
  // motion term
  std::vector<double> Qband;  // n_coeff x S blocks of dim x dim, Qband[(i*S + d)*D*D ...] = Q(i, i-d)
  int motion_order = 0;
  // structure (filled by finalize)
  std::vector<GroupRec> groups;       // all groups (terms numbered within the batch)
  std::vector<GroupRec> my_groups;    // the groups of this rank's time window
  std::vector<double> group_time;     // representative time per group (for sharding)
  // Static reprojection batches with landmark design variables.  `groups` holds one group per term over an active landmark
  // (footprint = the observation's blocks + the landmark: what the pattern, the J^T layout and the shard marks are built
  // from).  The evaluate / build kernel does not run one workgroup pass per such term: it takes the observation as ONE
  // group (obs_groups, footprint without the landmark) and adds a term's landmark rows x [core columns | landmark |
  // residual] products through the term's own footprint, term_fp[term] (-1: constant or inactive point).
  std::vector<GroupRec> obs_groups, my_obs_groups;
  std::vector<int> term_fp;
  // ... and the products of a landmark's rows with the camera-level columns (baselines, intrinsics), with themselves and with
  // the residual meet EVERY observation of that landmark: adding them per term would queue thousands of atomics on the same
  // entries.  They are formed in a second, landmark-major pass instead: lm_terms lists this rank's terms by landmark,
  // lm_groups cuts the lists into work items (term_first = offset into lm_terms; fp = camera-level blocks + landmark).
  std::vector<std::pair<int, int>> lm_fp;   // (landmark dv, footprint), sorted by dv
  // per term over an active landmark, where its landmark x target-pose blocks sit in the value array (the observation-major
  // pass adds them per term): [value offset of the (landmark, q_t_c) block pair, of (landmark, t_t_c)] (-1: not active) and
  // [2 h + o] for each, h = column height of the pair's column block, o = 1 when the landmark is the column block
  std::vector<int> lm_slots;
  std::vector<int> lm_terms;
  std::vector<GroupRec> lm_groups;
  enum { LM_GROUP_TERMS = 1024 };
  // stream kinds (lidar / IMU): time-sorted order, per-term frozen coefficient buffer, chunks
  std::vector<int> order, term_lo, term_nbuf;
  std::vector<ChunkRec> chunks, my_chunks;
  std::vector<double> chunk_time;
  int max_nb = 0, max_nob = 0, max_g = 0, max_terms = 0, max_tab = 0;   // over my_chunks: sizes the kernel's LDS
  // SURVEY 8(d) accounting (filled by finalize)
  double flops_dense = 0, flops_live = 0;
};

// "other blocks" footprint layout in `ofp_data` at ofp_off[f]:
//   [0] n blocks  [1] G (local columns)  [2] first buffered bias coefficient  [3] number of buffered bias coefficients
//   [4 .. 4+MC_MAX_ROLES) role -> local column (-1: inactive)
//   then n x (block index, dim, local column start, global column base)
enum { OFP_HDR = 4 + MC_MAX_ROLES };
// chunking limits: terms per chunk (rows must fit the LDS tile) and the target coefficient window
enum { CHUNK_TERMS_LIDAR = 160, CHUNK_TERMS_IMU = 40, CHUNK_TERMS_IMU_COMPACT = 50, CHUNK_TERMS_IMU_COMPACT_POSE = 64, CHUNK_WINDOW = 40, CHUNK_BIAS_WINDOW = 8 };  // terms: two workgroups per CU by LDS (the kernels sit at 2 waves/SIMD by registers anyway); measured 128->160: -6 %, 32->40: -3 %, 48: 1 workgroup per CU, +40 %

// footprint layout in the flat int array `fp_data` at fp_off[f]:
//   [0] nb   [1] C (local columns)   [2] first buffered spline coefficient, [3] number of buffered coefficients (time-offset kinds)
//   [4 .. 4+MC_MAX_ROLES) rolecol
//   then nb x (block index, local col start, block dim, global column base)
//   then nb(nb+1)/2 slot bases, pair (ia <= ib) at ib(ib+1)/2 + ia
//   then C ints: block position of each local column
//   then nb ints: col_h of each block
enum { FP_HDR = 4 + MC_MAX_ROLES };

struct Problem {
  std::vector<DvRec> dvs;
  std::vector<double> x;        // state vector (all parameters)
  std::vector<SplineRec> splines;
  std::vector<double> knots, basis;
  std::vector<CameraRec> cams;
  std::vector<Batch> batches;
  int rank = 0, world = 1;
  double shard_lo = -INFINITY, shard_hi = INFINITY;
  std::vector<double> shard_cuts;   // world + 1 window boundaries: rank r takes the work items with time in [cuts[r], cuts[r + 1])
  int shardOf(double t) const { int r = 0; while (r + 1 < world && t >= shard_cuts[r + 1]) ++r; return r; }
  // A BSplineMotionError term covers a whole spline.  Its sums are split by coefficient: the rank whose window holds the
  // middle of coefficient i's support takes band column i of Q (blocks (i - d, i)), row i of Q c and c_i^T (Q c)_i.
  // Knots are non-decreasing, so a rank's coefficients are the contiguous range [lo, hi).
  void motionRange(const SplineRec& s, int rnk, int& lo, int& hi) const {
    lo = 0; hi = s.n_coeff;
    if (world <= 1) return;
    lo = hi = -1;
    for (int i = 0; i < s.n_coeff; ++i)
      if (shardOf(knots[s.knots_off + i + s.order / 2]) == rnk) { if (lo < 0) lo = i; hi = i + 1; }
    if (lo < 0) lo = hi = 0;
  }

  // structure
  bool finalized = false;
  int n_blocks = 0, j_cols = 0, j_rows = 0, n_terms = 0, nnz = 0;
  std::vector<int> block_dim, block_col, block_dv;
  std::vector<std::vector<int>> col_rows;   // per block column: sorted row blocks (upper triangle incl. diagonal)
  std::vector<std::vector<int>> col_roff;   // scalar row offset of each of those blocks inside a scalar column
  std::vector<int> col_h;                   // total rows of the strictly-upper blocks of a block column
  std::vector<int> Cp, Ci;
  std::vector<int> fp_data, fp_off;
  std::unordered_map<std::string, int> fp_lookup;
  std::vector<int> update_dvs;              // active dvs in block order
  std::vector<int> ofp_data, ofp_off;       // other-block footprints of the chunks
  // Contended entries.  Hessian entries between two design variables that are not spline coefficients (extrinsics, time
  // offsets, target poses, gravity ...) are written by EVERY work item of a sensor stream: thousands of fp64 atomics
  // queue on the same few cache lines.  The kernels add those entries (and the matching rhs entries) into one of
  // REP_COPIES replicas of a small dense upper triangle over the "global" scalar columns instead; k_rep_reduce folds the
  // replicas into Cx / rhs at the end of buildSystem.  gcb[block] = first compact column of the block, -1 for spline
  // coefficients; disabled (all -1) when there are more than REP_MAX_COLS such columns (camera-calibration problems,
  // where every view's pose is a design variable and nothing is contended that much).
  enum { REP_COPIES = 16, REP_MAX_COLS = 512 };
  std::vector<int> gcb;                     // per block
  int rep_cols = 0;                         // compact global columns (0: replicas disabled)
  std::vector<int> rep_idx, rep_addr;       // dense triangle index -> Cx value index, for the pairs present in the pattern
  std::vector<int> rep_rhs_col;             // compact column -> scalar column of rhs
  // J^T layout of a term of a chunk (SparseCholesky backend): its design variables sorted by block index are the pose
  // buffer range, the 6 live bias coefficients and the single other blocks.  Per ofp, JT_STRIDE ints:
  //   [role]                    constant part of the role's first element index (-1: inactive)
  //   [MC_MAX_ROLES + role]     1 if the pose range sorts before the role (adds 6 * nbuf of the term)
  //   [2*MC_MAX_ROLES + role]   1 if the bias range sorts before the role (adds 18)
  //   [3*MC_MAX_ROLES]          elements per column without the pose range
  std::vector<int> ofp_jt;
  enum { JT_STRIDE = 3 * MC_MAX_ROLES + 1 };
  // per global term: first value index of its J^T columns and elements per column; total count
  std::vector<long long> jt_start;
  std::vector<int> jt_epc;
  long long jt_nnz = 0;
  bool jt_supported = true;
  std::map<std::vector<int>, int> ofp_lookup;
  std::vector<int> pat_colptr, pat_rows, pat_roff, pat_cp0;   // block pattern as CSR
  // Slot tables of this rank's chunks, one contiguous record per chunk (copied to LDS by one coalesced read):
  //   band[nb*6]   value offset of coefficient block pair (bj - d, bj) at [bj*6 + d], -1 if absent
  //   hcol[nb]     col_h of coefficient block bj
  //   arrow[nob*nb] value offset of (other block o, coefficient block b) in whichever order the pattern stores it
  //   oo[nob*nob]  value offset of other-block pair (o1, o2) when block(o1) <= block(o2), else -1
  //   oh[nob]      col_h of other block o
  //   ocolblk[G]   other block of each local other column
  //   oinfo[nob*3] (block index, local column start, global column base) of other block o
  std::vector<int> chunk_tab;
  // Multi-GPU: value indices (into Cx) of the Hessian blocks that more than one rank's terms can write -- window
  // halos of the coefficient band and the blocks between global design variables.  Only these need the all-reduce;
  // every other entry is written by exactly one rank.  Empty for world == 1.
  std::vector<int> shared_idx;
  // ... and the scalar columns of rhs that more than one rank writes (the blocks whose diagonal pair is shared): the
  // calibration parameters and the coefficients at the window boundaries.  Every other entry of rhs is complete on one rank.
  std::vector<int> shared_rhs;
  // world > 1: what this rank's launches write, as merged ascending (first, count) runs -- of Cx values, of residual rows.
  // A caller that keeps everything else zero may have the builds clear only these (mc_set_clear_mode).
  std::vector<std::pair<long long, long long>> written_cx, written_e;

  // ------------------------------------------------------------------ description
  int addDv(int kind, const double* params, int n, int active, double scaling) {
    DvRec d;
    d.kind = kind; d.active = active ? 1 : 0; d.scaling = scaling; d.off = (int)x.size();
    switch (kind) {
      case DV_EUCLIDEAN_POINT: if (n != 3) throw std::invalid_argument("EuclideanPoint needs 3 parameters"); d.dim = 3; d.nparam = 3; break;
      case DV_ROTATION_QUATERNION: if (n != 4) throw std::invalid_argument("RotationQuaternion needs 4 parameters"); d.dim = 3; d.nparam = 4; break;
      case DV_SCALAR: if (n != 1) throw std::invalid_argument("Scalar needs 1 parameter"); d.dim = 1; d.nparam = 1; break;
      case DV_EUCLIDEAN_DIRECTION: if (n != 3) throw std::invalid_argument("EuclideanDirection needs a 3-vector"); d.dim = 2; d.nparam = 10; break;
      case DV_PROJECTION: if (n < 4 || n > 6) throw std::invalid_argument("projection needs 4..6 parameters"); d.dim = n; d.nparam = n; break;
      case DV_DISTORTION: if (n != 4) throw std::invalid_argument("distortion needs 4 parameters"); d.dim = 4; d.nparam = 4; break;
      case DV_HOMOGENEOUS_POINT: if (n != 4) throw std::invalid_argument("HomogeneousPoint needs 4 parameters"); d.dim = 3; d.nparam = 4; break;
      default: throw std::invalid_argument("unknown design variable kind");
    }
    if (kind == DV_EUCLIDEAN_DIRECTION) {
      // EuclideanDirection constructor (ABE/src/EuclideanDirection.cpp:5-17): C = R_ypr(0, asin(-u0), atan2(u1,u2))^T
      double m = std::sqrt(params[0] * params[0] + params[1] * params[1] + params[2] * params[2]);
      if (!(m > 0.0)) throw std::invalid_argument("The pointing vector must have a magnitude greater than zero");
      double u0 = params[0] / m, u1 = params[1] / m, u2 = params[2] / m;
      double R[9];
      ypr0_to_r(std::asin(-u0), std::atan2(u1, u2), R);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) x.push_back(R[3 * j + i]);
      x.push_back(m);
    } else if (kind == DV_HOMOGENEOUS_POINT) {
      // HomogeneousPoint constructor (ABE/src/HomogeneousPoint.cpp:7-13): the point is kept with unit 4-norm
      const double nrm = std::sqrt(params[0] * params[0] + params[1] * params[1] + params[2] * params[2] + params[3] * params[3]);
      if (!(nrm > 0.0)) throw std::invalid_argument("HomogeneousPoint: the zero vector is not a point");
      for (int i = 0; i < 4; ++i) x.push_back(params[i] * (1.0 / nrm));
    } else {
      x.insert(x.end(), params, params + n);
    }
    dvs.push_back(d);
    return (int)dvs.size() - 1;
  }

  // MatrixBasic (ABE/src/MatrixBasic.cpp:10-47): A row-major; the entries with a non-zero pattern are the minimal
  // parameters, numbered in row-major order of the pattern (the column order of the selection matrix _B built at :17-25)
  int addMatrixDv(const double* A, const int* pattern, int active, double scaling) {
    DvRec d;
    d.kind = DV_MATRIX_BASIC; d.scaling = scaling; d.off = (int)x.size(); d.nparam = 9; d.dim = 0; d.mask = 0;
    for (int i = 0; i < 9; ++i) if (pattern[i] != 0) { d.mask |= 1 << i; ++d.dim; }
    // a matrix without free entries has minimal dimension 0 and never reaches the optimiser (the front-end keeps
    // Ix / Iy / Iz inactive, sensors_and_targets.py:1598-1606)
    d.active = (active && d.dim > 0) ? 1 : 0;
    x.insert(x.end(), A, A + 9);
    dvs.push_back(d);
    return (int)dvs.size() - 1;
  }

  // BSpline::M recursion (BSP/src/BSpline.cpp:70-149), evaluated bottom-up for one segment
  static void basisMatrix(const std::vector<double>& kn, int order, int i, double* out /* S x S row-major */) {
    std::vector<double> M(1, 1.0);
    for (int k = 2; k <= order; ++k) {
      std::vector<double> N((size_t)k * k, 0.0);
      // N = [M;0] A + [0;M] B with A,B (k-1) x k
      for (int idx = 0; idx < k - 1; ++idx) {
        int j = i - k + 2 + idx;
        double denom = kn[j + k - 1] - kn[j];
        double d0 = denom <= 0.0 ? 0.0 : (kn[i] - kn[j]) / denom;
        double d1 = denom <= 0.0 ? 0.0 : (kn[i + 1] - kn[i]) / denom;
        for (int r = 0; r < k - 1; ++r) {
          double m = M[(size_t)r * (k - 1) + idx];
          // [M;0] * A : row r
          N[(size_t)r * k + idx] += m * (1.0 - d0);
          N[(size_t)r * k + idx + 1] += m * d0;
          // [0;M] * B : row r+1
          N[(size_t)(r + 1) * k + idx] += m * (-d1);
          N[(size_t)(r + 1) * k + idx + 1] += m * d1;
        }
      }
      M.swap(N);
    }
    std::memcpy(out, M.data(), sizeof(double) * order * order);
  }

  int addSpline(int order, int dim, const double* kn, int n_knots, const double* coeffs, int n_coeff, int active) {
    if (order != S6) throw std::invalid_argument("the HIP path handles spline order 6 (Multical's splineOrder); got another order");
    if (dim != 6 && dim != 3) throw std::invalid_argument("spline dimension must be 6 (pose) or 3 (euclidean)");
    int n_seg = n_knots - 2 * order + 1;
    if (n_seg < 1 || n_seg + order - 1 != n_coeff) throw std::invalid_argument("knot / coefficient count mismatch");
    for (int i = 1; i < n_knots; ++i) if (kn[i - 1] > kn[i]) throw std::invalid_argument("The knot sequence must be nondecreasing");
    SplineRec s;
    s.order = order; s.dim = dim; s.n_knots = n_knots; s.n_coeff = n_coeff; s.n_seg = n_seg; s.active = active ? 1 : 0;
    s.first_dv = (int)dvs.size(); s.param_off = (int)x.size();
    s.knots_off = (int)knots.size(); s.basis_off = (int)basis.size();
    s.t_min = kn[order - 1]; s.t_max = kn[n_knots - order];
    knots.insert(knots.end(), kn, kn + n_knots);
    std::vector<double> kv(kn, kn + n_knots);
    basis.resize(basis.size() + (size_t)n_seg * order * order);
    for (int sgm = 0; sgm < n_seg; ++sgm) basisMatrix(kv, order, sgm + order - 1, &basis[s.basis_off + (size_t)sgm * order * order]);
    for (int i = 0; i < n_coeff; ++i) {
      DvRec d;
      d.kind = DV_SPLINE_COEFF; d.dim = dim; d.nparam = dim; d.active = s.active; d.scaling = 1.0;
      d.off = (int)x.size(); d.spline = (int)splines.size();
      x.insert(x.end(), coeffs + (size_t)i * dim, coeffs + (size_t)(i + 1) * dim);
      dvs.push_back(d);
    }
    splines.push_back(s);
    return (int)splines.size() - 1;
  }
  int addCamera(int model, int dist, int dv_proj, int dv_dist) {
    checkDv(dv_proj, DV_PROJECTION);
    int need = model == 0 ? 4 : (model == 1 ? 5 : 6);
    if (model < 0 || model > 2 || dvs[dv_proj].nparam != need) throw std::invalid_argument("projection parameter count does not match the model");
    if (dist < 0 || dist > 2) throw std::invalid_argument("unknown distortion model");
    if (model == 2 && dist != 0) throw std::invalid_argument("double sphere takes no distortion");
    if (dist != 0) checkDv(dv_dist, DV_DISTORTION);
    cams.push_back(CameraRec{model, dist, dv_proj, dist != 0 ? dv_dist : -1});
    return (int)cams.size() - 1;
  }
  void checkDv(int id, int kind) const {
    if (id < 0 || id >= (int)dvs.size()) throw std::invalid_argument("design variable id out of range");
    if (kind >= 0 && dvs[id].kind != kind) throw std::invalid_argument("design variable has the wrong kind for this role");
  }
  void checkSpline(int id, int dim) const {
    if (id < 0 || id >= (int)splines.size()) throw std::invalid_argument("spline id out of range");
    if (splines[id].dim != dim) throw std::invalid_argument("spline has the wrong dimension for this role");
  }

  // ------------------------------------------------------------------ views
  SplineView splineView(int id, const double* d_knots, const double* d_basis) const {
    SplineView v;
    std::memset(&v, 0, sizeof v);
    if (id < 0) return v;
    const SplineRec& s = splines[id];
    v.knots = d_knots + s.knots_off; v.basis = d_basis + s.basis_off;
    v.n_knots = s.n_knots; v.order = s.order; v.dim = s.dim; v.n_coeff = s.n_coeff; v.n_seg = s.n_seg;
    v.param_off = s.param_off;
    v.first_block = dvs[s.first_dv].block; v.first_col = dvs[s.first_dv].col;
    v.t_min = s.t_min; v.t_max = s.t_max;
    return v;
  }
  DvRef dvRef(int id) const {
    DvRef r{-1, -1, -1, 1.0};
    if (id < 0) return r;
    r.off = dvs[id].off; r.block = dvs[id].block; r.col = dvs[id].col; r.scale = dvs[id].scaling;
    return r;
  }
  CameraView cameraView(int id) const {
    const CameraRec& c = cams[id];
    CameraView v;
    v.model = c.model; v.dist = c.dist;
    v.proj_off = dvs[c.dv_proj].off; v.proj_block = dvs[c.dv_proj].block; v.proj_col = dvs[c.dv_proj].col;
    v.proj_dim = dvs[c.dv_proj].dim;
    v.dist_off = c.dv_dist >= 0 ? dvs[c.dv_dist].off : -1;
    v.dist_block = c.dv_dist >= 0 ? dvs[c.dv_dist].block : -1;
    v.dist_col = c.dv_dist >= 0 ? dvs[c.dv_dist].col : -1;
    return v;
  }

  // ------------------------------------------------------------------ structure
  // segment index of t on the host (BSpline::segmentIndex)
  int segmentOf(const SplineRec& s, double t) const {
    SplineView v = splineView((int)(&s - &splines[0]), knots.data(), basis.data());
    return spline_segment(v, t);
  }
  // TransformationTimeOffsetExpressionNode constructor (BSplineExpressions.cpp:406-437): coefficient range
  // [lo, hi] frozen from the initial time +- padding.
  void timeOffsetBuffer(const SplineRec& s, double t, double pad, int& lo, int& hi) const {
    if (!(t >= s.t_min && t <= s.t_max)) throw std::invalid_argument("measurement time outside the spline");
    int right = (pad + t > s.t_max) ? s.n_seg - 1 : segmentOf(s, t + pad);
    int left = (t - pad < s.t_min) ? 0 : segmentOf(s, t - pad);
    lo = left; hi = right + s.order - 1;
  }

  struct FpBuild {               // one footprint under construction
    std::vector<int> pat_blocks; // every block of the term's design-variable set (pattern, incl. structural zeros)
    std::vector<std::pair<int, int>> roles;  // (role, block index of the role's first block), active only
    std::vector<int> role_nblocks;
  };
  // Registers (or finds) a footprint: `written` = list of (role, first block, n blocks); `pattern` = all blocks.
  int internFootprint(const std::vector<int>& pattern, const std::vector<std::array<int, 3>>& written, int coefLo = 0, int nBuf = 0) {
    std::string key;
    key.reserve((pattern.size() + written.size() * 3 + 1) * 4);
    auto put = [&](int v) { key.append((const char*)&v, 4); };
    put((int)pattern.size());
    for (int b : pattern) put(b);
    for (auto& w : written) { put(w[0]); put(w[1]); put(w[2]); }
    auto it = fp_lookup.find(key);
    if (it != fp_lookup.end()) return it->second;
    int id = (int)fp_pattern.size();
    fp_lookup.emplace(std::move(key), id);
    fp_pattern.push_back(pattern);
    fp_written.push_back(written);
    fp_coef_lo.push_back(coefLo);
    fp_nbuf.push_back(nBuf);
    return id;
  }
  // Other-block footprint of a chunk.  `written` = (role, first block, n blocks), blocks of one role consecutive.
  // poseRole / biasRole: role ids of the two coefficient ranges; poseBlock0 / biasBlock0: first block of the splines (-1: inactive)
  int internOtherFootprint(const std::vector<std::array<int, 3>>& written, int biasLo, int biasNbuf, int poseRole = 0,
                           int poseBlock0 = -1, int biasRole = -1, int biasBlock0 = -1) {
    std::vector<int> key;
    key.push_back(biasLo); key.push_back(biasNbuf);
    for (auto& w : written) { key.push_back(w[0]); key.push_back(w[1]); key.push_back(w[2]); }
    auto it = ofp_lookup.find(key);
    if (it != ofp_lookup.end()) return it->second;
    int id = (int)ofp_off.size();
    ofp_lookup.emplace(std::move(key), id);
    ofp_off.push_back((int)ofp_data.size());
    std::vector<int> rolecol(MC_MAX_ROLES, -1);
    int nb = 0, G = 0;
    for (auto& w : written) { rolecol[w[0]] = G; for (int k = 0; k < w[2]; ++k) { G += block_dim[w[1] + k]; ++nb; } }
    ofp_data.push_back(nb); ofp_data.push_back(G); ofp_data.push_back(biasLo); ofp_data.push_back(biasNbuf);
    for (int r = 0; r < MC_MAX_ROLES; ++r) ofp_data.push_back(rolecol[r]);
    int c = 0;
    for (auto& w : written)
      for (int k = 0; k < w[2]; ++k) {
        int blkid = w[1] + k;
        ofp_data.push_back(blkid); ofp_data.push_back(block_dim[blkid]); ofp_data.push_back(c); ofp_data.push_back(block_col[blkid]);
        c += block_dim[blkid];
      }
    {  // J^T layout: sort the segments (pose range, live bias range, singles) by block index
      struct Seg { int block, role, dims; };
      std::vector<Seg> segs;
      if (poseBlock0 >= 0) segs.push_back(Seg{poseBlock0, poseRole, 0});
      for (auto& w : written) {
        if (w[0] == biasRole && biasBlock0 >= 0) segs.push_back(Seg{biasBlock0, biasRole, 18});   // 6 live coefficients x 3
        else segs.push_back(Seg{w[1], w[0], block_dim[w[1]] * w[2]});
      }
      std::sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b2) { return a.block < b2.block; });
      std::vector<int> jt(JT_STRIDE, 0);
      for (int r = 0; r < MC_MAX_ROLES; ++r) jt[r] = -1;
      int cst = 0, poseSeen = 0, biasSeen = 0;
      for (const Seg& sg : segs) {
        jt[sg.role] = cst; jt[MC_MAX_ROLES + sg.role] = poseSeen; jt[2 * MC_MAX_ROLES + sg.role] = 0;
        if (biasSeen && !(sg.role == biasRole && biasBlock0 >= 0)) jt[2 * MC_MAX_ROLES + sg.role] = 0;   // bias dims are part of cst
        if (poseBlock0 >= 0 && sg.role == poseRole && sg.block == poseBlock0) poseSeen = 1;
        else cst += sg.dims;
        (void)biasSeen;
      }
      jt[3 * MC_MAX_ROLES] = cst;
      ofp_jt.insert(ofp_jt.end(), jt.begin(), jt.end());
    }
    return id;
  }
  // pattern cliques of the stream kinds: (first coefficient block, count, second range first, count, other blocks...)
  std::map<std::vector<int>, int> clique_lookup;
  std::vector<std::vector<int>> cliques;
  void addClique(std::vector<int>& blocks) {
    auto it = clique_lookup.find(blocks);
    if (it != clique_lookup.end()) return;
    clique_lookup.emplace(blocks, (int)cliques.size());
    std::vector<int> sorted = blocks;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
      throw std::invalid_argument("a design variable plays two roles in one error term");
    cliques.push_back(std::move(sorted));
  }
  std::vector<int> fp_coef_lo, fp_nbuf;
  std::vector<std::vector<int>> fp_pattern;
  std::vector<std::vector<std::array<int, 3>>> fp_written;

  // IMU launches: the compact-bias row layout needs every batch of the launch without a general sqrtInvR
  bool imuLaunchCompact(int kind) const {
    if (std::getenv("MC_IMU_PLAIN_ROWS") != nullptr) return false;
    const bool sm = kind == KIND_ACCEL_SM || kind == KIND_GYRO_SM;
    for (const Batch& o : batches)
      if (kindIsImu(o.kind) && (o.kind == KIND_ACCEL_SM || o.kind == KIND_GYRO_SM) == sm && !o.S.empty()) return false;
    return true;
  }
  void finalize();
  void jtPattern(std::vector<int>& col_ptr, std::vector<int>& row_ind) const;
};

inline void Problem::finalize() {
  if (finalized) throw std::logic_error("mc_finalize called twice");
  const bool timing = std::getenv("MC_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tlast = tnow();
  auto lap = [&](const char* what) { if (timing) { double t = tnow(); std::fprintf(stderr, "[finalize] %-28s %.3f s\n", what, t - tlast); tlast = t; } };
  // ---- Optimizer2::initialize: active dvs in problem order get block index / column base
  n_blocks = 0; j_cols = 0;
  block_dim.clear(); block_col.clear(); block_dv.clear();
  for (size_t i = 0; i < dvs.size(); ++i) {
    DvRec& d = dvs[i];
    if (!d.active) { d.block = -1; d.col = -1; continue; }
    d.block = n_blocks++; d.col = j_cols; j_cols += d.dim;
    block_dim.push_back(d.dim); block_col.push_back(d.col); block_dv.push_back((int)i);
  }
  if (n_blocks == 0) throw std::runtime_error("It is illegal to run the optimizer with all marginalized design variables.");
  if (batches.empty()) throw std::runtime_error("It is illegal to run the optimizer with no error terms.");
  // ---- row bases
  j_rows = 0; n_terms = 0;
  for (Batch& b : batches) { b.row_base = j_rows; b.term_base = n_terms; j_rows += b.n_terms * b.dim; n_terms += b.n_terms; }

  auto blk = [&](int dv) { return dv >= 0 ? dvs[dv].block : -1; };
  // ---- groups + footprints per batch
  for (Batch& b : batches) {
    b.groups.clear(); b.group_time.clear(); b.obs_groups.clear(); b.term_fp.clear(); b.lm_fp.clear();
    auto pushGroup = [&](int first, int count, int fp, int aux, double time, int maxTerms) {
      for (int o = 0; o < count; o += maxTerms) {
        b.groups.push_back(GroupRec{first + o, std::min(maxTerms, count - o), fp, aux, 0});
        b.group_time.push_back(time);
      }
    };
    if (b.kind == KIND_REPROJ) {
      const SplineRec& s = splines[b.spline];
      int coef_block0 = dvs[s.first_dv].block;  // -1 if the spline is inactive
      int n_units = (int)b.obs_first.size() - 1;
      for (int u = 0; u < n_units; ++u) {
        double t0 = b.t[u] + (b.dv[2] >= 0 ? x[dvs[b.dv[2]].off] : 0.0);
        int lo, hi;
        timeOffsetBuffer(s, t0, b.pad, lo, hi);
        int dq = b.obs_dv_q[u], dt = b.obs_dv_t[u];
        std::vector<int> pattern;
        std::vector<std::array<int, 3>> written;
        if (coef_block0 >= 0) {
          // the active support moves with the time offset, so the whole frozen buffer [lo, hi] is laid out
          // and the kernel places the 6 live coefficient blocks inside it at run time
          for (int c = lo; c <= hi; ++c) pattern.push_back(coef_block0 + c);
          written.push_back({0, coef_block0 + lo, hi - lo + 1});
        }
        int gl[5] = {b.dv[0], b.dv[1], dq, dt, b.dv[2]};  // q_x_b, t_x_b, q_p_w, t_p_w, time offset
        for (int g = 0; g < 5; ++g)
          if (blk(gl[g]) >= 0) { pattern.push_back(blk(gl[g])); written.push_back({1 + g, blk(gl[g]), 1}); }
        std::sort(pattern.begin(), pattern.end());
        if (std::adjacent_find(pattern.begin(), pattern.end()) != pattern.end())
          throw std::invalid_argument("a design variable plays two roles in one error term");
        int fp = internFootprint(pattern, written, lo, coef_block0 >= 0 ? hi - lo + 1 : 0);
        pushGroup(b.obs_first[u], b.obs_first[u + 1] - b.obs_first[u], fp, u, b.t[u], 1 << 30);
      }
    } else if (kindIsStream(b.kind)) {
      const bool lidar = b.kind == KIND_LIDAR;
      const SplineRec& s = splines[b.spline];
      const int coef_block0 = dvs[s.first_dv].block;
      const int n = b.n_terms;
      b.order.resize(n);
      std::iota(b.order.begin(), b.order.end(), 0);
      if (!std::is_sorted(b.t.begin(), b.t.end()))
        std::stable_sort(b.order.begin(), b.order.end(), [&](int a, int c) { return b.t[a] < b.t[c]; });
      b.term_lo.assign(n, 0); b.term_nbuf.assign(n, 0);
      b.chunks.clear(); b.chunk_time.clear();
      b.flops_dense = b.flops_live = 0;
      const SplineRec* bs = (!lidar && b.bias_spline >= 0) ? &splines[b.bias_spline] : nullptr;
      const int bias_block0 = bs ? dvs[bs->first_dv].block : -1;
      // IMU rows without a general sqrtInvR are stored with compact bias columns (152 instead of 216 doubles per term on the
      // calibration problems; k_stream<.., CB>): 50 terms fit where 40 plain ones do (measured 44: 0.42, 48-54: 0.39, 56: 0.59 ms)
      // The row layout is chosen per LAUNCH (uploadStructure: all calibrated-model batches share one k_stream launch, all
      // scale-misalignment batches another): one batch with a general sqrtInvR, or MC_IMU_PLAIN_ROWS, turns the compact
      // layout off for every batch of that launch, so the chunk size follows the launch, not the batch
      // calibrated model: compact pose columns as well (98 doubles per term; k_stream's CP)
      const bool smKind = b.kind == KIND_ACCEL_SM || b.kind == KIND_GYRO_SM;
      int maxTerms = lidar ? (int)CHUNK_TERMS_LIDAR : imuLaunchCompact(b.kind) ? (smKind ? (int)CHUNK_TERMS_IMU_COMPACT : (int)CHUNK_TERMS_IMU_COMPACT_POSE) : (int)CHUNK_TERMS_IMU;
      if (const char* ov = std::getenv(lidar ? "MC_CHUNK_TERMS_LIDAR" : "MC_CHUNK_TERMS_IMU")) maxTerms = std::max(1, std::min(256, atoi(ov)));   // tuning aid
      // batch-level other roles
      std::vector<std::pair<int, int>> fixedRoles;  // (role, dv)
      if (lidar) fixedRoles = {{F5_ROLE_QLB, b.dv[0]}, {F5_ROLE_TLB, b.dv[1]}, {F5_ROLE_TOFF, b.dv[2]}};
      else {
        const bool sm = b.kind == KIND_ACCEL_SM || b.kind == KIND_GYRO_SM;
        if (!bs) fixedRoles.push_back({IMU_ROLE_BIAS, b.dv[0]});
        if (kindIsAccel(b.kind) || sm) fixedRoles.push_back({IMU_ROLE_G, b.dv[1]});
        fixedRoles.push_back({IMU_ROLE_QIB, b.dv[2]});
        if (kindIsAccel(b.kind) || sm) fixedRoles.push_back({IMU_ROLE_RB, b.dv[3]});
        if (sm) fixedRoles.push_back({IMU_ROLE_M, b.dv[4]});
        if (b.kind == KIND_GYRO_SM) { fixedRoles.push_back({IMU_ROLE_QGI, b.dv[5]}); fixedRoles.push_back({IMU_ROLE_MA, b.dv[6]}); }
        if (b.kind == KIND_ACCEL_SM && b.size_effect)
          for (int j = 0; j < 3; ++j) fixedRoles.push_back({IMU_ROLE_RX + j, b.dv[7 + j]});
      }
      double fixed_s1 = 0, fixed_s2 = 0;
      for (auto& fr : fixedRoles) if (blk(fr.second) >= 0) { double d = dvs[fr.second].dim; fixed_s1 += d; fixed_s2 += d * d; }
      const double fjac = lidar ? 3000.0 : (b.kind == KIND_ACCEL ? 6000.0 : (b.kind == KIND_GYRO ? 3000.0 : 9000.0));
      int c_first = 0, c_count = 0, c_lo = 0, c_hi = -1, c_blo = 0, c_bhi = -1, c_q = -1, c_t = -1;
      double c_time = 0;
      auto closeChunk = [&]() {
        if (c_count == 0) return;
        std::vector<std::array<int, 3>> written;
        int biasLo = 0, biasN = 0;
        if (bs && bias_block0 >= 0) { biasLo = c_blo; biasN = c_bhi - c_blo + 1; written.push_back({IMU_ROLE_BIAS, bias_block0 + c_blo, biasN}); }
        for (auto& fr : fixedRoles) if (blk(fr.second) >= 0) written.push_back({fr.first, blk(fr.second), 1});
        if (lidar) {
          if (blk(c_q) >= 0) written.push_back({F5_ROLE_QPW, blk(c_q), 1});
          if (blk(c_t) >= 0) written.push_back({F5_ROLE_TPW, blk(c_t), 1});
        }
        int ofp = internOtherFootprint(written, biasLo, biasN, lidar ? (int)F5_ROLE_COEF : (int)IMU_ROLE_POSE, coef_block0,
                                       lidar ? -1 : (int)IMU_ROLE_BIAS, (bs && bias_block0 >= 0) ? bias_block0 : -1);
        bool poseActive = coef_block0 >= 0;
        b.chunks.push_back(ChunkRec{c_first, c_count, poseActive ? c_lo : 0, poseActive ? c_hi - c_lo + 1 : 0, ofp, 0, 0, 0});
        b.chunk_time.push_back(c_time);
        c_count = 0;
      };
      int last_key[6] = {-2, -2, -2, -2, -2, -2};
      for (int pos = 0; pos < n; ++pos) {
        const int u = b.order[pos];
        int lo, hi;
        if (lidar) {
          double t0 = b.t[u] + (b.dv[2] >= 0 ? x[dvs[b.dv[2]].off] : 0.0);
          timeOffsetBuffer(s, t0, b.pad, lo, hi);
        } else {
          if (!(b.t[u] >= s.t_min && b.t[u] <= s.t_max)) throw std::invalid_argument("measurement time outside the pose spline");
          lo = segmentOf(s, b.t[u]); hi = lo + s.order - 1;
        }
        b.term_lo[u] = lo; b.term_nbuf[u] = hi - lo + 1;
        int blo = 0, bhi = -1;
        if (bs) {
          if (!(b.t[u] >= bs->t_min && b.t[u] <= bs->t_max)) throw std::invalid_argument("measurement time outside the bias spline");
          blo = segmentOf(*bs, b.t[u]); bhi = blo + bs->order - 1;
        }
        const int q = lidar ? b.term_dv_q[u] : -1, tt = lidar ? b.term_dv_t[u] : -1;
        // pattern clique of this term (every block of its design-variable set, padded buffer included)
        int key[6] = {lo, hi, blo, bhi, q, tt};
        if (std::memcmp(key, last_key, sizeof key) != 0) {
          std::memcpy(last_key, key, sizeof key);
          std::vector<int> clique;
          if (coef_block0 >= 0) for (int c = lo; c <= hi; ++c) clique.push_back(coef_block0 + c);
          if (bs && bias_block0 >= 0) for (int c = blo; c <= bhi; ++c) clique.push_back(bias_block0 + c);
          for (auto& fr : fixedRoles) if (blk(fr.second) >= 0) clique.push_back(blk(fr.second));
          if (lidar) { if (blk(q) >= 0) clique.push_back(blk(q)); if (blk(tt) >= 0) clique.push_back(blk(tt)); }
          addClique(clique);
        }
        {  // SURVEY 8(d) flops of this term: dense = whole clique, live = the 6 live coefficient blocks
          double s1 = fixed_s1, s2 = fixed_s2, l1, l2;
          if (lidar) { if (blk(q) >= 0) { s1 += 3; s2 += 9; } if (blk(tt) >= 0) { s1 += 3; s2 += 9; } }
          if (bs && bias_block0 >= 0) { s1 += 3.0 * (bhi - blo + 1); s2 += 9.0 * (bhi - blo + 1); }
          l1 = s1; l2 = s2;
          if (coef_block0 >= 0) { s1 += 6.0 * (hi - lo + 1); s2 += 36.0 * (hi - lo + 1); l1 += 36.0; l2 += 216.0; }
          double D = b.dim;
          b.flops_dense += 2.0 * D * 0.5 * (s1 * s1 + s2) + 2.0 * D * s1 + fjac;
          b.flops_live += 2.0 * D * 0.5 * (l1 * l1 + l2) + 2.0 * D * l1 + fjac;
        }
        if (c_count > 0) {
          int nlo = std::min(c_lo, lo), nhi = std::max(c_hi, hi), nblo = std::min(c_blo, blo), nbhi = std::max(c_bhi, bhi);
          bool fits = c_count < maxTerms && (nhi - nlo + 1 <= std::max((int)CHUNK_WINDOW, hi - lo + 1)) &&
                      (!bs || nbhi - nblo + 1 <= (int)CHUNK_BIAS_WINDOW) && q == c_q && tt == c_t;
          if (fits) { c_lo = nlo; c_hi = nhi; c_blo = nblo; c_bhi = nbhi; }
          else closeChunk();
        }
        if (c_count == 0) { c_first = pos; c_lo = lo; c_hi = hi; c_blo = blo; c_bhi = bhi; c_q = q; c_t = tt; c_time = b.t[u]; }
        ++c_count;
      }
      closeChunk();
    } else if (kindIsGlobal(b.kind)) {
      // roles = positions in b.dv (GP_ROLE_* / TE_ROLE_*); one footprint for the whole batch
      std::vector<int> pattern;
      std::vector<std::array<int, 3>> written;
      const int nrole = b.kind == KIND_GYRO_PRIOR ? 2 : 4;
      for (int r = 0; r < nrole; ++r)
        if (blk(b.dv[r]) >= 0) { pattern.push_back(blk(b.dv[r])); written.push_back({r, blk(b.dv[r]), 1}); }
      std::sort(pattern.begin(), pattern.end());
      if (std::adjacent_find(pattern.begin(), pattern.end()) != pattern.end())
        throw std::invalid_argument("a design variable plays two roles in one error term");
      int fp = internFootprint(pattern, written);
      pushGroup(0, b.n_terms, fp, 0, 0.0, (int)GLOBAL_GROUP_TERMS);
      // terms in order carry the time 'term index' for the shard cut (no time axis in these problems)
      for (size_t g = 0; g < b.group_time.size(); ++g) b.group_time[g] = (double)b.groups[g].term_first;
    } else if (b.kind == KIND_STATIC_REPROJ) {
      int n_obs = (int)b.obs_first.size() - 1;
      const CameraRec& c = cams[b.cam];
      if (!b.term_dv_lm.empty()) {
        // footprint of the landmark-major pass: the camera-level blocks of the batch + the landmark
        std::vector<int> seen;
        for (int dv : b.term_dv_lm) if (blk(dv) >= 0) seen.push_back(dv);
        std::sort(seen.begin(), seen.end());
        seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
        for (int dv : seen) {
          std::vector<int> pattern;
          std::vector<std::array<int, 3>> written;
          auto add = [&](int role, int d2) { if (blk(d2) >= 0) { pattern.push_back(blk(d2)); written.push_back({role, blk(d2), 1}); } };
          for (size_t j = 0; j < b.chain_q.size(); ++j) { add(F2_ROLE_CHAIN + 2 * (int)j, b.chain_q[j]); add(F2_ROLE_CHAIN + 2 * (int)j + 1, b.chain_t[j]); }
          add(F2_ROLE_PROJ, c.dv_proj);
          if (c.dv_dist >= 0) add(F2_ROLE_DIST, c.dv_dist);
          add(F2_ROLE_LM, dv);
          std::sort(pattern.begin(), pattern.end());
          if (std::adjacent_find(pattern.begin(), pattern.end()) != pattern.end())
            throw std::invalid_argument("a design variable plays two roles in one error term");
          b.lm_fp.push_back({dv, internFootprint(pattern, written)});
        }
      }
      for (int o = 0; o < n_obs; ++o) {
        std::vector<int> pattern;
        std::vector<std::array<int, 3>> written;
        auto add = [&](int role, int dv) { if (blk(dv) >= 0) { pattern.push_back(blk(dv)); written.push_back({role, blk(dv), 1}); } };
        add(F2_ROLE_QTC, b.obs_dv_q[o]); add(F2_ROLE_TTC, b.obs_dv_t[o]);
        for (size_t j = 0; j < b.chain_q.size(); ++j) { add(F2_ROLE_CHAIN + 2 * (int)j, b.chain_q[j]); add(F2_ROLE_CHAIN + 2 * (int)j + 1, b.chain_t[j]); }
        add(F2_ROLE_PROJ, c.dv_proj);
        if (c.dv_dist >= 0) add(F2_ROLE_DIST, c.dv_dist);
        std::sort(pattern.begin(), pattern.end());
        if (std::adjacent_find(pattern.begin(), pattern.end()) != pattern.end())
          throw std::invalid_argument("a design variable plays two roles in one error term");
        int fp = internFootprint(pattern, written);
        if (b.term_dv_lm.empty()) {
          pushGroup(b.obs_first[o], b.obs_first[o + 1] - b.obs_first[o], fp, o, (double)o, 1 << 30);
          continue;
        }
        // landmarks that are design variables: a term whose landmark is active has a Jacobian block of its own, so it
        // forms a group of its own (footprint = the observation's blocks + the landmark); runs of terms over constant
        // or inactive points stay together
        if (b.term_fp.empty()) b.term_fp.assign(b.n_terms, -1);
        b.obs_groups.push_back(GroupRec{b.obs_first[o], b.obs_first[o + 1] - b.obs_first[o], fp, o, 0});
        int run0 = b.obs_first[o];
        for (int k = b.obs_first[o]; k < b.obs_first[o + 1]; ++k) {
          const int lb = b.term_dv_lm[k] >= 0 ? blk(b.term_dv_lm[k]) : -1;
          if (lb < 0) continue;
          if (k > run0) pushGroup(run0, k - run0, fp, o, (double)o, 1 << 30);
          std::vector<int> pat2 = pattern;
          std::vector<std::array<int, 3>> wr2 = written;
          pat2.push_back(lb); wr2.push_back({F2_ROLE_LM, lb, 1});
          std::sort(pat2.begin(), pat2.end());
          if (std::adjacent_find(pat2.begin(), pat2.end()) != pat2.end())
            throw std::invalid_argument("a design variable plays two roles in one error term");
          b.term_fp[k] = internFootprint(pat2, wr2);
          pushGroup(k, 1, b.term_fp[k], o, (double)o, 1 << 30);
          run0 = k + 1;
        }
        if (b.obs_first[o + 1] > run0) pushGroup(run0, b.obs_first[o + 1] - run0, fp, o, (double)o, 1 << 30);
      }
    }
  }
  lap("groups + footprints");
  // ---- block pattern: union of all footprints' block pairs + every diagonal block
  // (BlockCholeskyLinearSystemSolver::solveSystem touches block(i,i,true) for all i before fillCCS runs,
  //  BlockCholeskyLinearSystemSolver.cpp:80-81)
  col_rows.assign(n_blocks, std::vector<int>());
  for (int i = 0; i < n_blocks; ++i) col_rows[i].push_back(i);
  {
    // Every pattern is a sorted block list: a few long runs of consecutive blocks (the frozen coefficient buffers) plus
    // single blocks (extrinsics, time offsets, board poses ...).  There are millions of distinct ones on a long sequence, so
    // their pair sets are accumulated by structure instead of pair by pair:
    //   run [a, b] x itself   column c of the run gets rows a .. c: the union over all runs through c is [min a, c]  -> reach[c]
    //   single s x run [a, b] one interval per occurrence, merged per single block afterwards                         -> touch[s]
    //   single x single       pushed directly (at most a handful per pattern)
    std::vector<int> reach(n_blocks);
    std::iota(reach.begin(), reach.end(), 0);
    std::vector<std::vector<std::pair<int, int>>> touch(n_blocks);
    std::vector<std::pair<int, int>> runs;
    std::vector<int> singles;
    auto addPattern = [&](const std::vector<int>& pat) {
      runs.clear(); singles.clear();
      for (size_t i = 0; i < pat.size();) {
        size_t j = i;
        while (j + 1 < pat.size() && pat[j + 1] == pat[j] + 1) ++j;
        if (j - i + 1 >= 3) runs.push_back({pat[i], pat[j]});
        else for (size_t k = i; k <= j; ++k) singles.push_back(pat[k]);
        i = j + 1;
      }
      for (auto& r : runs) for (int c = r.first; c <= r.second; ++c) reach[c] = std::min(reach[c], r.first);
      for (size_t a = 0; a < runs.size(); ++a)
        for (size_t b2 = a + 1; b2 < runs.size(); ++b2)          // two runs in one pattern (pose and bias windows)
          for (int c = runs[b2].first; c <= runs[b2].second; ++c)
            for (int r = runs[a].first; r <= runs[a].second; ++r) col_rows[c].push_back(r);
      for (int sgl : singles) for (auto& r : runs) touch[sgl].push_back(r);
      for (size_t j = 0; j < singles.size(); ++j)
        for (size_t i = 0; i < j; ++i) col_rows[singles[j]].push_back(singles[i]);
    };
    for (const std::vector<int>& pat : fp_pattern) addPattern(pat);
    for (const std::vector<int>& pat : cliques) addPattern(pat);
    for (int c = 0; c < n_blocks; ++c) for (int r = reach[c]; r < c; ++r) col_rows[c].push_back(r);
    for (int sgl = 0; sgl < n_blocks; ++sgl) {
      std::vector<std::pair<int, int>>& iv = touch[sgl];
      if (iv.empty()) continue;
      std::sort(iv.begin(), iv.end());
      int lo = iv[0].first, hi = iv[0].second;
      auto flush = [&]() { for (int k = lo; k <= hi; ++k) { if (k < sgl) col_rows[sgl].push_back(k); else if (k > sgl) col_rows[k].push_back(sgl); } };
      for (size_t i = 1; i < iv.size(); ++i) {
        if (iv[i].first <= hi + 1) hi = std::max(hi, iv[i].second);
        else { flush(); lo = iv[i].first; hi = iv[i].second; }
      }
      flush();
    }
  }
  // motion terms: band blocks between active coefficients (BSplineMotionError.hpp:122-156)
  for (const Batch& b : batches) {
    if (b.kind != KIND_MOTION) continue;
    const SplineRec& s = splines[b.spline];
    int b0 = dvs[s.first_dv].block;
    if (b0 < 0) continue;
    for (int i = 0; i < s.n_coeff; ++i)
      for (int d = 1; d < s.order && i - d >= 0; ++d) col_rows[b0 + i].push_back(b0 + i - d);
  }
  col_roff.assign(n_blocks, std::vector<int>());
  col_h.assign(n_blocks, 0);
  Cp.assign(j_cols + 1, 0);
  size_t total = 0;
  for (int c = 0; c < n_blocks; ++c) {
    std::vector<int>& rows = col_rows[c];
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    col_roff[c].resize(rows.size());
    int off = 0;
    for (size_t k = 0; k < rows.size(); ++k) { col_roff[c][k] = off; off += block_dim[rows[k]]; }
    col_h[c] = off - block_dim[c];  // the diagonal block is the last one
    for (int sc = 0; sc < block_dim[c]; ++sc) {
      Cp[block_col[c] + sc] = (int)total;
      total += (size_t)col_h[c] + sc + 1;
    }
    if (total > (size_t)0x7fffffff) throw std::overflow_error("Hessian pattern exceeds 2^31 scalar entries");
  }
  Cp[j_cols] = (int)total;
  nnz = (int)total;
  Ci.resize(nnz);
  for (int c = 0; c < n_blocks; ++c)
    for (int sc = 0; sc < block_dim[c]; ++sc) {
      int* ci = &Ci[Cp[block_col[c] + sc]];
      for (size_t k = 0; k < col_rows[c].size(); ++k) {
        int rb = col_rows[c][k], n = (rb == c) ? sc + 1 : block_dim[rb];
        for (int r = 0; r < n; ++r) *ci++ = block_col[rb] + r;
      }
    }
  lap("block pattern + Cp/Ci");
  // ---- replicas of the contended (global x global) entries
  gcb.assign(n_blocks, -1);
  rep_cols = 0; rep_idx.clear(); rep_addr.clear(); rep_rhs_col.clear();
  {
    int ng = 0;
    for (int c = 0; c < n_blocks; ++c) if (dvs[block_dv[c]].kind != DV_SPLINE_COEFF) ng += block_dim[c];
    if (ng > 0 && ng <= REP_MAX_COLS) {
      for (int c = 0; c < n_blocks; ++c) {
        if (dvs[block_dv[c]].kind == DV_SPLINE_COEFF) continue;
        gcb[c] = rep_cols;
        for (int sc = 0; sc < block_dim[c]; ++sc) rep_rhs_col.push_back(block_col[c] + sc);
        rep_cols += block_dim[c];
      }
      for (int c = 0; c < n_blocks; ++c) {
        if (gcb[c] < 0) continue;
        for (size_t k = 0; k < col_rows[c].size(); ++k) {
          const int rb = col_rows[c][k];
          if (gcb[rb] < 0) continue;
          for (int sc = 0; sc < block_dim[c]; ++sc) {
            const int n = (rb == c) ? sc + 1 : block_dim[rb];
            const int gj = gcb[c] + sc;
            for (int r = 0; r < n; ++r) {
              rep_idx.push_back(gj * (gj + 1) / 2 + gcb[rb] + r);
              rep_addr.push_back(Cp[block_col[c] + sc] + col_roff[c][k] + r);
            }
          }
        }
      }
    }
  }
  // ---- block pattern as CSR (device-side slot lookups of the chunk kernels)
  pat_colptr.assign(n_blocks + 1, 0); pat_rows.clear(); pat_roff.clear(); pat_cp0.assign(n_blocks, 0);
  for (int c = 0; c < n_blocks; ++c) {
    pat_cp0[c] = Cp[block_col[c]];
    pat_rows.insert(pat_rows.end(), col_rows[c].begin(), col_rows[c].end());
    pat_roff.insert(pat_roff.end(), col_roff[c].begin(), col_roff[c].end());
    pat_colptr[c + 1] = (int)pat_rows.size();
  }
  lap("replicas + pattern CSR");
  // ---- shard: contiguous time windows of equal estimated cost over all groups
  {
    struct GT { double t; double cost; };
    std::vector<GT> all;
    for (const Batch& b : batches) {
      double per = b.kind == KIND_REPROJ ? 1.0 : (b.kind == KIND_LIDAR ? 1.5 : (b.kind == KIND_STATIC_REPROJ ? 1.0 : 4.0));
      for (size_t g = 0; g < b.groups.size(); ++g) all.push_back(GT{b.group_time[g], 8.0 + per * b.groups[g].term_count});
      for (size_t g = 0; g < b.chunks.size(); ++g) all.push_back(GT{b.chunk_time[g], 8.0 + per * b.chunks[g].term_count});
    }
    std::vector<double> cuts(world + 1, 0.0);
    cuts[0] = -INFINITY; cuts[world] = INFINITY;
    if (world > 1 && !all.empty()) {
      std::sort(all.begin(), all.end(), [](const GT& a, const GT& b2) { return a.t < b2.t; });
      double tot = 0;
      for (auto& g : all) tot += g.cost;
      double acc = 0;
      int r = 1;
      for (size_t i = 0; i < all.size() && r < world; ++i) {
        acc += all[i].cost;
        while (r < world && acc >= tot * r / world) { cuts[r] = (i + 1 < all.size()) ? 0.5 * (all[i].t + all[i + 1].t) : INFINITY; ++r; }
      }
      for (; r < world; ++r) cuts[r] = INFINITY;
    }
    shard_lo = cuts[rank]; shard_hi = cuts[rank + 1];
    shard_cuts = cuts;
    for (Batch& b : batches) {
      // keep the groups whose time falls into [lo, hi); group order inside a batch is preserved
      std::vector<GroupRec> mine;
      for (size_t g = 0; g < b.groups.size(); ++g)
        if (b.group_time[g] >= shard_lo && b.group_time[g] < shard_hi) mine.push_back(b.groups[g]);
      b.my_groups.swap(mine);
      b.my_obs_groups.clear();
      for (const GroupRec& g : b.obs_groups)   // time of a static observation = its index (as for its groups above)
        if ((double)g.aux >= shard_lo && (double)g.aux < shard_hi) b.my_obs_groups.push_back(g);
      b.lm_terms.clear(); b.lm_groups.clear();
      if (!b.lm_fp.empty()) {
        std::vector<std::vector<int>> by_lm(b.lm_fp.size());
        for (const GroupRec& g : b.my_obs_groups)
          for (int k = g.term_first; k < g.term_first + g.term_count; ++k) {
            if (b.term_fp[k] < 0) continue;
            auto it = std::lower_bound(b.lm_fp.begin(), b.lm_fp.end(), std::make_pair(b.term_dv_lm[k], -1));
            by_lm[it - b.lm_fp.begin()].push_back(k);
          }
        for (size_t l = 0; l < by_lm.size(); ++l)
          for (size_t o = 0; o < by_lm[l].size(); o += Batch::LM_GROUP_TERMS) {
            const int n = (int)std::min<size_t>(Batch::LM_GROUP_TERMS, by_lm[l].size() - o);
            b.lm_groups.push_back(GroupRec{(int)b.lm_terms.size(), n, b.lm_fp[l].second, (int)l, 0});
            b.lm_terms.insert(b.lm_terms.end(), by_lm[l].begin() + o, by_lm[l].begin() + o + n);
          }
      }
      b.my_chunks.clear();
      b.max_nb = b.max_nob = b.max_g = b.max_terms = b.max_tab = 0;
      const int pose_block0 = (b.spline >= 0 && !b.chunks.empty()) ? dvs[splines[b.spline].first_dv].block : -1;
      for (size_t g = 0; g < b.chunks.size(); ++g)
        if (b.chunk_time[g] >= shard_lo && b.chunk_time[g] < shard_hi) {
          ChunkRec c = b.chunks[g];
          {
            const int* of = &ofp_data[ofp_off[c.ofp]];
            const int nob = of[0], G = of[1], nb = c.pose_nb, pb = pose_block0 + c.pose_lo;
            const int* ob = of + OFP_HDR;
            auto lookup = [&](int rb, int cb) {
              const std::vector<int>& rows = col_rows[cb];
              auto it = std::lower_bound(rows.begin(), rows.end(), rb);
              return (it != rows.end() && *it == rb) ? Cp[block_col[cb]] + col_roff[cb][it - rows.begin()] : -1;
            };
            c.tab_off = (int)chunk_tab.size();
            for (int bj = 0; bj < nb; ++bj)
              for (int d = 0; d < 6; ++d) chunk_tab.push_back(bj - d >= 0 ? lookup(pb + bj - d, pb + bj) : -1);
            for (int bj = 0; bj < nb; ++bj) chunk_tab.push_back(col_h[pb + bj]);
            for (int o = 0; o < nob; ++o)
              for (int bj = 0; bj < nb; ++bj) {
                int obk = ob[4 * o], pk = pb + bj;
                chunk_tab.push_back(obk < pk ? lookup(obk, pk) : lookup(pk, obk));
              }
            for (int o1 = 0; o1 < nob; ++o1)
              for (int o2 = 0; o2 < nob; ++o2) chunk_tab.push_back(ob[4 * o1] <= ob[4 * o2] ? lookup(ob[4 * o1], ob[4 * o2]) : -1);
            for (int o = 0; o < nob; ++o) chunk_tab.push_back(col_h[ob[4 * o]]);
            for (int o = 0; o < nob; ++o) for (int cc = 0; cc < ob[4 * o + 1]; ++cc) chunk_tab.push_back(o);
            for (int o = 0; o < nob; ++o) { chunk_tab.push_back(ob[4 * o]); chunk_tab.push_back(ob[4 * o + 2]); chunk_tab.push_back(ob[4 * o + 3]); }
            (void)G;
            c.tab_len = (int)chunk_tab.size() - c.tab_off;
            b.max_tab = std::max(b.max_tab, c.tab_len);
          }
          b.my_chunks.push_back(c);
          b.max_nb = std::max(b.max_nb, c.pose_nb);
          b.max_nob = std::max(b.max_nob, ofp_data[ofp_off[c.ofp]]);
          b.max_g = std::max(b.max_g, ofp_data[ofp_off[c.ofp] + 1]);
          b.max_terms = std::max(b.max_terms, c.term_count);
        }
    }
  }
  lap("shard + chunk tables");
  // ---- footprint tables (after the shard cut: see fp_used)
  auto slotBase = [&](int rb, int cb) {
    const std::vector<int>& rows = col_rows[cb];
    size_t k = std::lower_bound(rows.begin(), rows.end(), rb) - rows.begin();
    return Cp[block_col[cb]] + col_roff[cb][k];
  };
  for (Batch& b : batches) {
    b.lm_slots.clear();
    if (b.kind != KIND_STATIC_REPROJ || b.term_fp.empty()) continue;
    b.lm_slots.assign((size_t)4 * b.n_terms, -1);
    for (const GroupRec& g : b.my_obs_groups)
      for (int k = g.term_first; k < g.term_first + g.term_count; ++k) {
        if (b.term_fp[k] < 0) continue;
        const int lb = blk(b.term_dv_lm[k]);
        const int pose[2] = {blk(b.obs_dv_q[g.aux]), blk(b.obs_dv_t[g.aux])};
        for (int i = 0; i < 2; ++i) {
          if (pose[i] < 0) continue;
          const int rb = std::min(lb, pose[i]), cb = std::max(lb, pose[i]);
          b.lm_slots[4 * (size_t)k + i] = slotBase(rb, cb);
          b.lm_slots[4 * (size_t)k + 2 + i] = 2 * col_h[cb] + (cb == lb ? 1 : 0);
        }
      }
  }
  fp_off.assign(fp_written.size(), 0);
  fp_data.clear();
  // a shard only needs the tables of the footprints its own groups use (a long sequence has millions of footprints)
  std::vector<char> fp_used(fp_written.size(), world == 1 ? 1 : 0);
  if (world > 1) for (const Batch& b : batches) {
    for (const GroupRec& g : b.my_groups) fp_used[g.fp] = 1;
    for (const GroupRec& g : b.my_obs_groups) fp_used[g.fp] = 1;
    for (const GroupRec& g : b.lm_groups) fp_used[g.fp] = 1;
  }
  for (size_t f = 0; f < fp_written.size(); ++f) {
    fp_off[f] = (int)fp_data.size();
    if (!fp_used[f]) continue;
    // written blocks sorted by block index
    std::vector<std::array<int, 3>> w = fp_written[f];
    std::vector<std::pair<int, int>> blocks;  // (block index, role) expanded
    std::vector<int> rolecol(MC_MAX_ROLES, -1);
    for (auto& e : w) for (int k = 0; k < e[2]; ++k) blocks.push_back({e[1] + k, k == 0 ? e[0] : -1});
    std::sort(blocks.begin(), blocks.end());
    int nb = (int)blocks.size();
    std::vector<int> colstart(nb + 1, 0);
    for (int i = 0; i < nb; ++i) {
      colstart[i + 1] = colstart[i] + block_dim[blocks[i].first];
      if (blocks[i].second >= 0) rolecol[blocks[i].second] = colstart[i];
    }
    int C = colstart[nb];
    fp_data.push_back(nb); fp_data.push_back(C);
    fp_data.push_back(fp_coef_lo[f]); fp_data.push_back(fp_nbuf[f]);
    for (int r = 0; r < MC_MAX_ROLES; ++r) fp_data.push_back(rolecol[r]);
    for (int i = 0; i < nb; ++i) {
      fp_data.push_back(blocks[i].first); fp_data.push_back(colstart[i]);
      fp_data.push_back(block_dim[blocks[i].first]); fp_data.push_back(block_col[blocks[i].first]);
    }
    for (int ib = 0; ib < nb; ++ib)
      for (int ia = 0; ia <= ib; ++ia) fp_data.push_back(slotBase(blocks[ia].first, blocks[ib].first));
    for (int i = 0; i < nb; ++i)
      for (int c = colstart[i]; c < colstart[i + 1]; ++c) fp_data.push_back(i);
    for (int i = 0; i < nb; ++i) fp_data.push_back(col_h[blocks[i].first]);   // column height of each block (address stride)
  }
  fp_off.push_back((int)fp_data.size());   // sentinel: footprint f spans [fp_off[f], fp_off[f+1])
  lap("footprint tables");
  // ---- shared entries (world > 1): owner rank of every block pair over ALL ranks' work items
  shared_idx.clear();
  shared_rhs.clear();
  if (world > 1) {
    auto ownerOf = [&](double t) { return shardOf(t); };   // shard_cuts hold every rank's window
    std::vector<signed char> owner(pat_rows.size(), -1);   // -1 untouched, -2 shared, else rank
    std::vector<char> mine(pat_rows.size(), 0);            // this rank is among the writers
    auto mark = [&](int b1, int b2, int r) {
      int rb = std::min(b1, b2), cb = std::max(b1, b2);
      const std::vector<int>& rows = col_rows[cb];
      auto it = std::lower_bound(rows.begin(), rows.end(), rb);
      if (it == rows.end() || *it != rb) return;
      size_t k = pat_colptr[cb] + (it - rows.begin());
      if (r == rank) mine[k] = 1;
      if (owner[k] == -1) owner[k] = (signed char)r;
      else if (owner[k] != r) owner[k] = -2;
    };
    if (world > 120) throw std::invalid_argument("at most 120 shards");
    for (const Batch& b : batches) {
      for (size_t g = 0; g < b.groups.size(); ++g) {
        int r = ownerOf(b.group_time[g]);
        std::vector<int> blks;
        for (auto& w : fp_written[b.groups[g].fp]) for (int k = 0; k < w[2]; ++k) blks.push_back(w[1] + k);
        for (size_t i = 0; i < blks.size(); ++i) for (size_t j = i; j < blks.size(); ++j) mark(blks[i], blks[j], r);
      }
      const int pose_block0 = (b.spline >= 0 && !b.chunks.empty()) ? dvs[splines[b.spline].first_dv].block : -1;
      for (size_t g = 0; g < b.chunks.size(); ++g) {
        int r = ownerOf(b.chunk_time[g]);
        const ChunkRec& c = b.chunks[g];
        const int* of = &ofp_data[ofp_off[c.ofp]];
        const int nob = of[0];
        const int* ob = of + OFP_HDR;
        for (int bj = 0; bj < c.pose_nb; ++bj) {
          for (int d = 0; d < 6 && bj - d >= 0; ++d) mark(pose_block0 + c.pose_lo + bj - d, pose_block0 + c.pose_lo + bj, r);
          for (int o = 0; o < nob; ++o) mark(ob[4 * o], pose_block0 + c.pose_lo + bj, r);
        }
        for (int o1 = 0; o1 < nob; ++o1) for (int o2 = o1; o2 < nob; ++o2) mark(ob[4 * o1], ob[4 * o2], r);
      }
      if (b.kind == KIND_MOTION) {
        const SplineRec& sp = splines[b.spline];
        int b0 = dvs[sp.first_dv].block;
        if (b0 >= 0)
          for (int i = 0; i < sp.n_coeff; ++i) {
            const int r = shardOf(knots[sp.knots_off + i + sp.order / 2]);   // motionRange()'s rule
            for (int d = 0; d < sp.order && i - d >= 0; ++d) mark(b0 + i - d, b0 + i, r);
          }
      }
    }
    for (int cb = 0; cb < n_blocks; ++cb)
      for (int k = pat_colptr[cb]; k < pat_colptr[cb + 1]; ++k) {
        if (owner[k] != -2) continue;
        int rb = pat_rows[k], h = col_h[cb];
        if (rb == cb) for (int sc = 0; sc < block_dim[cb]; ++sc) shared_rhs.push_back(block_col[cb] + sc);
        for (int sc = 0; sc < block_dim[cb]; ++sc) {
          int base = pat_cp0[cb] + sc * h + (sc * (sc + 1)) / 2 + pat_roff[k];
          int n = (rb == cb) ? sc + 1 : block_dim[rb];
          for (int rr = 0; rr < n; ++rr) shared_idx.push_back(base + rr);
        }
      }
    std::sort(shared_idx.begin(), shared_idx.end());
    // the values this rank writes, in value order (column block, scalar column, row block)
    written_cx.clear();
    auto run = [](std::vector<std::pair<long long, long long>>& v, long long first, long long count) {
      if (!v.empty() && v.back().first + v.back().second == first) v.back().second += count;
      else v.push_back({first, count});
    };
    for (int cb = 0; cb < n_blocks; ++cb) {
      bool any = false;
      for (int k = pat_colptr[cb]; k < pat_colptr[cb + 1] && !any; ++k) any = mine[k] != 0;
      if (!any) continue;
      const int h = col_h[cb];
      for (int sc = 0; sc < block_dim[cb]; ++sc)
        for (int k = pat_colptr[cb]; k < pat_colptr[cb + 1]; ++k) {
          if (!mine[k]) continue;
          const int rb = pat_rows[k];
          run(written_cx, (long long)pat_cp0[cb] + (long long)sc * h + (sc * (sc + 1)) / 2 + pat_roff[k], (rb == cb) ? sc + 1 : block_dim[rb]);
        }
    }
    // ... and the residual rows of its terms
    written_e.clear();
    for (const Batch& b : batches) {
      if (b.kind == KIND_MOTION) { run(written_e, b.row_base, (long long)b.n_terms * b.dim); continue; }
      std::vector<int> idx;
      for (const GroupRec& g : b.my_groups) for (int k = 0; k < g.term_count; ++k) idx.push_back(g.term_first + k);
      for (const ChunkRec& c : b.my_chunks) for (int k = 0; k < c.term_count; ++k) idx.push_back(b.order[c.term_first + k]);
      std::sort(idx.begin(), idx.end());
      for (int i : idx) run(written_e, (long long)b.row_base + (long long)i * b.dim, b.dim);
    }
  }
  lap("shared entries");
  // ---- J^T (SparseCholesky backend) layout: elements per column and first value index of every term
  jt_start.assign(n_terms + 1, 0); jt_epc.assign(n_terms, 0); jt_supported = true;
  if (world > 1) jt_supported = false;   // the J^T backend is single-GPU (and the other shards' footprint tables are not built)
  for (const Batch& b : batches) {
    if (world > 1) break;
    if (b.kind == KIND_MOTION) { jt_supported = false; continue; }   // BSplineMotionError has no evaluateJacobians in the reference either
    for (const GroupRec& g : b.groups) {
      int C = fp_data[fp_off[g.fp] + 1];
      for (int k = 0; k < g.term_count; ++k) jt_epc[b.term_base + g.term_first + k] = C;
    }
    const bool poseActive = b.spline >= 0 && dvs[splines[b.spline].first_dv].block >= 0;
    for (const ChunkRec& c : b.chunks) {
      int fixed = ofp_jt[(size_t)c.ofp * JT_STRIDE + 3 * MC_MAX_ROLES];
      for (int k = 0; k < c.term_count; ++k) {
        int u = b.order[c.term_first + k];
        jt_epc[b.term_base + u] = fixed + (poseActive ? 6 * b.term_nbuf[u] : 0);
      }
    }
  }
  {
    int t = 0;
    long long acc = 0;
    for (const Batch& b : batches)
      for (int k = 0; k < b.n_terms; ++k, ++t) { jt_start[t] = acc; acc += (long long)jt_epc[t] * b.dim; }
    jt_start[n_terms] = acc;
    jt_nnz = acc;
  }
  lap("J^T layout");
  update_dvs = block_dv;
  finalized = true;
}

// Symbolic J^T of SparseCholeskyLinearSystemSolver (CompressedColumnMatrix::appendJacobiansSymbolic,
// AB/include/aslam/backend/implementation/CompressedColumnMatrix.hpp:236-285): per residual row one column holding the
// scalar columns of the term's active design variables in block-index order.
inline void Problem::jtPattern(std::vector<int>& col_ptr, std::vector<int>& row_ind) const {
  if (!jt_supported) throw std::runtime_error("This is currently unsupported");   // BSplineMotionError::evaluateJacobians
  if (jt_nnz > 0x7fffffffLL) throw std::overflow_error("J^T exceeds 2^31 entries");
  col_ptr.assign(j_rows + 1, 0);
  row_ind.assign((size_t)jt_nnz, 0);
  auto emit = [&](int gterm, int dim, std::vector<int>& blocks) {
    std::sort(blocks.begin(), blocks.end());
    long long at = jt_start[gterm];
    int epc = jt_epc[gterm];
    for (int r = 0; r < dim; ++r) {
      int k = 0;
      for (int bk : blocks) for (int c = 0; c < block_dim[bk]; ++c) row_ind[(size_t)(at + (long long)r * epc + k++)] = block_col[bk] + c;
      if (k != epc) throw std::logic_error("J^T layout mismatch");
    }
  };
  for (const Batch& b : batches) {
    for (const GroupRec& g : b.groups) {
      std::vector<int> blocks;
      for (auto& w : fp_written[g.fp]) for (int k = 0; k < w[2]; ++k) blocks.push_back(w[1] + k);
      for (int k = 0; k < g.term_count; ++k) { std::vector<int> bl = blocks; emit(b.term_base + g.term_first + k, b.dim, bl); }
    }
    if (b.chunks.empty()) continue;
    const SplineRec& sp = splines[b.spline];
    const int coef_block0 = dvs[sp.first_dv].block;
    const bool lidar = b.kind == KIND_LIDAR;
    const SplineRec* bs = (!lidar && b.bias_spline >= 0) ? &splines[b.bias_spline] : nullptr;
    const int bias_block0 = bs ? dvs[bs->first_dv].block : -1;
    const int biasRole = lidar ? -1 : (int)IMU_ROLE_BIAS;
    for (const ChunkRec& c : b.chunks) {
      const int* of = &ofp_data[ofp_off[c.ofp]];
      std::vector<int> fixed;
      // the ofp lists the bias WINDOW of the chunk first (if any), then the single blocks
      int nob = of[0], skip = (bs && bias_block0 >= 0) ? of[3] : 0;
      for (int o = skip; o < nob; ++o) fixed.push_back(of[OFP_HDR + 4 * o]);
      (void)biasRole;
      for (int k = 0; k < c.term_count; ++k) {
        int u = b.order[c.term_first + k];
        std::vector<int> blocks = fixed;
        if (coef_block0 >= 0) for (int i = 0; i < b.term_nbuf[u]; ++i) blocks.push_back(coef_block0 + b.term_lo[u] + i);
        if (bs && bias_block0 >= 0) { int bseg = segmentOf(*bs, b.t[u]); for (int i = 0; i < bs->order; ++i) blocks.push_back(bias_block0 + bseg + i); }
        emit(b.term_base + u, b.dim, blocks);
      }
    }
  }
  int col = 0;
  for (const Batch& b : batches)
    for (int k = 0; k < b.n_terms; ++k)
      for (int r = 0; r < b.dim; ++r, ++col) col_ptr[col + 1] = col_ptr[col] + jt_epc[b.term_base + k];
}

}  // namespace mc
