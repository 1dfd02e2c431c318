// TEST INFRASTRUCTURE ONLY.  Compiles the product's term closed forms (multical_amd/csrc/mc_terms.h) and
// structure analysis (mc_core.hpp) with g++ and evaluates single terms on the CPU, so that the `-m "not gpu"`
// suite can compare the exact source the kernels are built from against the DAG-walking oracle before any GPU
// time is spent.  It is not part of the package, is never imported by it, and has no role at run time: the
// product path is the HIP library only.
#include <memory>
#include "../../multical_amd/csrc/mc_describe.hpp"
#include "../../multical_amd/csrc/host/block_cholesky.hpp"
#include "../../multical_amd/csrc/host/lm_core.hpp"
#include "../../multical_amd/csrc/host/schur_plan.hpp"

using namespace mc;

struct Probe {
  Problem P;
  std::string err;
  std::vector<double> x_prev;   // one-level undo of the state (DesignVariable::revertUpdate)
  // host copies of per-batch offset arrays (the device path uploads the same)
  std::vector<std::vector<int>> obs_q_off, obs_t_off, term_q_off, term_t_off;
  std::vector<int> term_batch, term_local;
};

#define PR_TRY try {
#define PR_CATCH(p, rv) } catch (const std::exception& ex) { (p)->err = ex.what(); return rv; }

static std::vector<int> offs(const Problem& P, const std::vector<int>& ids) {
  std::vector<int> o(ids.size());
  for (size_t i = 0; i < ids.size(); ++i) o[i] = P.dvs[ids[i]].off;
  return o;
}

extern "C" {
void* probe_create() { return new Probe(); }
void probe_destroy(void* h) { delete (Probe*)h; }
const char* probe_last_error(void* h) { return ((Probe*)h)->err.c_str(); }
int probe_add_design_variable(void* h, int kind, const double* p, int n, int active, double scaling) {
  Probe* pr = (Probe*)h; PR_TRY return pr->P.addDv(kind, p, n, active, scaling); PR_CATCH(pr, -1)
}
int probe_add_spline(void* h, int order, int dim, const double* k, int nk, const double* c, int nc, int active) {
  Probe* pr = (Probe*)h; PR_TRY return pr->P.addSpline(order, dim, k, nk, c, nc, active); PR_CATCH(pr, -1)
}
int probe_add_camera(void* h, int model, int dist, int dp, int dd) {
  Probe* pr = (Probe*)h; PR_TRY return pr->P.addCamera(model, dist, dp, dd); PR_CATCH(pr, -1)
}
int probe_add_reprojection_batch(void* h, int cam, int spline, int dq, int dt, int dtoff, int n_obs, const double* ot, const int* oq,
                                 const int* otv, const int* of, const double* y, const double* p, const double* s, double pad, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeReprojection(pr->P, cam, spline, dq, dt, dtoff, n_obs, ot, oq, otv, of, y, p, s, pad, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_static_reprojection_batch(void* h, int cam, int nc, const int* cq, const int* ct, int n_obs, const int* oq, const int* ot,
                                        const int* of, const double* y, const double* p, const double* s, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeStaticReprojection(pr->P, cam, nc, cq, ct, n_obs, oq, ot, of, y, p, s, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_gyroscope_prior_batch(void* h, int dq, int db, int n, const double* wp, const double* me, const double* s, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeGyroPrior(pr->P, dq, db, n, wp, me, s, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_transformation_batch(void* h, int qa, int ta, int qb, int tb, int n, const double* pq, const double* pt, double wr,
                                   double wt, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeTransformation(pr->P, qa, ta, qb, tb, n, pq, pt, wr, wt, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_accelerometer_batch(void* h, int ps, int bs, int db, int dg, int dq, int dr, int n, const double* t, const double* me,
                                  const double* s, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeImu(pr->P, true, ps, bs, db, dg, dq, dr, n, t, me, s, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_gyroscope_batch(void* h, int ps, int bs, int db, int dq, int n, const double* t, const double* me, const double* s, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeImu(pr->P, false, ps, bs, db, -1, dq, -1, n, t, me, s, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_matrix_design_variable(void* h, const double* A, const int* pat, int active, double scaling) {
  Probe* pr = (Probe*)h; PR_TRY return pr->P.addMatrixDv(A, pat, active, scaling); PR_CATCH(pr, -1)
}
int probe_add_accelerometer_batch_sm(void* h, int ps, int bs, int db, int dg, int dq, int dr, int dM, const int* se, int n, const double* t,
                                     const double* me, const double* s, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeImuSm(pr->P, true, ps, bs, db, dg, dq, dr, dM, -1, -1, se, n, t, me, s, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_gyroscope_batch_sm(void* h, int ps, int bs, int db, int dq, int dqg, int dM, int dMa, int dg, int dr, int n, const double* t,
                                 const double* me, const double* s, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeImuSm(pr->P, false, ps, bs, db, dg, dq, dr, dM, dqg, dMa, nullptr, n, t, me, s, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_lidar_batch(void* h, int sp, int dtoff, int dq, int dt, int n, const double* t, const double* dir, const double* rg,
                          const double* s, const int* q, const int* tv, double pad, int m, double ma) {
  Probe* pr = (Probe*)h; PR_TRY describeLidar(pr->P, sp, dtoff, dq, dt, n, t, dir, rg, s, q, tv, pad, m, ma); return 0; PR_CATCH(pr, -1)
}
int probe_add_motion_term(void* h, int sp, const double* W, int order) {
  Probe* pr = (Probe*)h; PR_TRY describeMotion(pr->P, sp, W, order); return 0; PR_CATCH(pr, -1)
}
int probe_set_last_batch_landmarks(void* h, int n, const int* dv) {
  Probe* pr = (Probe*)h; PR_TRY describeLandmarks(pr->P, n, dv); return 0; PR_CATCH(pr, -1)
}
int probe_set_last_batch_sqrt_inv_r(void* h, int dim, int n, const double* S) {
  Probe* pr = (Probe*)h; PR_TRY describeSqrtInvR(pr->P, dim, n, S); return 0; PR_CATCH(pr, -1)
}
int probe_set_shard(void* h, int rank, int world) { Probe* pr = (Probe*)h; pr->P.rank = rank; pr->P.world = world; return 0; }
int probe_finalize(void* h) {
  Probe* pr = (Probe*)h;
  PR_TRY
  Problem& P = pr->P;
  P.finalize();
  for (size_t b = 0; b < P.batches.size(); ++b) {
    const Batch& B = P.batches[b];
    pr->obs_q_off.push_back(offs(P, B.obs_dv_q)); pr->obs_t_off.push_back(offs(P, B.obs_dv_t));
    pr->term_q_off.push_back(offs(P, B.term_dv_q)); pr->term_t_off.push_back(offs(P, B.term_dv_t));
    for (int k = 0; k < B.n_terms; ++k) { pr->term_batch.push_back((int)b); pr->term_local.push_back(k); }
  }
  return 0;
  PR_CATCH(pr, -1)
}
int probe_dims(void* h, int* jr, int* jc, int* nb, int* nt, int* nnz) {
  Problem& P = ((Probe*)h)->P;
  *jr = P.j_rows; *jc = P.j_cols; *nb = P.n_blocks; *nt = P.n_terms; *nnz = P.nnz;
  return 0;
}
int probe_get_pattern(void* h, int* Cp, int* Ci) {
  Problem& P = ((Probe*)h)->P;
  std::memcpy(Cp, P.Cp.data(), sizeof(int) * P.Cp.size());
  std::memcpy(Ci, P.Ci.data(), sizeof(int) * P.Ci.size());
  return 0;
}
int probe_get_block_bases(void* h, int* base) {
  Problem& P = ((Probe*)h)->P;
  int c = 0;
  for (int b = 0; b < P.n_blocks; ++b) { base[b] = c; c += P.block_dim[b]; }
  base[P.n_blocks] = c;
  return 0;
}
int probe_get_block_times(void* h, double* t) {
  Problem& P = ((Probe*)h)->P;
  for (int b = 0; b < P.n_blocks; ++b) t[b] = std::nan("");
  for (const SplineRec& s : P.splines)
    for (int i = 0; i < s.n_coeff; ++i) {
      const int b = P.dvs[s.first_dv + i].block;
      if (b >= 0) t[b] = P.knots[s.knots_off + i + s.order];
    }
  return 0;
}
// spline of every coefficient block (-1: not a spline coefficient)
static std::vector<int> blockSplines(const Problem& P) {
  std::vector<int> sp(P.n_blocks, -1);
  for (size_t s = 0; s < P.splines.size(); ++s)
    for (int i = 0; i < P.splines[s].n_coeff; ++i) {
      const int b = P.dvs[P.splines[s].first_dv + i].block;
      if (b >= 0) sp[b] = (int)s;
    }
  return sp;
}
// The factorisation's GPU stage on the host: plan (host/schur_plan.hpp) + its reference executor + the host factorisation of
// the reduced system.  Solves (H + shift I) x = g for the system in Cx (this problem's pattern; the plan is made on the
// non-zero blocks of Cx_plan when that is given).  info: [plan ok, interiors,
// r, reduced columns, reduced nnz, pairs that must stay zero]; returns 0 ok, 1 not positive definite, 2 structure grew,
// 3 no plan for this problem.
int probe_schur_solve(void* h, const double* Cx_plan, const double* Cx, const double* g, double shift, int parts, double* x, int* info) {
  Probe* pr = (Probe*)h;
  PR_TRY
  Problem& P = pr->P;
  std::vector<int> base(P.n_blocks + 1, 0);
  for (int b = 0; b < P.n_blocks; ++b) base[b + 1] = base[b] + P.block_dim[b];
  const std::vector<int> sp = blockSplines(P);
  multical_amd::SchurPlan plan;
  plan.build(P.j_cols, P.Cp.data(), P.Ci.data(), P.n_blocks, base.data(), sp.data(), Cx_plan ? Cx_plan : Cx, parts);
  info[0] = plan.ok ? 1 : 0;
  if (!plan.ok) { pr->err = plan.why; return 3; }
  info[1] = (int)plan.parts.size(); info[2] = plan.r; info[3] = plan.R.n; info[4] = (int)plan.R.nnz(); info[5] = (int)plan.check.size();
  std::vector<double> S, bS;
  multical_amd::SchurPlan::HostWork wk;
  if (!plan.eliminate(Cx, g, shift, S, bS, wk)) return 2;
  std::vector<int> CpR, CiR;
  plan.R.toScalar(CpR, CiR);
  multical_amd::BlockSparseCholesky chol;
  chol.analyze(plan.R.n, CpR.data(), CiR.data(), plan.R.nb, plan.R.base.data());
  std::vector<double> when(P.n_blocks), whenR(plan.R.nb);
  probe_get_block_times(h, when.data());
  for (int i = 0; i < plan.R.nb; ++i) whenR[i] = when[plan.rblocks[i]];
  chol.setBlockCoordinates(whenR.data(), 2);
  if (!chol.factorize(S.data())) return 1;
  std::vector<double> xR(plan.R.n);
  chol.solve(bS.data(), xR.data());
  plan.back(xR.data(), wk, x);
  return 0;
  PR_CATCH(pr, -1)
}
// Timing aid (scripts/schur_host_profile.py): the host factorisation of the reduced system alone, as SchurSolver runs it
// after the device stage -- here behind the reference executor.  out = {factorize best ms, solve best ms, reduced columns}.
int probe_schur_reduced_bench(void* h, const double* Cx, const double* g, double shift, int parts, int threads, int reps, double* out) {
  Probe* pr = (Probe*)h;
  PR_TRY
  Problem& P = pr->P;
  std::vector<int> base(P.n_blocks + 1, 0);
  for (int b = 0; b < P.n_blocks; ++b) base[b + 1] = base[b] + P.block_dim[b];
  const std::vector<int> sp = blockSplines(P);
  multical_amd::SchurPlan plan;
  plan.build(P.j_cols, P.Cp.data(), P.Ci.data(), P.n_blocks, base.data(), sp.data(), Cx, parts);
  if (!plan.ok) { pr->err = plan.why; return 3; }
  std::vector<double> S, bS;
  multical_amd::SchurPlan::HostWork wk;
  if (!plan.eliminate(Cx, g, shift, S, bS, wk)) return 2;
  std::vector<int> CpR, CiR;
  plan.R.toScalar(CpR, CiR);
  multical_amd::BlockSparseCholesky chol;
  chol.analyze(plan.R.n, CpR.data(), CiR.data(), plan.R.nb, plan.R.base.data());
  std::vector<double> when(P.n_blocks), whenR(plan.R.nb);
  probe_get_block_times(h, when.data());
  for (int i = 0; i < plan.R.nb; ++i) whenR[i] = when[plan.rblocks[i]];
  chol.setBlockCoordinates(whenR.data(), threads);
  std::vector<double> xR(plan.R.n);
  double bf = 1e30, bs = 1e30;
  for (int r = 0; r < reps; ++r) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!chol.factorize(S.data())) return 1;
    const auto t1 = std::chrono::steady_clock::now();
    chol.solve(bS.data(), xR.data());
    const auto t2 = std::chrono::steady_clock::now();
    bf = std::min(bf, std::chrono::duration<double, std::milli>(t1 - t0).count());
    bs = std::min(bs, std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  out[0] = bf; out[1] = bs; out[2] = plan.R.n;
  double chk = 0.0;
  for (double v : xR) chk += v * v;
  out[3] = chk;
  std::fprintf(stderr, "reduced factor: %lld entries, %.1f Mflop, %d panels, %d parts, %d top blocks\n", chol.stats().nnz_factor, chol.stats().flops * 1e-6, chol.stats().n_panels, chol.stats().n_parts, chol.stats().n_top_blocks);
  return 0;
  PR_CATCH(pr, -1)
}
// number of groups kept by this rank / total (sharding checks)
int probe_group_counts(void* h, int* mine, int* total, int* my_terms) {
  Problem& P = ((Probe*)h)->P;
  *mine = *total = *my_terms = 0;
  for (const Batch& b : P.batches) {
    *mine += (int)b.my_groups.size() + (int)b.my_chunks.size(); *total += (int)b.groups.size() + (int)b.chunks.size();
    for (const GroupRec& g : b.my_groups) *my_terms += g.term_count;
    for (const ChunkRec& c : b.my_chunks) *my_terms += c.term_count;
  }
  return 0;
}
// (first, count) runs of the values / residual rows this rank writes (Problem::written_cx / written_e)
int probe_written_counts(void* h, int* n_cx, int* n_e) {
  Problem& P = ((Probe*)h)->P;
  *n_cx = (int)P.written_cx.size(); *n_e = (int)P.written_e.size();
  return 0;
}
int probe_written(void* h, long long* cx, long long* e) {
  Problem& P = ((Probe*)h)->P;
  for (size_t i = 0; i < P.written_cx.size(); ++i) { cx[2 * i] = P.written_cx[i].first; cx[2 * i + 1] = P.written_cx[i].second; }
  for (size_t i = 0; i < P.written_e.size(); ++i) { e[2 * i] = P.written_e[i].first; e[2 * i + 1] = P.written_e[i].second; }
  return 0;
}
int probe_shared_count(void* h) { return (int)((Probe*)h)->P.shared_idx.size(); }
int probe_shared_rhs_count(void* h) { return (int)((Probe*)h)->P.shared_rhs.size(); }
int probe_shared_rhs_indices(void* h, int* out) {
  Problem& P = ((Probe*)h)->P;
  std::memcpy(out, P.shared_rhs.data(), sizeof(int) * P.shared_rhs.size());
  return 0;
}
int probe_shared_indices(void* h, int* out) {
  Problem& P = ((Probe*)h)->P;
  std::memcpy(out, P.shared_idx.data(), sizeof(int) * P.shared_idx.size());
  return 0;
}
// per-term flag: 1 if the term belongs to this rank's groups
int probe_my_terms(void* h, int* flags) {
  Problem& P = ((Probe*)h)->P;
  int base = 0;
  for (const Batch& b : P.batches) {
    for (int k = 0; k < b.n_terms; ++k) flags[base + k] = (b.kind == KIND_MOTION && P.rank == 0) ? 1 : 0;
    for (const GroupRec& g : b.my_groups) for (int k = 0; k < g.term_count; ++k) flags[base + g.term_first + k] = 1;
    for (const ChunkRec& c : b.my_chunks) for (int k = 0; k < c.term_count; ++k) flags[base + b.order[c.term_first + k]] = 1;
    base += b.n_terms;
  }
  return 0;
}

// Dense weighted Jacobian rows (dim x j_cols, row-major), weighted residual and raw chi2 of one term, evaluated
// with the product's closed forms at the current state P.x.  Returns the term dimension.
int probe_term(void* h, int term, int useM, double* Jd, double* e, double* chi2) {
  Probe* pr = (Probe*)h;
  PR_TRY
  Problem& P = pr->P;
  int bi = pr->term_batch[term], k = pr->term_local[term];
  const Batch& B = P.batches[bi];
  const double* x = P.x.data();
  int JC = P.j_cols;
  std::memset(Jd, 0, sizeof(double) * B.dim * JC);
  TermOut out{0, 0};
  int rolecol[MC_MAX_ROLES];
  for (int r = 0; r < MC_MAX_ROLES; ++r) rolecol[r] = -1;
  auto col = [&](int dv) { return dv >= 0 ? P.dvs[dv].col : -1; };
  if (B.kind == KIND_REPROJ) {
    ReprojBatchView v;
    v.cam = P.cameraView(B.cam); v.sp = P.splineView(B.spline, P.knots.data(), P.basis.data());
    v.q_cb = P.dvRef(B.dv[0]); v.t_cb = P.dvRef(B.dv[1]); v.toff = P.dvRef(B.dv[2]);
    v.coef_scale = v.qpw_scale = v.tpw_scale = 1.0;
    v.obs_time = B.t.data(); v.obs_first = B.obs_first.data(); v.obs_qpw_off = pr->obs_q_off[bi].data(); v.obs_tpw_off = pr->obs_t_off[bi].data();
    v.y = B.y.data(); v.p = B.p.data(); v.sinv = B.sinv.data(); v.S = B.S.empty() ? nullptr : B.S.data(); v.mest = B.mest; v.mest_a = B.mest_a;
    int obs = (int)(std::upper_bound(B.obs_first.begin(), B.obs_first.end(), k) - B.obs_first.begin()) - 1;
    F1Common cm;
    f1_common(v, x, obs, cm);
    if (!cm.ok) throw std::runtime_error("time out of spline");
    double core[2 * F1_CORE], ee[2];
    f1_term(v, x, cm, k, useM != 0, true, core, F1_CORE, ee, out);
    rolecol[F1_ROLE_COEF] = v.sp.first_col >= 0 ? v.sp.first_col + 6 * cm.seg : -1;
    rolecol[F1_ROLE_QCB] = v.q_cb.col; rolecol[F1_ROLE_TCB] = v.t_cb.col;
    rolecol[F1_ROLE_QPW] = col(B.obs_dv_q[obs]); rolecol[F1_ROLE_TPW] = col(B.obs_dv_t[obs]); rolecol[F1_ROLE_TOFF] = v.toff.col;
    for (int r = 0; r < 2; ++r) { f1_expand_row(core + r * F1_CORE, cm.w, rolecol, Jd + (size_t)r * JC); e[r] = ee[r]; }
  } else if (B.kind == KIND_LIDAR) {
    LidarBatchView v;
    v.sp = P.splineView(B.spline, P.knots.data(), P.basis.data());
    v.q_lb = P.dvRef(B.dv[0]); v.t_lb = P.dvRef(B.dv[1]); v.toff = P.dvRef(B.dv[2]);
    v.coef_scale = v.qpw_scale = v.tpw_scale = 1.0;
    v.t = B.t.data(); v.dir = B.p.data(); v.range = B.range.data(); v.sinv = B.sinv.data();
    v.qpw_off = pr->term_q_off[bi].data(); v.tpw_off = pr->term_t_off[bi].data(); v.mest = B.mest; v.mest_a = B.mest_a;
    double t = B.t[k] + (v.toff.off >= 0 ? x[v.toff.off] : 0.0);
    int seg = spline_segment(v.sp, t);
    rolecol[F5_ROLE_COEF] = v.sp.first_col >= 0 ? v.sp.first_col + 6 * seg : -1;
    rolecol[F5_ROLE_QLB] = v.q_lb.col; rolecol[F5_ROLE_TLB] = v.t_lb.col;
    rolecol[F5_ROLE_QPW] = col(B.term_dv_q[k]); rolecol[F5_ROLE_TPW] = col(B.term_dv_t[k]); rolecol[F5_ROLE_TOFF] = v.toff.col;
    if (!f5_term(v, x, k, useM != 0, true, rolecol, Jd, JC, e, out)) throw std::runtime_error("time out of spline");
  } else if (B.kind == KIND_ACCEL || B.kind == KIND_GYRO) {
    ImuBatchView v;
    v.pose = P.splineView(B.spline, P.knots.data(), P.basis.data());
    v.bias = P.splineView(B.bias_spline, P.knots.data(), P.basis.data());
    v.bias_dv = P.dvRef(B.dv[0]); v.g = P.dvRef(B.dv[1]); v.q_ib = P.dvRef(B.dv[2]); v.r_b = P.dvRef(B.dv[3]);
    v.g_is_direction = (B.dv[1] >= 0 && P.dvs[B.dv[1]].kind == DV_EUCLIDEAN_DIRECTION) ? 1 : 0;
    v.pose_scale = 1.0; v.bias_scale = B.bias_spline >= 0 ? 1.0 : v.bias_dv.scale;
    v.t = B.t.data(); v.meas = B.meas.data(); v.sinv = B.sinv.data(); v.S = B.S.empty() ? nullptr : B.S.data(); v.mest = B.mest; v.mest_a = B.mest_a;
    v.is_accel = B.kind == KIND_ACCEL;
    int seg = spline_segment(v.pose, B.t[k]);
    rolecol[IMU_ROLE_POSE] = v.pose.first_col >= 0 ? v.pose.first_col + 6 * seg : -1;
    if (B.bias_spline >= 0) { int bs = spline_segment(v.bias, B.t[k]); rolecol[IMU_ROLE_BIAS] = v.bias.first_col >= 0 ? v.bias.first_col + 3 * bs : -1; }
    else rolecol[IMU_ROLE_BIAS] = v.bias_dv.col;
    rolecol[IMU_ROLE_G] = v.g.col; rolecol[IMU_ROLE_QIB] = v.q_ib.col; rolecol[IMU_ROLE_RB] = v.r_b.col;
    if (!imu_term(v, x, k, useM != 0, true, rolecol, Jd, JC, e, out)) throw std::runtime_error("time out of spline");
  } else if (B.kind == KIND_ACCEL_SM || B.kind == KIND_GYRO_SM) {
    ImuSmBatchView v;
    v.pose = P.splineView(B.spline, P.knots.data(), P.basis.data());
    v.bias = P.splineView(B.bias_spline, P.knots.data(), P.basis.data());
    v.bias_dv = P.dvRef(B.dv[0]); v.g = P.dvRef(B.dv[1]); v.q_ib = P.dvRef(B.dv[2]); v.r_b = P.dvRef(B.dv[3]);
    v.g_is_direction = P.dvs[B.dv[1]].kind == DV_EUCLIDEAN_DIRECTION ? 1 : 0;
    v.pose_scale = 1.0; v.bias_scale = B.bias_spline >= 0 ? 1.0 : v.bias_dv.scale;
    v.t = B.t.data(); v.meas = B.meas.data(); v.sinv = B.sinv.data(); v.S = B.S.empty() ? nullptr : B.S.data(); v.mest = B.mest; v.mest_a = B.mest_a;
    v.is_accel = B.kind == KIND_ACCEL_SM;
    v.M = P.dvRef(B.dv[4]); v.M_mask = P.dvs[B.dv[4]].mask;
    v.q_gi = P.dvRef(B.dv[5]); v.Ma = P.dvRef(B.dv[6]); v.Ma_mask = B.dv[6] >= 0 ? P.dvs[B.dv[6]].mask : 0;
    v.size_effect = B.size_effect;
    for (int j = 0; j < 3; ++j) { v.arm[j] = P.dvRef(B.dv[7 + j]); v.I_off[j] = B.dv[10 + j] >= 0 ? P.dvs[B.dv[10 + j]].off : -1; }
    int seg = spline_segment(v.pose, B.t[k]);
    rolecol[IMU_ROLE_POSE] = v.pose.first_col >= 0 ? v.pose.first_col + 6 * seg : -1;
    if (B.bias_spline >= 0) { int bs = spline_segment(v.bias, B.t[k]); rolecol[IMU_ROLE_BIAS] = v.bias.first_col >= 0 ? v.bias.first_col + 3 * bs : -1; }
    else rolecol[IMU_ROLE_BIAS] = v.bias_dv.col;
    rolecol[IMU_ROLE_G] = v.g.col; rolecol[IMU_ROLE_QIB] = v.q_ib.col; rolecol[IMU_ROLE_RB] = v.r_b.col;
    rolecol[IMU_ROLE_M] = v.M.col; rolecol[IMU_ROLE_QGI] = v.q_gi.col; rolecol[IMU_ROLE_MA] = v.Ma.col;
    for (int j = 0; j < 3; ++j) rolecol[IMU_ROLE_RX + j] = v.arm[j].col;
    if (!imu_sm_term(v, x, k, useM != 0, true, rolecol, Jd, JC, e, out)) throw std::runtime_error("time out of spline");
  } else if (B.kind == KIND_STATIC_REPROJ) {
    StaticReprojBatchView v;
    v.cam = P.cameraView(B.cam);
    v.n_chain = (int)B.chain_q.size();
    for (int j = 0; j < v.n_chain; ++j) { v.chain_q[j] = P.dvRef(B.chain_q[j]); v.chain_t[j] = P.dvRef(B.chain_t[j]); }
    v.qtc_scale = v.ttc_scale = 1.0;
    v.proj_scale = P.dvs[P.cams[B.cam].dv_proj].scaling;
    v.dist_scale = P.cams[B.cam].dv_dist >= 0 ? P.dvs[P.cams[B.cam].dv_dist].scaling : 1.0;
    v.obs_first = B.obs_first.data(); v.obs_qtc_off = pr->obs_q_off[bi].data(); v.obs_ttc_off = pr->obs_t_off[bi].data();
    v.term_obs = B.term_obs.data(); v.y = B.y.data(); v.p = B.p.data(); v.sinv = B.sinv.data(); v.S = B.S.empty() ? nullptr : B.S.data(); v.mest = B.mest; v.mest_a = B.mest_a;
    int obs = B.term_obs[k];
    rolecol[F2_ROLE_QTC] = col(B.obs_dv_q[obs]); rolecol[F2_ROLE_TTC] = col(B.obs_dv_t[obs]);
    for (int j = 0; j < v.n_chain; ++j) { rolecol[F2_ROLE_CHAIN + 2 * j] = v.chain_q[j].col; rolecol[F2_ROLE_CHAIN + 2 * j + 1] = v.chain_t[j].col; }
    rolecol[F2_ROLE_PROJ] = v.cam.proj_col; rolecol[F2_ROLE_DIST] = v.cam.dist_col;
    std::vector<int> lm_off;
    if (!B.term_dv_lm.empty()) {
      lm_off.assign(B.term_dv_lm.size(), -1);
      const int lm = B.term_dv_lm[k];
      if (lm >= 0) { lm_off[k] = P.dvs[lm].off; v.lm_scale = P.dvs[lm].scaling; rolecol[F2_ROLE_LM] = col(lm); }
      v.lm_off = lm_off.data();
    }
    f2_term(v, x, k, useM != 0, true, rolecol, Jd, JC, e, out);
  } else if (B.kind == KIND_GYRO_PRIOR) {
    GyroPriorBatchView v;
    v.q = P.dvRef(B.dv[0]); v.bias = P.dvRef(B.dv[1]);
    v.w_pred = B.p.data(); v.meas = B.meas.data(); v.sinv = B.sinv.data(); v.S = B.S.empty() ? nullptr : B.S.data(); v.mest = B.mest; v.mest_a = B.mest_a;
    rolecol[GP_ROLE_Q] = v.q.col; rolecol[GP_ROLE_BIAS] = v.bias.col;
    gyro_prior_term(v, x, k, useM != 0, true, rolecol, Jd, JC, e, out);
  } else if (B.kind == KIND_TRANSFORMATION) {
    TransformBatchView v;
    v.qa = P.dvRef(B.dv[0]); v.ta = P.dvRef(B.dv[1]); v.qb = P.dvRef(B.dv[2]); v.tb = P.dvRef(B.dv[3]);
    v.prior_inv = B.y.data(); v.sw_t = std::sqrt(B.w_t); v.sw_r = std::sqrt(B.w_r); v.mest = B.mest; v.mest_a = B.mest_a;
    rolecol[TE_ROLE_QA] = v.qa.col; rolecol[TE_ROLE_TA] = v.ta.col; rolecol[TE_ROLE_QB] = v.qb.col; rolecol[TE_ROLE_TB] = v.tb.col;
    transform_term(v, x, k, useM != 0, true, rolecol, Jd, JC, e, out);
  } else {
    throw std::runtime_error("probe_term: motion terms have no Jacobian");
  }
  *chi2 = out.chi2;
  return B.dim;
  PR_CATCH(pr, -1)
}
int probe_term_kind(void* h, int term) { Probe* pr = (Probe*)h; return pr->P.batches[pr->term_batch[term]].kind; }
}

// ---- unit-level access to the product's math (tests/test_oracle_pins.py): the same functions the kernels call
extern "C" {
// spline value / derivative of an order-6 spline through spline_weights (mc_math.h) + Problem::basisMatrix; coeffs D x nc column major
int probe_u_spline_evald(int nk, const double* knots, int D, int nc, const double* coeffs, double t, int d, double* out) {
  if (d < 0 || d > 2) return -1;
  std::vector<double> kn(knots, knots + nk);
  int n_seg = nk - 2 * S6 + 1;
  if (n_seg < 1 || n_seg + S6 - 1 != nc) return -1;
  std::vector<double> basis((size_t)n_seg * S6 * S6);
  for (int s = 0; s < n_seg; ++s) Problem::basisMatrix(kn, S6, s + S6 - 1, &basis[(size_t)s * S6 * S6]);
  SplineView v;
  std::memset(&v, 0, sizeof v);
  v.knots = kn.data(); v.basis = basis.data(); v.n_knots = nk; v.order = S6; v.dim = D; v.n_coeff = nc; v.n_seg = n_seg;
  v.t_min = kn[S6 - 1]; v.t_max = kn[nk - S6];
  double w[3][S6];
  int seg = spline_weights<S6, 3>(v, t, w);
  if (seg < 0) return -1;
  for (int i = 0; i < D; ++i) {
    double a = 0.0;
    for (int j = 0; j < S6; ++j) a += w[d][j] * coeffs[(size_t)(seg + j) * D + i];
    out[i] = a;
  }
  return seg;
}
// S(theta) and d(S(theta) v)/dtheta (rotvec_C_S, rotvec_dSv_dtheta): omega = S v, D 3x3 row major
int probe_u_rotvec_dSv(const double* th, const double* v, double* omega, double* D) {
  double Cm[9], S[9];
  rotvec_C_S(th, Cm, S);
  m3_mulv(S, v, omega);
  rotvec_dSv_dtheta(th, v, D);
  return 0;
}
int probe_u_distort(int kind, const double* k, const double* y, double* yd, double* Jy, double* Jk) {
  double yy[2] = {y[0], y[1]};
  distort<true>(kind, k, yy, Jy);
  yd[0] = yy[0]; yd[1] = yy[1];
  distort_param_jacobian(kind, y, Jk);
  return 0;
}
int probe_u_mest_weights(int kind, double a, int n, const double* e2, double* w) {
  for (int i = 0; i < n; ++i) w[i] = mest_weight<true>(kind, a, e2[i]);
  return 0;
}
int probe_u_update_quat(const double* q, const double* dq, double* out) { update_quat(q, dq, out); return 0; }
int probe_u_quat2r(const double* q, double* R) { quat2r(q, R); return 0; }
int probe_u_rotvec_C(const double* th, double* Cm) { double S[9]; rotvec_C_S(th, Cm, S); return 0; }
// project / intrinsics_jacobian / distortion_jacobian of mc_math.h for one point: kp[2], Jp 2 x 3, Ji 2 x projDims,
// Jd 2 x 4 (untouched without a distortion), row major
int probe_u_project(int model, int dist, const double* proj, const double* dk, const double* p, double* kp, double* Jp, double* Ji,
                    double* Jd) {
  project<true>(model, dist, proj, dk, p, kp, Jp);
  intrinsics_jacobian(model, dist, proj, dk, p, Ji);
  if (dist != 0) distortion_jacobian(model, dist, proj, p, Jd);
  return 0;
}
// Q of the n-th BSplineMotionError batch (describeMotion's band) as a dense (D nc) x (D nc) row-major matrix
int probe_motion_q(void* h, int which, double* out, int N) {
  Probe* pr = (Probe*)h;
  int seen = 0;
  for (const Batch& b : pr->P.batches) {
    if (b.kind != KIND_MOTION) continue;
    if (seen++ != which) continue;
    const SplineRec& s = pr->P.splines[b.spline];
    const int D = s.dim, S = s.order;
    if (N != D * s.n_coeff) return -1;
    std::fill(out, out + (size_t)N * N, 0.0);
    for (int i = 0; i < s.n_coeff; ++i)
      for (int d = 0; d < S && i - d >= 0; ++d) {
        const double* blk = &b.Qband[((size_t)i * S + d) * D * D];
        for (int r = 0; r < D; ++r)
          for (int c = 0; c < D; ++c) {
            out[(size_t)(i * D + r) * N + (i - d) * D + c] = blk[r * D + c];
            out[(size_t)((i - d) * D + c) * N + i * D + r] = blk[r * D + c];
          }
      }
    return 0;
  }
  return -1;
}
}
#include "cpu_optimised.inc"

// ---------------------------------------------------------------------------------------------------------------------
// Host twin of the N-rank iteration (mc_optimize_sharded): the SAME loop (multical_amd/csrc/host/lm_core.hpp) and the same
// factorisation (host/block_cholesky.hpp) over the g++ build of the term math, so that world_size-2 gloo ranks can run
// Optimizer2::optimize over time-window shards on CPUs.  The state update restates k_update (mc_kernels.hip).
static void twinApplyUpdate(Probe* pr, const double* dx) {
  Problem& P = pr->P;
  pr->x_prev = P.x;
  for (int dvi : P.update_dvs) {
    const DvRec& u = P.dvs[dvi];
    const double* d = dx + u.col;
    double* p = &P.x[u.off];
    if (u.kind == DV_ROTATION_QUATERNION || u.kind == DV_HOMOGENEOUS_POINT) {
      const double dq[3] = {d[0] * u.scaling, d[1] * u.scaling, d[2] * u.scaling};
      const double q[4] = {p[0], p[1], p[2], p[3]};
      double o[4];
      update_quat(q, dq, o);
      for (int k = 0; k < 4; ++k) p[k] = o[k];
    } else if (u.kind == DV_EUCLIDEAN_DIRECTION) {
      double R[9], Cm[9], O[9];
      ypr0_to_r(d[0] * u.scaling, d[1] * u.scaling, R);
      for (int k = 0; k < 9; ++k) Cm[k] = p[k];
      m3_mult(Cm, R, O);
      for (int k = 0; k < 9; ++k) p[k] = O[k];
    } else if (u.kind == DV_MATRIX_BASIC) {
      int k = 0;
      for (int idx = 0; idx < 9; ++idx) if ((u.mask >> idx) & 1) p[idx] += d[k++] * u.scaling;
    } else {
      for (int k = 0; k < u.dim; ++k) p[k] += d[k] * u.scaling;
    }
  }
}
typedef int (*twin_allreduce_fn)(void* user, double* buf, long long n);
typedef int (*twin_broadcast_fn)(void* user, double* buf, long long n, int root);
extern "C" int probe_cpu_optimised(void* h, int useM, int build, int n_threads, double* Cx, double* rhs, double* J_out);

namespace {
class TwinEquations : public multical_amd::NormalEquations {
 public:
  TwinEquations(Probe* pr, int threads, twin_allreduce_fn ar, twin_broadcast_fn bc, void* user) : pr_(pr), threads_(threads), ar_(ar), bc_(bc), user_(user) {}
  void prepare(bool damped) override {
    damped_ = damped;
    Problem& P = pr_->P;
    n_ = P.j_cols; nnz_ = P.nnz;
    std::vector<int> base(P.n_blocks + 1, 0);
    for (int b = 0; b < P.n_blocks; ++b) base[b + 1] = base[b] + P.block_dim[b];
    chol_.analyze(n_, P.Cp.data(), P.Ci.data(), P.n_blocks, base.data());
    for (int k = 0; k < 2; ++k) { H_[k].assign(nnz_, 0.0); G_[k].assign(n_, 0.0); }
    image_.assign(nnz_, 0.0);
    g_.assign(n_, 0.0);
  }
  double trial() override {
    const int t = 1 - cur_;
    double cost[2] = {0.0, 0.0};
    if (probe_cpu_optimised(pr_, 1, 1, threads_, H_[t].data(), G_[t].data(), &cost[0]) != 0) cost[1] = 1.0;   // the failure travels with the cost
    if (sharded() && ar_(user_, cost, 2) != 0) throw std::runtime_error("all-reduce of the cost failed");
    if (cost[1] != 0.0) throw std::runtime_error(pr_->err.empty() ? "a term failed on another rank" : pr_->err);
    pending_ = true;
    return cost[0];
  }
  void adopt() override {
    if (!pending_) return;
    pending_ = false;
    cur_ = 1 - cur_;
    if (sharded() && (ar_(user_, H_[cur_].data(), nnz_) != 0 || ar_(user_, G_[cur_].data(), n_) != 0))
      throw std::runtime_error("all-reduce of the normal equations failed");
    image_ = H_[cur_];
    g_ = G_[cur_];
  }
  bool step(double damping, std::vector<double>& dx) override {
    const std::vector<int>& Cp = pr_->P.Cp;
    if (damped_) for (int c = 0; c < n_; ++c) image_[Cp[c + 1] - 1] += damping * damping;
    dx.assign(n_, 0.0);
    bool good = true;
    if (!sharded() || pr_->P.rank == 0) {
      good = chol_.factorize(image_.data());
      if (good) {
        chol_.solve(g_.data(), dx.data());
        for (double v : dx) if (!std::isfinite(v)) { good = false; break; }
      }
    }
    if (sharded()) {
      dx.push_back(good ? 1.0 : 0.0);
      if (bc_(user_, dx.data(), (long long)dx.size(), 0) != 0) throw std::runtime_error("broadcast of the step failed");
      good = dx.back() != 0.0;
      dx.pop_back();
    }
    if (damped_) for (int c = 0; c < n_; ++c) image_[Cp[c + 1] - 1] -= damping;   // the reference's residue (BlockCholeskyLinearSystemSolver.cpp:92-97)
    return good;
  }
  const std::vector<double>& gradient() const override { return g_; }
  void applyUpdate(const std::vector<double>& dx) override { twinApplyUpdate(pr_, dx.data()); }
  void revertUpdate() override { pr_->P.x = pr_->x_prev; }

 private:
  bool sharded() const { return pr_->P.world > 1; }
  Probe* pr_;
  int threads_;
  twin_allreduce_fn ar_;
  twin_broadcast_fn bc_;
  void* user_;
  bool damped_ = true, pending_ = false;
  int n_ = 0, nnz_ = 0, cur_ = 0;
  std::vector<double> H_[2], G_[2], image_, g_;
  multical_amd::BlockSparseCholesky chol_;
};
}  // namespace

extern "C" {
// out5: JStart, JFinal, iterations, failedIterations, linearSolverFailure
int probe_optimize(void* h, int max_iterations, double lambda0, double dx_tol, double ratio_tol, int damped, int n_threads,
                   twin_allreduce_fn ar, twin_broadcast_fn bc, void* user, double* out5) {
  Probe* pr = (Probe*)h;
  PR_TRY
  multical_amd::MinimiserSettings cfg;
  cfg.maxSteps = max_iterations; cfg.initialDamping = lambda0; cfg.stepTolerance = dx_tol; cfg.costRatioTolerance = ratio_tol;
  cfg.damped = damped != 0;
  TwinEquations eq(pr, n_threads, ar, bc, user);
  const multical_amd::MinimiserReport r = multical_amd::minimise(&eq, cfg);
  out5[0] = r.costStart; out5[1] = r.costFinal; out5[2] = r.trials; out5[3] = r.failures; out5[4] = r.solverFailed ? 1.0 : 0.0;
  return 0;
  PR_CATCH(pr, -1)
}
int probe_get_dv_params(void* h, int dv, double* out) {
  Probe* pr = (Probe*)h;
  if (dv < 0 || dv >= (int)pr->P.dvs.size()) return -1;
  const DvRec& d = pr->P.dvs[dv];
  for (int k = 0; k < d.nparam; ++k) out[k] = pr->P.x[d.off + k];
  return d.nparam;
}
int probe_spline_first_dv(void* h, int sid) { return ((Probe*)h)->P.splines[sid].first_dv; }
}
