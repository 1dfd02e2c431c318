// TEST INFRASTRUCTURE ONLY (see omat.hpp).  Included at the end of oracle_capi.cpp.
// Unit-level entry points of the oracle, so that tests/test_oracle_pins.py can run the identities the reference's own
// tests assert (and the vectors produced by the reference's own generated expressions, tests/golden/) against single
// functions of the restatement instead of whole problems:
//   bsplines_python/test/BSplineTests.py:213-257            evalD against fitpack spalde, orders 2..9
//   bsplines/test/BSplinePoseTests.cpp:265-321              curveQuadraticIntegralSparse == curveQuadraticIntegral, 1e-10
//   aslam_backend/test/JacobianContainer.cpp:205-319        evaluateHessian blocks == J1^T invR J2, upper blocks only, doubling
//   aslam_backend/test/LinearSolverTests.cpp:18-141 with test/SampleDvAndError.hpp:9-233   solver-vs-solver equality
//   sm_eigen/include/sm/eigen/matrix_sqrt.hpp:21-40         computeMatrixSqrt (ErrorTermFs::setInvR)

namespace {

// sm::eigen::computeMatrixSqrt (matrix_sqrt.hpp:33-36): S = P^T L sqrt(D) from Eigen::LDLT, whose in-place algorithm
// (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked) pivots symmetrically on the largest remaining |diagonal|
// entry.  Eigen is not in this image: the algorithm is restated from its published description; positive definite input.
inline Mat computeMatrixSqrt(const Mat& Ain) {
  const int n = Ain.r;
  Mat A = Ain;                       // works on the lower triangle
  std::vector<int> tr(n);
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, i)) > best) { best = std::fabs(A(i, i)); piv = i; }
    tr[k] = piv;
    if (piv != k) {                  // symmetric swap of rows / columns k and piv, lower triangle only
      int s = n - piv - 1;
      for (int j = 0; j < k; ++j) std::swap(A(k, j), A(piv, j));
      for (int i = 0; i < s; ++i) std::swap(A(piv + 1 + i, k), A(piv + 1 + i, piv));
      std::swap(A(k, k), A(piv, piv));
      for (int i = k + 1; i < piv; ++i) std::swap(A(i, k), A(piv, i));
    }
    std::vector<double> temp(k);
    for (int j = 0; j < k; ++j) temp[j] = A(j, j) * A(k, j);
    for (int j = 0; j < k; ++j) A(k, k) -= A(k, j) * temp[j];
    for (int i = k + 1; i < n; ++i) for (int j = 0; j < k; ++j) A(i, k) -= A(i, j) * temp[j];
    double akk = A(k, k);
    if (std::fabs(akk) > 0.0) for (int i = k + 1; i < n; ++i) A(i, k) /= akk;
  }
  Mat S(n, n);
  for (int i = 0; i < n; ++i) { S(i, i) = 1.0; for (int j = 0; j < i; ++j) S(i, j) = A(i, j); }   // matrixL
  for (int k = n - 1; k >= 0; --k)                                                                  // transpositionsP().transpose() *
    if (tr[k] != k) for (int j = 0; j < n; ++j) std::swap(S(k, j), S(tr[k], j));
  for (int j = 0; j < n; ++j) { double d = std::sqrt(A(j, j)); for (int i = 0; i < n; ++i) S(i, j) *= d; }   // * sqrt(D)
  return S;
}

// test/SampleDvAndError.hpp:46-173 (LinearErr / LinearErr2 / LinearErr3) in one class: e = p - sum_i J_i v_i over
// vector design variables (Point2d = a 2-vector with additive update; DummyDesignVariable<N> likewise)
struct LinearError : ErrorTerm {
  std::vector<ParamVectorDv*> dv;
  std::vector<Mat> J;
  Mat p;
  LinearError(int dim, const Mat& p_) : ErrorTerm(dim), p(p_) {}
  double evaluateErrorImplementation() override {
    error = p;
    for (size_t i = 0; i < dv.size(); ++i) {
      Mat v((int)dv[i]->p.size(), 1);
      v.v = dv[i]->p;
      error -= J[i] * v;
    }
    return evaluateChiSquaredError();
  }
  void evaluateJacobiansImplementation(JacobianContainer& jc) override { for (size_t i = 0; i < dv.size(); ++i) jc.add(dv[i], -J[i]); }
  void getDvs(std::set<DesignVariable*>& s) override { for (auto* d : dv) s.insert(d); }
};

BSpline unitSpline(int order, int nk, const double* knots, int D, int nc, const double* coeffs) {
  BSpline s;
  Mat c(D, nc);
  c.v.assign(coeffs, coeffs + (size_t)D * nc);
  s.set(order, std::vector<double>(knots, knots + nk), c);
  return s;
}

}  // namespace

extern "C" {

int orc_u_matrix_sqrt(int n, const double* A /* row major */, double* S /* row major */) {
  try {
    Mat a(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a(i, j) = A[i * n + j];
    Mat s = computeMatrixSqrt(a);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[i * n + j] = s(i, j);
    return 0;
  } catch (...) { return -1; }
}
// BSpline::evalD (bsplines/src/BSpline.cpp:331-349); coeffs D x nc column major
int orc_u_spline_evald(int order, int nk, const double* knots, int D, int nc, const double* coeffs, double t, int d, double* out) {
  try {
    BSpline s = unitSpline(order, nk, knots, D, nc, coeffs);
    Mat v = s.evalD(t, d);
    for (int i = 0; i < D; ++i) out[i] = v[i];
    return 0;
  } catch (...) { return -1; }
}
// evalDAndJacobian's Jacobian: Phi (D x D*order) for the order local coefficients, and the index of the first one
int orc_u_spline_phi(int order, int nk, const double* knots, int D, int nc, const double* coeffs, double t, int d, double* phi /* row major */,
                     int* first) {
  try {
    BSpline s = unitSpline(order, nk, knots, D, nc, coeffs);
    Mat J;
    s.evalDAndJacobian(t, d, &J);
    for (int i = 0; i < D; ++i) for (int j = 0; j < D * order; ++j) phi[i * D * order + j] = J(i, j);
    *first = s.localCoefficientStart(t);
    return 0;
  } catch (...) { return -1; }
}
// curveQuadraticIntegral (dense, BSpline.cpp:1669-1685: Q.block(sD, sD) += segment integral) when sparse == 0,
// curveQuadraticIntegralSparse (:1585-1622: D x D blocks scattered into a SparseBlockMatrix) -> toDense when sparse == 1.
// out: (D nc) x (D nc) row major.
int orc_u_curve_quadratic_integral(int order, int nk, const double* knots, int D, int nc, const double* coeffs, const double* W,
                                   int derivativeOrder, int sparse, double* out) {
  try {
    BSpline s = unitSpline(order, nk, knots, D, nc, coeffs);
    Mat Wm(D, D);
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) Wm(i, j) = W[i * D + j];
    const int N = D * nc;
    std::fill(out, out + (size_t)N * N, 0.0);
    if (!sparse) {
      for (int seg = 0; seg < s.numValidTimeSegments(); ++seg) {
        Mat Q = s.segmentQuadraticIntegral(Wm, seg, derivativeOrder);
        for (int i = 0; i < Q.r; ++i) for (int j = 0; j < Q.c; ++j) out[(size_t)(seg * D + i) * N + seg * D + j] += Q(i, j);
      }
    } else {
      SparseBlockMatrix Qs;
      Qs.init(std::vector<int>(nc, D));
      for (int seg = 0; seg < s.numValidTimeSegments(); ++seg) {
        Mat Q = s.segmentQuadraticIntegral(Wm, seg, derivativeOrder);
        for (int i = 0; i < order; ++i)
          for (int j = 0; j < order; ++j) *Qs.block(seg + i, seg + j, true) += Q.block(i * D, j * D, D, D);
      }
      for (size_t c = 0; c < Qs.cols.size(); ++c)
        for (auto& kv : Qs.cols[c])
          for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) out[(size_t)(kv.first * D + i) * N + c * D + j] = kv.second(i, j);
    }
    return 0;
  } catch (...) { return -1; }
}
// RotationVector::angularVelocityAndJacobian (sm_kinematics/src/RotationVector.cpp:108-183): omega[3], J 3 x 6 row major
int orc_u_rv_angular_velocity_and_jacobian(const double* p, const double* pdot, double* omega, double* J) {
  Mat P = Mat::Vec({p[0], p[1], p[2]}), V = Mat::Vec({pdot[0], pdot[1], pdot[2]}), Jm;
  Mat w = rvAngularVelocityAndJacobian(P, V, &Jm);
  for (int i = 0; i < 3; ++i) { omega[i] = w[i]; for (int j = 0; j < 6; ++j) J[i * 6 + j] = Jm(i, j); }
  return 0;
}
// {RadialTangential, Equidistant}Distortion::distort with Jacobian and distortParameterJacobian; kind 1 / 2
int orc_u_distort(int kind, const double* k, const double* y, double* yd, double* Jy /* 2x2 row major */, double* Jk /* 2x4 row major */) {
  Distortion d;
  d.kind = kind;
  for (int i = 0; i < 4; ++i) d.k[i] = k[i];
  double yy[2] = {y[0], y[1]};
  Mat J;
  d.distort(yy, &J);
  yd[0] = yy[0]; yd[1] = yy[1];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) Jy[2 * i + j] = J(i, j);
  Mat P = d.parameterJacobian(y);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) Jk[4 * i + j] = P(i, j);
  return 0;
}

// quat2r (quaternion_algebra.cpp:77-101) and RotationVector::parametersToRotationMatrix (RotationVector.cpp:10-55): 3 x 3 row major
int orc_u_quat2r(const double* q, double* R) {
  Mat M = quat2r(Mat::Vec({q[0], q[1], q[2], q[3]}));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = M(i, j);
  return 0;
}
// MEstimator::getWeight (MEstimatorPolicies.cpp:17-129): n weights for squared errors e2[i], kind / parameter as in oterms.hpp
int orc_u_mest_weights(int kind, double a, int n, const double* e2, double* w) {
  MEstimator m;
  m.kind = kind; m.a = a;
  for (int i = 0; i < n; ++i) w[i] = m.getWeight(e2[i]);
  return 0;
}
// sm::kinematics::updateQuat (quaternion_algebra.cpp:302-318): what RotationQuaternion / HomogeneousPoint::update apply
int orc_u_update_quat(const double* q, const double* dq, double* out) {
  Mat r = updateQuat(Mat::Vec({q[0], q[1], q[2], q[3]}), Mat::Vec({dq[0], dq[1], dq[2]}));
  for (int i = 0; i < 4; ++i) out[i] = r[i];
  return 0;
}
int orc_u_rv_rotation_matrix(const double* p, double* Cm) {
  Mat M = rvRotationMatrix(Mat::Vec({p[0], p[1], p[2]}));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Cm[3 * i + j] = M(i, j);
  return 0;
}
// {Pinhole,Omni,DoubleSphere}Projection::euclideanToKeypoint with Jacobian, euclideanToKeypointIntrinsicsJacobian and
// euclideanToKeypointDistortionJacobian of one point.  model 0 / 1 / 2, dist 0 / 1 / 2 as in ocam.hpp; proj in update()
// order; kp[2], Jp 2 x 3, Ji 2 x projDims, Jd 2 x 4 (untouched without a distortion), all row major.  Returns e2k's flag.
int orc_u_project(int model, int dist, const double* proj, const double* dk, const double* p, double* kp, double* Jp, double* Ji,
                  double* Jd) {
  Camera c;
  c.model = model;
  c.proj.assign(proj, proj + (model == PROJ_PINHOLE ? 4 : (model == PROJ_OMNI ? 5 : 6)));
  c.dist.kind = dist;
  for (int i = 0; i < 4; ++i) c.dist.k[i] = dist != DIST_NONE ? dk[i] : 0.0;
  Mat J;
  const bool ok = c.e2k(p, kp, &J);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) Jp[3 * i + j] = J(i, j);
  Mat ph = Mat::Vec({p[0], p[1], p[2], 1.0});
  Mat I = c.intrinsicsJacobian(ph);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < c.projDims(); ++j) Ji[c.projDims() * i + j] = I(i, j);
  if (dist != DIST_NONE) {
    Mat D = c.distortionJacobian(ph);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) Jd[4 * i + j] = D(i, j);
  }
  return ok ? 1 : 0;
}

// aslam::calibration::OptimizationProblem (incremental_calibration/src/core/OptimizationProblem.cpp): the design-variable
// enumeration Optimizer2::initialize walks.  addDesignVariable (:123-137) appends the variable to its group's vector and
// the group id to _groupsOrdering on first sight; designVariable(idx) (:215-221) resolves idx through getGroupId (:272-292):
// groups in first-insertion order, insertion order inside a group.  groupIds[i] = group of the i-th addDesignVariable call;
// order[idx] = insertion index of the variable the problem reports at position idx.
int orc_u_problem_design_variable_order(int n, const int* groupIds, int* order) {
  std::vector<int> groupsOrdering;
  std::map<int, std::vector<int>> designVariables;
  for (int i = 0; i < n; ++i) {
    if (!designVariables.count(groupIds[i])) groupsOrdering.push_back(groupIds[i]);
    designVariables[groupIds[i]].push_back(i);
  }
  for (int idx = 0; idx < n; ++idx) {
    size_t idxRunning = 0;
    int groupIdRunning = 0;
    for (int g : groupsOrdering) {
      const size_t groupSize = designVariables.at(g).size();
      if (idxRunning + groupSize > (size_t)idx) { groupIdRunning = g; break; }
      idxRunning += groupSize;
    }
    order[idx] = designVariables.at(groupIdRunning)[idx - idxRunning];
  }
  return 0;
}

// ---- the linear fixture of SampleDvAndError.hpp on the oracle's Problem
// Point2d / DummyDesignVariable<N>: an n-vector with additive update
int orc_add_vector_dv(void* h, int n, const double* v, int active) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  DesignVariable* d = new ParamVectorDv(v, n);
  d->active = active != 0;
  d->id = (int)P->dvs.size();
  P->ownedDvs.emplace_back(d);
  P->dvs.push_back(d);
  P->dvKind.push_back(DV_PROJECTION);
  return d->id;
  ORC_CATCH(P, -1)
}
// LinearErr{,2,3}: e = p - sum_i J_i v_i, Jacobians -J_i, setInvR(invR) (through computeMatrixSqrt).  Js: the J_i one after
// the other, each dim x dims(dv_i) row major.
int orc_add_linear_error(void* h, int dim, int nDvs, const int* dvIds, const double* Js, const double* p, const double* invR) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  Mat pm(dim, 1);
  for (int i = 0; i < dim; ++i) pm[i] = p[i];
  std::unique_ptr<LinearError> t(new LinearError(dim, pm));
  const double* at = Js;
  for (int k = 0; k < nDvs; ++k) {
    ParamVectorDv* d = P->dvAs<ParamVectorDv>(dvIds[k]);
    int n = (int)d->p.size();
    Mat J(dim, n);
    for (int i = 0; i < dim; ++i) for (int j = 0; j < n; ++j) J(i, j) = at[i * n + j];
    at += dim * n;
    t->dv.push_back(d);
    t->J.push_back(J);
  }
  Mat R(dim, dim);
  for (int i = 0; i < dim; ++i) for (int j = 0; j < dim; ++j) R(i, j) = invR[i * dim + j];
  t->sqrtInvR = computeMatrixSqrt(R);
  P->terms.emplace_back(t.release());
  return (int)P->terms.size() - 1;
  ORC_CATCH(P, -1)
}
// ErrorTermFs::setInvR on the n most recently added terms (general D x D covariances): invR n x D x D row major
int orc_set_last_terms_inv_r(void* h, int n, int dim, const double* invR) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  if (n > (int)P->terms.size()) throw std::runtime_error("more matrices than terms");
  for (int k = 0; k < n; ++k) {
    ErrorTerm* t = P->terms[P->terms.size() - n + k].get();
    if (t->dim != dim) throw std::runtime_error("The covariance matrix does not match the size of the error");
    Mat R(dim, dim);
    for (int i = 0; i < dim; ++i) for (int j = 0; j < dim; ++j) R(i, j) = invR[((size_t)k * dim + i) * dim + j];
    t->sqrtInvR = computeMatrixSqrt(R);
  }
  return 0;
  ORC_CATCH(P, -1)
}
// one term's buildHessian accumulated into the solver's H / rhs WITHOUT clearing (JacobianContainer.cpp test: "Do this
// twice! I should get double the values"), and the clear
int orc_clear_system(void* h) {
  Problem* P = (Problem*)h;
  P->opt.solver.H.clearValues();
  std::fill(P->opt.solver.rhs.begin(), P->opt.solver.rhs.end(), 0.0);
  return 0;
}
int orc_accumulate_term(void* h, int term, int useMest) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->terms[term]->buildHessian(P->opt.solver.H, P->opt.solver.rhs, useMest != 0);
  return 0;
  ORC_CATCH(P, -1)
}
// is block (r, c) allocated in H (SparseBlockMatrix::isBlockSet)
int orc_is_block_set(void* h, int r, int c) {
  Problem* P = (Problem*)h;
  return P->opt.solver.H.block(r, c, false) != nullptr ? 1 : 0;
}
// LinearSystemSolver::setConditioner + BlockCholeskyLinearSystemSolver::solveSystem
int orc_solve_system_with_conditioner(void* h, const double* diag, double* dx) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->opt.solver.diagCond.assign(diag, diag + P->opt.solver.JCols);
  std::vector<double> d;
  bool ok = P->opt.solver.solveSystem(d);
  if (ok) std::memcpy(dx, d.data(), sizeof(double) * d.size());
  return ok ? 0 : 1;
  ORC_CATCH(P, -1)
}

}  // extern "C"
