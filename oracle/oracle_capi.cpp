// TEST INFRASTRUCTURE ONLY (see omat.hpp).  C API over the CPU oracle, loaded with ctypes by
// tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  The call surface
// mirrors include/multical_hip.h so the same problem description can be fed to both; inside,
// every error term is built as the expression DAG the Python front-end would compose
// (sensors_and_targets.py:274-305, 556-630, 1133-1246) and walked node by node.
#include <cstring>
#include "oterms.hpp"

using namespace orc;

namespace {

enum { DV_EUCLIDEAN_POINT = 0, DV_ROTATION_QUATERNION = 1, DV_SCALAR = 2, DV_EUCLIDEAN_DIRECTION = 3,
       DV_SPLINE_COEFF = 4, DV_PROJECTION = 5, DV_DISTORTION = 6, DV_MATRIX_BASIC = 7, DV_HOMOGENEOUS_POINT = 8 };

struct Problem {
  std::vector<std::unique_ptr<DesignVariable>> ownedDvs;
  std::vector<DesignVariable*> dvs;      // registration (= problem) order, incl. spline coefficients
  std::vector<int> dvKind;
  std::vector<std::unique_ptr<SplineDv>> splines;
  std::vector<int> splineFirstDv;
  std::vector<std::unique_ptr<CameraDv>> cameras;
  std::vector<std::unique_ptr<ErrorTerm>> terms;
  Optimizer2 opt;
  int nextShareGroup = 0;
  bool finalized = false;
  std::string err;

  DesignVariable* dv(int id) { if (id < 0 || id >= (int)dvs.size()) throw std::runtime_error("bad dv id"); return dvs[id]; }
  template <class T> T* dvAs(int id) {
    T* p = dynamic_cast<T*>(dv(id));
    if (!p) throw std::runtime_error("dv kind mismatch");
    return p;
  }
  TraP basicT(int dq, int dt) {
    RotP r(new RotationQuaternionNode(dvAs<RotationQuaternion>(dq)));
    EucP t(new EuclideanPointNode(dvAs<EuclideanPoint>(dt)));
    return TraP(new TransformationBasicNode(r, t));
  }
  ScaP timeExpr(int dvToff, double t) {
    if (dvToff < 0) return ScaP(new ScalarConstantNode(t));
    ScaP d(new ScalarDvNode(dvAs<ScalarDv>(dvToff)));
    return ScaP(new ScalarAddNode(d, ScaP(new ScalarConstantNode(t))));
  }
};

#define ORC_TRY(P) try {
#define ORC_CATCH(P, rv) } catch (const std::exception& ex) { (P)->err = ex.what(); return rv; }

}  // namespace

extern "C" {

void* orc_create(void) { return new Problem(); }
void orc_destroy(void* h) { delete (Problem*)h; }
const char* orc_last_error(void* h) { return ((Problem*)h)->err.c_str(); }

int orc_add_dv(void* h, int kind, const double* params, int n, int active, double scaling) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  DesignVariable* d = nullptr;
  switch (kind) {
    case DV_EUCLIDEAN_POINT: d = new EuclideanPoint(params); break;
    case DV_ROTATION_QUATERNION: d = new RotationQuaternion(params); break;
    case DV_SCALAR: d = new ScalarDv(params[0]); break;
    case DV_EUCLIDEAN_DIRECTION: d = new EuclideanDirection(params); break;
    case DV_PROJECTION: case DV_DISTORTION: d = new ParamVectorDv(params, n); break;
    case DV_HOMOGENEOUS_POINT: d = new HomogeneousPointDv(params); break;
    default: throw std::runtime_error("unknown dv kind");
  }
  d->active = active != 0;
  d->scaling = scaling;
  d->id = (int)P->dvs.size();
  P->ownedDvs.emplace_back(d);
  P->dvs.push_back(d);
  P->dvKind.push_back(kind);
  return d->id;
  ORC_CATCH(P, -1)
}

int orc_add_matrix_dv(void* h, const double* A, const int* pattern, int active, double scaling) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  MatrixBasic* d = new MatrixBasic(A, pattern);
  d->active = active != 0 && d->minimalDimensions() > 0;
  d->scaling = scaling;
  d->id = (int)P->dvs.size();
  P->ownedDvs.emplace_back(d);
  P->dvs.push_back(d);
  P->dvKind.push_back(DV_MATRIX_BASIC);
  return d->id;
  ORC_CATCH(P, -1)
}

int orc_add_spline(void* h, int order, int dim, const double* knots, int nknots, const double* coeffs, int ncoef, int active) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  std::unique_ptr<SplineDv> s(new SplineDv());
  Mat c(dim, ncoef);
  std::memcpy(c.v.data(), coeffs, sizeof(double) * dim * ncoef);
  s->init(order, std::vector<double>(knots, knots + nknots), c);
  P->splineFirstDv.push_back((int)P->dvs.size());
  for (auto& d : s->dvs) {
    d->active = active != 0;
    d->id = (int)P->dvs.size();
    P->dvs.push_back(d.get());
    P->dvKind.push_back(DV_SPLINE_COEFF);
  }
  P->splines.push_back(std::move(s));
  return (int)P->splines.size() - 1;
  ORC_CATCH(P, -1)
}
int orc_spline_first_dv(void* h, int sid) { return ((Problem*)h)->splineFirstDv[sid]; }

int orc_add_camera(void* h, int model, int dist, int dvProj, int dvDist) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  std::unique_ptr<CameraDv> c(new CameraDv());
  c->cam.model = model;
  c->cam.dist.kind = dist;
  c->projDv = P->dvAs<ParamVectorDv>(dvProj);
  if ((int)c->projDv->p.size() != c->cam.projDims()) throw std::runtime_error("projection parameter count");
  c->distDv = dist == DIST_NONE ? nullptr : P->dvAs<ParamVectorDv>(dvDist);
  c->sync();
  P->cameras.push_back(std::move(c));
  return (int)P->cameras.size() - 1;
  ORC_CATCH(P, -1)
}

// F1: Camera.addCameraErrorTerms, sensors_and_targets.py:556-630
int orc_add_reprojection_batch(void* h, int cam, int spline, int dvQcb, int dvTcb, int dvToff, int nObs,
                               const double* obsTime, const int* obsDvQpw, const int* obsDvTpw, const int* obsFirst,
                               const double* y, const double* p, const double* sinv, double pad,
                               int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  TraP T_c_b = P->basicT(dvQcb, dvTcb);
  for (int o = 0; o < nObs; ++o) {
    ScaP frameTime = P->timeExpr(dvToff, obsTime[o]);
    TraP T_w_b(new TimeOffsetTransformationNode(P->splines[spline].get(), frameTime, pad, pad));
    TraP T_b_w(new TransformationInverseNode(T_w_b));
    TraP T_w_p(new TransformationInverseNode(P->basicT(obsDvQpw[o], obsDvTpw[o])));
    TraP T_c_p(new TransformationMultiplyNode(TraP(new TransformationMultiplyNode(T_c_b, T_b_w)), T_w_p));
    for (int k = obsFirst[o]; k < obsFirst[o + 1]; ++k) {
      Mat ph(4, 1);
      ph[0] = p[3 * k]; ph[1] = p[3 * k + 1]; ph[2] = p[3 * k + 2]; ph[3] = 1.0;
      HomP pt(new HomogeneousMultiplyNode(T_c_p, HomP(new HomogeneousConstantNode(ph))));
      std::unique_ptr<ReprojectionError> e(new ReprojectionError(&y[2 * k], sinv[k], pt, P->cameras[cam].get(), false));
      e->mest.kind = mestKind; e->mest.a = mestParam;
      e->shareGroup = P->nextShareGroup;
      P->terms.push_back(std::move(e));
    }
    ++P->nextShareGroup;
  }
  return 0;
  ORC_CATCH(P, -1)
}

// F2: CalibrationTargetOptimizationProblem.fromTargetViewObservations, CameraCalibrator.py:195-219
// p_cam = B_{n-1} * ... * B_0 * inverse(T_target_camera) * p ; ReprojectionError with camera dvs.
int orc_add_static_reprojection_batch(void* h, int cam, int nChain, const int* chainDvQ, const int* chainDvT, int nObs,
                                      const int* obsDvQtc, const int* obsDvTtc, const int* obsFirst,
                                      const double* y, const double* p, const double* sinv, int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  for (int o = 0; o < nObs; ++o) {
    TraP T(new TransformationInverseNode(P->basicT(obsDvQtc[o], obsDvTtc[o])));
    for (int b = 0; b < nChain; ++b) T = TraP(new TransformationMultiplyNode(P->basicT(chainDvQ[b], chainDvT[b]), T));
    for (int k = obsFirst[o]; k < obsFirst[o + 1]; ++k) {
      Mat ph(4, 1);
      ph[0] = p[3 * k]; ph[1] = p[3 * k + 1]; ph[2] = p[3 * k + 2]; ph[3] = 1.0;
      HomP pt(new HomogeneousMultiplyNode(T, HomP(new HomogeneousConstantNode(ph))));
      std::unique_ptr<ReprojectionError> e(new ReprojectionError(&y[2 * k], sinv[k], pt, P->cameras[cam].get(), true));
      e->mest.kind = mestKind; e->mest.a = mestParam;
      e->shareGroup = P->nextShareGroup;
      P->terms.push_back(std::move(e));
    }
    ++P->nextShareGroup;
  }
  return 0;
  ORC_CATCH(P, -1)
}

// estimateLandmarks: the point of the last n ReprojectionErrors becomes T * HomogeneousPointDv (CameraCalibrator.py:195-219)
int orc_set_last_terms_landmarks(void* h, int n, const int* dvLm) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  if (n > (int)P->terms.size()) throw std::runtime_error("more landmarks than terms");
  for (int k = 0; k < n; ++k) {
    if (dvLm[k] < 0) continue;
    ReprojectionError* e = dynamic_cast<ReprojectionError*>(P->terms[P->terms.size() - n + k].get());
    if (!e) throw std::runtime_error("landmarks follow reprojection errors");
    HomogeneousMultiplyNode* m = dynamic_cast<HomogeneousMultiplyNode*>(e->point.get());
    if (!m) throw std::runtime_error("the point is not T * p");
    m->r = HomP(new HomogeneousPointNode(P->dvAs<HomogeneousPointDv>(dvLm[k])));
  }
  return 0;
  ORC_CATCH(P, -1)
}

static EucP biasExpr(Problem* P, int biasSpline, int dvBias, double t) {
  if (biasSpline >= 0) return EucP(new SplineEuclidean3Node(P->splines[biasSpline].get(), t, 0));
  return EucP(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvBias)));
}

// F3: Imu.addAccelerometerErrorTerms, sensors_and_targets.py:1133-1179
int orc_add_accel_batch(void* h, int poseSpline, int biasSpline, int dvBias, int dvGravity, int dvQib, int dvRb, int n,
                        const double* t, const double* meas, const double* sinv, int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  SplineDv* ps = P->splines[poseSpline].get();
  EucP g_w;
  if (P->dvKind[dvGravity] == DV_EUCLIDEAN_DIRECTION) g_w.reset(new EuclideanDirectionNode(P->dvAs<EuclideanDirection>(dvGravity)));
  else g_w.reset(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvGravity)));
  RotP C_i_b(new RotationQuaternionNode(P->dvAs<RotationQuaternion>(dvQib)));
  EucP r_b(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvRb)));
  for (int k = 0; k < n; ++k) {
    double tk = t[k];
    RotP C_b_w(new RotationInverseNode(RotP(new SplineRotationNode(ps, tk))));
    EucP a_w(new SplineEuclideanDerivNode(ps, tk, 2));
    EucP b_i = biasExpr(P, biasSpline, dvBias, tk);
    EucP w_b(new SplineAngularRateBodyNode(ps, tk, 1));
    EucP w_dot_b(new SplineAngularRateBodyNode(ps, tk, 2));
    // a = C_i_b * (C_b_w * (a_w - g_w) + w_dot_b.cross(r_b) + w_b.cross(w_b.cross(r_b)))
    EucP t1(new EuclideanRotateNode(C_b_w, EucP(new EuclideanSubtractNode(a_w, g_w))));
    EucP t2(new EuclideanCrossNode(w_dot_b, r_b));
    EucP t3(new EuclideanCrossNode(w_b, EucP(new EuclideanCrossNode(w_b, r_b))));
    EucP inner(new EuclideanAddNode(EucP(new EuclideanAddNode(t1, t2)), t3));
    EucP a(new EuclideanRotateNode(C_i_b, inner));
    EucP pred(new EuclideanAddNode(a, b_i));
    std::unique_ptr<EuclideanError> e(new EuclideanError(&meas[3 * k], sinv[k], pred));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}

// F4: Imu.addGyroscopeErrorTerms, sensors_and_targets.py:1181-1221
int orc_add_gyro_batch(void* h, int poseSpline, int biasSpline, int dvBias, int dvQib, int n,
                       const double* t, const double* meas, const double* sinv, int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  SplineDv* ps = P->splines[poseSpline].get();
  RotP C_i_b(new RotationQuaternionNode(P->dvAs<RotationQuaternion>(dvQib)));
  for (int k = 0; k < n; ++k) {
    double tk = t[k];
    EucP w_b(new SplineAngularRateBodyNode(ps, tk, 1));
    EucP b_i = biasExpr(P, biasSpline, dvBias, tk);
    EucP w(new EuclideanRotateNode(C_i_b, w_b));
    EucP pred(new EuclideanAddNode(w, b_i));
    std::unique_ptr<EuclideanError> e(new EuclideanError(&meas[3 * k], sinv[k], pred));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}

// F6: ket.GyroscopeError(omega_measured, invR, R_i_c * EuclideanExpression(const), bias) in
// findExtrinsicPriorSensorsToCamera (sensors_and_targets.py:858-879); GyroscopeError = EuclideanError on
// angularVelocity + bias (KAL/src/GyroscopeError.cpp:5-17)
int orc_add_gyroscope_error_batch(void* h, int dvQ, int dvBias, int n, const double* wPred, const double* meas, const double* sinv,
                                  int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  RotP R_i_c(new RotationQuaternionNode(P->dvAs<RotationQuaternion>(dvQ)));
  EucP bias(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvBias)));
  for (int k = 0; k < n; ++k) {
    EucP wc(new EuclideanConstantNode(Mat::Vec({wPred[3 * k], wPred[3 * k + 1], wPred[3 * k + 2]})));
    EucP w(new EuclideanRotateNode(R_i_c, wc));
    EucP pred(new EuclideanAddNode(w, bias));
    std::unique_ptr<EuclideanError> e(new EuclideanError(&meas[3 * k], sinv[k], pred));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}
// F6: aopt.ErrorTermTransformation(T_a [* T_b.inverse()], prior, weightRotation, weightTranslation)
// (sensors_and_targets.py:812-817, 886-919); the prior's inverse as sm::kinematics::Transformation::inverse() forms it
// (conjugate quaternion, rotated and negated translation)
int orc_add_transformation_error_batch(void* h, int dvQa, int dvTa, int dvQb, int dvTb, int n, const double* priorQ,
                                       const double* priorT, double weightRotation, double weightTranslation, int mestKind,
                                       double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  TraP T = P->basicT(dvQa, dvTa);
  if (dvQb >= 0) T = TraP(new TransformationMultiplyNode(T, TraP(new TransformationInverseNode(P->basicT(dvQb, dvTb)))));
  for (int k = 0; k < n; ++k) {
    Mat qi = Mat::Vec({-priorQ[4 * k], -priorQ[4 * k + 1], -priorQ[4 * k + 2], priorQ[4 * k + 3]});
    Mat Ci = quat2r(qi);
    Mat t = Mat::Vec({priorT[3 * k], priorT[3 * k + 1], priorT[3 * k + 2]});
    Mat Pi = Mat::Identity(4);
    Pi.setBlock(0, 0, Ci);
    Pi.setBlock(0, 3, (Ci * t) * -1.0);
    std::unique_ptr<ErrorTermTransformation> e(new ErrorTermTransformation(T, Pi, weightRotation, weightTranslation));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}

static EucP gravityExpr(Problem* P, int dvGravity) {
  if (P->dvKind[dvGravity] == DV_EUCLIDEAN_DIRECTION) return EucP(new EuclideanDirectionNode(P->dvAs<EuclideanDirection>(dvGravity)));
  return EucP(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvGravity)));
}
static EucP leverTerms(EucP w_dot_b, EucP w_b, EucP r) {   // w_dot_b.cross(r) + w_b.cross(w_b.cross(r))
  EucP t2(new EuclideanCrossNode(w_dot_b, r));
  EucP t3(new EuclideanCrossNode(w_b, EucP(new EuclideanCrossNode(w_b, r))));
  return EucP(new EuclideanAddNode(t2, t3));
}

// F3, ScaledMisalignedImu.addAccelerometerErrorTerms (sensors_and_targets.py:1449-1497) and, with sizeEffect != NULL
// (rx_i, ry_i, rz_i, Ix, Iy, Iz), ScaledMisalignedSizeEffectImu.addAccelerometerErrorTerms (:1608-1661)
int orc_add_accel_sm_batch(void* h, int poseSpline, int biasSpline, int dvBias, int dvGravity, int dvQib, int dvRb, int dvM,
                           const int* sizeEffect, int n, const double* t, const double* meas, const double* sinv, int mestKind,
                           double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  SplineDv* ps = P->splines[poseSpline].get();
  EucP g_w = gravityExpr(P, dvGravity);
  RotP C_i_b(new RotationQuaternionNode(P->dvAs<RotationQuaternion>(dvQib)));
  EucP r_b(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvRb)));
  MatrixBasic* M = P->dvAs<MatrixBasic>(dvM);
  for (int k = 0; k < n; ++k) {
    double tk = t[k];
    RotP C_b_w(new RotationInverseNode(RotP(new SplineRotationNode(ps, tk))));
    EucP a_w(new SplineEuclideanDerivNode(ps, tk, 2));
    EucP b_i = biasExpr(P, biasSpline, dvBias, tk);
    EucP w_b(new SplineAngularRateBodyNode(ps, tk, 1));
    EucP w_dot_b(new SplineAngularRateBodyNode(ps, tk, 2));
    EucP t1(new EuclideanRotateNode(C_b_w, EucP(new EuclideanSubtractNode(a_w, g_w))));
    EucP a;
    if (!sizeEffect) {
      // a = M * (C_i_b * (C_b_w * (a_w - g_w) + w_dot_b.cross(r_b) + w_b.cross(w_b.cross(r_b))))
      EucP t2(new EuclideanCrossNode(w_dot_b, r_b));
      EucP t3(new EuclideanCrossNode(w_b, EucP(new EuclideanCrossNode(w_b, r_b))));
      EucP inner(new EuclideanAddNode(EucP(new EuclideanAddNode(t1, t2)), t3));
      a.reset(new EuclideanMatrixMultiplyNode(M, EucP(new EuclideanRotateNode(C_i_b, inner))));
    } else {
      // rx_b = r_b + C_i_b.inverse() * rx_i ;  a = M * (C_i_b * t1 + Ix * (C_i_b * (...rx_b...)) + Iy * (...) + Iz * (...))
      EucP sum(new EuclideanRotateNode(C_i_b, t1));
      for (int j = 0; j < 3; ++j) {
        EucP rj_i(new EuclideanPointNode(P->dvAs<EuclideanPoint>(sizeEffect[j])));
        EucP rj_b(new EuclideanAddNode(r_b, EucP(new EuclideanRotateNode(RotP(new RotationInverseNode(C_i_b)), rj_i))));
        EucP sj(new EuclideanRotateNode(C_i_b, leverTerms(w_dot_b, w_b, rj_b)));
        sum.reset(new EuclideanAddNode(sum, EucP(new EuclideanMatrixMultiplyNode(P->dvAs<MatrixBasic>(sizeEffect[3 + j]), sj))));
      }
      a.reset(new EuclideanMatrixMultiplyNode(M, sum));
    }
    EucP pred(new EuclideanAddNode(a, b_i));
    std::unique_ptr<EuclideanError> e(new EuclideanError(&meas[3 * k], sinv[k], pred));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}

// F4, ScaledMisalignedImu.addGyroscopeErrorTerms (sensors_and_targets.py:1499-1550)
int orc_add_gyro_sm_batch(void* h, int poseSpline, int biasSpline, int dvBias, int dvQib, int dvQgi, int dvM, int dvMa, int dvGravity,
                          int dvRb, int n, const double* t, const double* meas, const double* sinv, int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  SplineDv* ps = P->splines[poseSpline].get();
  EucP g_w = gravityExpr(P, dvGravity);
  RotP C_i_b(new RotationQuaternionNode(P->dvAs<RotationQuaternion>(dvQib)));
  RotP C_gyro_i(new RotationQuaternionNode(P->dvAs<RotationQuaternion>(dvQgi)));
  EucP r_b(new EuclideanPointNode(P->dvAs<EuclideanPoint>(dvRb)));
  MatrixBasic* M = P->dvAs<MatrixBasic>(dvM);
  MatrixBasic* Ma = P->dvAs<MatrixBasic>(dvMa);
  for (int k = 0; k < n; ++k) {
    double tk = t[k];
    EucP w_b(new SplineAngularRateBodyNode(ps, tk, 1));
    EucP w_dot_b(new SplineAngularRateBodyNode(ps, tk, 2));
    EucP b_i = biasExpr(P, biasSpline, dvBias, tk);
    RotP C_b_w(new RotationInverseNode(RotP(new SplineRotationNode(ps, tk))));
    EucP a_w(new SplineEuclideanDerivNode(ps, tk, 2));
    // a_b = C_b_w * (a_w - g_w) + w_dot_b.cross(r_b) + w_b.cross(w_b.cross(r_b))
    EucP t1(new EuclideanRotateNode(C_b_w, EucP(new EuclideanSubtractNode(a_w, g_w))));
    EucP t2(new EuclideanCrossNode(w_dot_b, r_b));
    EucP t3(new EuclideanCrossNode(w_b, EucP(new EuclideanCrossNode(w_b, r_b))));
    EucP a_b(new EuclideanAddNode(EucP(new EuclideanAddNode(t1, t2)), t3));
    RotP C_gyro_b(new RotationMultiplyNode(C_gyro_i, C_i_b));
    // w = M * (C_gyro_b * w_b) + Ma * (C_gyro_b * a_b)
    EucP w(new EuclideanAddNode(EucP(new EuclideanMatrixMultiplyNode(M, EucP(new EuclideanRotateNode(C_gyro_b, w_b)))),
                                EucP(new EuclideanMatrixMultiplyNode(Ma, EucP(new EuclideanRotateNode(C_gyro_b, a_b))))));
    EucP pred(new EuclideanAddNode(w, b_i));
    std::unique_ptr<EuclideanError> e(new EuclideanError(&meas[3 * k], sinv[k], pred));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}

// F5: LiDAR.addLiDARErrorTerms, sensors_and_targets.py:274-305
int orc_add_lidar_batch(void* h, int spline, int dvToff, int dvQlb, int dvTlb, int n, const double* t, const double* dir,
                        const double* range, const double* sinv, const int* dvQpw, const int* dvTpw, double pad,
                        int mestKind, double mestParam) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  Mat nrm = Mat::Vec({0.0, 0.0, 1.0});
  EucP plane_normal(new EuclideanConstantNode(nrm));
  TraP T_l_b = P->basicT(dvQlb, dvTlb);
  for (int k = 0; k < n; ++k) {
    ScaP tk = P->timeExpr(dvToff, t[k]);
    TraP T_w_b(new TimeOffsetTransformationNode(P->splines[spline].get(), tk, pad, pad));
    TraP T_b_l(new TransformationInverseNode(T_l_b));
    TraP T_w_l(new TransformationMultiplyNode(T_w_b, T_b_l));
    TraP T_p_l(new TransformationMultiplyNode(P->basicT(dvQpw[k], dvTpw[k]), T_w_l));
    RotP C_p_l(new RotationFromTransformationNode(T_p_l));
    EucP t_p(new EuclideanTranslationNode(T_p_l));
    Mat d3 = Mat::Vec({dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]});
    ScaP d(new ScalarDotNode(plane_normal, t_p));
    ScaP theta(new ScalarDotNode(plane_normal, EucP(new EuclideanRotateConstantNode(C_p_l, d3))));
    ScaP pred(new ScalarMultiplyNode(ScaP(new ScalarDivideNode(d, theta)), ScaP(new ScalarConstantNode(-1.0))));
    std::unique_ptr<ScalarError> e(new ScalarError(range[k], sinv[k], pred));
    e->mest.kind = mestKind; e->mest.a = mestParam;
    P->terms.push_back(std::move(e));
  }
  return 0;
  ORC_CATCH(P, -1)
}

// LiDAR.findPointsOnTarget, sensors_and_targets.py:199-235: transformMeasurementsToWorldFrame (per reading,
// T_w_b(t + offset) from the spline, general inverse of T_l_b) then _onPlane per target.  mask bit p: on target p;
// bit 31: time strictly inside the spline.  margin[i] (optional): smallest distance of any tested coordinate to a box face.
int orc_lidar_find_points_on_targets(void* h, int spline, int dvToff, int dvQlb, int dvTlb, int nPlanes, const int* dvQpw,
                                     const int* dvTpw, const double* planeMin, const double* planeMax, double threshold, int n,
                                     const double* pts, unsigned* mask, double* world, double* margin) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  const BSplinePose& sp = P->splines[spline]->spline;
  const double off = dvToff >= 0 ? P->dvAs<ScalarDv>(dvToff)->s : 0.0;
  Mat T_b_l = inverse(P->basicT(dvQlb, dvTlb)->toT());
  std::vector<Mat> T_p_w;
  for (int k = 0; k < nPlanes; ++k) T_p_w.push_back(P->basicT(dvQpw[k], dvTpw[k])->toT());
  for (int i = 0; i < n; ++i) {
    mask[i] = 0;
    if (margin) margin[i] = 1e300;
    double tk = pts[4 * i + 3] + off;
    if (!(tk > sp.t_min() && tk < sp.t_max())) { if (world) world[3 * i] = world[3 * i + 1] = world[3 * i + 2] = 0.0; continue; }
    mask[i] = 0x80000000u;
    Mat p(4, 1);
    p[0] = pts[4 * i]; p[1] = pts[4 * i + 1]; p[2] = pts[4 * i + 2]; p[3] = 1.0;
    Mat pw = sp.transformation(tk) * (T_b_l * p);
    if (world) { world[3 * i] = pw[0]; world[3 * i + 1] = pw[1]; world[3 * i + 2] = pw[2]; }
    for (int k = 0; k < nPlanes; ++k) {
      Mat q = T_p_w[k] * pw;
      bool in = true;
      for (int a = 0; a < 3; ++a) {
        double lo = planeMin[3 * k + a] - (a == 2 ? threshold : 0.0), hi = planeMax[3 * k + a] + (a == 2 ? threshold : 0.0);
        in = in && q[a] > lo && q[a] < hi;
        if (margin) margin[i] = std::fmin(margin[i], std::fmin(std::fabs(q[a] - lo), std::fabs(q[a] - hi)));
      }
      if (in) mask[i] |= 1u << k;
    }
  }
  return 0;
  ORC_CATCH(P, -1)
}

// F7: asp.BSplineEuclideanMotionError / addMotionErrorTerms
int orc_add_motion_term(void* h, int spline, const double* W, int order) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  SplineDv* s = P->splines[spline].get();
  int D = s->spline.dim();
  Mat Wm(D, D);
  for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) Wm(i, j) = W[i * D + j];
  P->terms.emplace_back(new BSplineMotionError(s, Wm, order));
  return 0;
  ORC_CATCH(P, -1)
}

int orc_finalize(void* h) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->opt.problemDvs = P->dvs;
  P->opt.problemTerms.clear();
  for (auto& t : P->terms) P->opt.problemTerms.push_back(t.get());
  P->opt.initialize();
  P->finalized = true;
  return 0;
  ORC_CATCH(P, -1)
}
int orc_dims(void* h, int* JRows, int* JCols, int* nBlocks, int* nTerms) {
  Problem* P = (Problem*)h;
  *JRows = P->opt.solver.JRows; *JCols = P->opt.solver.JCols;
  *nBlocks = (int)P->opt.dvs.size(); *nTerms = (int)P->terms.size();
  return 0;
}
int orc_dv_block(void* h, int dv, int* blockIndex, int* columnBase) {
  Problem* P = (Problem*)h;
  *blockIndex = P->dvs[dv]->blockIndex; *columnBase = P->dvs[dv]->active ? P->dvs[dv]->columnBase : -1;
  return 0;
}
double orc_evaluate_error(void* h, int nThreads, int useMest, double* eOut) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  for (auto& c : P->cameras) c->sync();   // serially, before any thread reads the camera parameters
  double J = P->opt.solver.evaluateError((size_t)nThreads, useMest != 0);
  if (eOut) std::memcpy(eOut, P->opt.solver.e.data(), sizeof(double) * P->opt.solver.JRows);
  return J;
  ORC_CATCH(P, NAN)
}
int orc_term_chi2(void* h, double* out) {
  Problem* P = (Problem*)h;
  for (size_t i = 0; i < P->terms.size(); ++i) out[i] = P->terms[i]->squaredError;
  return 0;
}
int orc_build_system(void* h, int useMest) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->opt.solver.buildSystem(useMest != 0);
  return 0;
  ORC_CATCH(P, -1)
}
// The reference's first solveSystem allocates every diagonal block (BlockCholeskyLinearSystemSolver.cpp:80-81)
// before fillCCS runs, so the exported pattern contains them all.
int orc_ccs_nnz(void* h) {
  Problem* P = (Problem*)h;
  for (int i = 0; i < P->opt.solver.H.bRows(); ++i) P->opt.solver.H.block(i, i, true);
  std::vector<int> Cp, Ci; std::vector<double> Cx;
  return P->opt.solver.H.fillCCS(Cp, Ci, Cx, true);
}
int orc_get_ccs(void* h, int* Cp, int* Ci, double* Cx, double* rhs) {
  Problem* P = (Problem*)h;
  for (int i = 0; i < P->opt.solver.H.bRows(); ++i) P->opt.solver.H.block(i, i, true);
  std::vector<int> cp, ci; std::vector<double> cx;
  P->opt.solver.H.fillCCS(cp, ci, cx, true);
  if (Cp) std::memcpy(Cp, cp.data(), sizeof(int) * cp.size());
  if (Ci) std::memcpy(Ci, ci.data(), sizeof(int) * ci.size());
  if (Cx) std::memcpy(Cx, cx.data(), sizeof(double) * cx.size());
  if (rhs) std::memcpy(rhs, P->opt.solver.rhs.data(), sizeof(double) * P->opt.solver.JCols);
  return (int)cx.size();
}
// SparseCholeskyLinearSystemSolver's J^T in compressed-column form
// (AB/include/aslam/backend/implementation/CompressedColumnJacobianTransposeBuilder.hpp:20-56, 94-104;
//  CompressedColumnMatrix.hpp:236-304): one CSC column per residual row; its entries are the ACTIVE design variables of
// the term's design-variable set sorted by block index (structural zeros of padded spline coefficients included);
// values = getWeightedJacobians written in JacobianContainer (= block index) order.  Pass NULL arrays to get the count.
int orc_build_jacobian_transpose(void* h, int useMest, int* col_ptr, int* row_ind, double* values) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  size_t nnz = 0;
  int col = 0;
  if (col_ptr) col_ptr[0] = 0;
  for (auto& tp : P->terms) {
    ErrorTerm* t = tp.get();
    std::set<DesignVariable*> ds;
    t->getDvs(ds);
    std::vector<DesignVariable*> act;
    for (DesignVariable* d : ds) if (d->active) act.push_back(d);
    std::sort(act.begin(), act.end(), [](DesignVariable* a, DesignVariable* b) { return a->blockIndex < b->blockIndex; });
    int epc = 0;
    for (DesignVariable* d : act) epc += d->minimalDimensions();
    if (row_ind)
      for (int r = 0; r < t->dim; ++r) {
        int k = 0;
        for (DesignVariable* d : act) for (int c = 0; c < d->minimalDimensions(); ++c) row_ind[nnz + (size_t)r * epc + k++] = d->columnBase + c;
      }
    if (values) {
      JacobianContainer jc(t->dim);
      t->getWeightedJacobians(jc, useMest != 0);
      int rowOffset = 0;
      for (auto& kv : jc.jm) {
        for (int c = 0; c < kv.second.r; ++c)
          for (int r = 0; r < kv.second.c; ++r) values[nnz + (size_t)c * epc + rowOffset + r] = kv.second(c, r);
        rowOffset += kv.second.c;
      }
      if (rowOffset != epc) throw std::runtime_error("Jacobian container does not cover the design-variable set of the term");
    }
    if (col_ptr) for (int r = 0; r < t->dim; ++r) { col_ptr[col + 1] = col_ptr[col] + epc; ++col; }
    nnz += (size_t)epc * t->dim;
  }
  return (int)nnz;
  ORC_CATCH(P, -1)
}
// dense weighted Jacobian rows + weighted error of one term (getWeightedJacobians / getWeightedError)
int orc_term_weighted_jacobian(void* h, int term, int useMest, double* Jdense /* dim x JCols, row-major */, double* e) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  ErrorTerm* t = P->terms[term].get();
  JacobianContainer jc(t->dim);
  t->getWeightedJacobians(jc, useMest != 0);
  int JCols = P->opt.solver.JCols;
  std::memset(Jdense, 0, sizeof(double) * t->dim * JCols);
  for (auto& kv : jc.jm)
    for (int r = 0; r < kv.second.r; ++r)
      for (int c = 0; c < kv.second.c; ++c) Jdense[(size_t)r * JCols + kv.first->columnBase + c] = kv.second(r, c);
  Mat we;
  t->getWeightedError(we, useMest != 0);
  for (int r = 0; r < t->dim; ++r) e[r] = we[r];
  return t->dim;
  ORC_CATCH(P, -1)
}
int orc_term_info(void* h, int term, int* dim, int* rowBase) {
  Problem* P = (Problem*)h;
  *dim = P->terms[term]->dim; *rowBase = P->terms[term]->rowBase;
  return 0;
}
int orc_update(void* h, const double* dx) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->opt.dx.assign(dx, dx + P->opt.solver.JCols);
  P->opt.applyStateUpdate();
  return 0;
  ORC_CATCH(P, -1)
}
int orc_revert(void* h) { ((Problem*)h)->opt.revertLastStateUpdate(); return 0; }
int orc_get_dv_params(void* h, int dv, double* out) {
  std::vector<double> p;
  ((Problem*)h)->dvs[dv]->getParameters(p);
  std::memcpy(out, p.data(), sizeof(double) * p.size());
  return (int)p.size();
}
int orc_solve_system(void* h, double lambda, double* dx) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->opt.solver.setConstantConditioner(lambda);
  std::vector<double> d;
  bool ok = P->opt.solver.solveSystem(d);
  if (ok) std::memcpy(dx, d.data(), sizeof(double) * d.size());
  return ok ? 0 : 1;
  ORC_CATCH(P, -1)
}
// out[0..6] = JStart JFinal iterations failedIterations dXFinal dJFinal linearSolverFailure
int orc_optimize(void* h, int maxIterations, double lambdaInit, double convergenceDeltaX, double convergenceJDescentRatio,
                 int nThreads, double* out) {
  Problem* P = (Problem*)h;
  ORC_TRY(P)
  P->opt.opt.maxIterations = maxIterations;
  P->opt.opt.lambdaInit = lambdaInit;
  P->opt.opt.convergenceDeltaX = convergenceDeltaX;
  P->opt.opt.convergenceJDescentRatioThreshold = convergenceJDescentRatio;
  P->opt.opt.nThreads = nThreads;
  SolutionReturnValue s = P->opt.optimize();
  out[0] = s.JStart; out[1] = s.JFinal; out[2] = s.iterations; out[3] = s.failedIterations;
  out[4] = s.dXFinal; out[5] = s.dJFinal; out[6] = s.linearSolverFailure ? 1 : 0;
  return 0;
  ORC_CATCH(P, -1)
}

}  // extern "C"

#include "oracle_unit_capi.hpp"
