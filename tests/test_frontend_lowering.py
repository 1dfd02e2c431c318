"""The lowering pass (multical_amd/frontend.py): the small Multical problem is built a second time the way the reference's
front-end builds it -- one expression tree and one error-term object per measurement, in the statement order of
kalibr_sensor_calibration/sensors_and_targets.py:274-305, 556-630, 1133-1221 and calibrator.py:94-123 -- then lowered
to typed batches.  The lowered problem must be THE SAME problem as the one synth.build_spec writes down directly:
identical dimensions, block indices and sparsity pattern, and the same J, e, H, b (oracle on both)."""
import numpy as np
import pytest

from multical_amd import frontend as F
from multical_amd import spec as S
from multical_amd import synth


def build_like_the_reference(W, state, meas, imu_model="calibrated", extras=None):
    """imu_model: "calibrated" (sens.Imu), "scale_misalignment" (sens.ScaledMisalignedImu) or "size_effect"
    (sens.ScaledMisalignedSizeEffectImu); extras[k]: values of IMU k's additional design variables (the reference starts
    them at identity / zero, sensors_and_targets.py:1431-1447, 1586-1606)."""
    cfg = W["cfg"]
    problem = F.OptimizationProblem()
    # --- calibrator.initDesignVariables
    poseSpline = F.BSplinePose(synth.SPLINE_ORDER, W["pose_knots"], state["pose_coeffs"])
    poseDv = F.BSplinePoseDesignVariable(poseSpline)
    F.addSplineDesignVariables(problem, poseDv)
    gravityDv = F.EuclideanDirection(state["gravity"])
    gravityDv.setActive(True)
    problem.addDesignVariable(gravityDv, F.HELPER_GROUP_ID)
    g_w = gravityDv.toExpression()
    targets = []
    for b, (q, t) in enumerate(state["boards"]):
        dv = F.TransformationDv(F.Transformation(q, t), rotationActive=b != 0, translationActive=b != 0)
        for i in range(dv.numDesignVariables()):
            problem.addDesignVariable(dv.getDesignVariable(i), F.HELPER_GROUP_ID)
        targets.append(dv)
    imus = []
    for k, imu in enumerate(W["imus"]):
        si = state["imus"][k]
        gyroBiasDv = F.EuclideanBSplineDesignVariable(F.BSpline(synth.SPLINE_ORDER, imu["bias_knots"], si["gyro_bias"]))
        accelBiasDv = F.EuclideanBSplineDesignVariable(F.BSpline(synth.SPLINE_ORDER, imu["bias_knots"], si["accel_bias"]))
        F.addSplineDesignVariables(problem, gyroBiasDv, setActive=True, group_id=F.HELPER_GROUP_ID)
        F.addSplineDesignVariables(problem, accelBiasDv, setActive=True, group_id=F.HELPER_GROUP_ID)
        q_i_b = F.RotationQuaternionDv(si["q"])
        problem.addDesignVariable(q_i_b, F.HELPER_GROUP_ID)
        r_b = F.EuclideanPointDv(si["r"])
        problem.addDesignVariable(r_b, F.HELPER_GROUP_ID)
        q_i_b.setActive(not imu["reference"])
        r_b.setActive(not imu["reference"])
        sm = None
        if imu_model != "calibrated":            # ScaledMisalignedImu.addDesignVariables
            ex = extras[k]
            lower = np.array([[1, 0, 0], [1, 1, 0], [1, 1, 1]], dtype=int)
            sm = dict(q_gyro_i=F.RotationQuaternionDv(ex["q_gi"]), M_accel=F.MatrixBasicDv(ex["M_a"], lower),
                      M_gyro=F.MatrixBasicDv(ex["M_g"], lower), M_accel_gyro=F.MatrixBasicDv(ex["M_ag"], np.ones((3, 3), dtype=int)))
            for name in ("q_gyro_i", "M_accel", "M_gyro", "M_accel_gyro"):
                problem.addDesignVariable(sm[name], F.HELPER_GROUP_ID)
                sm[name].setActive(True)
            if imu_model == "size_effect":       # ScaledMisalignedSizeEffectImu.addDesignVariables
                sm["arms"] = [F.EuclideanPointDv(v) for v in ex["arms"]]
                for j, dv in enumerate(sm["arms"]):
                    problem.addDesignVariable(dv, F.HELPER_GROUP_ID)
                    dv.setActive(j > 0)          # rx_i stays fixed
                sm["sel"] = [F.MatrixBasicDv(np.diag(np.eye(3)[j]), np.zeros((3, 3), dtype=int)) for j in range(3)]
                for dv in sm["sel"]:
                    problem.addDesignVariable(dv, F.HELPER_GROUP_ID)
                    dv.setActive(False)
        imus.append((gyroBiasDv, accelBiasDv, q_i_b, r_b, sm))
    cams = []
    for k, cam in enumerate(W["cams"]):
        sc = state["cams"][k]
        T_c_b = F.TransformationDv(F.Transformation(sc["q"], sc["t"]))
        for i in range(T_c_b.numDesignVariables()):
            problem.addDesignVariable(T_c_b.getDesignVariable(i), F.HELPER_GROUP_ID)
        toff = F.Scalar(sc["toff"])
        toff.setActive(cfg.time_calibration)
        problem.addDesignVariable(toff, F.CALIBRATION_GROUP_ID)
        cams.append((T_c_b, toff, F.CameraGeometry(cam["model"], cam["dist"], sc["proj"], sc["dk"])))
    lidars = []
    for k, lid in enumerate(W["lidars"]):
        sl = state["lidars"][k]
        off = F.Scalar(sl["toff"])
        off.setActive(cfg.time_calibration)
        problem.addDesignVariable(off, F.HELPER_GROUP_ID)
        T_l_b = F.TransformationDv(F.Transformation(sl["q"], sl["t"]))
        for i in range(T_l_b.numDesignVariables()):
            problem.addDesignVariable(T_l_b.getDesignVariable(i), F.CALIBRATION_GROUP_ID)
        lidars.append((off, T_l_b))
    # --- Camera.addCameraErrorTerms
    pts = W["board_points"]
    for k, cam in enumerate(W["cams"]):
        T_c_b_dv, toff, geom = cams[k]
        T_c_b = T_c_b_dv.toExpression()
        first = cam["obs_first"]
        for o in range(len(cam["obs_time"])):
            frameTime = toff.toExpression() + float(cam["obs_time"][o]) + 0.0
            assert poseSpline.t_min() < frameTime.toScalar() < poseSpline.t_max()
            T_w_b = poseDv.transformationAtTime(frameTime, cfg.time_padding, cfg.time_padding)
            T_b_w = T_w_b.inverse()
            T_w_p = targets[cam["obs_board"][o]].toExpression().inverse()
            T_c_p = T_c_b * T_b_w * T_w_p
            frame = F.Frame()
            frame.setGeometry(geom)
            invR = np.linalg.inv(np.eye(2) * cfg.reproj_sigma * cfg.reproj_sigma)
            ys = meas["cams"][k][first[o]:first[o + 1]]
            for y in ys:
                kp = F.Keypoint()
                kp.setMeasurement(y)
                kp.setInverseMeasurementCovariance(invR)
                frame.addKeypoint(kp)
            for pidx, ci in enumerate(cam["corner_idx"][o]):
                p = T_c_p * F.HomogeneousExpression(np.insert(pts[ci], 3, 1))
                rerr = F.SimpleReprojectionError(frame, pidx, p)
                if cfg.blake_zisserman_cam > 0:
                    rerr.setMEstimatorPolicy(F.BlakeZissermanMEstimator(cfg.blake_zisserman_cam))
                problem.addErrorTerm(rerr)
    # --- Imu.addAccelerometerErrorTerms / addGyroscopeErrorTerms / addBiasMotionTerms
    for k, imu in enumerate(W["imus"]):
        gyroBiasDv, accelBiasDv, q_i_b, r_b_dv, sm = imus[k]
        mest_a = F.HuberMEstimator(cfg.huber_accel) if cfg.huber_accel > 0 else F.NoMEstimator()
        mest_g = F.HuberMEstimator(cfg.huber_gyro) if cfg.huber_gyro > 0 else F.NoMEstimator()
        for j, tk in enumerate(imu["t"]):
            tk = float(tk)
            C_b_w = poseDv.orientation(tk).inverse()
            a_w = poseDv.linearAcceleration(tk)
            b_i = accelBiasDv.toEuclideanExpression(tk, 0)
            w_b = poseDv.angularVelocityBodyFrame(tk)
            w_dot_b = poseDv.angularAccelerationBodyFrame(tk)
            C_i_b = q_i_b.toExpression()
            r_b = r_b_dv.toExpression()
            if imu_model == "calibrated":
                a = C_i_b * (C_b_w * (a_w - g_w) + w_dot_b.cross(r_b) + w_b.cross(w_b.cross(r_b)))
            elif imu_model == "scale_misalignment":
                M = sm["M_accel"].toExpression()
                a = M * (C_i_b * (C_b_w * (a_w - g_w) + w_dot_b.cross(r_b) + w_b.cross(w_b.cross(r_b))))
            else:
                M = sm["M_accel"].toExpression()
                rx_b = r_b_dv.toExpression() + C_i_b.inverse() * sm["arms"][0].toExpression()
                ry_b = r_b_dv.toExpression() + C_i_b.inverse() * sm["arms"][1].toExpression()
                rz_b = r_b_dv.toExpression() + C_i_b.inverse() * sm["arms"][2].toExpression()
                Ix, Iy, Iz = [dv.toExpression() for dv in sm["sel"]]
                a = M * (C_i_b * (C_b_w * (a_w - g_w)) +
                         Ix * (C_i_b * (w_dot_b.cross(rx_b) + w_b.cross(w_b.cross(rx_b)))) +
                         Iy * (C_i_b * (w_dot_b.cross(ry_b) + w_b.cross(w_b.cross(ry_b)))) +
                         Iz * (C_i_b * (w_dot_b.cross(rz_b) + w_b.cross(w_b.cross(rz_b)))))
            aerr = F.EuclideanError(meas["accel"][k][j], np.eye(3) * (1.0 / cfg.accel_sigma) ** 2, a + b_i)
            aerr.setMEstimatorPolicy(mest_a)
            problem.addErrorTerm(aerr)
        for j, tk in enumerate(imu["t"]):
            tk = float(tk)
            w_b = poseDv.angularVelocityBodyFrame(tk)
            b_i = gyroBiasDv.toEuclideanExpression(tk, 0)
            if imu_model == "calibrated":
                w = q_i_b.toExpression() * w_b
            else:
                w_dot_b = poseDv.angularAccelerationBodyFrame(tk)
                C_b_w = poseDv.orientation(tk).inverse()
                a_w = poseDv.linearAcceleration(tk)
                r_b = r_b_dv.toExpression()
                a_b = C_b_w * (a_w - g_w) + w_dot_b.cross(r_b) + w_b.cross(w_b.cross(r_b))
                C_gyro_b = sm["q_gyro_i"].toExpression() * q_i_b.toExpression()
                w = sm["M_gyro"].toExpression() * (C_gyro_b * w_b) + sm["M_accel_gyro"].toExpression() * (C_gyro_b * a_b)
            gerr = F.EuclideanError(meas["gyro"][k][j], np.eye(3) * (1.0 / cfg.gyro_sigma) ** 2, w + b_i)
            gerr.setMEstimatorPolicy(mest_g)
            problem.addErrorTerm(gerr)
        if cfg.bias_motion_terms:
            problem.addErrorTerm(F.BSplineEuclideanMotionError(gyroBiasDv, np.eye(3) / (1.9393e-5 ** 2), 1))
            problem.addErrorTerm(F.BSplineEuclideanMotionError(accelBiasDv, np.eye(3) / (3.0e-3 ** 2), 1))
    # --- LiDAR.addLiDARErrorTerms
    for k, lid in enumerate(W["lidars"]):
        off, T_l_b_dv = lidars[k]
        for b in range(cfg.n_boards):
            plane_normal = F.EuclideanExpression(np.array([0.0, 0.0, 1.0]))
            for i in np.nonzero(lid["board"] == b)[0]:
                tk = off.toExpression() + float(lid["t"][i])
                T_w_b = poseDv.transformationAtTime(tk, cfg.time_padding, cfg.time_padding)
                T_b_l = T_l_b_dv.toExpression().inverse()
                T_w_l = T_w_b * T_b_l
                T_p_l = targets[b].toExpression() * T_w_l
                C_p_l = T_p_l.toRotationExpression()
                t_p = T_p_l.toEuclideanExpression()
                d = plane_normal.dot(t_p)
                theta = plane_normal.dot(C_p_l * lid["dir"][i])
                predicted = d / theta * -1.0
                assert predicted.toScalar() > 0.0          # the front-end's "Swapped sign!" check
                err = F.ScalarError(meas["lidar"][k][i], (1.0 / cfg.lidar_sigma) ** 2, predicted)
                if cfg.lidar_mest:
                    err.setMEstimatorPolicy({S.MEST_CAUCHY: F.CauchyMEstimator, S.MEST_HUBER: F.HuberMEstimator}[cfg.lidar_mest[0]](cfg.lidar_mest[1]))
                problem.addErrorTerm(err)
    return problem


@pytest.fixture(scope="module")
def lowered(small_problem):
    W, meas = small_problem["W"], small_problem["meas"]
    problem = build_like_the_reference(W, synth.initial_state(W), meas)
    spec, handles = F.lower(problem)
    return problem, spec


def test_lowered_problem_is_the_same_problem(lowered, small_problem, small_oracle):
    from oracle_api import OracleApi
    problem, spec = lowered
    ref, _ = small_oracle
    o = OracleApi()
    spec.emit(o)
    o.finalize()
    assert len(problem.terms) == ref.n_terms
    assert (o.j_rows, o.j_cols, o.n_blocks, o.n_terms) == (ref.j_rows, ref.j_cols, ref.n_blocks, ref.n_terms)
    J0, e0 = ref.evaluate_error(True)
    J1, e1 = o.evaluate_error(True)
    Cp0, Ci0, Cx0, rhs0 = ref.build_system(True)
    Cp1, Ci1, Cx1, rhs1 = o.build_system(True)
    assert np.array_equal(Cp0, Cp1) and np.array_equal(Ci0, Ci1)
    assert abs(J1 - J0) <= 1e-12 * abs(J0)
    assert np.abs(e1 - e0).max() <= 1e-12 * np.abs(e0).max()
    assert np.abs(Cx1 - Cx0).max() <= 1e-12 * np.abs(Cx0).max()
    assert np.abs(rhs1 - rhs0).max() <= 1e-12 * np.abs(rhs0).max()


def test_batches_follow_the_sensor_streams(lowered, small_problem):
    _, spec = lowered
    kinds = [k for _, k, _ in spec.batch_ops()]
    want = [k for _, k, _ in small_problem["spec"].batch_ops()]
    assert kinds == want


def test_unsupported_terms_are_refused():
    problem = F.OptimizationProblem()
    p = F.EuclideanPointDv(np.zeros(3))
    p.setActive(True)
    problem.addDesignVariable(p)
    problem.addErrorTerm(F.EuclideanError(np.zeros(3), np.eye(3), p.toExpression() + p.toExpression()))
    with pytest.raises(F.UnsupportedErrorTerm):
        F.lower(problem)
    problem = F.OptimizationProblem()
    problem.addDesignVariable(p)
    problem.addErrorTerm(F.EuclideanError(np.zeros(3), np.diag([1.0, 2.0, 3.0]), p.toExpression()))
    with pytest.raises(F.UnsupportedErrorTerm):
        F.lower(problem)


def test_front_end_numeric_evaluation_matches_the_oracle(lowered, small_problem, small_oracle):
    """ScalarError.error() evaluated by the front-end mirror (SciPy B-spline) vs the oracle's residual: a third, independent
    evaluation of the LiDAR forward model."""
    problem, spec = lowered
    ref, _ = small_oracle
    J, e = ref.evaluate_error(False)
    rows = 0
    checked = 0
    W = small_problem["W"]
    for t in problem.terms:
        if isinstance(t, F.ScalarError):
            if checked % 7 == 0:
                mine = t.error() * np.sqrt(t.invR)
                assert abs(-e[rows] - mine) <= 1e-9 * max(1.0, abs(mine)), (e[rows], mine)
            checked += 1
            rows += 1
        elif isinstance(t, F.SimpleReprojectionError):
            rows += 2
        elif isinstance(t, F.EuclideanError):
            rows += 3
        else:
            rows += 1
    assert checked > 0


@pytest.mark.gpu
def test_lowered_problem_on_the_gpu(lowered, small_oracle):
    from multical_amd.backend import HipApi
    _, spec = lowered
    ref, _ = small_oracle
    g = HipApi(0)
    spec.emit(g)
    g.finalize()
    J0, e0 = ref.evaluate_error(True)
    Cp0, Ci0, Cx0, rhs0 = ref.build_system(True)
    J1, e1 = g.evaluate_error(True)
    Cx1, rhs1 = g.build_system(True)
    Cp1, Ci1 = g.pattern()
    assert np.array_equal(Cp0, Cp1) and np.array_equal(Ci0, Ci1)
    assert abs(J1 - J0) <= 1e-9 * abs(J0)
    assert np.abs(e1 - e0).max() <= 1e-9 * np.abs(e0).max()
    assert np.abs(Cx1 - Cx0).max() <= 1e-9 * np.abs(Cx0).max()
    assert np.abs(rhs1 - rhs0).max() <= 1e-9 * np.abs(rhs0).max()


@pytest.mark.parametrize("model", ["scale_misalignment", "size_effect"])
def test_imu_models_lower_to_the_same_problem(small_problem, model):
    """The expression trees of ScaledMisalignedImu / ScaledMisalignedSizeEffectImu (sensors_and_targets.py:1449-1550,
    1608-1661), built statement by statement, lower to accel_sm / gyro_sm batches that are the same problem as the one
    written down directly (test_variants.make_variant).  The two descriptions add the extra design variables at different
    places of the problem, so block indices differ; what must agree is everything that does not depend on the order:
    dimensions, the cost, the residual vector, and the norms / traces of b and H."""
    from oracle_api import OracleApi
    from test_variants import make_variant
    direct = make_variant(small_problem["spec"], model)
    val = lambda h: direct.ops[h][1]["params"]
    extras = []
    for _, k, d in direct.batch_ops():
        if k == "gyro_sm":
            acc = next(dd for _, kk, dd in direct.batch_ops() if kk == "accel_sm" and dd["dv_q_ib"] == d["dv_q_ib"])
            ex = dict(q_gi=val(d["dv_q_gi"]), M_g=val(d["dv_M"]), M_ag=val(d["dv_Ma"]), M_a=val(acc["dv_M"]))
            if acc["size_effect"]:
                ex["arms"] = [val(h) for h in acc["size_effect"][:3]]
            extras.append(ex)
    W, meas = small_problem["W"], small_problem["meas"]
    problem = build_like_the_reference(W, synth.initial_state(W), meas, imu_model=model, extras=extras)
    spec, _ = F.lower(problem)
    assert [k for _, k, _ in spec.batch_ops()] == [k for _, k, _ in direct.batch_ops()]
    o0, o1 = OracleApi(), OracleApi()
    direct.emit(o0)
    o0.finalize()
    spec.emit(o1)
    o1.finalize()
    assert (o0.j_rows, o0.j_cols, o0.n_blocks, o0.n_terms) == (o1.j_rows, o1.j_cols, o1.n_blocks, o1.n_terms)
    J0, e0 = o0.evaluate_error(True)
    J1, e1 = o1.evaluate_error(True)
    assert abs(J1 - J0) <= 1e-12 * abs(J0)
    assert np.abs(e1 - e0).max() <= 1e-12 * np.abs(e0).max()
    Cp0, Ci0, Cx0, rhs0 = o0.build_system(True)
    Cp1, Ci1, Cx1, rhs1 = o1.build_system(True)
    assert Cx0.size == Cx1.size
    assert abs(np.linalg.norm(rhs1) - np.linalg.norm(rhs0)) <= 1e-10 * np.linalg.norm(rhs0)
    d0, d1 = Cx0[Cp0[1:] - 1], Cx1[Cp1[1:] - 1]
    assert abs(d1.sum() - d0.sum()) <= 1e-10 * d0.sum()
    assert abs(np.linalg.norm(Cx1) - np.linalg.norm(Cx0)) <= 1e-10 * np.linalg.norm(Cx0)
