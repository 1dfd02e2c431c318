"""Problem description, one level above the C ABI.

`ProblemSpec` records, in call order, what the Python front-end of the reference does with
``problem.addDesignVariable`` / ``problem.addErrorTerm`` (kalibr_sensor_calibration/calibrator.py:94-234,
sensors_and_targets.py:123-130, 260-305, 542-630, 1103-1246) -- but in batches instead of one Python
object per error term.  The order of `add_*` calls is the design-variable / error-term order of the
problem and therefore defines block indices and row bases.  `emit(api)` replays the description on any
object that offers the C-ABI call surface (the HIP backend here; the CPU oracle in tests/).
"""
import numpy as np

DV_EUCLIDEAN_POINT, DV_ROTATION_QUATERNION, DV_SCALAR, DV_EUCLIDEAN_DIRECTION = 0, 1, 2, 3
DV_SPLINE_COEFF, DV_PROJECTION, DV_DISTORTION, DV_MATRIX_BASIC, DV_HOMOGENEOUS_POINT = 4, 5, 6, 7, 8
PROJ_PINHOLE, PROJ_OMNI, PROJ_DS = 0, 1, 2
DIST_NONE, DIST_RADTAN, DIST_EQUI = 0, 1, 2
MEST_NONE, MEST_HUBER, MEST_CAUCHY, MEST_GEMAN_MCCLURE, MEST_BZ, MEST_BLAKE_ZISSERMAN = range(6)

# incremental_calibration group ids used by the front-end (calibrator.py:16-18)
CALIBRATION_GROUP_ID, HELPER_GROUP_ID = 0, 1


def blake_zisserman_epsilon(df, p_cut=0.999, w_cut=0.1):
    """BlakeZissermanMEstimator::computeEpsilon (aslam_backend/src/MEstimatorPolicies.cpp:122-129)."""
    from scipy.stats import chi2
    return (1.0 - w_cut) / w_cut * np.exp(-chi2.ppf(p_cut, df))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class ProblemSpec(object):
    def __init__(self):
        self.ops = []          # ("dv"|"spline"|"camera"|batch kinds, dict)
        self.n_dv_handles = 0  # handles for dvs (non-spline)
        self.groups = []       # per op index: group id for dvs/splines (ordering like inc.OptimizationProblem)

    # ---- design variables (handles are indices into self.ops) -------------------------------
    def add_dv(self, kind, params, active=True, scaling=1.0, group=HELPER_GROUP_ID):
        self.ops.append(("dv", dict(kind=kind, params=_f64(params).ravel().copy(), active=bool(active),
                                    scaling=float(scaling), group=group)))
        return len(self.ops) - 1

    def add_matrix_dv(self, A, pattern, active=True, scaling=1.0, group=HELPER_GROUP_ID):
        """aopt.MatrixBasicDv(A, updatePattern) (sensors_and_targets.py:1435-1447)"""
        self.ops.append(("dv", dict(kind=DV_MATRIX_BASIC, params=_f64(A).reshape(9).copy(), pattern=_i32(pattern).reshape(9).copy(),
                                    active=bool(active), scaling=float(scaling), group=group)))
        return len(self.ops) - 1

    def add_spline(self, order, knots, coeffs, active=True, group=HELPER_GROUP_ID):
        coeffs = _f64(coeffs)            # dim x n_coeff (column i = coefficient i)
        self.ops.append(("spline", dict(order=int(order), dim=coeffs.shape[0], knots=_f64(knots).copy(),
                                        coeffs=coeffs.copy(), active=bool(active), group=group)))
        return len(self.ops) - 1

    def add_camera(self, model, dist, dv_proj, dv_dist=-1):
        self.ops.append(("camera", dict(model=model, dist=dist, dv_proj=dv_proj, dv_dist=dv_dist)))
        return len(self.ops) - 1

    # ---- error-term batches ------------------------------------------------------------------
    def add_reprojection_batch(self, camera, spline, dv_q_cb, dv_t_cb, dv_toff, obs_time, obs_dv_q_pw, obs_dv_t_pw,
                               obs_first, y, p, sinv, pad, mest=MEST_NONE, mest_param=0.0):
        self.ops.append(("reproj", dict(camera=camera, spline=spline, dv_q_cb=dv_q_cb, dv_t_cb=dv_t_cb, dv_toff=dv_toff,
                                        obs_time=_f64(obs_time), obs_dv_q_pw=list(obs_dv_q_pw), obs_dv_t_pw=list(obs_dv_t_pw),
                                        obs_first=_i32(obs_first), y=_f64(y).reshape(-1, 2), p=_f64(p).reshape(-1, 3),
                                        sinv=_f64(sinv), pad=float(pad), mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_static_reprojection_batch(self, camera, chain_dv_q, chain_dv_t, obs_dv_q_tc, obs_dv_t_tc, obs_first, y, p, sinv,
                                      mest=MEST_NONE, mest_param=0.0):
        self.ops.append(("static_reproj", dict(camera=camera, chain_dv_q=list(chain_dv_q), chain_dv_t=list(chain_dv_t),
                                               obs_dv_q_tc=list(obs_dv_q_tc), obs_dv_t_tc=list(obs_dv_t_tc),
                                               obs_first=_i32(obs_first), y=_f64(y).reshape(-1, 2), p=_f64(p).reshape(-1, 3),
                                               sinv=_f64(sinv), mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_accelerometer_batch(self, pose_spline, bias_spline, dv_bias, dv_gravity, dv_q_ib, dv_r_b, t, meas, sinv,
                                mest=MEST_NONE, mest_param=0.0):
        self.ops.append(("accel", dict(pose_spline=pose_spline, bias_spline=bias_spline, dv_bias=dv_bias, dv_gravity=dv_gravity,
                                       dv_q_ib=dv_q_ib, dv_r_b=dv_r_b, t=_f64(t), meas=_f64(meas).reshape(-1, 3),
                                       sinv=_f64(sinv), mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_gyroscope_batch(self, pose_spline, bias_spline, dv_bias, dv_q_ib, t, meas, sinv, mest=MEST_NONE, mest_param=0.0):
        self.ops.append(("gyro", dict(pose_spline=pose_spline, bias_spline=bias_spline, dv_bias=dv_bias, dv_q_ib=dv_q_ib,
                                      t=_f64(t), meas=_f64(meas).reshape(-1, 3), sinv=_f64(sinv), mest=mest,
                                      mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_accelerometer_batch_sm(self, pose_spline, bias_spline, dv_bias, dv_gravity, dv_q_ib, dv_r_b, dv_M, size_effect, t, meas,
                                   sinv, mest=MEST_NONE, mest_param=0.0):
        """size_effect: None or the handles (rx_i, ry_i, rz_i, Ix, Iy, Iz)"""
        self.ops.append(("accel_sm", dict(pose_spline=pose_spline, bias_spline=bias_spline, dv_bias=dv_bias, dv_gravity=dv_gravity,
                                          dv_q_ib=dv_q_ib, dv_r_b=dv_r_b, dv_M=dv_M,
                                          size_effect=None if size_effect is None else list(size_effect), t=_f64(t),
                                          meas=_f64(meas).reshape(-1, 3), sinv=_f64(sinv), mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_gyroscope_batch_sm(self, pose_spline, bias_spline, dv_bias, dv_q_ib, dv_q_gi, dv_M, dv_Ma, dv_gravity, dv_r_b, t, meas,
                               sinv, mest=MEST_NONE, mest_param=0.0):
        self.ops.append(("gyro_sm", dict(pose_spline=pose_spline, bias_spline=bias_spline, dv_bias=dv_bias, dv_q_ib=dv_q_ib,
                                         dv_q_gi=dv_q_gi, dv_M=dv_M, dv_Ma=dv_Ma, dv_gravity=dv_gravity, dv_r_b=dv_r_b, t=_f64(t),
                                         meas=_f64(meas).reshape(-1, 3), sinv=_f64(sinv), mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_lidar_batch(self, spline, dv_toff, dv_q_lb, dv_t_lb, t, direction, rng, sinv, dv_q_pw, dv_t_pw, pad,
                        mest=MEST_NONE, mest_param=0.0):
        self.ops.append(("lidar", dict(spline=spline, dv_toff=dv_toff, dv_q_lb=dv_q_lb, dv_t_lb=dv_t_lb, t=_f64(t),
                                       dir=_f64(direction).reshape(-1, 3), range=_f64(rng), sinv=_f64(sinv),
                                       dv_q_pw=list(dv_q_pw), dv_t_pw=list(dv_t_pw), pad=float(pad), mest=mest,
                                       mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_gyroscope_prior_batch(self, dv_q, dv_bias, w_pred, meas, sinv, mest=MEST_NONE, mest_param=0.0):
        """ket.GyroscopeError(omega_measured, invR, R_i_c * const, bias) (sensors_and_targets.py:858-879)"""
        self.ops.append(("gyro_prior", dict(dv_q=dv_q, dv_bias=dv_bias, w_pred=_f64(w_pred).reshape(-1, 3),
                                            meas=_f64(meas).reshape(-1, 3), sinv=_f64(sinv), mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_transformation_batch(self, dv_q_a, dv_t_a, dv_q_b, dv_t_b, prior_q, prior_t, weight_rotation=1.0, weight_translation=0.1,
                                 mest=MEST_NONE, mest_param=0.0):
        """aopt.ErrorTermTransformation(T_a [* T_b.inverse()], prior, weightRotation, weightTranslation)
        (sensors_and_targets.py:812-817, 886-919); dv_q_b = dv_t_b = None for the single-transformation form"""
        prior_q, prior_t = _f64(prior_q).reshape(-1, 4), _f64(prior_t).reshape(-1, 3)
        self.ops.append(("transformation", dict(dv_q_a=dv_q_a, dv_t_a=dv_t_a, dv_q_b=dv_q_b, dv_t_b=dv_t_b, prior_q=prior_q,
                                                prior_t=prior_t, weight_rotation=float(weight_rotation),
                                                weight_translation=float(weight_translation), sinv=np.ones(prior_t.shape[0]),
                                                mest=mest, mest_param=float(mest_param))))
        return len(self.ops) - 1

    def add_motion_term(self, spline, W, order):
        self.ops.append(("motion", dict(spline=spline, W=_f64(W), order=int(order))))
        return len(self.ops) - 1

    def set_landmarks(self, op, handles):
        """landmarks as design variables for static reprojection batch `op`: per term the handle of a
        DV_HOMOGENEOUS_POINT, or None / -1 to keep the constant target point"""
        k, d = self.ops[op]
        assert k == "static_reproj" and len(handles) == len(d["sinv"])
        d["landmarks"] = [(-1 if h is None else h) for h in handles]

    def set_inv_r(self, op, invR):
        """ErrorTermFs.setInvR with a general matrix per term of batch `op`: invR [n, D, D].  It replaces the batch's
        isotropic scalars (which stay as placeholders)."""
        k, d = self.ops[op]
        invR = _f64(invR)
        assert invR.ndim == 3 and invR.shape[0] == len(d["sinv"]) and invR.shape[1] == invR.shape[2]
        d["invR"] = invR.copy()

    # ---- bookkeeping ---------------------------------------------------------------------------
    def batch_ops(self):
        return [(i, k, d) for i, (k, d) in enumerate(self.ops) if k not in ("dv", "spline", "camera")]

    def n_terms(self):
        n = 0
        for _, k, d in self.batch_ops():
            n += 1 if k == "motion" else (len(d["sinv"]))
        return n

    def dv_order(self):
        """Design-variable op indices in problem order: groups in first-insertion order, insertion order inside a
        group (incremental_calibration/src/core/OptimizationProblem.cpp:123-137, 272-292)."""
        order, seen = [], []
        for i, (k, d) in enumerate(self.ops):
            if k in ("dv", "spline") and d["group"] not in seen:
                seen.append(d["group"])
        for g in seen:
            order += [i for i, (k, d) in enumerate(self.ops) if k in ("dv", "spline") and d["group"] == g]
        return order

    def emit(self, api):
        """Replay on `api`.  Returns {op index: backend id} (dv id / spline id / camera id)."""
        ids = {}

        def dvid(h):
            return -1 if h is None or h < 0 else ids[h]

        for i in self.dv_order():
            k, d = self.ops[i]
            if k == "dv" and d["kind"] == DV_MATRIX_BASIC:
                ids[i] = api.add_matrix_design_variable(d["params"], d["pattern"], d["active"], d["scaling"])
            elif k == "dv":
                ids[i] = api.add_design_variable(d["kind"], d["params"], d["active"], d["scaling"])
            else:
                ids[i] = ("spline", api.add_spline(d["order"], d["dim"], d["knots"], d["coeffs"], d["active"]))
        sid = lambda h: -1 if h is None or h < 0 else ids[h][1]
        for i, (k, d) in enumerate(self.ops):
            if k == "camera":
                ids[i] = api.add_camera(d["model"], d["dist"], dvid(d["dv_proj"]), dvid(d["dv_dist"]))
        for i, k, d in self.batch_ops():
            if k == "reproj":
                api.add_reprojection_batch(ids[d["camera"]], sid(d["spline"]), dvid(d["dv_q_cb"]), dvid(d["dv_t_cb"]),
                                           dvid(d["dv_toff"]), d["obs_time"], _i32([dvid(h) for h in d["obs_dv_q_pw"]]),
                                           _i32([dvid(h) for h in d["obs_dv_t_pw"]]), d["obs_first"], d["y"], d["p"], d["sinv"],
                                           d["pad"], d["mest"], d["mest_param"])
            elif k == "static_reproj":
                api.add_static_reprojection_batch(ids[d["camera"]], _i32([dvid(h) for h in d["chain_dv_q"]]),
                                                  _i32([dvid(h) for h in d["chain_dv_t"]]),
                                                  _i32([dvid(h) for h in d["obs_dv_q_tc"]]),
                                                  _i32([dvid(h) for h in d["obs_dv_t_tc"]]), d["obs_first"], d["y"], d["p"],
                                                  d["sinv"], d["mest"], d["mest_param"])
            elif k == "accel":
                api.add_accelerometer_batch(sid(d["pose_spline"]), sid(d["bias_spline"]), dvid(d["dv_bias"]),
                                            dvid(d["dv_gravity"]), dvid(d["dv_q_ib"]), dvid(d["dv_r_b"]), d["t"], d["meas"],
                                            d["sinv"], d["mest"], d["mest_param"])
            elif k == "gyro":
                api.add_gyroscope_batch(sid(d["pose_spline"]), sid(d["bias_spline"]), dvid(d["dv_bias"]), dvid(d["dv_q_ib"]),
                                        d["t"], d["meas"], d["sinv"], d["mest"], d["mest_param"])
            elif k == "accel_sm":
                se = None if d["size_effect"] is None else _i32([dvid(h) for h in d["size_effect"]])
                api.add_accelerometer_batch_sm(sid(d["pose_spline"]), sid(d["bias_spline"]), dvid(d["dv_bias"]),
                                               dvid(d["dv_gravity"]), dvid(d["dv_q_ib"]), dvid(d["dv_r_b"]), dvid(d["dv_M"]), se,
                                               d["t"], d["meas"], d["sinv"], d["mest"], d["mest_param"])
            elif k == "gyro_sm":
                api.add_gyroscope_batch_sm(sid(d["pose_spline"]), sid(d["bias_spline"]), dvid(d["dv_bias"]), dvid(d["dv_q_ib"]),
                                           dvid(d["dv_q_gi"]), dvid(d["dv_M"]), dvid(d["dv_Ma"]), dvid(d["dv_gravity"]),
                                           dvid(d["dv_r_b"]), d["t"], d["meas"], d["sinv"], d["mest"], d["mest_param"])
            elif k == "lidar":
                api.add_lidar_batch(sid(d["spline"]), dvid(d["dv_toff"]), dvid(d["dv_q_lb"]), dvid(d["dv_t_lb"]), d["t"],
                                    d["dir"], d["range"], d["sinv"], _i32([dvid(h) for h in d["dv_q_pw"]]),
                                    _i32([dvid(h) for h in d["dv_t_pw"]]), d["pad"], d["mest"], d["mest_param"])
            elif k == "gyro_prior":
                api.add_gyroscope_prior_batch(dvid(d["dv_q"]), dvid(d["dv_bias"]), d["w_pred"], d["meas"], d["sinv"], d["mest"],
                                              d["mest_param"])
            elif k == "transformation":
                api.add_transformation_batch(dvid(d["dv_q_a"]), dvid(d["dv_t_a"]), dvid(d["dv_q_b"]), dvid(d["dv_t_b"]),
                                             d["prior_q"], d["prior_t"], d["weight_rotation"], d["weight_translation"], d["mest"],
                                             d["mest_param"])
            elif k == "motion":
                api.add_motion_term(sid(d["spline"]), d["W"], d["order"])
            if "landmarks" in d:
                api.set_last_batch_landmarks(_i32([dvid(h) for h in d["landmarks"]]))
            if "invR" in d:
                api.set_last_batch_inv_r(d["invR"])
        return ids
