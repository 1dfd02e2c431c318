"""Synthetic versions of the front-end's initialisation problems (the F6 error terms of SURVEY.md section 8a):

* `make_target_pose_prior`  -- Calibrator.findTargetPosePrior (kalibr_sensor_calibration/sensors_and_targets.py:760-843):
  one TransformationDv per target (target 0 fixed), `ErrorTermTransformation(T_ti * T_tj.inverse(), measured, 1.0, 0.1)`
  for every simultaneous observation of two targets;
* `make_extrinsic_prior`    -- findExtrinsicPriorSensorsToCamera (:845-935): rotation q_i_c + gyroscope bias from
  `GyroscopeError(omega_measured, invR, R_i_c * const, bias)`, one TransformationDv per further camera / LiDAR with
  `ErrorTermTransformation(T.toExpression(), measured, 1.0, 0.1)`.

All of them are small (hundreds to thousands of terms over a handful of global design variables, no spline)."""
import numpy as np

from . import spec as S
from .synth import quat_from_matrix, rotvec_to_matrix


def _rand_T(rng, rot, tra):
    return rotvec_to_matrix(rng.uniform(-rot, rot, 3)), rng.uniform(-tra, tra, 3)


def _mul(A, B):
    return A[0] @ B[0], A[0] @ B[1] + A[1]


def _inv(A):
    return A[0].T, -A[0].T @ A[1]


def make_target_pose_prior(n_targets=3, n_meas=12, seed=7, mest=S.MEST_NONE, mest_param=0.0):
    rng = np.random.default_rng(20221023 + seed)
    truth = [(np.eye(3), np.zeros(3))] + [_rand_T(rng, 1.0, 2.0) for _ in range(n_targets - 1)]
    sp = S.ProblemSpec()
    H = dict(targets=[], truth=truth)
    for i, T in enumerate(truth):
        guess = T if i == 0 else _mul(_rand_T(rng, 0.05, 0.05), T)
        q = sp.add_dv(S.DV_ROTATION_QUATERNION, quat_from_matrix(guess[0]), active=(i != 0), group=S.CALIBRATION_GROUP_ID)
        t = sp.add_dv(S.DV_EUCLIDEAN_POINT, guess[1], active=(i != 0), group=S.CALIBRATION_GROUP_ID)
        H["targets"].append((q, t))
    for i in range(n_targets):
        for j in range(i + 1, n_targets):
            n = n_meas + 3 * i + j          # ragged batches
            pq, pt = np.zeros((n, 4)), np.zeros((n, 3))
            for k in range(n):
                M = _mul(_rand_T(rng, 0.01, 0.005), _mul(truth[i], _inv(truth[j])))
                pq[k], pt[k] = quat_from_matrix(M[0]), M[1]
            sp.add_transformation_batch(H["targets"][i][0], H["targets"][i][1], H["targets"][j][0], H["targets"][j][1], pq, pt, 1.0, 0.1,
                                        mest, mest_param)
    return sp, H


def make_extrinsic_prior(n_gyro=300, n_other=2, n_obs=25, seed=11, gyro_sigma=0.01, huber_gyro=None, mest_transform=None):
    rng = np.random.default_rng(20221023 + seed)
    sp = S.ProblemSpec()
    C_i_c = rotvec_to_matrix(rng.uniform(-0.4, 0.4, 3))
    b_true = rng.uniform(-0.02, 0.02, 3)
    q = sp.add_dv(S.DV_ROTATION_QUATERNION, np.array([0.0, 0.0, 0.0, 1.0]), group=S.CALIBRATION_GROUP_ID)
    b = sp.add_dv(S.DV_EUCLIDEAN_POINT, np.zeros(3), group=S.CALIBRATION_GROUP_ID)
    tt = np.linspace(0.0, 20.0, n_gyro)
    w_pred = np.stack([0.5 * np.sin(0.9 * tt), 0.4 * np.sin(0.7 * tt + 0.3), 0.6 * np.cos(0.5 * tt)], 1)
    meas = w_pred @ C_i_c.T + b_true + rng.normal(0.0, gyro_sigma, (n_gyro, 3))
    gm = (S.MEST_HUBER, huber_gyro) if huber_gyro else (S.MEST_NONE, 0.0)
    sp.add_gyroscope_prior_batch(q, b, w_pred, meas, np.full(n_gyro, 1.0 / gyro_sigma), gm[0], gm[1])
    H = dict(q_i_c=q, bias=b, others=[], C_i_c=C_i_c, bias_true=b_true)
    tm = mest_transform or (S.MEST_NONE, 0.0)
    for o in range(n_other):
        T = _rand_T(rng, 0.8, 0.5)
        hq = sp.add_dv(S.DV_ROTATION_QUATERNION, np.array([0.0, 0.0, 0.0, 1.0]), group=S.CALIBRATION_GROUP_ID)
        ht = sp.add_dv(S.DV_EUCLIDEAN_POINT, np.zeros(3), group=S.CALIBRATION_GROUP_ID)
        n = n_obs + 7 * o
        pq, pt = np.zeros((n, 4)), np.zeros((n, 3))
        for k in range(n):
            M = _mul(_rand_T(rng, 0.02, 0.01), T)
            pq[k], pt[k] = quat_from_matrix(M[0]), M[1]
        sp.add_transformation_batch(hq, ht, None, None, pq, pt, 1.0, 0.1, tm[0], tm[1])
        H["others"].append((hq, ht, T))
    return sp, H
