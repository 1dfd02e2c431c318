"""ctypes adapter over oracle/liboracle.so (CPU oracle; TEST INFRASTRUCTURE ONLY).

Offers the same call surface as multical_amd.backend.HipApi so a ProblemSpec can be emitted to either.
Imported only by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build_oracle()
        _lib = C.CDLL(ORACLE_LIB)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_evaluate_error.restype = C.c_double
        for name in dir(_lib):
            pass
    return _lib


class OracleApi(object):
    def __init__(self):
        self.lib = _load()
        self.h = C.c_void_p(self.lib.orc_create())
        self.finalized = False

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            self.lib.orc_last_error.argtypes = [C.c_void_p]
            raise RuntimeError("oracle: " + self.lib.orc_last_error(self.h).decode())
        return rc

    def add_design_variable(self, kind, params, active=True, scaling=1.0):
        p = _f64(params)
        return self._chk(self.lib.orc_add_dv(self.h, C.c_int(kind), _d(p), C.c_int(p.size), C.c_int(int(bool(active))),
                                             C.c_double(scaling)))

    def add_spline(self, order, dim, knots, coeffs, active=True):
        k = _f64(knots)
        c = np.ascontiguousarray(_f64(coeffs).T)
        return self._chk(self.lib.orc_add_spline(self.h, C.c_int(order), C.c_int(dim), _d(k), C.c_int(k.size), _d(c),
                                                 C.c_int(c.shape[0]), C.c_int(int(bool(active)))))

    def spline_first_dv(self, sid):
        return self.lib.orc_spline_first_dv(self.h, C.c_int(sid))

    def add_camera(self, model, dist, dv_proj, dv_dist=-1):
        return self._chk(self.lib.orc_add_camera(self.h, C.c_int(model), C.c_int(dist), C.c_int(dv_proj), C.c_int(dv_dist)))

    def add_reprojection_batch(self, cam, spline, dv_q_cb, dv_t_cb, dv_toff, obs_time, obs_q, obs_t, obs_first, y, p, sinv, pad,
                               mest=0, mest_param=0.0):
        obs_time, y, p, sinv = _f64(obs_time), _f64(y), _f64(p), _f64(sinv)
        obs_q, obs_t, obs_first = _i32(obs_q), _i32(obs_t), _i32(obs_first)
        self._chk(self.lib.orc_add_reprojection_batch(self.h, C.c_int(cam), C.c_int(spline), C.c_int(dv_q_cb), C.c_int(dv_t_cb),
                                                      C.c_int(dv_toff), C.c_int(obs_time.size), _d(obs_time), _i(obs_q),
                                                      _i(obs_t), _i(obs_first), _d(y), _d(p), _d(sinv), C.c_double(pad),
                                                      C.c_int(mest), C.c_double(mest_param)))

    def add_static_reprojection_batch(self, cam, chain_q, chain_t, obs_q, obs_t, obs_first, y, p, sinv, mest=0, mest_param=0.0):
        y, p, sinv = _f64(y), _f64(p), _f64(sinv)
        chain_q, chain_t, obs_q, obs_t, obs_first = _i32(chain_q), _i32(chain_t), _i32(obs_q), _i32(obs_t), _i32(obs_first)
        self._chk(self.lib.orc_add_static_reprojection_batch(self.h, C.c_int(cam), C.c_int(chain_q.size), _i(chain_q), _i(chain_t),
                                                             C.c_int(obs_q.size), _i(obs_q), _i(obs_t), _i(obs_first), _d(y),
                                                             _d(p), _d(sinv), C.c_int(mest), C.c_double(mest_param)))

    def add_accelerometer_batch(self, pose_spline, bias_spline, dv_bias, dv_g, dv_q_ib, dv_r_b, t, meas, sinv, mest=0,
                                mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        self._chk(self.lib.orc_add_accel_batch(self.h, C.c_int(pose_spline), C.c_int(bias_spline), C.c_int(dv_bias), C.c_int(dv_g),
                                               C.c_int(dv_q_ib), C.c_int(dv_r_b), C.c_int(t.size), _d(t), _d(meas), _d(sinv),
                                               C.c_int(mest), C.c_double(mest_param)))

    def add_gyroscope_batch(self, pose_spline, bias_spline, dv_bias, dv_q_ib, t, meas, sinv, mest=0, mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        self._chk(self.lib.orc_add_gyro_batch(self.h, C.c_int(pose_spline), C.c_int(bias_spline), C.c_int(dv_bias),
                                              C.c_int(dv_q_ib), C.c_int(t.size), _d(t), _d(meas), _d(sinv), C.c_int(mest),
                                              C.c_double(mest_param)))

    def add_gyroscope_prior_batch(self, dv_q, dv_bias, w_pred, meas, sinv, mest=0, mest_param=0.0):
        w_pred, meas, sinv = _f64(w_pred), _f64(meas), _f64(sinv)
        self._chk(self.lib.orc_add_gyroscope_error_batch(self.h, C.c_int(dv_q), C.c_int(dv_bias), C.c_int(sinv.size), _d(w_pred),
                                                         _d(meas), _d(sinv), C.c_int(mest), C.c_double(mest_param)))

    def add_transformation_batch(self, dv_q_a, dv_t_a, dv_q_b, dv_t_b, prior_q, prior_t, weight_rotation, weight_translation,
                                 mest=0, mest_param=0.0):
        prior_q, prior_t = _f64(prior_q), _f64(prior_t)
        self._chk(self.lib.orc_add_transformation_error_batch(self.h, C.c_int(dv_q_a), C.c_int(dv_t_a), C.c_int(dv_q_b), C.c_int(dv_t_b),
                                                              C.c_int(prior_t.size // 3), _d(prior_q), _d(prior_t),
                                                              C.c_double(weight_rotation), C.c_double(weight_translation),
                                                              C.c_int(mest), C.c_double(mest_param)))

    def add_matrix_design_variable(self, A, pattern, active=True, scaling=1.0):
        A, pattern = _f64(A).reshape(9), _i32(pattern).reshape(9)
        return self._chk(self.lib.orc_add_matrix_dv(self.h, _d(A), _i(pattern), C.c_int(int(bool(active))), C.c_double(scaling)))

    def add_accelerometer_batch_sm(self, pose_spline, bias_spline, dv_bias, dv_g, dv_q_ib, dv_r_b, dv_M, size_effect, t, meas, sinv,
                                   mest=0, mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        se = None if size_effect is None else _i32(size_effect)
        self._chk(self.lib.orc_add_accel_sm_batch(self.h, C.c_int(pose_spline), C.c_int(bias_spline), C.c_int(dv_bias), C.c_int(dv_g),
                                                  C.c_int(dv_q_ib), C.c_int(dv_r_b), C.c_int(dv_M), None if se is None else _i(se),
                                                  C.c_int(t.size), _d(t), _d(meas), _d(sinv), C.c_int(mest), C.c_double(mest_param)))

    def add_gyroscope_batch_sm(self, pose_spline, bias_spline, dv_bias, dv_q_ib, dv_q_gi, dv_M, dv_Ma, dv_g, dv_r_b, t, meas, sinv,
                               mest=0, mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        self._chk(self.lib.orc_add_gyro_sm_batch(self.h, C.c_int(pose_spline), C.c_int(bias_spline), C.c_int(dv_bias),
                                                 C.c_int(dv_q_ib), C.c_int(dv_q_gi), C.c_int(dv_M), C.c_int(dv_Ma), C.c_int(dv_g),
                                                 C.c_int(dv_r_b), C.c_int(t.size), _d(t), _d(meas), _d(sinv), C.c_int(mest),
                                                 C.c_double(mest_param)))

    def add_lidar_batch(self, spline, dv_toff, dv_q_lb, dv_t_lb, t, direction, rng, sinv, dv_q_pw, dv_t_pw, pad, mest=0,
                        mest_param=0.0):
        t, direction, rng, sinv = _f64(t), _f64(direction), _f64(rng), _f64(sinv)
        dv_q_pw, dv_t_pw = _i32(dv_q_pw), _i32(dv_t_pw)
        self._chk(self.lib.orc_add_lidar_batch(self.h, C.c_int(spline), C.c_int(dv_toff), C.c_int(dv_q_lb), C.c_int(dv_t_lb),
                                               C.c_int(t.size), _d(t), _d(direction), _d(rng), _d(sinv), _i(dv_q_pw),
                                               _i(dv_t_pw), C.c_double(pad), C.c_int(mest), C.c_double(mest_param)))

    def add_motion_term(self, spline, W, order):
        W = _f64(W)
        self._chk(self.lib.orc_add_motion_term(self.h, C.c_int(spline), _d(W), C.c_int(order)))

    def set_last_batch_landmarks(self, dv_landmark):
        dv = _i32(dv_landmark)
        self._chk(self.lib.orc_set_last_terms_landmarks(self.h, C.c_int(dv.size), _i(dv)))

    def set_last_batch_inv_r(self, invR):
        """ErrorTermFs::setInvR on the terms of the batch added last (oracle_unit_capi.hpp)"""
        invR = _f64(invR)
        self._chk(self.lib.orc_set_last_terms_inv_r(self.h, C.c_int(invR.shape[0]), C.c_int(invR.shape[1]), _d(invR)))

    def finalize(self):
        self._chk(self.lib.orc_finalize(self.h))
        v = [C.c_int() for _ in range(4)]
        self.lib.orc_dims(self.h, *[C.byref(x) for x in v])
        self.j_rows, self.j_cols, self.n_blocks, self.n_terms = [x.value for x in v]
        self.finalized = True

    def dv_block(self, dv):
        b, c = C.c_int(), C.c_int()
        self.lib.orc_dv_block(self.h, C.c_int(dv), C.byref(b), C.byref(c))
        return b.value, c.value

    def evaluate_error(self, use_mest=True, want_e=True, n_threads=1):
        e = np.empty(self.j_rows) if want_e else None
        J = self.lib.orc_evaluate_error(self.h, C.c_int(n_threads), C.c_int(int(use_mest)), _d(e))
        if J != J:
            self._chk(-1)
        return J, e

    def term_chi2(self):
        out = np.empty(self.n_terms)
        self.lib.orc_term_chi2(self.h, _d(out))
        return out

    def build_system(self, use_mest=True):
        """Returns (Cp, Ci, Cx, rhs) exactly as fillCCS(upper) exports them after buildSystem."""
        self._chk(self.lib.orc_build_system(self.h, C.c_int(int(use_mest))))
        nnz = self.lib.orc_ccs_nnz(self.h)
        Cp = np.empty(self.j_cols + 1, np.int32)
        Ci = np.empty(nnz, np.int32)
        Cx = np.empty(nnz)
        rhs = np.empty(self.j_cols)
        self.lib.orc_get_ccs(self.h, _i(Cp), _i(Ci), _d(Cx), _d(rhs))
        return Cp, Ci, Cx, rhs

    def build_only(self, use_mest=True):
        self._chk(self.lib.orc_build_system(self.h, C.c_int(int(use_mest))))

    def jacobian_transpose(self, use_mest=True):
        """(col_ptr[j_rows+1], row_ind, values) of J^T as SparseCholeskyLinearSystemSolver builds it."""
        self.lib.orc_build_jacobian_transpose.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        nnz = self._chk(self.lib.orc_build_jacobian_transpose(self.h, int(use_mest), None, None, None))
        cp, ri, v = np.zeros(self.j_rows + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        self._chk(self.lib.orc_build_jacobian_transpose(self.h, int(use_mest), cp.ctypes.data, ri.ctypes.data, v.ctypes.data))
        return cp, ri, v

    def term_weighted_jacobian(self, term, use_mest=True):
        dim, rb = C.c_int(), C.c_int()
        self.lib.orc_term_info(self.h, C.c_int(term), C.byref(dim), C.byref(rb))
        J = np.zeros((dim.value, self.j_cols))
        e = np.zeros(dim.value)
        self._chk(self.lib.orc_term_weighted_jacobian(self.h, C.c_int(term), C.c_int(int(use_mest)), _d(J), _d(e)))
        return J, e

    def apply_update(self, dx):
        dx = _f64(dx)
        self._chk(self.lib.orc_update(self.h, _d(dx)))

    def revert_update(self):
        self.lib.orc_revert(self.h)

    def dv_params(self, dv):
        out = np.empty(16)
        n = self.lib.orc_get_dv_params(self.h, C.c_int(dv), _d(out))
        return out[:n].copy()

    def solve_system(self, lam):
        dx = np.zeros(self.j_cols)
        rc = self._chk(self.lib.orc_solve_system(self.h, C.c_double(lam), _d(dx)))
        return rc == 0, dx

    def lidar_find_points_on_targets(self, spline, dv_toff, dv_q_lb, dv_t_lb, dv_q_pw, dv_t_pw, plane_min, plane_max, threshold, points):
        pts = _f64(points).reshape(-1, 4)
        q, t = _i32(dv_q_pw), _i32(dv_t_pw)
        lo, hi = _f64(plane_min).reshape(-1, 3), _f64(plane_max).reshape(-1, 3)
        mask, world, margin = np.zeros(pts.shape[0], np.uint32), np.zeros((pts.shape[0], 3)), np.zeros(pts.shape[0])
        self._chk(self.lib.orc_lidar_find_points_on_targets(self.h, C.c_int(spline), C.c_int(dv_toff), C.c_int(dv_q_lb), C.c_int(dv_t_lb),
                                                            C.c_int(q.size), _i(q), _i(t), _d(lo), _d(hi), C.c_double(threshold),
                                                            C.c_int(pts.shape[0]), _d(pts), mask.ctypes.data_as(C.POINTER(C.c_uint)),
                                                            _d(world), _d(margin)))
        return mask, world, margin

    def optimize(self, max_iterations=30, lambda_init=10.0, convergence_delta_x=1e-5, convergence_j_ratio=1e-6, n_threads=1):
        out = np.zeros(7)
        self._chk(self.lib.orc_optimize(self.h, C.c_int(max_iterations), C.c_double(lambda_init), C.c_double(convergence_delta_x),
                                        C.c_double(convergence_j_ratio), C.c_int(n_threads), _d(out)))
        return dict(JStart=out[0], JFinal=out[1], iterations=int(out[2]), failedIterations=int(out[3]), dXFinal=out[4],
                    dJFinal=out[5], linearSolverFailure=bool(out[6]))
