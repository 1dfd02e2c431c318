"""Host-side mirror of the reference's solver interface for the HIP backend.

`HipApi` is a 1:1 wrapper over the C ABI (include/multical_hip.h).  `HipBlockCholeskyLinearSystemSolver`,
`LevenbergMarquardtTrustRegionPolicy`, `Optimizer2Options` and `Optimizer2` keep the names, argument meaning and
control flow of the reference classes they stand in for:

  aslam_backend/include/aslam/backend/LinearSystemSolver.hpp:16-109
  aslam_backend/src/BlockCholeskyLinearSystemSolver.cpp:34-106,172-176
  aslam_backend/src/LevenbergMarquardtTrustRegionPolicy.cpp:38-113, src/TrustRegionPolicy.cpp:29-57
  aslam_backend/src/Optimizer2.cpp:183-320, include/aslam/backend/Optimizer2Options.hpp:9-42

The residual / Jacobian / normal-equation work runs on the GPU; the sparse Cholesky solve stays on the host
(north_star: "timed separately").  CHOLMOD is not installed here, so `solveSystem` factorises the same CCS
matrix with the C++ block Cholesky of csrc/host/block_cholesky.hpp; the reference's own tests pin the solver only
through solver-vs-solver agreement of dx (aslam_backend/test/LinearSolverTests.cpp:42-57).  The optimiser loop is C++
(csrc/host/lm_driver.hpp behind `mc_optimize` / `HipApi.optimize`); `Optimizer2` here only translates the options object.
"""
import ctypes as C

import numpy as np

from . import _lib


class BackendError(RuntimeError):
    """aslam::Exception stand-in (SM_DEFINE_EXCEPTION, LinearSystemSolver.hpp:18)."""


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class HipApi(object):
    """Thin object wrapper over the C ABI; every method is one C call."""

    def __init__(self, device=-1):
        self.lib = _lib.load()
        self.ctx = self.lib.mc_create(device)
        if not self.ctx:
            raise BackendError("mc_create failed: %s" % self.lib.mc_last_error(None).decode())
        self.finalized = False

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise BackendError(self.lib.mc_last_error(self.ctx).decode())
        return rc

    # ---- description
    def add_design_variable(self, kind, params, active=True, scaling=1.0):
        p = _f64(params)
        return self._chk(self.lib.mc_add_design_variable(self.ctx, kind, _dp(p), p.size, int(bool(active)), float(scaling)))

    def add_spline(self, order, dim, knots, coeffs, active=True):
        k = _f64(knots)
        c = np.ascontiguousarray(_f64(coeffs).T)   # dim x n -> n x dim rows == column-major dim x n
        return self._chk(self.lib.mc_add_spline(self.ctx, order, dim, _dp(k), k.size, _dp(c), c.shape[0], int(bool(active))))

    def spline_first_dv(self, sid):
        return self.lib.mc_spline_first_dv(self.ctx, sid)

    def add_camera(self, model, dist, dv_proj, dv_dist=-1):
        return self._chk(self.lib.mc_add_camera(self.ctx, model, dist, dv_proj, dv_dist))

    def add_reprojection_batch(self, cam, spline, dv_q_cb, dv_t_cb, dv_toff, obs_time, obs_q, obs_t, obs_first, y, p, sinv, pad,
                               mest=0, mest_param=0.0):
        obs_time, y, p, sinv = _f64(obs_time), _f64(y), _f64(p), _f64(sinv)
        obs_q, obs_t, obs_first = _i32(obs_q), _i32(obs_t), _i32(obs_first)
        assert obs_first.size == obs_time.size + 1 and y.size == 2 * sinv.size and p.size == 3 * sinv.size
        self._chk(self.lib.mc_add_reprojection_batch(self.ctx, cam, spline, dv_q_cb, dv_t_cb, dv_toff, obs_time.size,
                                                     _dp(obs_time), _ip(obs_q), _ip(obs_t), _ip(obs_first), _dp(y), _dp(p),
                                                     _dp(sinv), float(pad), mest, float(mest_param)))

    def add_static_reprojection_batch(self, cam, chain_q, chain_t, obs_q, obs_t, obs_first, y, p, sinv, mest=0, mest_param=0.0):
        y, p, sinv = _f64(y), _f64(p), _f64(sinv)
        chain_q, chain_t, obs_q, obs_t, obs_first = _i32(chain_q), _i32(chain_t), _i32(obs_q), _i32(obs_t), _i32(obs_first)
        self._chk(self.lib.mc_add_static_reprojection_batch(self.ctx, cam, chain_q.size, _ip(chain_q), _ip(chain_t), obs_q.size,
                                                            _ip(obs_q), _ip(obs_t), _ip(obs_first), _dp(y), _dp(p), _dp(sinv),
                                                            mest, float(mest_param)))

    def add_accelerometer_batch(self, pose_spline, bias_spline, dv_bias, dv_g, dv_q_ib, dv_r_b, t, meas, sinv, mest=0,
                                mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        self._chk(self.lib.mc_add_accelerometer_batch(self.ctx, pose_spline, bias_spline, dv_bias, dv_g, dv_q_ib, dv_r_b, t.size,
                                                      _dp(t), _dp(meas), _dp(sinv), mest, float(mest_param)))

    def add_gyroscope_batch(self, pose_spline, bias_spline, dv_bias, dv_q_ib, t, meas, sinv, mest=0, mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        self._chk(self.lib.mc_add_gyroscope_batch(self.ctx, pose_spline, bias_spline, dv_bias, dv_q_ib, t.size, _dp(t), _dp(meas),
                                                  _dp(sinv), mest, float(mest_param)))

    def add_gyroscope_prior_batch(self, dv_q, dv_bias, w_pred, meas, sinv, mest=0, mest_param=0.0):
        w_pred, meas, sinv = _f64(w_pred), _f64(meas), _f64(sinv)
        assert w_pred.size == 3 * sinv.size and meas.size == 3 * sinv.size
        self._chk(self.lib.mc_add_gyroscope_prior_batch(self.ctx, dv_q, dv_bias, sinv.size, _dp(w_pred), _dp(meas), _dp(sinv), mest,
                                                        float(mest_param)))

    def add_transformation_batch(self, dv_q_a, dv_t_a, dv_q_b, dv_t_b, prior_q, prior_t, weight_rotation, weight_translation,
                                 mest=0, mest_param=0.0):
        prior_q, prior_t = _f64(prior_q), _f64(prior_t)
        n = prior_t.size // 3
        assert prior_q.size == 4 * n and prior_t.size == 3 * n
        self._chk(self.lib.mc_add_transformation_batch(self.ctx, dv_q_a, dv_t_a, dv_q_b, dv_t_b, n, _dp(prior_q), _dp(prior_t),
                                                       float(weight_rotation), float(weight_translation), mest, float(mest_param)))

    def add_matrix_design_variable(self, A, pattern, active=True, scaling=1.0):
        A, pattern = _f64(A).reshape(9), _i32(pattern).reshape(9)
        return self._chk(self.lib.mc_add_matrix_design_variable(self.ctx, _dp(A), _ip(pattern), int(bool(active)), float(scaling)))

    def add_accelerometer_batch_sm(self, pose_spline, bias_spline, dv_bias, dv_g, dv_q_ib, dv_r_b, dv_M, size_effect, t, meas, sinv,
                                   mest=0, mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        se = None if size_effect is None else _i32(size_effect)
        assert se is None or se.size == 6
        self._chk(self.lib.mc_add_accelerometer_batch_sm(self.ctx, pose_spline, bias_spline, dv_bias, dv_g, dv_q_ib, dv_r_b, dv_M,
                                                         _ip(se), t.size, _dp(t), _dp(meas), _dp(sinv), mest, float(mest_param)))

    def add_gyroscope_batch_sm(self, pose_spline, bias_spline, dv_bias, dv_q_ib, dv_q_gi, dv_M, dv_Ma, dv_g, dv_r_b, t, meas, sinv,
                               mest=0, mest_param=0.0):
        t, meas, sinv = _f64(t), _f64(meas), _f64(sinv)
        self._chk(self.lib.mc_add_gyroscope_batch_sm(self.ctx, pose_spline, bias_spline, dv_bias, dv_q_ib, dv_q_gi, dv_M, dv_Ma, dv_g,
                                                     dv_r_b, t.size, _dp(t), _dp(meas), _dp(sinv), mest, float(mest_param)))

    def add_lidar_batch(self, spline, dv_toff, dv_q_lb, dv_t_lb, t, direction, rng, sinv, dv_q_pw, dv_t_pw, pad, mest=0,
                        mest_param=0.0):
        t, direction, rng, sinv = _f64(t), _f64(direction), _f64(rng), _f64(sinv)
        dv_q_pw, dv_t_pw = _i32(dv_q_pw), _i32(dv_t_pw)
        self._chk(self.lib.mc_add_lidar_batch(self.ctx, spline, dv_toff, dv_q_lb, dv_t_lb, t.size, _dp(t), _dp(direction),
                                              _dp(rng), _dp(sinv), _ip(dv_q_pw), _ip(dv_t_pw), float(pad), mest,
                                              float(mest_param)))

    def add_motion_term(self, spline, W, order):
        W = _f64(W)
        self._chk(self.lib.mc_add_motion_term(self.ctx, spline, _dp(W), order))

    def set_last_batch_landmarks(self, dv_landmark):
        """per term of the static reprojection batch added last: the HomogeneousPoint dv id, or -1"""
        dv = _i32(dv_landmark)
        self._chk(self.lib.mc_set_last_batch_landmarks(self.ctx, dv.size, _ip(dv)))

    def compute_matrix_sqrt(self, A):
        """sm::eigen::computeMatrixSqrt: S with A = S S^T (what ErrorTermFs.setInvR applies to invR)"""
        A = _f64(A)
        S = np.empty_like(A)
        if self.lib.mc_compute_matrix_sqrt(A.shape[0], _dp(A), _dp(S)) != 0:
            raise ValueError("mc_compute_matrix_sqrt: bad size")
        return S

    def set_last_batch_sqrt_inv_r(self, S):
        """general per-term sqrtInvR [n, D, D] for the batch added last (ErrorTermFs.setSqrtInvR)"""
        S = _f64(S)
        assert S.ndim == 3 and S.shape[1] == S.shape[2]
        self._chk(self.lib.mc_set_last_batch_sqrt_inv_r(self.ctx, S.shape[1], S.shape[0], _dp(S)))

    def set_last_batch_inv_r(self, invR):
        """ErrorTermFs.setInvR per term: invR [n, D, D]"""
        invR = _f64(invR)
        self.set_last_batch_sqrt_inv_r(np.stack([self.compute_matrix_sqrt(R) for R in invR]))

    def set_shard(self, rank, world):
        self._chk(self.lib.mc_set_shard(self.ctx, rank, world))

    def set_clear_mode(self, written_only):
        """With shards: later passes clear only the values this rank writes (mc_set_clear_mode; the caller keeps the rest
        of every buffer it passes at zero)."""
        self._chk(self.lib.mc_set_clear_mode(self.ctx, int(bool(written_only))))

    # ---- structure
    def finalize(self):
        self._chk(self.lib.mc_finalize(self.ctx))
        v = [C.c_int() for _ in range(5)]
        self.lib.mc_dims(self.ctx, *[C.byref(x) for x in v])
        self.j_rows, self.j_cols, self.n_blocks, self.n_terms, self.nnz = [x.value for x in v]
        self.finalized = True

    def dv_block(self, dv):
        b, c = C.c_int(), C.c_int()
        self._chk(self.lib.mc_dv_block(self.ctx, dv, C.byref(b), C.byref(c)))
        return b.value, c.value

    def block_bases(self):
        out = np.empty(self.n_blocks + 1, np.int32)
        self._chk(self.lib.mc_get_block_bases(self.ctx, _ip(out)))
        return out

    def block_times(self):
        """per block: middle of the support of its spline coefficient, NaN for the other blocks"""
        out = np.empty(self.n_blocks)
        self._chk(self.lib.mc_get_block_times(self.ctx, _dp(out)))
        return out

    def optimize(self, **kw):
        """Optimizer2::optimize in C++ (mc_optimize): keyword arguments are mc_optimizer_options fields.  Returns the
        mc_solution_return_value as a dict."""
        o = _lib.OptimizerOptions()
        self.lib.mc_optimizer_options_default(C.byref(o))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError("unknown option %s" % k)
            setattr(o, k, v)
        r = _lib.SolutionReturnValue()
        if self.lib.mc_optimize(self.ctx, C.byref(o), C.byref(r)) != 0:
            raise BackendError(self.lib.mc_host_last_error().decode())
        self.finalized = True
        return {f: getattr(r, f) for f, _ in r._fields_}

    def optimize_sharded(self, allreduce, broadcast, **kw):
        """mc_optimize_sharded: Optimizer2::optimize over time-window shards (this context was given set_shard(rank, world)).
        allreduce(ptr, n): sum n doubles of DEVICE memory at address ptr over the ranks, in place; broadcast(ptr, n, root): n
        doubles of HOST memory from rank root to all.  torch_collectives() makes both from torch.distributed."""
        o = _lib.OptimizerOptions()
        self.lib.mc_optimizer_options_default(C.byref(o))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError("unknown option %s" % k)
            setattr(o, k, v)
        failure = []

        def _ar(user, ptr, n):
            try:
                allreduce(ptr, n)
                return 0
            except Exception as ex:   # an exception must not unwind through the C frames
                failure.append(ex)
                return -1

        def _bc(user, ptr, n, root):
            try:
                broadcast(ptr, n, root)
                return 0
            except Exception as ex:
                failure.append(ex)
                return -1
        ar, bc = _lib.ALLREDUCE_FN(_ar), _lib.BROADCAST_FN(_bc)
        r = _lib.SolutionReturnValue()
        rc = self.lib.mc_optimize_sharded(self.ctx, C.byref(o), C.byref(r), ar, bc, None)
        if failure:
            raise failure[0]
        if rc != 0:
            raise BackendError(self.lib.mc_host_last_error().decode())
        self.finalized = True
        return {f: getattr(r, f) for f, _ in r._fields_}

    def pattern(self):
        Cp = np.empty(self.j_cols + 1, np.int32)
        Ci = np.empty(self.nnz, np.int32)
        self._chk(self.lib.mc_get_pattern(self.ctx, _ip(Cp), _ip(Ci)))
        return Cp, Ci

    # ---- hot path
    def evaluate_error(self, use_mest=True, want_e=True):
        J = C.c_double()
        e = np.empty(self.j_rows) if want_e else None
        self._chk(self.lib.mc_evaluate_error(self.ctx, int(use_mest), C.byref(J), _dp(e)))
        self.check_status()
        return J.value, e

    def build_system(self, use_mest=True):
        Cx = np.empty(self.nnz)
        rhs = np.empty(self.j_cols)
        self._chk(self.lib.mc_build_system(self.ctx, int(use_mest), _dp(Cx), _dp(rhs)))
        self.check_status()
        return Cx, rhs

    def evaluate_error_device(self, use_mest, d_J, d_e, stream):
        self._chk(self.lib.mc_evaluate_error_device(self.ctx, int(use_mest), d_J, d_e, stream))

    def evaluate_error_and_build_system_device(self, use_mest, d_J, d_e, d_Cx, d_rhs, stream):
        self._chk(self.lib.mc_evaluate_error_and_build_system_device(self.ctx, int(use_mest), d_J, d_e, d_Cx, d_rhs, stream))

    def build_system_device(self, use_mest, d_Cx, d_rhs, stream):
        self._chk(self.lib.mc_build_system_device(self.ctx, int(use_mest), d_Cx, d_rhs, stream))

    def build_system_device_noalloc(self, use_mest, d_Cx, d_rhs, stream):
        """mc_build_system_device with its scratch in the library's static device array: makes no allocation."""
        self._chk(self.lib.mc_build_system_device_noalloc(self.ctx, int(use_mest), d_Cx, d_rhs, stream))

    def term_chi2(self):
        out = np.empty(self.n_terms)
        self._chk(self.lib.mc_get_term_chi2(self.ctx, _dp(out)))
        return out

    def apply_update(self, dx):
        dx = _f64(dx)
        assert dx.size == self.j_cols
        self._chk(self.lib.mc_apply_update(self.ctx, _dp(dx)))

    def revert_update(self):
        self._chk(self.lib.mc_revert_update(self.ctx))

    def dv_params(self, dv):
        out = np.empty(16)
        n = self._chk(self.lib.mc_get_dv_params(self.ctx, dv, _dp(out)))
        return out[:n].copy()

    def lidar_find_points_on_targets(self, pose_spline, dv_toff, dv_q_lb, dv_t_lb, dv_q_pw, dv_t_pw, plane_min, plane_max, threshold,
                                     points, want_world=False):
        """LiDAR.findPointsOnTarget at the current state.  Returns (kept, inliers, world): `kept` = indices of the readings
        whose time lies inside the spline (the reference's filtered lidarData), `inliers[p]` = indices INTO `kept` of the
        readings on target p (what TargetObservation.inliers holds), world = kept readings in the world frame or None."""
        pts = _f64(points).reshape(-1, 4)
        q, t = _i32(dv_q_pw), _i32(dv_t_pw)
        lo, hi = _f64(plane_min).reshape(-1, 3), _f64(plane_max).reshape(-1, 3)
        mask = np.zeros(pts.shape[0], np.uint32)
        world = np.zeros((pts.shape[0], 3)) if want_world else None
        self._chk(self.lib.mc_lidar_find_points_on_targets(self.ctx, pose_spline, dv_toff, dv_q_lb, dv_t_lb, q.size, _ip(q), _ip(t),
                                                           _dp(lo), _dp(hi), float(threshold), pts.shape[0], _dp(pts),
                                                           mask.ctypes.data_as(C.POINTER(C.c_uint)), _dp(world)))
        kept = np.nonzero(mask >> 31)[0]
        km = mask[kept]
        inliers = [np.nonzero((km >> p) & 1)[0] for p in range(q.size)]
        return kept, inliers, (world[kept] if want_world else None)

    def check_status(self):
        s = self._chk(self.lib.mc_device_status(self.ctx))
        if s:
            raise BackendError(self.lib.mc_last_error(self.ctx).decode())

    def jt_nnz(self):
        n = C.c_longlong()
        rc = self.lib.mc_jt_nnz(self.ctx, C.byref(n))
        if rc == -2:
            raise BackendError("This is currently unsupported")      # BSplineMotionError::evaluateJacobians
        self._chk(rc)
        return n.value

    def jt_pattern(self):
        nnz = self.jt_nnz()
        cp, ri = np.zeros(self.j_rows + 1, np.int32), np.zeros(nnz, np.int32)
        self._chk(self.lib.mc_get_jt_pattern(self.ctx, cp.ctypes.data_as(C.POINTER(C.c_int)), ri.ctypes.data_as(C.POINTER(C.c_int))))
        return cp, ri

    def build_jacobian_transpose(self, use_mest=True):
        out = np.zeros(self.jt_nnz())
        self._chk(self.lib.mc_build_jacobian_transpose(self.ctx, int(use_mest), _dp(out)))
        self.check_status()
        return out

    def shared_indices(self):
        n = self.lib.mc_shared_count(self.ctx)
        out = np.zeros(max(n, 0), np.int32)
        if n > 0:
            self._chk(self.lib.mc_get_shared_indices(self.ctx, out.ctypes.data_as(C.POINTER(C.c_int))))
        return out

    def pack_shared(self, d_cx, d_buf, stream=None):
        self._chk(self.lib.mc_pack_shared(self.ctx, C.c_void_p(d_cx), C.c_void_p(d_buf), C.c_void_p(stream or 0)))

    def unpack_shared(self, d_buf, d_cx, stream=None):
        self._chk(self.lib.mc_unpack_shared(self.ctx, C.c_void_p(d_buf), C.c_void_p(d_cx), C.c_void_p(stream or 0)))

    def shared_rhs_indices(self):
        n = self.lib.mc_shared_rhs_count(self.ctx)
        out = np.zeros(max(n, 0), np.int32)
        if n > 0:
            self._chk(self.lib.mc_get_shared_rhs_indices(self.ctx, out.ctypes.data_as(C.POINTER(C.c_int))))
        return out

    def pack_shared_rhs(self, d_rhs, d_buf, stream=None):
        self._chk(self.lib.mc_pack_shared_rhs(self.ctx, C.c_void_p(d_rhs), C.c_void_p(d_buf), C.c_void_p(stream or 0)))

    def unpack_shared_rhs(self, d_buf, d_rhs, stream=None):
        self._chk(self.lib.mc_unpack_shared_rhs(self.ctx, C.c_void_p(d_buf), C.c_void_p(d_rhs), C.c_void_p(stream or 0)))

    LAUNCH_NAMES = ("k_reproj", "k_static", "k_stream<Imu>", "k_stream<Lidar>", "k_motion", "k_stream<ImuSm>",
                    "k_small<GyroPrior>", "k_small<Transformation>")

    def launch_info(self):
        """[(kernel name, batches, terms, workgroups, flops_dense, flops_live, record_bytes)] per kernel launch of a build"""
        out = []
        for i in range(self.lib.mc_num_launches(self.ctx)):
            nb, n, g = C.c_int(), C.c_int(), C.c_int()
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            self._chk(self.lib.mc_launch_info(self.ctx, i, C.byref(nb), C.byref(n), C.byref(g)))
            self._chk(self.lib.mc_launch_counts(self.ctx, i, C.byref(a), C.byref(b), C.byref(c)))
            out.append((self.LAUNCH_NAMES[i], nb.value, n.value, g.value, a.value, b.value, c.value))
        return out

    def build_system_timed(self, use_mest=True, build=True):
        ms = np.zeros(self.lib.mc_num_launches(self.ctx), np.float32)
        self._chk(self.lib.mc_build_system_timed(self.ctx, int(use_mest), int(build), ms.ctypes.data_as(C.POINTER(C.c_float))))
        return ms

    def phase_cycles(self, use_mest=True, build=1):
        out = (C.c_ulonglong * 24)()
        self._chk(self.lib.mc_phase_cycles(self.ctx, int(use_mest), int(build), out))
        return np.array(list(out), dtype=np.float64)

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, gib = C.c_int(), C.c_double()
        self._chk(self.lib.mc_device_info(self.ctx, name, 256, C.byref(ncu), C.byref(gib)))
        return name.value.decode(), ncu.value, gib.value


class _DevicePtr(object):
    """n doubles of device memory at `ptr` as seen through the CUDA array interface (torch.as_tensor aliases it)"""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(ptr), False), version=2, strides=None)


def torch_collectives(device, through_host=False):
    """(allreduce, broadcast) for HipApi.optimize_sharded from torch.distributed's default process group: RCCL on the device
    buffer itself (backend "nccl"), or -- through_host=True, for a gloo group (CPU rehearsal of the N-rank path on one
    GPU) -- staged through host memory."""
    import torch
    import torch.distributed as dist

    def allreduce(ptr, n):
        t = torch.as_tensor(_DevicePtr(ptr, n), device=device)
        if through_host:
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
            torch.cuda.synchronize(device)
        else:
            dist.all_reduce(t)
            torch.cuda.synchronize(device)

    def broadcast(ptr, n, root):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(int(n),))
        if through_host:
            t = torch.from_numpy(a)
            dist.broadcast(t, src=root)
        else:
            t = torch.from_numpy(a).to(device)
            dist.broadcast(t, src=root)
            a[:] = t.cpu().numpy()
    return allreduce, broadcast


def host_threads():
    """threads of the host factorisation: MC_CHOLESKY_THREADS, else the CPUs this process may run on, at most 16"""
    import os
    v = os.environ.get("MC_CHOLESKY_THREADS")
    if v:
        return max(1, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


# ------------------------------------------------------------------------------------------------
class HostSparseCholesky(object):
    """sparse_block_matrix::LinearSolverCholmod::solve (linear_solver_cholmod.h:70-112) on the host: the C++ block
    Cholesky of multical_amd/csrc/host/block_cholesky.hpp through its C entry points (CHOLMOD is not in this image).
    Symbolic analysis once per pattern; `solve` returns None when the matrix is not positive definite."""

    def __init__(self):
        self.lib = _lib.load()
        self.h = self.lib.mc_cholesky_create()
        self._pattern = None

    def __del__(self):
        try:
            if self.h:
                self.lib.mc_cholesky_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def analyze(self, n, Cp, Ci, block_base=None):
        Cp, Ci = _i32(Cp), _i32(Ci)
        bb = None if block_base is None else _i32(block_base)
        rc = self.lib.mc_cholesky_analyze(self.h, n, _ip(Cp), _ip(Ci), 0 if bb is None else bb.size - 1, _ip(bb))
        if rc != 0:
            raise BackendError(self.lib.mc_cholesky_last_error(self.h).decode())
        self._pattern = (n, Cp.ctypes.data, Cp.size, Ci.size)
        self._keep = (Cp, Ci)

    def set_block_coordinates(self, coord, threads=None):
        """time-axis dissection: independent parts factorised / solved by one thread each (see multical_hip.h)"""
        coord = _f64(coord)
        if threads is None:
            threads = host_threads()
        rc = self.lib.mc_cholesky_set_block_coordinates(self.h, _dp(coord), int(threads))
        if rc != 0:
            raise BackendError(self.lib.mc_cholesky_last_error(self.h).decode())

    def stats(self):
        nd, nnz, fl = C.c_int(), C.c_longlong(), C.c_double()
        self.lib.mc_cholesky_stats(self.h, C.byref(nd), C.byref(nnz), C.byref(fl))
        return dict(dense_blocks=nd.value, factor_nnz=nnz.value, factor_flops=fl.value)

    def covariance_blocks(self, pairs, block_base):
        """blocks (i, j) of the inverse of the last factorised matrix (LinearSolverCholmod::solvePattern)"""
        bb = _i32(block_base)
        bi, bj = _i32([p[0] for p in pairs]), _i32([p[1] for p in pairs])
        sizes = (bb[bi + 1] - bb[bi]).astype(np.int64) * (bb[bj + 1] - bb[bj])
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        out = np.zeros(int(off[-1]))
        rc = self.lib.mc_cholesky_covariance_blocks(self.h, bi.size, _ip(bi), _ip(bj), off.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(out))
        if rc != 0:
            raise BackendError(self.lib.mc_cholesky_last_error(self.h).decode())
        return [out[off[k]:off[k + 1]].reshape(bb[bi[k] + 1] - bb[bi[k]], bb[bj[k] + 1] - bb[bj[k]]) for k in range(bi.size)]

    def solve(self, n, Cp, Ci, Cx, b, block_base=None):
        if self._pattern is None or self._pattern[0] != n or self._pattern[3] != len(Ci):
            self.analyze(n, Cp, Ci, block_base)
        Cx, b = _f64(Cx), _f64(b)
        rc = self.lib.mc_cholesky_factorize(self.h, _dp(Cx))
        if rc == 1:
            return None
        if rc != 0:
            raise BackendError(self.lib.mc_cholesky_last_error(self.h).decode())
        x = np.empty(n)
        self.lib.mc_cholesky_solve(self.h, _dp(b), _dp(x))
        return x if np.all(np.isfinite(x)) else None


class DeviceSchurSolver(object):
    """The factorisation's GPU stage (multical_amd/csrc/mc_schur.hip through mc_schur_*): the interiors of the pose-spline
    band are eliminated on the device, the reduced system is factorised on the host.  `planned` is False when the problem
    has no band the kernels take (`why` says which); the caller keeps HostSparseCholesky then."""

    def __init__(self, api, d_Cx, parts=0):
        self.lib = api.lib
        self.n = api.j_cols
        self.h = self.lib.mc_schur_create(api.ctx, d_Cx, int(parts))
        if not self.h:
            raise BackendError("mc_schur_create failed")
        self.planned = bool(self.lib.mc_schur_planned(self.h))
        self.why = "" if self.planned else self.lib.mc_schur_last_error(self.h).decode()

    def __del__(self):
        try:
            if self.h:
                self.lib.mc_schur_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self, d_Cx, d_rhs, diag_shift=0.0, stream=None):
        """(H + diag_shift I) x = rhs; returns (code, x): 0 ok, 1 not positive definite, 2 plan again"""
        x = np.zeros(self.n)
        rc = self.lib.mc_schur_solve(self.h, d_Cx, d_rhs, float(diag_shift), _dp(x), stream)
        if rc < 0:
            raise BackendError(self.lib.mc_schur_last_error(self.h).decode())
        return rc, x

    def info(self):
        c = np.zeros(10, np.int32)
        t = np.zeros(9)
        if self.lib.mc_schur_info(self.h, _ip(c), _dp(t)) != 0:
            return None
        keys = ("interiors", "r", "block_dim", "columns", "reduced_columns", "reduced_nnz", "watched_zero_blocks", "largest_boundary",
                "largest_interior", "reduced_factor_nnz")
        out = dict(zip(keys, [int(v) for v in c]))
        out["seconds"] = dict(zip(("gather", "eliminate", "products", "read_back", "host_reduced", "back_substitution"), [float(v) for v in t[:6]]))
        out["plan_seconds"] = float(t[6])
        out["work_bytes"] = float(t[7])
        out["reduced_factor_flops"] = float(t[8])
        return out


class HipBlockCholeskyLinearSystemSolver(object):
    """LinearSystemSolver backend: same virtual surface as BlockCholeskyLinearSystemSolver."""

    def __init__(self, api):
        self.api = api
        self._host = HostSparseCholesky()
        self._useDiagonalConditioner = False
        self._diag = None
        self._Cx = None
        self._rhs = None
        self._e = None

    def name(self):
        return "hip_block_cholesky"

    # LinearSystemSolver::initMatrixStructure -- the design-variable / error-term lists live in the context
    def initMatrixStructure(self, useDiagonalConditioner):
        if not self.api.finalized:
            self.api.finalize()
        self._useDiagonalConditioner = bool(useDiagonalConditioner)
        self._Cp, self._Ci = self.api.pattern()
        n = self.api.j_cols
        self._diag_pos = self._Cp[1:] - 1          # last entry of every upper-triangular column = diagonal
        self._host.analyze(n, self._Cp, self._Ci, self.api.block_bases())
        self._host.set_block_coordinates(self.api.block_times())
        self._diag = np.zeros(n)
        self._Cx = np.zeros(self.api.nnz)
        self._rhs = np.zeros(n)

    def JRows(self):
        return self.api.j_rows

    def JCols(self):
        return self.api.j_cols

    def evaluateError(self, nThreads=1, useMEstimator=True):
        J, self._e = self.api.evaluate_error(useMEstimator, want_e=True)
        return J

    def buildSystem(self, nThreads=1, useMEstimator=True):
        self._Cx, self._rhs = self.api.build_system(useMEstimator)

    def setConstantConditioner(self, diag):
        self._diag = np.full(self.api.j_cols, float(diag))

    def setConditioner(self, diag):
        diag = _f64(diag)
        if diag.size != self.api.j_cols:
            raise BackendError("The diagonal conditioner must have the same number of rows as the Hessian matrix")
        self._diag = diag.copy()

    def solveSystem(self):
        """Returns (success, dx).  Reproduces the augment-by-lambda^2 / un-augment-by-lambda sequence of
        BlockCholeskyLinearSystemSolver.cpp:76-98 on the host copy of H."""
        if self._useDiagonalConditioner:
            self._Cx[self._diag_pos] += self._diag * self._diag
        x = self._host.solve(self.api.j_cols, self._Cp, self._Ci, self._Cx, self._rhs)
        if self._useDiagonalConditioner:
            self._Cx[self._diag_pos] -= self._diag
        if x is None:
            return False, np.zeros(self.api.j_cols)
        return True, x

    def computeCovarianceBlocks(self, blockIndices):
        """BlockCholeskyLinearSystemSolver::computeCovarianceBlocks (:122-148)"""
        if self._useDiagonalConditioner:
            self._Cx[self._diag_pos] += self._diag * self._diag
        rc = self._host.lib.mc_cholesky_factorize(self._host.h, _dp(self._Cx))
        if self._useDiagonalConditioner:
            self._Cx[self._diag_pos] -= self._diag
        if rc != 0:
            raise BackendError("Unable to retrieve covariance")
        return self._host.covariance_blocks(blockIndices, self.api.block_bases())

    def rhs(self):
        return self._rhs

    def e(self):
        return self._e

    def Hessian(self):
        import scipy.sparse as sp
        n = self.api.j_cols
        U = sp.csc_matrix((self._Cx, self._Ci, self._Cp), shape=(n, n))
        return U

    def rhsJtJrhs(self):
        import scipy.sparse as sp
        U = self.Hessian()
        A = U + sp.triu(U, 1).T
        return float(self._rhs @ (A @ self._rhs))


class HipSparseCholeskyLinearSystemSolver(object):
    """Same surface as SparseCholeskyLinearSystemSolver (aslam_backend/src/SparseCholeskyLinearSystemSolver.cpp): keeps
    J^T in compressed-column form (built on the GPU), rhs = J^T e, and factors J^T J (+ conditioner^2) on the host."""

    def __init__(self, api):
        self.api = api
        self._useDiagonalConditioner = False
        self._diag = None
        self._values = None
        self._rhs = None
        self._e = None

    def name(self):
        return "hip_sparse_cholesky"

    def initMatrixStructure(self, useDiagonalConditioner):
        if not self.api.finalized:
            self.api.finalize()
        self._useDiagonalConditioner = bool(useDiagonalConditioner)
        self._cp, self._ri = self.api.jt_pattern()
        self._diag = np.ones(self.api.j_cols)
        self._rhs = np.zeros(self.api.j_cols)

    def JRows(self):
        return self.api.j_rows

    def JCols(self):
        return self.api.j_cols

    def evaluateError(self, nThreads=1, useMEstimator=True):
        J, self._e = self.api.evaluate_error(useMEstimator, want_e=True)
        return J

    def Jacobian(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self._values, self._ri, self._cp), shape=(self.api.j_cols, self.api.j_rows))   # J^T

    def buildSystem(self, nThreads=1, useMEstimator=True):
        self._values = self.api.build_jacobian_transpose(useMEstimator)
        self._rhs = self.Jacobian() @ self._e                      # J_transpose.rightMultiply(_e, _rhs)

    def setConstantConditioner(self, diag):
        self._diag = np.full(self.api.j_cols, float(diag))

    def setConditioner(self, diag):
        self._diag = _f64(diag).copy()

    def solveSystem(self):
        """cholmod factors [J^T diag] [J^T diag]^T = J^T J + diag^2 (pushDiagonalBlock, :56-58)."""
        import scipy.sparse as sp
        Jt = self.Jacobian()
        A = (Jt @ Jt.T).tocsc()
        if self._useDiagonalConditioner:
            A = A + sp.diags(self._diag * self._diag)
        U = sp.triu(A).tocsc()
        x = HostSparseCholesky().solve(self.api.j_cols, U.indptr, U.indices, U.data, self._rhs)
        if x is None:
            return False, np.zeros(self.api.j_cols)
        return True, x

    def rhs(self):
        return self._rhs

    def e(self):
        return self._e

    def rhsJtJrhs(self):
        return float(np.sum((self.Jacobian().T @ self._rhs) ** 2))   # |J rhs|^2 (leftMultiply, :121-126)


class LevenbergMarquardtTrustRegionPolicy(object):
    """Selector for Optimizer2Options.trustRegionPolicy (aslam_backend LevenbergMarquardtTrustRegionPolicy(lambdaInit))."""

    def __init__(self, lambdaInit=1e-3):
        self.lambdaInit = float(lambdaInit)

    def name(self):
        return "levenberg_marquardt"


class GaussNewtonTrustRegionPolicy(object):
    def name(self):
        return "gauss_newton"


class Optimizer2Options(object):
    """aslam::backend::Optimizer2Options as the Python front-end fills it (calibrator.py:27-58)."""

    def __init__(self):
        self.convergenceDeltaX = 1e-3
        self.convergenceJDescentRatioThreshold = 1e-5
        self.maxIterations = 20
        self.levenbergMarquardtLambdaInit = 1e-3
        self.verbose = False
        self.nThreads = 4
        self.linearSolver = None
        self.trustRegionPolicy = None


class SolutionReturnValue(object):
    def __init__(self, r):
        self.JStart, self.JFinal, self.dXFinal, self.dJFinal = r["JStart"], r["JFinal"], r["dXFinal"], r["dJFinal"]
        self.iterations, self.failedIterations = r["iterations"], r["failedIterations"]
        self.linearSolverFailure = bool(r["linearSolverFailure"])
        self.timings = {k: v for k, v in r.items() if k.startswith("seconds_")}


class Optimizer2(object):
    """`aopt.Optimizer2(options).optimize()` as calibrator.py:41-46 calls it.  The loop itself is C++ (mc_optimize,
    csrc/host/lm_driver.hpp); this class only translates the options object."""

    def __init__(self, options):
        self._options = options

    def optimize(self):
        o = self._options
        solver = o.linearSolver
        if solver is None or not hasattr(solver, "api"):
            raise BackendError("Optimizer2Options.linearSolver must be a Hip*LinearSystemSolver")
        policy = o.trustRegionPolicy or LevenbergMarquardtTrustRegionPolicy(o.levenbergMarquardtLambdaInit)
        kw = dict(maxIterations=int(o.maxIterations), convergenceDeltaX=float(o.convergenceDeltaX),
                  convergenceJDescentRatioThreshold=float(o.convergenceJDescentRatioThreshold), verbose=int(bool(o.verbose)),
                  linearSolver=1 if isinstance(solver, HipSparseCholeskyLinearSystemSolver) else 0)
        if isinstance(policy, GaussNewtonTrustRegionPolicy):
            kw["trustRegionPolicy"] = 1
        else:
            kw["trustRegionPolicy"] = 0
            kw["levenbergMarquardtLambdaInit"] = float(policy.lambdaInit)
        return SolutionReturnValue(solver.api.optimize(**kw))
