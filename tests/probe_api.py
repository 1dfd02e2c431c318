"""ctypes adapter over tests/hostprobe/libhostprobe.so: the product's term closed forms (multical_amd/csrc/mc_terms.h) and
structure analysis (mc_core.hpp) compiled with g++ for CPU-side checks.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle_api import _d, _i, _f64, _i32, ROOT

PROBE_DIR = os.path.join(ROOT, "tests", "hostprobe")
PROBE_LIB = os.path.join(PROBE_DIR, "libhostprobe.so")
_L = None


def _load():
    global _L
    if _L is None:
        if not os.path.exists(PROBE_LIB):
            subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wno-unknown-pragmas",
                                   "-o", PROBE_LIB, os.path.join(PROBE_DIR, "hostprobe.cpp")])
        _L = C.CDLL(PROBE_LIB)
        _L.probe_create.restype = C.c_void_p
        _L.probe_last_error.restype = C.c_char_p
        _L.probe_last_error.argtypes = [C.c_void_p]
    return _L


class ProbeApi(object):
    def __init__(self):
        self.L = _load()
        self.h = C.c_void_p(self.L.probe_create())
        self.finalized = False

    def __del__(self):
        try:
            self.L.probe_destroy(self.h)
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError("probe: " + self.L.probe_last_error(self.h).decode())
        return rc

    def add_design_variable(self, kind, params, active=True, scaling=1.0):
        p = _f64(params)
        return self._chk(self.L.probe_add_design_variable(self.h, C.c_int(kind), _d(p), C.c_int(p.size), C.c_int(int(bool(active))),
                                                          C.c_double(scaling)))

    def add_spline(self, order, dim, knots, coeffs, active=True):
        k = _f64(knots)
        c = np.ascontiguousarray(_f64(coeffs).T)
        return self._chk(self.L.probe_add_spline(self.h, C.c_int(order), C.c_int(dim), _d(k), C.c_int(k.size), _d(c),
                                                 C.c_int(c.shape[0]), C.c_int(int(bool(active)))))

    def add_camera(self, m, d, a, b=-1):
        return self._chk(self.L.probe_add_camera(self.h, C.c_int(m), C.c_int(d), C.c_int(a), C.c_int(b)))

    def add_reprojection_batch(self, cam, spline, dq, dt, dtoff, ot, oq, otv, of, y, p, s, pad, m=0, ma=0.0):
        ot, y, p, s = _f64(ot), _f64(y), _f64(p), _f64(s)
        oq, otv, of = _i32(oq), _i32(otv), _i32(of)
        self._chk(self.L.probe_add_reprojection_batch(self.h, C.c_int(cam), C.c_int(spline), C.c_int(dq), C.c_int(dt), C.c_int(dtoff),
                                                      C.c_int(ot.size), _d(ot), _i(oq), _i(otv), _i(of), _d(y), _d(p), _d(s),
                                                      C.c_double(pad), C.c_int(m), C.c_double(ma)))

    def add_static_reprojection_batch(self, cam, cq, ct, oq, ot, of, y, p, s, m=0, ma=0.0):
        y, p, s = _f64(y), _f64(p), _f64(s)
        cq, ct, oq, ot, of = _i32(cq), _i32(ct), _i32(oq), _i32(ot), _i32(of)
        self._chk(self.L.probe_add_static_reprojection_batch(self.h, C.c_int(cam), C.c_int(cq.size), _i(cq), _i(ct), C.c_int(oq.size),
                                                             _i(oq), _i(ot), _i(of), _d(y), _d(p), _d(s), C.c_int(m), C.c_double(ma)))

    def add_gyroscope_prior_batch(self, dq, db, wp, me, s, m=0, ma=0.0):
        wp, me, s = _f64(wp), _f64(me), _f64(s)
        self._chk(self.L.probe_add_gyroscope_prior_batch(self.h, C.c_int(dq), C.c_int(db), C.c_int(s.size), _d(wp), _d(me), _d(s),
                                                         C.c_int(m), C.c_double(ma)))

    def add_transformation_batch(self, qa, ta, qb, tb, pq, pt, wr, wt, m=0, ma=0.0):
        pq, pt = _f64(pq), _f64(pt)
        self._chk(self.L.probe_add_transformation_batch(self.h, C.c_int(qa), C.c_int(ta), C.c_int(qb), C.c_int(tb),
                                                        C.c_int(pt.size // 3), _d(pq), _d(pt), C.c_double(wr), C.c_double(wt),
                                                        C.c_int(m), C.c_double(ma)))

    def add_accelerometer_batch(self, ps, bs, db, dg, dq, dr, t, me, s, m=0, ma=0.0):
        t, me, s = _f64(t), _f64(me), _f64(s)
        self._chk(self.L.probe_add_accelerometer_batch(self.h, C.c_int(ps), C.c_int(bs), C.c_int(db), C.c_int(dg), C.c_int(dq),
                                                       C.c_int(dr), C.c_int(t.size), _d(t), _d(me), _d(s), C.c_int(m), C.c_double(ma)))

    def add_gyroscope_batch(self, ps, bs, db, dq, t, me, s, m=0, ma=0.0):
        t, me, s = _f64(t), _f64(me), _f64(s)
        self._chk(self.L.probe_add_gyroscope_batch(self.h, C.c_int(ps), C.c_int(bs), C.c_int(db), C.c_int(dq), C.c_int(t.size), _d(t),
                                                   _d(me), _d(s), C.c_int(m), C.c_double(ma)))

    def add_matrix_design_variable(self, A, pattern, active=True, scaling=1.0):
        A, pattern = _f64(A).reshape(9), _i32(pattern).reshape(9)
        return self._chk(self.L.probe_add_matrix_design_variable(self.h, _d(A), _i(pattern), C.c_int(int(bool(active))),
                                                                 C.c_double(scaling)))

    def add_accelerometer_batch_sm(self, ps, bs, db, dg, dq, dr, dM, se, t, me, s, m=0, ma=0.0):
        t, me, s = _f64(t), _f64(me), _f64(s)
        se = None if se is None else _i32(se)
        self._chk(self.L.probe_add_accelerometer_batch_sm(self.h, C.c_int(ps), C.c_int(bs), C.c_int(db), C.c_int(dg), C.c_int(dq),
                                                          C.c_int(dr), C.c_int(dM), None if se is None else _i(se), C.c_int(t.size),
                                                          _d(t), _d(me), _d(s), C.c_int(m), C.c_double(ma)))

    def add_gyroscope_batch_sm(self, ps, bs, db, dq, dqg, dM, dMa, dg, dr, t, me, s, m=0, ma=0.0):
        t, me, s = _f64(t), _f64(me), _f64(s)
        self._chk(self.L.probe_add_gyroscope_batch_sm(self.h, C.c_int(ps), C.c_int(bs), C.c_int(db), C.c_int(dq), C.c_int(dqg),
                                                      C.c_int(dM), C.c_int(dMa), C.c_int(dg), C.c_int(dr), C.c_int(t.size), _d(t),
                                                      _d(me), _d(s), C.c_int(m), C.c_double(ma)))

    def add_lidar_batch(self, sp, dtoff, dq, dt, t, dr, rg, s, q, tv, pad, m=0, ma=0.0):
        t, dr, rg, s = _f64(t), _f64(dr), _f64(rg), _f64(s)
        q, tv = _i32(q), _i32(tv)
        self._chk(self.L.probe_add_lidar_batch(self.h, C.c_int(sp), C.c_int(dtoff), C.c_int(dq), C.c_int(dt), C.c_int(t.size), _d(t),
                                               _d(dr), _d(rg), _d(s), _i(q), _i(tv), C.c_double(pad), C.c_int(m), C.c_double(ma)))

    def add_motion_term(self, sp, W, order):
        W = _f64(W)
        self._chk(self.L.probe_add_motion_term(self.h, C.c_int(sp), _d(W), C.c_int(order)))

    def set_last_batch_landmarks(self, dv_landmark):
        dv = _i32(dv_landmark)
        self._chk(self.L.probe_set_last_batch_landmarks(self.h, C.c_int(dv.size), _i(dv)))

    def set_last_batch_inv_r(self, invR):
        """sqrtInvR from the product's own mc_compute_matrix_sqrt (the host library needs no GPU for it)"""
        from multical_amd._lib import load
        lib = load()
        invR = _f64(invR)
        S = np.empty_like(invR)
        for k in range(invR.shape[0]):
            A = np.ascontiguousarray(invR[k]); o = np.empty_like(A)
            assert lib.mc_compute_matrix_sqrt(A.shape[0], A.ctypes.data_as(C.POINTER(C.c_double)), o.ctypes.data_as(C.POINTER(C.c_double))) == 0
            S[k] = o
        self._chk(self.L.probe_set_last_batch_sqrt_inv_r(self.h, C.c_int(S.shape[1]), C.c_int(S.shape[0]), _d(S)))

    def set_shard(self, rank, world):
        self.L.probe_set_shard(self.h, C.c_int(rank), C.c_int(world))

    def finalize(self):
        self._chk(self.L.probe_finalize(self.h))
        v = [C.c_int() for _ in range(5)]
        self.L.probe_dims(self.h, *[C.byref(x) for x in v])
        self.j_rows, self.j_cols, self.n_blocks, self.n_terms, self.nnz = [x.value for x in v]
        self.finalized = True

    def pattern(self):
        Cp = np.empty(self.j_cols + 1, np.int32)
        Ci = np.empty(self.nnz, np.int32)
        self.L.probe_get_pattern(self.h, _i(Cp), _i(Ci))
        return Cp, Ci

    def block_bases(self):
        out = np.empty(self.n_blocks + 1, np.int32)
        self.L.probe_get_block_bases(self.h, _i(out))
        return out

    def schur_solve(self, Cx, g, shift=0.0, parts=4, Cx_plan=None):
        """(H + shift I) x = g through the interior elimination of host/schur_plan.hpp (reference executor) and the host
        factorisation of the reduced system.  Returns (code, x, info)."""
        x = np.zeros(self.j_cols)
        info = np.zeros(6, np.int32)
        Cx = np.ascontiguousarray(Cx, np.float64)
        g = np.ascontiguousarray(g, np.float64)
        self.L.probe_schur_solve.restype = C.c_int
        if Cx_plan is not None:
            Cx_plan = np.ascontiguousarray(Cx_plan, np.float64)
        rc = self.L.probe_schur_solve(self.h, _d(Cx_plan) if Cx_plan is not None else None, _d(Cx), _d(g), C.c_double(shift),
                                      C.c_int(parts), _d(x), _i(info))
        if rc < 0:
            self._chk(rc)
        return rc, x, info

    def block_times(self):
        out = np.empty(self.n_blocks)
        self.L.probe_get_block_times(self.h, _d(out))
        return out

    def term(self, t, use_mest=True):
        J = np.zeros((6, self.j_cols))
        e = np.zeros(6)
        c = C.c_double()
        d = self._chk(self.L.probe_term(self.h, C.c_int(t), C.c_int(int(use_mest)), _d(J), _d(e), C.byref(c)))
        return J[:d], e[:d], c.value

    def cpu_optimised(self, use_mest=True, build=True, n_threads=1):
        """evaluateError (+ buildSystem) of the whole problem on the host with the product's formulation and OpenMP
        (tests/hostprobe/cpu_optimised.inc).  Returns (J, Cx, rhs); Cx / rhs are None for build=False."""
        Cx = np.zeros(self.nnz if build else 1)
        rhs = np.zeros(self.j_cols if build else 1)
        J = C.c_double()
        self._chk(self.L.probe_cpu_optimised(self.h, C.c_int(int(use_mest)), C.c_int(int(build)), C.c_int(int(n_threads)), _d(Cx), _d(rhs),
                                             C.byref(J)))
        return (J.value, Cx, rhs) if build else (J.value, None, None)

    def term_kind(self, t):
        return self.L.probe_term_kind(self.h, C.c_int(t))

    def shared_indices(self):
        n = self.L.probe_shared_count(self.h)
        out = np.zeros(n, np.int32)
        if n:
            self.L.probe_shared_indices(self.h, _i(out))
        return out

    def written_runs(self):
        """(first, count) runs of the Cx values and of the residual rows this rank's launches write."""
        a, b = C.c_int(), C.c_int()
        self.L.probe_written_counts(self.h, C.byref(a), C.byref(b))
        cx = np.zeros((max(1, a.value), 2), np.int64)
        e = np.zeros((max(1, b.value), 2), np.int64)
        self.L.probe_written(self.h, cx.ctypes.data_as(C.POINTER(C.c_longlong)), e.ctypes.data_as(C.POINTER(C.c_longlong)))
        return cx[:a.value], e[:b.value]

    def shared_rhs_indices(self):
        n = self.L.probe_shared_rhs_count(self.h)
        out = np.zeros(n, np.int32)
        if n:
            self.L.probe_shared_rhs_indices(self.h, _i(out))
        return out

    def my_terms(self):
        f = np.zeros(self.n_terms, np.int32)
        self.L.probe_my_terms(self.h, _i(f))
        return f.astype(bool)

    def dv_params(self, dv):
        out = np.zeros(16)
        n = self.L.probe_get_dv_params(self.h, C.c_int(int(dv)), _d(out))
        return out[:n].copy()

    def spline_first_dv(self, sid):
        return self.L.probe_spline_first_dv(self.h, C.c_int(int(sid)))

    def optimize(self, max_iterations, lambda0, dx_tol, ratio_tol, damped=True, n_threads=1, allreduce=None, broadcast=None):
        """Host twin of mc_optimize / mc_optimize_sharded (tests/hostprobe: the product's minimiser loop and factorisation over
        the g++ build of the term math).  allreduce(array) sums a float64 numpy array over the ranks in place,
        broadcast(array, root) sends it from rank root; both only needed for a sharded probe."""
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong)
        BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong, C.c_int)
        failure = []

        def _ar(user, ptr, n):
            try:
                allreduce(np.ctypeslib.as_array(ptr, shape=(int(n),)))
                return 0
            except Exception as ex:
                failure.append(ex)
                return -1

        def _bc(user, ptr, n, root):
            try:
                broadcast(np.ctypeslib.as_array(ptr, shape=(int(n),)), int(root))
                return 0
            except Exception as ex:
                failure.append(ex)
                return -1
        ar, bc = AR(_ar), BC(_bc)
        out = np.zeros(5)
        self.L.probe_optimize.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, AR, BC, C.c_void_p,
                                          C.POINTER(C.c_double)]
        rc = self.L.probe_optimize(self.h, int(max_iterations), float(lambda0), float(dx_tol), float(ratio_tol), int(bool(damped)),
                                   int(n_threads), ar, bc, None, _d(out))
        if failure:
            raise failure[0]
        self._chk(rc)
        return dict(JStart=out[0], JFinal=out[1], iterations=int(out[2]), failedIterations=int(out[3]), linearSolverFailure=int(out[4]))

