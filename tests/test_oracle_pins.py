"""Pins for the CPU oracle (and, where the function exists there, for the product's math compiled for the host).

The reference cannot be built or imported here and holds no known-answer vectors for this path, so the
oracle is pinned with what the reference DOES hold:

  1. golden vectors produced by the reference's own machine-generated expressions, evaluated from its source text where
     it lies (tests/golden/make_reference_vectors.py): RotationVector::angularVelocityAndJacobian's d(S p')/dp and
     EquidistantDistortion's distort / parameter Jacobians -- the two places where oracle and product replace a symbolic
     dump by an analytic form
  2. the identities its own tests assert, re-run here against the restatement:
       bsplines_python/test/BSplineTests.py:213-257       evalD == fitpack spalde, orders 2..9, uniform / repeated knots
       bsplines/test/BSplinePoseTests.cpp:265-321          curveQuadraticIntegralSparse == curveQuadraticIntegral (1e-10)
       aslam_backend/test/JacobianContainer.cpp:205-319    evaluateHessian blocks, upper triangle only, re-accumulation doubles
       aslam_backend/test/LinearSolverTests.cpp:18-141     compareSolvers on the SampleDvAndError.hpp system
  3. a 50-digit mpmath evaluation of one term per family from first principles (SURVEY.md 8c promised it)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle_api import OracleApi, _d, _i, _f64, _i32, ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "reference_generated_jacobians.npz")


def _olib():
    return OracleApi().lib


def _plib():
    import probe_api
    return probe_api._load()


# ------------------------------------------------------------------------------------------ 1. reference-generated vectors
def test_golden_vectors_are_what_the_reference_source_gives():
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference tree is not on this machine")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_reference_vectors.py"), "--check"])


def _rv_tol(p):
    # the generated expression divides differences of O(1) products by |p|^5: its own rounding error grows like eps/|p|^5
    ang = np.linalg.norm(p)
    return max(1e-12, 4e-16 / ang ** 5 * 50)


def test_rotation_vector_jacobian_against_the_reference_expression():
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    worst_o = worst_p = 0.0
    for k in range(g["rv_p"].shape[0]):
        th, v, D = _f64(g["rv_p"][k]), _f64(g["rv_pdot"][k]), g["rv_dSv_dp"][k]
        om, J = np.zeros(3), np.zeros((3, 6))
        o.orc_u_rv_angular_velocity_and_jacobian(_d(th), _d(v), _d(om), _d(J))
        om2, D2 = np.zeros(3), np.zeros((3, 3))
        p.probe_u_rotvec_dSv(_d(th), _d(v), _d(om2), _d(D2))
        scale, tol = np.abs(D).max(), _rv_tol(th)
        eo, ep = np.abs(J[:, :3] - D).max() / scale, np.abs(D2 - D).max() / scale
        assert eo <= tol and ep <= tol, (k, np.linalg.norm(th), eo, ep, tol)
        assert np.abs(om - om2).max() <= 1e-14 * max(1.0, np.abs(om).max())
        if np.linalg.norm(th) >= 0.3:
            worst_o, worst_p = max(worst_o, eo), max(worst_p, ep)
    assert worst_o <= 1e-12 and worst_p <= 1e-12, (worst_o, worst_p)


def test_equidistant_jacobians_against_the_reference_expression():
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    for k in range(g["eq_y"].shape[0]):
        y, kk = _f64(g["eq_y"][k]), _f64(g["eq_k"][k])
        r = np.linalg.norm(y)
        tol = max(1e-12, 1e-15 / r ** 2)          # the dump subtracts two terms of size 1/r^2 near the centre
        for lib, fn in ((o, "orc_u_distort"), (p, "probe_u_distort")):
            yd, Jy, Jk = np.zeros(2), np.zeros((2, 2)), np.zeros((2, 4))
            getattr(lib, fn)(C.c_int(2), _d(kk), _d(y), _d(yd), _d(Jy), _d(Jk))
            assert np.abs(Jy - g["eq_Jy"][k]).max() <= tol * np.abs(g["eq_Jy"][k]).max(), (fn, k, r)
            assert np.abs(Jk - g["eq_Jk"][k]).max() <= 1e-12 * np.abs(g["eq_Jk"][k]).max() + 1e-300, (fn, k, r)
            # value: y * thetad / r  (EquidistantDistortion.hpp:17-28)
            th = np.arctan(r)
            thd = th * (1 + kk[0] * th ** 2 + kk[1] * th ** 4 + kk[2] * th ** 6 + kk[3] * th ** 8)
            assert np.abs(yd - y * thd / r).max() <= 1e-15


CAMERA_PAIRS = [(0, 0, "pinhole_none"), (0, 1, "pinhole_radtan"), (0, 2, "pinhole_equi"), (1, 0, "omni_none"),
                (1, 1, "omni_radtan"), (1, 2, "omni_equi"), (2, 0, "ds_none")]


@pytest.mark.parametrize("model,dist,tag", CAMERA_PAIRS)
def test_camera_models_against_the_reference_source(model, dist, tag):
    """Rows E1, E2, E4, E5 (and E3 through the projections): keypoint, point Jacobian, intrinsics Jacobian and distortion
    Jacobian of every projection x distortion pair of north_star, oracle AND product, against what the reference's own
    straight-line code gives at the same inputs (PinholeProjection.hpp:123-141, 340-351; OmniProjection.hpp:147-177, 399-418;
    DoubleSphereProjection.hpp:166-217, 460-501; RadialTangentialDistortion.hpp:47-63, 162-180; EquidistantDistortion.hpp:17-28,
    54-170, 258-273 -- evaluated from the source text by tests/golden/make_reference_vectors.py)."""
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    pd = 4 + model
    n = g["cam_%s_p" % tag].shape[0]
    for k in range(n):
        proj, dk, pt = _f64(g["cam_%s_proj" % tag][k]), _f64(g["cam_%s_dist" % tag][k]), _f64(g["cam_%s_p" % tag][k])
        # the equidistant dump subtracts two terms of size 1 / r^2 near the optical axis (see the test above)
        r = np.hypot(pt[0], pt[1]) / pt[2]
        tol = max(1e-12, 1e-15 / r ** 2) if dist == 2 else 1e-12
        for lib, fn in ((o, "orc_u_project"), (p, "probe_u_project")):
            kp, Jp, Ji, Jd = np.zeros(2), np.zeros((2, 3)), np.zeros((2, pd)), np.zeros((2, 4))
            getattr(lib, fn)(C.c_int(model), C.c_int(dist), _d(proj), _d(dk), _d(pt), _d(kp), _d(Jp), _d(Ji), _d(Jd))
            for name, got in (("kp", kp), ("Jp", Jp), ("Ji", Ji), ("Jd", Jd)):
                want = g["cam_%s_%s" % (tag, name)][k]
                assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-300) + (0.0 if np.abs(want).max() > 0 else 1e-300), \
                    (fn, tag, name, k, got, want)


def test_radial_tangential_distortion_against_the_reference_source():
    """Row E2 on its own: distort, its Jacobian (symmetric by construction, J(0,1) = J(1,0)) and distortParameterJacobian."""
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    for k in range(g["rt_y"].shape[0]):
        y, kk = _f64(g["rt_y"][k]), _f64(g["rt_k"][k])
        for lib, fn in ((o, "orc_u_distort"), (p, "probe_u_distort")):
            yd, Jy, Jk = np.zeros(2), np.zeros((2, 2)), np.zeros((2, 4))
            getattr(lib, fn)(C.c_int(1), _d(kk), _d(y), _d(yd), _d(Jy), _d(Jk))
            assert np.abs(yd - g["rt_yd"][k]).max() <= 1e-14 * max(1.0, np.abs(g["rt_yd"][k]).max()), (fn, k)
            assert np.abs(Jy - g["rt_Jy"][k]).max() <= 1e-13 * np.abs(g["rt_Jy"][k]).max(), (fn, k)
            assert np.abs(Jk - g["rt_Jk"][k]).max() <= 1e-13 * np.abs(g["rt_Jk"][k]).max() + 1e-300, (fn, k)
            assert Jy[0, 1] == Jy[1, 0]


def test_quat2r_and_rodrigues_against_the_reference_source():
    """Rows C7 / D4: quat2r's nine element formulas (quaternion_algebra.cpp:90-98) and the Rodrigues matrix of
    RotationVector::parametersToRotationMatrix (RotationVector.cpp:31-43), oracle and product."""
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    for k in range(g["kin_q"].shape[0]):
        q, th = _f64(g["kin_q"][k]), _f64(g["kin_rv"][k])
        for lib, fq, fc in ((o, "orc_u_quat2r", "orc_u_rv_rotation_matrix"), (p, "probe_u_quat2r", "probe_u_rotvec_C")):
            R, Cm = np.zeros((3, 3)), np.zeros((3, 3))
            getattr(lib, fq)(_d(q), _d(R))
            getattr(lib, fc)(_d(th), _d(Cm))
            assert np.abs(R - g["kin_quat2r"][k]).max() <= 1e-15, (fq, k)
            assert np.abs(Cm - g["kin_rv_C"][k]).max() <= 1e-14, (fc, k, np.linalg.norm(th))


# ------------------------------------------------------------------------------------------ 2a. evalD == spalde
def test_update_quat_against_the_reference_source():
    """sm::kinematics::updateQuat (quaternion_algebra.cpp:302-318, with axisAngle2quat's regular branch :216-219): the state
    update of every RotationQuaternion and HomogeneousPoint design variable (row A4), oracle and product (the product's
    device kernel k_update calls the same update_quat)."""
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    for q, dq, want in zip(g["uq_q"], g["uq_dq"], g["uq_out"]):
        a, b = np.zeros(4), np.zeros(4)
        o.orc_u_update_quat(_d(q.copy()), _d(dq.copy()), _d(a))
        p.probe_u_update_quat(_d(q.copy()), _d(dq.copy()), _d(b))
        assert np.abs(a - want).max() <= 1e-15 and np.abs(b - want).max() <= 1e-15


def test_mestimator_weights_against_the_reference_source():
    """MEstimator::getWeight of the five robust kernels as MEstimatorPolicies.cpp:30-33, 46-49, 62-64, 77-79, 114-116 state
    them (golden vectors from the source text), oracle and product, on a 40 x 40 grid of squared errors and parameters."""
    g = np.load(GOLD)
    o, p = _olib(), _plib()
    kinds = dict(huber=1, cauchy=2, geman_mcclure=3, bz=4, blake_zisserman=5)
    e2 = np.ascontiguousarray(g["mest_e2"])
    for name, kind in kinds.items():
        want = g["mest_w_" + name]
        for j, a in enumerate(g["mest_a"]):
            wo, wp = np.zeros(e2.size), np.zeros(e2.size)
            o.orc_u_mest_weights(kind, C.c_double(float(a)), e2.size, _d(e2), _d(wo))
            p.probe_u_mest_weights(kind, C.c_double(float(a)), e2.size, _d(e2), _d(wp))
            assert np.abs(wo - want[:, j]).max() <= 1e-15 * max(1.0, np.abs(want[:, j]).max()), (name, j)
            assert np.abs(wp - want[:, j]).max() <= 1e-15 * max(1.0, np.abs(want[:, j]).max()), (name, j)


def _spline_case(order, knots, rng):
    kr = knots.size
    kc = kr - order                       # numCoefficientsRequired: knots - order
    return rng.random(kc)


@pytest.mark.parametrize("mode", ["uniform", "repeated", "random"])
def test_evald_equals_fitpack_spalde(mode):
    """BSplineTests.py:213-287 (test_uniform / test_repeated / test_random): every derivative of a 1-D spline of order
    2..9 on 3 segments equals scipy.interpolate.spalde."""
    import scipy.interpolate as fp
    o, p = _olib(), _plib()
    rng = np.random.default_rng({"uniform": 1, "repeated": 2, "random": 3}[mode])
    for order in range(2, 10):
        kr = 3 + 2 * order - 1            # numKnotsRequired(3 segments) = segments + 2 order - 1
        if mode == "uniform":
            knots = np.linspace(0.0, kr * 1.0, kr)
        elif mode == "repeated":
            knots = np.where(np.arange(kr) >= kr * 0.5, 1.0, 0.0)
        else:
            knots = np.sort(rng.random(kr) * 10)
        cp = _spline_case(order, knots, rng)
        t_min, t_max = knots[order - 1], knots[kr - order]
        # The reference's later "HACK" (BSpline.cpp:241-244) snaps |t_max - t| < 1e-10 onto t_max, whose knot segment is the
        # zero-width one when the end knots repeat (computeU then returns 0): its own last sample, t_max - 1e-15, no longer
        # evaluates the curve there.  The restatement keeps the hack; repeated knots stop at t_max - 1e-9.
        for t in np.linspace(t_min, t_max - (1e-9 if mode == "repeated" else 1e-15), 10):
            f = np.atleast_1d(fp.spalde(float(t), (knots, cp, order - 1)))
            for d in range(f.shape[0]):
                out = np.zeros(1)
                assert o.orc_u_spline_evald(order, kr, _d(knots), 1, cp.size, _d(cp), C.c_double(t), d, _d(out)) == 0
                assert abs(out[0] - f[d]) <= 1e-7 * max(1.0, abs(f[d])), (mode, order, t, d, out[0], f[d])   # assertAlmostEqual: 7 places
                if order == 6 and d <= 2 and mode != "repeated":
                    out2 = np.zeros(1)
                    assert p.probe_u_spline_evald(kr, _d(knots), 1, cp.size, _d(cp), C.c_double(t), d, _d(out2)) >= 0
                    assert abs(out2[0] - f[d]) <= 1e-7 * max(1.0, abs(f[d]))
                    assert abs(out2[0] - out[0]) <= 1e-13 * max(1.0, abs(out[0]))


def test_evald_on_the_multical_spline_shape():
    """order 6, dim 6, 100 knots/s: oracle and product against scipy.interpolate.BSpline derivatives at 1e-12."""
    import scipy.interpolate as fp
    o, p = _olib(), _plib()
    rng = np.random.default_rng(7)
    order, nseg = 6, 40
    dt = 0.01
    knots = (np.arange(nseg + 2 * order - 1) - (order - 1)) * dt
    nc = nseg + order - 1
    c = rng.standard_normal((nc, 6))       # coefficient-major == D x nc column major
    for t in rng.uniform(0.0, nseg * dt, 40):
        for d in range(3):
            want = np.array([fp.BSpline(knots, c[:, i], order - 1)(t, d) for i in range(6)])
            a, b = np.zeros(6), np.zeros(6)
            o.orc_u_spline_evald(order, knots.size, _d(knots), 6, nc, _d(c), C.c_double(t), d, _d(a))
            p.probe_u_spline_evald(knots.size, _d(knots), 6, nc, _d(c), C.c_double(t), d, _d(b))
            scale = max(1.0, np.abs(want).max())
            assert np.abs(a - want).max() <= 1e-11 * scale and np.abs(b - want).max() <= 1e-11 * scale, (t, d)


# ------------------------------------------------------------------------------------------ 2b. quadratic integral
def _gauss_quadratic_integral(o, order, knots, D, c, W, dorder):
    """the DEFINITION: Q = sum_segments int Phi_d(t)^T W Phi_d(t) dt by Gauss-Legendre (exact: polynomial of degree < 2 order)"""
    nc = c.shape[0]
    N = D * nc
    Q = np.zeros((N, N))
    xs, ws = np.polynomial.legendre.leggauss(order + 1)
    nseg = knots.size - 2 * order + 1
    for s in range(nseg):
        a, b = knots[order - 1 + s], knots[order + s]
        if b <= a:
            continue
        for x, w in zip(xs, ws):
            t = 0.5 * (a + b) + 0.5 * (b - a) * x
            phi, first = np.zeros((D, D * order)), C.c_int()
            assert o.orc_u_spline_phi(order, knots.size, _d(knots), D, nc, _d(c), C.c_double(t), dorder, _d(phi), C.byref(first)) == 0
            i0 = first.value * D
            Q[i0:i0 + D * order, i0:i0 + D * order] += 0.5 * (b - a) * w * (phi.T @ W @ phi)
    return Q


@pytest.mark.parametrize("dorder", [0, 1, 2])
def test_curve_quadratic_integral_sparse_equals_dense_and_the_definition(dorder):
    """BSplinePoseTests.cpp:265-321 (sparse == dense at 1e-10, orders 2..9, 20 segments, random symmetric W), plus the
    integral evaluated from its definition by quadrature."""
    o = _olib()
    rng = np.random.default_rng(11 + dorder)
    for order in range(max(2, dorder + 1), 10):
        nseg, D = 20, 6
        knots = np.concatenate([np.zeros(order - 1), np.sort(rng.uniform(0.0, 2.0, nseg + 1)), np.full(order - 1, 2.0)])
        knots[order - 1:order + nseg] = np.linspace(0.0, 2.0, nseg + 1) + np.concatenate([[0], rng.uniform(-0.02, 0.02, nseg - 1), [0]])
        knots[:order - 1] = knots[order - 1] - np.arange(order - 1, 0, -1) * 0.1
        knots[order + nseg:] = knots[order + nseg - 1] + np.arange(1, order) * 0.1
        nc = nseg + order - 1
        c = rng.standard_normal((nc, D))
        W = rng.uniform(-1, 1, (D, D))
        W = np.tril(W) + np.tril(W, -1).T
        N = D * nc
        Qd, Qs = np.zeros((N, N)), np.zeros((N, N))
        assert o.orc_u_curve_quadratic_integral(order, knots.size, _d(knots), D, nc, _d(c), _d(W), dorder, 0, _d(Qd)) == 0
        assert o.orc_u_curve_quadratic_integral(order, knots.size, _d(knots), D, nc, _d(c), _d(W), dorder, 1, _d(Qs)) == 0
        assert np.abs(Qd - Qs).max() <= 1e-10 * max(1.0, np.abs(Qd).max())
        Qg = _gauss_quadratic_integral(o, order, knots, D, c, W, dorder)
        assert np.abs(Qd - Qg).max() <= 1e-9 * np.abs(Qg).max(), (order, dorder)


@pytest.mark.parametrize("dim,dorder", [(3, 1), (6, 2)])
def test_product_motion_band_equals_the_oracle_integral(dim, dorder):
    """describeMotion (the product's BSplineMotionError band) == the oracle's curveQuadraticIntegral on a Multical-shaped
    spline (order 6; bias splines: dim 3 order 1, pose spline: dim 6 order 2)."""
    from probe_api import ProbeApi
    o = _olib()
    rng = np.random.default_rng(5)
    order, nseg = 6, 12
    knots = (np.arange(nseg + 2 * order - 1) - (order - 1)) * 0.2
    nc = nseg + order - 1
    c = rng.standard_normal((nc, dim))
    W = rng.uniform(-1, 1, (dim, dim))
    W = W @ W.T
    pr = ProbeApi()
    sid = pr.add_spline(order, dim, knots, c.T, True)
    pr.add_motion_term(sid, W, dorder)
    N = dim * nc
    Qp, Qo = np.zeros((N, N)), np.zeros((N, N))
    assert pr.L.probe_motion_q(pr.h, 0, _d(Qp), N) == 0
    assert o.orc_u_curve_quadratic_integral(order, knots.size, _d(knots), dim, nc, _d(c), _d(W), dorder, 1, _d(Qo)) == 0
    assert np.abs(Qp - Qo).max() <= 1e-12 * np.abs(Qo).max()


# ------------------------------------------------------------------------------------------ 2c. JacobianContainer identities
def _random_covariance(rng, n):
    A = rng.uniform(-1, 1, (n, n))
    return A @ A.T + 0.1 * np.eye(n)


def test_matrix_sqrt_reconstructs_the_covariance():
    """JacobianContainer.cpp:238-246: sqrtInvR sqrtInvR^T == invR (1e-9); and the LDLT structure: a row permutation of a
    lower-triangular matrix with positive diagonal."""
    o = _olib()
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 6):
        for _ in range(20):
            R = _random_covariance(rng, n) * rng.choice([1.0, 100.0])
            S = np.zeros((n, n))
            assert o.orc_u_matrix_sqrt(n, _d(R), _d(S)) == 0
            assert np.abs(S @ S.T - R).max() <= 1e-9 * np.abs(R).max()
            # some row order makes it lower triangular
            nz_first = [np.max(np.nonzero(np.abs(row) > 0)[0]) for row in S]
            assert sorted(nz_first) == list(range(n))
    # isotropic covariance -> s * I exactly (what Multical's terms pass)
    S = np.zeros((3, 3))
    o.orc_u_matrix_sqrt(3, _d(np.eye(3) * 4.0), _d(S))
    assert np.array_equal(S, 2.0 * np.eye(3))


def _linear_problem(rng, dims, terms, scale_invR=1.0):
    """dims: design-variable sizes; terms: list of (error dim, [dv indices]).  Returns (oracle, Js, ps, invRs)."""
    o = OracleApi()
    ids = [o.lib.orc_add_vector_dv(o.h, n, _d(rng.uniform(-1, 1, n)), 1) for n in dims]
    Js, invRs = [], []
    for dim, dvs in terms:
        J = [rng.uniform(-1, 1, (dim, dims[k])) for k in dvs]
        R = _random_covariance(rng, dim) * scale_invR
        p = rng.standard_normal(dim)
        flat = np.concatenate([j.ravel() for j in J])
        assert o.lib.orc_add_linear_error(o.h, dim, len(dvs), _i(_i32([ids[k] for k in dvs])), _d(flat), _d(p), _d(R)) >= 0
        Js.append(J)
        invRs.append(R)
    o.finalize()
    return o, Js, invRs


def _dense_upper(o):
    nnz = o.lib.orc_ccs_nnz(o.h)
    Cp, Ci, Cx, rhs = np.empty(o.j_cols + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz), np.empty(o.j_cols)
    o.lib.orc_get_ccs(o.h, _i(Cp), _i(Ci), _d(Cx), _d(rhs))
    H = np.zeros((o.j_cols, o.j_cols))
    H[Ci, np.repeat(np.arange(o.j_cols), np.diff(Cp))] = Cx
    return H, rhs


@pytest.mark.parametrize("scale", [100.0, 1.0])
def test_evaluate_hessian_block_identities(scale):
    """JacobianContainer.cpp:205-319 (testBuildHessian / testBuildHessian2): a 2-row term over a 3- and a 2-dimensional
    design variable; blocks (0,0), (0,1), (1,1) == Ji^T invR Jj at 1e-9, block (1,0) never allocated, rhs == -Ji^T invR e,
    and evaluating twice gives double the values."""
    rng = np.random.default_rng(17)
    o, Js, invRs = _linear_problem(rng, [3, 2], [(2, [0, 1])], scale)
    (J1, J2), R = Js[0], invRs[0]
    J1, J2 = -J1, -J2                       # the term's Jacobians are -J_i
    o.evaluate_error(False)
    e = np.zeros(2)
    Jd = np.zeros((2, 5))
    o.lib.orc_clear_system(o.h)
    o.lib.orc_accumulate_term(o.h, 0, 0)
    assert o.lib.orc_is_block_set(o.h, 0, 0) and o.lib.orc_is_block_set(o.h, 0, 1) and o.lib.orc_is_block_set(o.h, 1, 1)
    assert not o.lib.orc_is_block_set(o.h, 1, 0)
    H, rhs = _dense_upper(o)
    # raw error of the term: weighted error = sqrtInvR^T e -> recover e through the chi2 identity instead: use the API
    Jw, ew = o.term_weighted_jacobian(0, False)
    S = np.zeros((2, 2))
    o.lib.orc_u_matrix_sqrt(2, _d(R), _d(S))
    e = np.linalg.solve(S.T, ew)
    for k in (1, 2):
        assert np.abs(H[0:3, 0:3] - np.triu(k * J1.T @ R @ J1)).max() <= 1e-9 * scale
        assert np.abs(H[0:3, 3:5] - k * J1.T @ R @ J2).max() <= 1e-9 * scale
        assert np.abs(H[3:5, 3:5] - np.triu(k * J2.T @ R @ J2)).max() <= 1e-9 * scale
        assert np.all(H[3:5, 0:3] == 0.0)
        assert np.abs(rhs[0:3] + k * J1.T @ R @ e).max() <= 1e-9 * scale
        assert np.abs(rhs[3:5] + k * J2.T @ R @ e).max() <= 1e-9 * scale
        o.lib.orc_accumulate_term(o.h, 0, 0)       # "Do this twice! I should get double the values"
        H, rhs = _dense_upper(o)


# ------------------------------------------------------------------------------------------ 2d. compareSolvers
@pytest.mark.parametrize("use_diag", [False, True])
def test_compare_solvers_on_the_sample_system(use_diag):
    """LinearSolverTests.cpp:18-141 with SampleDvAndError.hpp:175-206: D = 4 Point2d, E = 20 terms cycling LinearErr (one
    dv, 2 rows) / LinearErr2 (two dvs, 2 rows) / LinearErr3 (three dvs, 4 rows), random covariances.  The block-Cholesky
    path (H, rhs assembled term by term) and the sparse-Cholesky path (J^T in compressed columns, rhs = J^T e, factor of
    J^T J + diag^2) agree: cost 1e-9, e / rhs / dx 1e-6 (the reference's tolerances; observed ~1e-13)."""
    rng = np.random.default_rng(23)
    D, E = 4, 20
    terms = []
    for i in range(E):
        m = i % 3
        terms.append((2, [i % D]) if m == 0 else (2, [i % D, (i + 1) % D]) if m == 1 else (4, [i % D, (i + 1) % D, (i + 2) % D]))
    o, _, _ = _linear_problem(rng, [2] * D, terms)
    assert (o.j_rows, o.j_cols) == (sum(t[0] for t in terms), 2 * D)
    diag = rng.uniform(-1, 1, o.j_cols)
    # S1 = BlockCholeskyLinearSystemSolver
    J1, e1 = o.evaluate_error(False)
    Cp, Ci, Cx, rhs1 = o.build_system(False)
    dx1 = np.zeros(o.j_cols)
    if use_diag:
        assert o.lib.orc_solve_system_with_conditioner(o.h, _d(diag), _d(dx1)) == 0
    else:
        assert o.lib.orc_solve_system_with_conditioner(o.h, _d(np.zeros(o.j_cols)), _d(dx1)) == 0
    # S2 = SparseCholeskyLinearSystemSolver: J^T CSC, rhs = J^T e, (J^T J + diag^2) dx = rhs
    import scipy.sparse as sp
    J2, e2 = o.evaluate_error(False)
    cp, ri, v = o.jacobian_transpose(False)
    Jt = sp.csc_matrix((v, ri, cp), shape=(o.j_cols, o.j_rows))
    rhs2 = Jt @ e2
    A = (Jt @ Jt.T).toarray() + (np.diag(diag * diag) if use_diag else 0.0)
    dx2 = np.linalg.solve(A, rhs2)
    assert abs(J1 - J2) <= 1e-9
    assert np.abs(e1 - e2).max() <= 1e-6 and np.abs(rhs1 - rhs2).max() <= 1e-6 and np.abs(dx1 - dx2).max() <= 1e-6
    assert np.abs(rhs1 - rhs2).max() <= 1e-11 * np.abs(rhs1).max() and np.abs(dx1 - dx2).max() <= 1e-9 * np.abs(dx1).max()
    # and the cost is the sum of e^T invR e (no M-estimator): equals |e|^2 of the stacked weighted error
    assert abs(J1 - e1 @ e1) <= 1e-9 * J1


# ------------------------------------------------------------------------------------------ 3. mpmath, first principles
def test_spline_pose_kinematics_against_mpmath():
    """50-digit evaluation of the gyroscope prediction's spline part from first principles: omega_b = -C(theta)^T S(theta)
    theta', with C / S from their series-free definitions and the spline value / derivative from de Boor's recursion in
    mpmath, against oracle and product at 1e-12 (rows D1-D4)."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    o, p = _olib(), _plib()
    rng = np.random.default_rng(31)
    order, nseg = 6, 8
    knots = (np.arange(nseg + 2 * order - 1) - (order - 1)) * 0.01
    nc = nseg + order - 1
    c = np.concatenate([rng.standard_normal((nc, 3)), 0.7 * rng.standard_normal((nc, 3))], 1)

    def basis(i, k, t):          # Cox - de Boor, order k (degree k-1)
        if k == 1:
            return mp.mpf(1) if knots[i] <= t < knots[i + 1] else mp.mpf(0)
        a = mp.mpf(0)
        d1, d2 = mp.mpf(knots[i + k - 1]) - mp.mpf(knots[i]), mp.mpf(knots[i + k]) - mp.mpf(knots[i + 1])
        if d1 > 0:
            a += (t - mp.mpf(knots[i])) / d1 * basis(i, k - 1, t)
        if d2 > 0:
            a += (mp.mpf(knots[i + k]) - t) / d2 * basis(i + 1, k - 1, t)
        return a

    def curve(t):
        return [sum(basis(i, order, t) * mp.mpf(c[i, d]) for i in range(nc)) for d in range(6)]

    for t in rng.uniform(0.0, nseg * 0.01, 6):
        tm = mp.mpf(float(t))
        val = curve(tm)
        der = [mp.diff(lambda s, d=d: curve(s)[d], tm) for d in range(6)]
        th, thd = mp.matrix(val[3:]), mp.matrix(der[3:])
        phi = mp.norm(th)
        K = mp.matrix([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0]])
        I = mp.eye(3)
        # RotationVector::parametersToRotationMatrix gives exp(-theta^) (the transpose of the usual exponential map) and
        # S(theta) = I - (1 - cos)/phi^2 theta^ + (phi - sin)/phi^3 theta^ theta^   (RotationVector.cpp:10-106)
        Cm = I - mp.sin(phi) / phi * K + (1 - mp.cos(phi)) / phi ** 2 * (K * K)
        Sm = I - (1 - mp.cos(phi)) / phi ** 2 * K + (phi - mp.sin(phi)) / phi ** 3 * (K * K)
        w_b = -(Cm.T * (Sm * thd))
        # oracle / product: value and derivative, then omega = S theta'
        a0, a1 = np.zeros(6), np.zeros(6)
        o.orc_u_spline_evald(order, knots.size, _d(knots), 6, nc, _d(c), C.c_double(t), 0, _d(a0))
        o.orc_u_spline_evald(order, knots.size, _d(knots), 6, nc, _d(c), C.c_double(t), 1, _d(a1))
        assert max(abs(float(val[d]) - a0[d]) for d in range(6)) <= 1e-13 * max(1.0, np.abs(a0).max())
        assert max(abs(float(der[d]) - a1[d]) for d in range(6)) <= 1e-12 * max(1.0, np.abs(a1).max())
        b0, b1 = np.zeros(6), np.zeros(6)
        p.probe_u_spline_evald(knots.size, _d(knots), 6, nc, _d(c), C.c_double(t), 0, _d(b0))
        p.probe_u_spline_evald(knots.size, _d(knots), 6, nc, _d(c), C.c_double(t), 1, _d(b1))
        assert np.abs(b0 - a0).max() <= 1e-13 * max(1.0, np.abs(a0).max()) and np.abs(b1 - a1).max() <= 1e-12 * max(1.0, np.abs(a1).max())
        om_o, J = np.zeros(3), np.zeros((3, 6))
        o.orc_u_rv_angular_velocity_and_jacobian(_d(a0[3:].copy()), _d(a1[3:].copy()), _d(om_o), _d(J))
        om_p, Dp = np.zeros(3), np.zeros((3, 3))
        p.probe_u_rotvec_dSv(_d(a0[3:].copy()), _d(a1[3:].copy()), _d(om_p), _d(Dp))
        Sv = Sm * thd
        want = np.array([float(Sv[i]) for i in range(3)])
        assert np.abs(om_o - want).max() <= 1e-12 * np.abs(want).max() and np.abs(om_p - want).max() <= 1e-12 * np.abs(want).max()
        # d(S theta')/dtheta by 50-digit differentiation
        def Sv_of(x, y, z):
            tt = mp.matrix([x, y, z])
            ph = mp.norm(tt)
            KK = mp.matrix([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0]])
            return (I - (1 - mp.cos(ph)) / ph ** 2 * KK + (ph - mp.sin(ph)) / ph ** 3 * (KK * KK)) * thd
        Dm = np.zeros((3, 3))
        for j in range(3):
            for i in range(3):
                args = [th[0], th[1], th[2]]
                Dm[i, j] = float(mp.diff(lambda s: Sv_of(*[s if q == j else args[q] for q in range(3)])[i], args[j]))
        assert np.abs(J[:, :3] - Dm).max() <= 1e-11 * np.abs(Dm).max(), np.abs(J[:, :3] - Dm).max()
        assert np.abs(Dp - Dm).max() <= 1e-11 * np.abs(Dm).max()
        assert float(mp.norm(w_b)) > 0


# ------------------------------------------------------------------------------------------ design-variable order (row A2)
def test_problem_order_is_group_first_insertion_order(small_problem):
    """The block indices of the whole path hang on the order in which inc.CalibrationOptimizationProblem reports its
    design variables (OptimizationProblem.cpp:123-137, 215-221, 272-292).  multical_amd/spec.py orders them in Python
    (`dv_order`); here the ORACLE restates the reference's container logic from the raw insertion sequence, and both must
    agree -- on the Multical problem (pose spline in the HELPER group first, calibration parameters interleaved) and on
    shuffled group sequences."""
    o = _olib()

    def oracle_order(groups):
        g = _i32(groups)
        out = np.zeros(g.size, np.int32)
        assert o.orc_u_problem_design_variable_order(g.size, _i(g), _i(out)) == 0
        return out.tolist()

    spec = small_problem["spec"]
    ins = [i for i, (k, d) in enumerate(spec.ops) if k in ("dv", "spline")]
    groups = [spec.ops[i][1]["group"] for i in ins]
    assert len(set(groups)) >= 2 and groups != sorted(groups) and groups != sorted(groups, reverse=True)
    assert [ins[j] for j in oracle_order(groups)] == spec.dv_order()
    rng = np.random.default_rng(1)
    for _ in range(20):
        groups = rng.integers(0, 4, 30).tolist()
        want = []
        for g in dict.fromkeys(groups):
            want += [i for i, x in enumerate(groups) if x == g]
        assert oracle_order(groups) == want
    # and the block indices that follow from it: feed the oracle in ITS order, by hand, and compare with spec.emit
    from oracle_api import OracleApi
    a = OracleApi()
    ids = spec.emit(a)
    a.finalize()
    groups = [spec.ops[i][1]["group"] for i in ins]
    blk = 0
    for j in oracle_order(groups):
        k, d = spec.ops[ins[j]]
        if k == "spline":
            first = a.spline_first_dv(ids[ins[j]][1])
            n = d["coeffs"].shape[1]
            if d["active"]:
                assert [a.dv_block(first + c)[0] for c in range(n)] == list(range(blk, blk + n))
                blk += n
        elif d["active"] and not (d["kind"] == 7 and not np.any(d.get("pattern", 1))):
            assert a.dv_block(ids[ins[j]])[0] == blk, (j, k)
            blk += 1
    assert blk == a.n_blocks
