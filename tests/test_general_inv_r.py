"""Row B2: a general (anisotropic, per-term) inverse covariance -- ErrorTermFs<D>::setInvR -> computeMatrixSqrt ->
weighted error S^T e, weighted Jacobian S^T J (aslam_backend/include/aslam/backend/implementation/ErrorTerm.hpp:96-192,
sm_eigen/include/sm/eigen/matrix_sqrt.hpp:21-40).  The product's square root and its mixed rows against the oracle's."""
import copy
import ctypes as C

import numpy as np
import pytest

from oracle_api import OracleApi, _d
from probe_api import ProbeApi

DIMS = {"reproj": 2, "static_reproj": 2, "accel": 3, "gyro": 3, "accel_sm": 3, "gyro_sm": 3, "gyro_prior": 3}


def anisotropic_spec(spec, seed=11):
    """a copy of the spec in which every 2- and 3-row term has its own random symmetric positive definite invR, scaled
    like the isotropic one it replaces (about half of them make the pivoted LDL^T swap rows)"""
    rng = np.random.RandomState(seed)
    s2 = copy.copy(spec)
    s2.ops = []
    for k, d in spec.ops:
        if k in DIMS:
            d = dict(d)
            D, n = DIMS[k], len(d["sinv"])
            A = rng.standard_normal((n, D, D))
            invR = (A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)) * (d["sinv"] ** 2)[:, None, None]
            d["invR"] = invR
        s2.ops.append((k, d))
    return s2


def _product_sqrt(A):
    from multical_amd._lib import load
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float64)
    S = np.empty_like(A)
    assert lib.mc_compute_matrix_sqrt(A.shape[0], A.ctypes.data_as(C.POINTER(C.c_double)), S.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return S


@pytest.mark.parametrize("n", [1, 2, 3, 6])
def test_matrix_sqrt_equals_the_oracles(n):
    o = OracleApi()
    rng = np.random.RandomState(n)
    swapped = 0
    for trial in range(40):
        A = rng.standard_normal((n, n))
        R = A @ A.T + 0.1 * np.eye(n)
        S = _product_sqrt(R)
        So = np.empty((n, n))
        assert o.lib.orc_u_matrix_sqrt(n, _d(R), _d(So)) == 0
        assert np.abs(S - So).max() <= 1e-13 * np.abs(So).max()
        assert np.abs(S @ S.T - R).max() <= 1e-13 * np.abs(R).max()
        swapped += int(np.argmax(np.diag(R)) != 0)
    assert n == 1 or swapped > 5          # the pivoting branch was exercised
    S = _product_sqrt(4.0 * np.eye(n))
    assert np.array_equal(S, 2.0 * np.eye(n))   # Multical's isotropic case: s I exactly


@pytest.fixture(scope="module")
def aniso(small_problem):
    spec = anisotropic_spec(small_problem["spec"])
    o = OracleApi()
    spec.emit(o)
    o.finalize()
    return spec, o


@pytest.mark.parametrize("use_mest", [True, False])
def test_every_term_with_general_inv_r_matches_oracle(aniso, use_mest):
    spec, o = aniso
    p = ProbeApi()
    spec.emit(p)
    p.finalize()
    o.evaluate_error(use_mest)
    chi = o.term_chi2()
    worst_j = worst_e = worst_c = 0.0
    for t in range(o.n_terms):
        if p.term_kind(t) == 5:
            continue
        Ja, ea = o.term_weighted_jacobian(t, use_mest)
        Jp, ep, c = p.term(t, use_mest)
        worst_j = max(worst_j, np.abs(Ja - Jp).max() / max(np.abs(Ja).max(), 1e-300))
        worst_e = max(worst_e, np.abs(ea - ep).max() / max(1e-9, np.abs(ea).max()))
        worst_c = max(worst_c, abs(c - chi[t]) / max(1e-9, chi[t]))
    assert worst_j < 1e-9 and worst_e < 1e-9 and worst_c < 1e-9, (worst_j, worst_e, worst_c)


def test_general_inv_r_changes_the_system(aniso, small_oracle):
    """guards the test above against a setter that silently does nothing"""
    _, o = aniso
    o0, _ = small_oracle
    J1, _ = o.evaluate_error(False)
    J0, _ = o0.evaluate_error(False)
    assert abs(J1 - J0) > 1e-3 * abs(J0)


def test_wrong_dimension_is_refused(small_problem):
    p = ProbeApi()
    spec = small_problem["spec"]
    bad = copy.copy(spec)
    bad.ops = []
    done = False
    for k, d in spec.ops:
        if k == "accel" and not done:
            d = dict(d)
            d["invR"] = np.tile(np.eye(2), (len(d["sinv"]), 1, 1))
            done = True
        bad.ops.append((k, d))
    with pytest.raises(RuntimeError, match="does not match the size of the error"):
        bad.emit(p)


def test_front_end_lowers_a_general_covariance():
    """the lowering pass hands a non-isotropic keypoint covariance on instead of refusing the term"""
    from multical_amd import frontend as F
    R = np.array([[4.0, 1.0], [1.0, 2.0]])
    assert isinstance(F._isotropic(R, "x"), np.ndarray)
    assert F._isotropic(9.0 * np.eye(3), "x") == 3.0
    s, invR = F._weights([2.0, R], 2)
    assert invR.shape == (2, 2, 2) and np.array_equal(invR[0], 4.0 * np.eye(2)) and np.array_equal(invR[1], R)
    s, invR = F._weights([2.0, 3.0], 2)
    assert invR is None and np.array_equal(s, [2.0, 3.0])


@pytest.mark.gpu
@pytest.mark.parametrize("use_mest", [True, False])
def test_device_build_with_general_inv_r(aniso, use_mest):
    from multical_amd.backend import HipApi
    spec, o = aniso
    g = HipApi(0)
    spec.emit(g)
    g.finalize()
    Jo, eo = o.evaluate_error(use_mest)
    Jg, eg = g.evaluate_error(use_mest)
    assert abs(Jg - Jo) <= 1e-9 * abs(Jo)
    assert np.abs(eg - eo).max() <= 1e-9 * np.abs(eo).max()
    Cp, Ci, Cx, rhs = o.build_system(use_mest)
    Cxg, rhsg = g.build_system(use_mest)
    assert np.abs(Cxg - Cx).max() <= 1e-9 * np.abs(Cx).max()
    assert np.abs(rhsg - rhs).max() <= 1e-9 * np.abs(rhs).max()


# ---- the other term families: F2 (static reprojection), the scale-misalignment IMU models, the gyroscope prior ----
def _family_spec(name, small_problem):
    if name == "static":
        from multical_amd import synth_static as SS
        cfg = SS.StaticConfig(n_cams=3, n_boards=2, n_views=6, corner_stride=5, cam_models=[(0, 1), (1, 1), (0, 2)])
        return SS.make_static_problem(cfg, OracleApi)[0]
    if name == "extrinsic_prior":
        from multical_amd import synth_init as SI
        return SI.make_extrinsic_prior(n_gyro=120, n_other=1, n_obs=20, seed=5, huber_gyro=1.0)[0]
    from test_variants import make_variant
    return make_variant(small_problem["spec"], name)


FAMILIES = ["static", "extrinsic_prior", "scale_misalignment", "size_effect"]


@pytest.fixture(scope="module")
def families(small_problem):
    out = {}
    for name in FAMILIES:
        spec = anisotropic_spec(_family_spec(name, small_problem), seed=5)
        assert any("invR" in d for k, d in spec.ops)
        o = OracleApi()
        spec.emit(o)
        o.finalize()
        out[name] = (spec, o)
    return out


@pytest.mark.parametrize("name", FAMILIES)
def test_other_families_product_math(families, name):
    spec, o = families[name]
    p = ProbeApi()
    spec.emit(p)
    p.finalize()
    o.evaluate_error(True)
    chi = o.term_chi2()
    for t in range(o.n_terms):
        if p.term_kind(t) == 5:
            continue
        Ja, ea = o.term_weighted_jacobian(t, True)
        Jp, ep, c = p.term(t, True)
        assert np.abs(Ja - Jp).max() <= 1e-9 * max(np.abs(Ja).max(), 1e-300), (name, t)
        assert np.abs(ea - ep).max() <= 1e-9 * max(1e-9, np.abs(ea).max()), (name, t)
        assert abs(c - chi[t]) <= 1e-9 * max(1e-9, chi[t]), (name, t)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FAMILIES)
def test_other_families_on_the_device(families, name):
    from multical_amd.backend import HipApi
    spec, o = families[name]
    g = HipApi(0)
    spec.emit(g)
    g.finalize()
    Jo, eo = o.evaluate_error(True)
    Jg, eg = g.evaluate_error(True)
    assert abs(Jg - Jo) <= 1e-9 * abs(Jo)
    assert np.abs(eg - eo).max() <= 1e-9 * np.abs(eo).max()
    Cp, Ci, Cx, rhs = o.build_system(True)
    Cxg, rhsg = g.build_system(True)
    assert np.abs(Cxg - Cx).max() <= 1e-9 * np.abs(Cx).max()
    assert np.abs(rhsg - rhs).max() <= 1e-9 * np.abs(rhs).max()
    if name in ("static", "extrinsic_prior"):        # no motion term: the Jacobian-transpose backend applies
        cp, ri, v = o.jacobian_transpose(True)
        g.evaluate_error(True)
        vg = g.build_jacobian_transpose(True)
        assert np.abs(vg - v).max() <= 1e-9 * np.abs(v).max()
