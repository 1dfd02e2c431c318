"""Row C8: the HomogeneousPoint design variable (aslam_backend_expressions/src/HomogeneousPoint.cpp) -- unit-norm 4-vector,
minimal dimension 3, update by updateQuat, Jacobian quatInvS -- and the one place kalibr uses it: target corners as
landmarks of the static-camera ReprojectionError (estimateLandmarks).  Oracle identities, product math, device parity."""
import copy

import numpy as np
import pytest

from multical_amd import spec as S
from oracle_api import OracleApi
from probe_api import ProbeApi
from test_oracle_identities import _fd_check


from multical_amd.synth_static import with_landmarks  # noqa: E402


@pytest.fixture(scope="module")
def lm_problem():
    from multical_amd import synth_static as SS
    cfg = SS.StaticConfig(n_cams=3, n_boards=2, n_views=5, corner_stride=7, cam_models=[(0, 1), (1, 1), (2, 0)])
    base = SS.make_static_problem(cfg, OracleApi)[0]
    out = {}
    for name, kw in (("landmarks", {}), ("negative_w", dict(flip_w_every=3))):
        sp = with_landmarks(base, **kw)
        o = OracleApi()
        ids = sp.emit(o)
        o.finalize()
        out[name] = (sp, o, ids)
    return out


NAMES = ["landmarks", "negative_w"]


def test_design_variable_is_normalised_and_updates_like_a_quaternion():
    sp = S.ProblemSpec()
    h = sp.add_dv(S.DV_HOMOGENEOUS_POINT, [0.3, -0.2, 2.0, 1.0], True, 1.0, S.HELPER_GROUP_ID)
    for Api in (OracleApi, ProbeApi):
        a = Api()
        ids = sp.emit(a)
        if hasattr(a, "dv_params"):
            p = a.dv_params(ids[h])
            assert abs(np.linalg.norm(p) - 1.0) < 1e-15
            assert np.allclose(p * (1.0 / p[3]), [0.3, -0.2, 2.0, 1.0], rtol=1e-14)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_jacobians_match_finite_differences(lm_problem, name):
    sp, o, _ = lm_problem[name]
    _fd_check(o, range(0, o.n_terms, 9), 1e-6)


@pytest.mark.parametrize("name", NAMES)
def test_landmark_blocks_are_in_the_problem(lm_problem, name):
    sp, o, ids = lm_problem[name]
    n_active = sum(1 for k, d in sp.ops if k == "dv" and d["kind"] == S.DV_HOMOGENEOUS_POINT and d["active"])
    n_all = sum(1 for k, d in sp.ops if k == "dv" and d["kind"] == S.DV_HOMOGENEOUS_POINT)
    assert 0 < n_active < n_all
    base_cols = sum(len(d["params"]) if d["kind"] not in (S.DV_ROTATION_QUATERNION, S.DV_HOMOGENEOUS_POINT) else 3
                    for k, d in sp.ops if k == "dv" and d["active"])
    assert o.j_cols == base_cols


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("use_mest", [True, False])
def test_product_math_matches_oracle(lm_problem, name, use_mest):
    sp, o, _ = lm_problem[name]
    p = ProbeApi()
    sp.emit(p)
    p.finalize()
    assert (p.j_rows, p.j_cols, p.n_blocks, p.n_terms) == (o.j_rows, o.j_cols, o.n_blocks, o.n_terms)
    o.evaluate_error(use_mest)
    Cp, Ci, Cx, rhs = o.build_system(use_mest)
    Cp2, Ci2 = p.pattern()
    assert np.array_equal(Cp, Cp2) and np.array_equal(Ci, Ci2)        # the landmark blocks sit where the reference puts them
    chi = o.term_chi2()
    for t in range(o.n_terms):
        Ja, ea = o.term_weighted_jacobian(t, use_mest)
        Jp, ep, c = p.term(t, use_mest)
        assert np.abs(Ja - Jp).max() <= 1e-9 * max(np.abs(Ja).max(), 1e-300), t
        assert np.abs(ea - ep).max() <= 1e-9 * max(1e-9, np.abs(ea).max()), t
        assert abs(c - chi[t]) <= 1e-9 * max(1e-9, chi[t]), t


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("use_mest", [True, False])
def test_device_parity(lm_problem, name, use_mest):
    from multical_amd.backend import HipApi
    sp, o, _ = lm_problem[name]
    g = HipApi(0)
    sp.emit(g)
    g.finalize()
    Cp, Ci = g.pattern()
    Jo, eo = o.evaluate_error(use_mest)
    Cpo, Cio, Cx, rhs = o.build_system(use_mest)
    assert np.array_equal(Cp, Cpo) and np.array_equal(Ci, Cio)
    Jg, eg = g.evaluate_error(use_mest)
    assert abs(Jg - Jo) <= 1e-9 * abs(Jo)
    assert np.abs(eg - eo).max() <= 1e-9 * np.abs(eo).max()
    Cxg, rhsg = g.build_system(use_mest)
    assert np.abs(Cxg - Cx).max() <= 1e-9 * np.abs(Cx).max()
    assert np.abs(rhsg - rhs).max() <= 1e-9 * np.abs(rhs).max()
    cp, ri, v = o.jacobian_transpose(use_mest)
    cpg, rig = g.jt_pattern()
    assert np.array_equal(cp, cpg) and np.array_equal(ri, rig)
    vg = g.build_jacobian_transpose(use_mest)
    assert np.abs(vg - v).max() <= 1e-9 * np.abs(v).max()


@pytest.mark.gpu
def test_device_update_and_revert(lm_problem):
    """applyStateUpdate / revert on the landmarks: updateQuat on the device against the oracle's, then the system again"""
    from multical_amd.backend import HipApi
    sp, o, ids = lm_problem["landmarks"]
    o2 = OracleApi()
    sp.emit(o2)
    o2.finalize()
    g = HipApi(0)
    gids = sp.emit(g)
    g.finalize()
    rng = np.random.RandomState(0)
    dx = 1e-3 * rng.standard_normal(o2.j_cols)
    o2.apply_update(dx)
    g.apply_update(dx)
    for h, (k, d) in enumerate(sp.ops):
        if k == "dv" and d["kind"] == S.DV_HOMOGENEOUS_POINT:
            assert np.abs(g.dv_params(gids[h]) - o2.dv_params(ids[h])).max() < 1e-14
    Jo, eo = o2.evaluate_error(True)
    Jg, eg = g.evaluate_error(True)
    assert abs(Jg - Jo) <= 1e-9 * abs(Jo)
    g.revert_update()
    o2.revert_update()
    Jo, _ = o2.evaluate_error(True)
    Jg, _ = g.evaluate_error(True)
    assert abs(Jg - Jo) <= 1e-9 * abs(Jo)


@pytest.mark.gpu
def test_device_parity_with_many_views_of_each_landmark():
    """The landmark-major pass (k_static_lm) takes a landmark's terms of one camera in work items of <= 1024 terms and
    passes of 128: 1 200 views of 5 corners put ~1 200 terms on every landmark, i.e. two work items and ten passes each;
    the observation-major pass adds the landmark x target-pose blocks per term from the precomputed slots."""
    from multical_amd import synth_static as SS
    from multical_amd.backend import HipApi
    cfg = SS.StaticConfig(n_cams=2, n_boards=1, n_views=1200, corner_stride=29, cam_models=[(0, 1), (2, 0)])
    base = SS.make_static_problem(cfg, OracleApi)[0]
    sp = SS.with_landmarks(base, inactive_every=4, constant_every=0)
    o = OracleApi()
    sp.emit(o)
    o.finalize()
    g = HipApi(0)
    sp.emit(g)
    g.finalize()
    Cp, Ci = g.pattern()
    Jo, eo = o.evaluate_error(True)
    Cpo, Cio, Cx, rhs = o.build_system(True)
    assert np.array_equal(Cp, Cpo) and np.array_equal(Ci, Cio)
    Jg, eg = g.evaluate_error(True)
    assert abs(Jg - Jo) <= 1e-9 * abs(Jo)
    Cxg, rhsg = g.build_system(True)
    assert np.abs(Cxg - Cx).max() <= 1e-9 * np.abs(Cx).max()
    assert np.abs(rhsg - rhs).max() <= 1e-9 * np.abs(rhs).max()
    assert np.array_equal(Cxg == 0.0, Cx == 0.0)
