"""Camera-model matrix.  north_star lists pinhole / omni / double sphere x radtan / equidistant; the small spline problem
(conftest.py) carries pinhole-radtan, omni-radtan and double sphere through F1 and the static problem
(test_static_problem.py) pinhole-radtan, omni-radtan and pinhole-equidistant through F2.  Here the remaining pairs go
through BOTH term families, so that every (projection, distortion) pair the reference instantiates
(PinholeProjection / OmniProjection x RadialTangential / Equidistant / NoDistortion, DoubleSphereProjection<NoDistortion>,
aslam_cameras/include/aslam/cameras.hpp) is checked in SimpleReprojectionError and in ReprojectionError:
oracle vs finite differences (1e-6, ErrorTermTestHarness.hpp:20-35), the product's closed forms on the host against the
oracle term by term, and the HIP path through the C ABI (1e-9)."""
import numpy as np
import pytest

from test_oracle_identities import _fd_check

RTOL = 1e-9
P, O, DS = 0, 1, 2
NONE, RT, EQ = 0, 1, 2
F1_MODELS = [(P, EQ), (O, EQ), (O, NONE), (P, NONE)]
F2_MODELS = [(DS, NONE), (O, NONE), (P, NONE), (O, EQ)]


@pytest.fixture(scope="module")
def problems():
    from multical_amd import synth, synth_static as SS
    from oracle_api import OracleApi
    out = {}
    cfg = synth.SynthConfig(duration=1.0, n_cams=4, n_imus=1, n_lidars=0, corner_stride=9, cam_models=F1_MODELS,
                            blake_zisserman_cam=2)
    spec, _, _, _ = synth.make_problem(cfg, OracleApi)
    out["spline"] = spec
    scfg = SS.StaticConfig(n_cams=4, n_boards=2, n_views=6, corner_stride=6, cam_models=F2_MODELS)
    sspec, _, _, _ = SS.make_static_problem(scfg, OracleApi)
    out["static"] = sspec
    res = {}
    for name, sp in out.items():
        o = OracleApi()
        sp.emit(o)
        o.finalize()
        res[name] = (sp, o)
    return res


def _reproj_terms(sp):
    """(first term, count, camera model pair) of every reprojection batch"""
    cams = [d for k, d in sp.ops if k == "camera"]
    out, base = [], 0
    for _, k, d in sp.batch_ops():
        n = 1 if k == "motion" else len(d["sinv"])
        if k in ("reproj", "static_reproj"):
            out.append((base, n))
        base += n
    return out, cams


@pytest.mark.parametrize("name", ["spline", "static"])
def test_every_model_pair_is_present(problems, name):
    sp, o = problems[name]
    batches, cams = _reproj_terms(sp)
    assert len(batches) == 4 and all(n > 0 for _, n in batches)
    want = F1_MODELS if name == "spline" else F2_MODELS
    assert [(c["model"], c["dist"]) for c in cams] == want


@pytest.mark.parametrize("name", ["spline", "static"])
def test_oracle_jacobians_match_finite_differences(problems, name):
    sp, o = problems[name]
    batches, _ = _reproj_terms(sp)
    terms = []
    for first, n in batches:
        terms += [first, first + n // 2, first + n - 1]
    _fd_check(o, terms, 1e-6)


@pytest.mark.parametrize("use_mest", [True, False])
@pytest.mark.parametrize("name", ["spline", "static"])
def test_product_math_matches_oracle(problems, name, use_mest):
    from probe_api import ProbeApi
    sp, o = problems[name]
    p = ProbeApi()
    sp.emit(p)
    p.finalize()
    assert (p.j_rows, p.j_cols, p.n_blocks, p.n_terms) == (o.j_rows, o.j_cols, o.n_blocks, o.n_terms)
    o.evaluate_error(use_mest)
    Cp, Ci, _, _ = o.build_system(use_mest)
    Cp2, Ci2 = p.pattern()
    assert np.array_equal(Cp, Cp2) and np.array_equal(Ci, Ci2)
    chi = o.term_chi2()
    batches, _ = _reproj_terms(sp)
    for first, n in batches:
        wj = we = wc = 0.0
        for t in range(first, first + n, max(1, n // 60)):
            Ja, ea = o.term_weighted_jacobian(t, use_mest)
            Jp, ep, c = p.term(t, use_mest)
            wj = max(wj, np.abs(Ja - Jp).max() / max(1e-300, np.abs(Ja).max()))
            we = max(we, np.abs(ea - ep).max() / max(1e-9, np.abs(ea).max()))
            wc = max(wc, abs(c - chi[t]) / max(1e-9, chi[t]))
        assert wj < RTOL and we < RTOL and wc < RTOL, (first, wj, we, wc)


@pytest.mark.gpu
@pytest.mark.parametrize("use_mest", [True, False])
@pytest.mark.parametrize("name", ["spline", "static"])
def test_gpu_parity(problems, name, use_mest):
    from multical_amd.backend import HipApi
    sp, o = problems[name]
    g = HipApi(0)
    sp.emit(g)
    g.finalize()
    Jo, eo = o.evaluate_error(use_mest)
    Cp, Ci, Cx, rhs = o.build_system(use_mest)
    Cpg, Cig = g.pattern()
    assert np.array_equal(Cp, Cpg) and np.array_equal(Ci, Cig)
    Jg, eg = g.evaluate_error(use_mest)
    assert abs(Jg - Jo) <= RTOL * abs(Jo)
    assert np.abs(eg - eo).max() <= RTOL * np.abs(eo).max()
    cg, co = g.term_chi2(), o.term_chi2()
    assert np.abs(cg - co).max() <= RTOL * np.abs(co).max()
    Cxg, rhsg = g.build_system(use_mest)
    assert np.abs(Cxg - Cx).max() <= RTOL * np.abs(Cx).max()
    assert np.linalg.norm(Cxg - Cx) <= RTOL * np.linalg.norm(Cx)
    assert np.abs(rhsg - rhs).max() <= RTOL * np.abs(rhs).max()
    g.check_status()


@pytest.mark.gpu
def test_gpu_jacobian_transpose_static(problems):
    """second backend (SparseCholeskyLinearSystemSolver's J^T) over the same model pairs"""
    from multical_amd.backend import HipApi
    sp, o = problems["static"]
    g = HipApi(0)
    sp.emit(g)
    g.finalize()
    o.evaluate_error(True)
    g.evaluate_error(True)
    cp, ri, vals = o.jacobian_transpose(True)
    cpg, rig = g.jt_pattern()
    assert np.array_equal(cp, cpg) and np.array_equal(ri, rig)
    vg = g.build_jacobian_transpose(True)
    assert np.abs(vg - vals).max() <= RTOL * np.abs(vals).max()
