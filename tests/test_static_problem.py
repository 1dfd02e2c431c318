"""F2: static-camera calibration problems (BASELINE configs[0] / configs[2] shape: ReprojectionError with camera design
variables over a baseline chain, CameraCalibrator.py:195-219).  CPU part: oracle identities + the product's closed forms
(tests/hostprobe) against the oracle; GPU part: the HIP path through the C ABI against the oracle (1e-9)."""
import numpy as np
import pytest

from test_oracle_identities import _fd_check

RTOL = 1e-9


@pytest.fixture(scope="module")
def static_problem():
    from multical_amd import synth_static as SS
    from oracle_api import OracleApi
    # three cameras with three different model pairs, two boards, ragged corner counts, Blake-Zisserman(2)
    cfg = SS.StaticConfig(n_cams=3, n_boards=2, n_views=8, corner_stride=5, cam_models=[(0, 1), (1, 1), (0, 2)])
    spec, H, W, meas = SS.make_static_problem(cfg, OracleApi)
    o = OracleApi()
    ids = spec.emit(o)
    o.finalize()
    return dict(spec=spec, H=H, W=W, oracle=o, ids=ids)


def test_static_oracle_jacobians_match_finite_differences(static_problem):
    o = static_problem["oracle"]
    terms = list(range(0, o.n_terms, max(1, o.n_terms // 12)))
    _fd_check(o, terms, 1e-6)


def test_static_oracle_system_is_dense_normal_equations(static_problem):
    o = static_problem["oracle"]
    o.evaluate_error(True)
    Cp, Ci, Cx, rhs = o.build_system(True)
    n = o.j_cols
    Hd, g = np.zeros((n, n)), np.zeros(n)
    for t in range(o.n_terms):
        J, e = o.term_weighted_jacobian(t, True)
        Hd += J.T @ J
        g -= J.T @ e
    cols = np.repeat(np.arange(n), np.diff(Cp))
    assert np.abs(Hd[Ci, cols] - Cx).max() <= 1e-12 * np.abs(Cx).max()
    assert np.abs(g - rhs).max() <= 1e-12 * np.abs(rhs).max()
    # every structurally non-zero entry of the dense upper triangle is in the pattern
    mask = np.zeros((n, n), bool)
    mask[Ci, cols] = True
    assert not np.any(np.triu(Hd != 0.0) & ~mask)


@pytest.mark.parametrize("use_mest", [True, False])
def test_static_product_math_matches_oracle(static_problem, use_mest):
    from probe_api import ProbeApi
    o = static_problem["oracle"]
    p = ProbeApi()
    static_problem["spec"].emit(p)
    p.finalize()
    o.evaluate_error(use_mest)
    Cp, Ci, _, _ = o.build_system(use_mest)
    Cp2, Ci2 = p.pattern()
    assert np.array_equal(Cp, Cp2) and np.array_equal(Ci, Ci2)
    chi = o.term_chi2()
    wj = we = wc = 0.0
    for t in range(o.n_terms):
        Ja, ea = o.term_weighted_jacobian(t, use_mest)
        Jp, ep, c = p.term(t, use_mest)
        wj = max(wj, np.abs(Ja - Jp).max() / np.abs(Ja).max())
        we = max(we, np.abs(ea - ep).max() / max(1e-9, np.abs(ea).max()))
        wc = max(wc, abs(c - chi[t]) / max(1e-9, chi[t]))
    assert wj < RTOL and we < RTOL and wc < RTOL, (wj, we, wc)


@pytest.mark.gpu
@pytest.mark.parametrize("use_mest", [True, False])
def test_static_gpu_parity(static_problem, use_mest):
    from multical_amd.backend import HipApi
    o = static_problem["oracle"]
    g = HipApi(0)
    static_problem["spec"].emit(g)
    g.finalize()
    Jo, eo = o.evaluate_error(use_mest)
    Cp, Ci, Cx, rhs = o.build_system(use_mest)
    Cpg, Cig = g.pattern()
    assert np.array_equal(Cp, Cpg) and np.array_equal(Ci, Cig)
    Jg, eg = g.evaluate_error(use_mest)
    assert abs(Jg - Jo) <= RTOL * abs(Jo)
    assert np.abs(eg - eo).max() <= RTOL * np.abs(eo).max()
    Cxg, rhsg = g.build_system(use_mest)
    assert np.abs(Cxg - Cx).max() <= RTOL * np.abs(Cx).max()
    assert np.linalg.norm(Cxg - Cx) <= RTOL * np.linalg.norm(Cx)
    assert np.abs(rhsg - rhs).max() <= RTOL * np.abs(rhs).max()
    g.check_status()


@pytest.mark.gpu
def test_static_gpu_levenberg_marquardt_matches_oracle():
    """configs[0] shape (one pinhole-radtan camera, one board), reduced to 40 views: the Python Optimizer2 over the HIP
    solver against the oracle's Optimizer2, estimates within 1e-7."""
    from multical_amd import synth_static as SS
    from multical_amd.backend import (HipApi, HipBlockCholeskyLinearSystemSolver, LevenbergMarquardtTrustRegionPolicy, Optimizer2,
                                      Optimizer2Options)
    from oracle_api import OracleApi
    cfg = SS.baseline_static_config(0, n_views=40, corner_stride=3)
    spec, H, W, meas = SS.make_static_problem(cfg, lambda: HipApi(0))
    o = OracleApi()
    oids = spec.emit(o)
    o.finalize()
    g = HipApi(0)
    gids = spec.emit(g)
    g.finalize()
    iters = 5
    opt = Optimizer2Options()
    opt.maxIterations, opt.convergenceDeltaX, opt.convergenceJDescentRatioThreshold = iters, 1e-5, 1e-6
    opt.linearSolver = HipBlockCholeskyLinearSystemSolver(g)
    opt.trustRegionPolicy = LevenbergMarquardtTrustRegionPolicy(10.0)
    rv = Optimizer2(opt).optimize()
    ro = o.optimize(iters, 10.0, 1e-5, 1e-6, 1)
    assert rv.iterations == ro["iterations"] and rv.failedIterations == ro["failedIterations"]
    assert abs(rv.JFinal - ro["JFinal"]) <= 1e-7 * abs(ro["JFinal"])
    assert rv.JFinal < 0.5 * rv.JStart
    for h, v in gids.items():
        if isinstance(v, int) and spec.ops[h][0] == "dv":
            a, b = g.dv_params(v), o.dv_params(oids[h])
            assert np.abs(a - b).max() <= 1e-7 * max(1.0, np.abs(b).max()), (h, a, b)


# ---------------------------------------------------------------------------------------------------------------
# second backend: J^T in compressed-column form (SparseCholeskyLinearSystemSolver, the default solver of Optimizer2 and
# the one kalibr_calibrate_cameras runs on this problem shape)
def test_oracle_jacobian_transpose_is_consistent(static_problem):
    import scipy.sparse as sp
    o = static_problem["oracle"]
    J, e = o.evaluate_error(True)
    cp, ri, v = o.jacobian_transpose(True)
    Jt = sp.csc_matrix((v, ri, cp), shape=(o.j_cols, o.j_rows))
    Cp, Ci, Cx, rhs = o.build_system(True)
    U = sp.csc_matrix((Cx, Ci, Cp), shape=(o.j_cols, o.j_cols))
    assert abs(sp.triu(Jt @ Jt.T) - U).max() <= 1e-13 * np.abs(Cx).max()
    assert np.abs(Jt @ e - rhs).max() <= 1e-13 * np.abs(rhs).max()
    assert np.all(np.diff(cp) > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("use_mest", [True, False])
def test_static_gpu_jacobian_transpose(static_problem, use_mest):
    from multical_amd.backend import HipApi
    o = static_problem["oracle"]
    g = HipApi(0)
    static_problem["spec"].emit(g)
    g.finalize()
    o.evaluate_error(use_mest)
    cp, ri, v = o.jacobian_transpose(use_mest)
    cpg, rig = g.jt_pattern()
    assert np.array_equal(cp, cpg) and np.array_equal(ri, rig)           # bit-exact structure
    g.evaluate_error(use_mest)
    vg = g.build_jacobian_transpose(use_mest)
    assert np.abs(vg - v).max() <= RTOL * np.abs(v).max()


@pytest.mark.gpu
def test_static_gpu_sparse_cholesky_solver_matches_block_solver(static_problem):
    """LinearSolverTests.cpp:42-57 of the reference: the two solvers agree on J, e, rhs and dx."""
    from multical_amd.backend import HipApi, HipBlockCholeskyLinearSystemSolver, HipSparseCholeskyLinearSystemSolver
    g = HipApi(0)
    static_problem["spec"].emit(g)
    g.finalize()
    a, b = HipBlockCholeskyLinearSystemSolver(g), HipSparseCholeskyLinearSystemSolver(g)
    out = []
    for s in (a, b):
        s.initMatrixStructure(True)
        J = s.evaluateError(1, True)
        s.buildSystem(1, True)
        s.setConstantConditioner(1.0)
        ok, dx = s.solveSystem()
        assert ok
        out.append((J, s.e().copy(), s.rhs().copy(), dx, s.rhsJtJrhs()))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[0][2] - out[1][2]).max() <= 1e-9 * np.abs(out[0][2]).max()
    assert np.abs(out[0][3] - out[1][3]).max() <= 1e-6 * np.abs(out[0][3]).max()
    assert abs(out[0][4] - out[1][4]) <= 1e-9 * abs(out[0][4])
