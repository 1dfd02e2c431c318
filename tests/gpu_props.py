"""Size-independent property checks of an assembled system on the GPU (no oracle needed): used at the BASELINE
configurations' full sizes, where the CPU oracle would take minutes per buildSystem.

  gradient    for J(x) = sum e^2 (no M-estimator) the directional derivative of the cost along a random update v, by
              central differences of evaluateError through applyStateUpdate / revert, equals -2 rhs . v
  curvature   v^T H v equals |J v|^2 with J v = (e(x + h v) - e(x - h v)) / 2h: ties the assembled Hessian (atomics, slot
              tables, chunk windows) to the residual path, which shares none of that machinery
  psd         v^T H v >= 0 on random vectors, H symmetric by construction of the upper storage
"""
import numpy as np


def sym_matrix(g, Cx):
    import scipy.sparse as sp
    Cp, Ci = g.pattern()
    n = g.j_cols
    U = sp.csc_matrix((Cx, Ci, Cp), shape=(n, n))
    return U + sp.triu(U, 1).T


def check_gradient_and_curvature(g, trials=2, h=1e-5, rtol=1e-6, seed=3, scale=None):
    """g: a finalized HipApi whose terms carry no M-estimator and no BSplineMotionError (its e row is 0)."""
    Cx, rhs = g.build_system(False)
    A = sym_matrix(g, Cx)
    n = g.j_cols
    rng = np.random.default_rng(seed)
    for _ in range(trials):
        v = rng.standard_normal(n)
        if scale is not None:
            v *= scale
        v /= np.linalg.norm(v)
        g.apply_update(h * v)
        Jp, ep = g.evaluate_error(False)
        g.revert_update()
        g.apply_update(-h * v)
        Jm, em = g.evaluate_error(False)
        g.revert_update()
        dJ = (Jp - Jm) / (2 * h)
        want_g = -2.0 * rhs @ v
        assert abs(dJ - want_g) <= rtol * abs(want_g), (dJ, want_g)
        Jv = (ep - em) / (2 * h)
        q, want = v @ (A @ v), Jv @ Jv
        assert abs(q - want) <= rtol * want, (q, want)
    g.check_status()
    return Cx, rhs


def check_psd(g, Cx, trials=5, seed=0):
    A = sym_matrix(g, Cx)
    rng = np.random.default_rng(seed)
    for _ in range(trials):
        v = rng.standard_normal(g.j_cols)
        assert v @ (A @ v) >= -1e-9 * np.abs(Cx).max()


def scaled_sinv_spec(spec, factor):
    """a copy of the spec with every sqrtInvR multiplied by `factor` (H and b must scale by factor^2 without M-estimators)"""
    import copy
    s2 = copy.copy(spec)
    s2.ops = []
    for k, d in spec.ops:
        if isinstance(d, dict) and "sinv" in d:
            d = dict(d)
            d["sinv"] = d["sinv"] * factor
        s2.ops.append((k, d))
    return s2
