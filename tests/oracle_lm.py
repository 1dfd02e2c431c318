"""TEST INFRASTRUCTURE: Optimizer2::optimize with the Levenberg-Marquardt policy driven over the CPU oracle's
evaluateError / buildSystem / applyStateUpdate, with the linear solve done by SciPy's SuperLU on the exported CCS matrix.

The oracle's own C++ loop (orc_optimize) factors a DENSE copy of H, which stops being usable beyond a few thousand
columns; this driver is the same control flow (aslam_backend/src/Optimizer2.cpp:183-275,
LevenbergMarquardtTrustRegionPolicy.cpp:50-91, BlockCholeskyLinearSystemSolver.cpp:74-106 incl. the augment-by-lambda^2 /
un-augment-by-lambda sequence on the solver's copy of H) at sizes where only a sparse factorisation will do.  The solve
itself is outside the parity scope (CHOLMOD in the reference; its own tests pin it only through dx agreement at 1e-6,
aslam_backend/test/LinearSolverTests.cpp:55-57)."""
import numpy as np


def _solve(Cp, Ci, Cx, rhs):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    n = rhs.size
    U = sp.csc_matrix((Cx, Ci, Cp), shape=(n, n))
    A = (U + sp.triu(U, 1).T).tocsc()
    try:
        lu = spl.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    except RuntimeError:
        return None
    d = lu.U.diagonal()
    if not np.all(np.isfinite(d)) or np.any(d <= 0.0):     # not positive definite: LinearSolverCholmod::solve returns false
        return None
    x = lu.solve(rhs)
    return x if np.all(np.isfinite(x)) else None


def optimize_lm(o, max_iterations=20, lambda_init=10.0, convergence_delta_x=1e-5, convergence_j_ratio=1e-6, log=None):
    """Returns dict(JStart, JFinal, iterations, failedIterations, dXFinal, dJFinal, linearSolverFailure, trace) where
    trace lists (accepted, J, lambda) per iteration."""
    gamma, beta, p = 3.0, 2.0, 3
    lam, mu = lambda_init, 2.0
    J, _ = o.evaluate_error(True, False)
    JStart = pJ = J
    trPJ = trLastOkJ = J
    deltaX, ratio, deltaJ = convergence_delta_x + 1.0, convergence_j_ratio + 1.0, 0.0
    prevFailed = lsFail = False
    first = True
    it = failed = 0
    Cp = Ci = Cx = rhs = dx = None
    diag_pos = None
    trace = []
    while it < max_iterations and failed < max_iterations and \
            ((deltaX > convergence_delta_x and abs(ratio) > convergence_j_ratio) or lsFail):
        # ---- LevenbergMarquardtTrustRegionPolicy::solveSystemImplementation
        if not prevFailed:
            trPJ, trLastOkJ = trLastOkJ, J
        if first:
            Cp, Ci, Cx, rhs = o.build_system(True)
            diag_pos = Cp[1:] - 1
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                rho = np.float64(trPJ - J) / np.float64(dx @ (lam * dx + rhs))
            if prevFailed:
                mu *= 2
                lam *= mu
            elif rho <= 0:
                mu *= 10
                lam *= mu
            else:
                Cp, Ci, Cx, rhs = o.build_system(True)
                if lam > 1e-16:
                    u1, u2 = 1.0 / gamma, 1.0 - (beta - 1.0) * (2.0 * rho - 1.0) ** p
                    lam *= u1 if u1 > u2 else u2
                    mu = beta
                else:
                    lam = 1e-15
        first = False
        # ---- BlockCholeskyLinearSystemSolver::solveSystem: H += lambda^2 I, solve, H -= lambda I (sic)
        Cx[diag_pos] += lam * lam
        dx = _solve(Cp, Ci, Cx, rhs)
        Cx[diag_pos] -= lam
        if dx is None:
            prevFailed = lsFail = True
            failed += 1
            dx = np.zeros(rhs.size)
            continue
        # ---- Optimizer2::optimize
        o.apply_update(dx)
        deltaX = float(np.max(np.abs(dx)))
        J, _ = o.evaluate_error(True, False)
        deltaJ = pJ - J
        ratio = deltaJ / pJ
        if ratio < 0.0:
            o.revert_update()
            o.evaluate_error(True, False)     # the oracle's terms cache chi^2 of the last evaluation: bring it back too
            failed += 1
            prevFailed = True
            trace.append((False, J, lam))
        else:
            pJ = J
            prevFailed = False
            trace.append((True, J, lam))
        it += 1
        if log:
            log("[%d] J %.9g dJ %.3g dX %.3g lambda %.3g mu %g %s" % (it, J, deltaJ, deltaX, lam, mu, "" if not prevFailed else "REJECTED"))
    return dict(JStart=JStart, JFinal=pJ, iterations=it, failedIterations=failed, dXFinal=deltaX, dJFinal=deltaJ,
                linearSolverFailure=lsFail, trace=trace)
