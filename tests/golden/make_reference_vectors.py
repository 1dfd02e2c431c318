"""Golden vectors from the reference's OWN machine-generated expressions, evaluated where they lie.

Two pieces of the hot path exist in the reference only as symbolic-algebra dumps:

  * RotationVector::angularVelocityAndJacobian, the 3 x 3 d(S(p) pdot)/dp
    (Schweizer-Messer/sm_kinematics/src/RotationVector.cpp:126-176: `double t5 = ...; ... J(2,2) = ...;`)
  * EquidistantDistortion::distort's "MATLAB generated Jacobian" and distortParameterJacobian
    (aslam_cv/aslam_cameras/include/aslam/cameras/implementation/EquidistantDistortion.hpp:54-170, 258-273)

and the camera models of rows E1, E2, E4, E5 are straight-line scalar code there:

  * RadialTangentialDistortion::distort (+ Jacobian), distortParameterJacobian
    (.../implementation/RadialTangentialDistortion.hpp:47-63, 162-180)
  * PinholeProjection::euclideanToKeypoint (+ Jacobian), the intrinsics / distortion Jacobians (PinholeProjection.hpp:123-141, 340-351)
  * OmniProjection::euclideanToKeypoint (+ Jacobian), euclideanToKeypointIntrinsicsJacobian (OmniProjection.hpp:147-177, 399-418)
  * DoubleSphereProjection::euclideanToKeypoint (+ Jacobian), euclideanToKeypointIntrinsicsJacobian
    (DoubleSphereProjection.hpp:166-217, 460-501)

The oracle and the product restate all of them (the first two in closed analytic form).  This script READS those statement blocks as text out of
/root/reference (nothing is copied into the repository), checks that they consist of nothing but scalar arithmetic over
the expected names (numbers, + - * / ( ) , and pow / sqrt / sin / cos / atan), evaluates them in IEEE double at seeded
inputs and stores inputs + outputs in tests/golden/reference_generated_jacobians.npz.  tests/test_oracle_pins.py compares
oracle and product against the stored vectors (on any machine) and, where /root/reference is present, re-runs this
script and checks that the stored file is what it produces.

Run:  python tests/golden/make_reference_vectors.py [--check]
"""
import math
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_generated_jacobians.npz")
REF = os.environ.get("MULTICAL_REFERENCE", "/root/reference")
RV_FILE = "Schweizer-Messer/sm_kinematics/src/RotationVector.cpp"
EQ_FILE = "aslam_cv/aslam_cameras/include/aslam/cameras/implementation/EquidistantDistortion.hpp"

_FUNCS = {"pow": math.pow, "sqrt": math.sqrt, "sin": math.sin, "cos": math.cos, "atan": math.atan, "exp": math.exp}
_TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\d+)|([A-Za-z_]\w*)|([-+*/(),]))")


def _lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        return "".join(f.readlines()[lo - 1:hi])


def _statements(text, subst):
    """[(target, python expression)] of every `[double] target = expr;` in text.  `subst` renames C lvalues / inputs
    (e.g. 'p[0]' -> 'p0').  Anything that is not plain scalar arithmetic over known names raises."""
    text = re.sub(r"//[^\n]*", "", text)
    for a, b in subst:
        text = text.replace(a, b)
    out = []
    for m in re.finditer(r"(?:double\s+)?([A-Za-z_]\w*(?:\(\s*\d\s*,\s*\d\s*\))?)\s*=\s*([^;=]+);", text):
        target, expr = re.sub(r"\s", "", m.group(1)), " ".join(m.group(2).split())
        out.append((target, expr))
    return out


def _check_expr(expr, names):
    pos = 0
    while pos < len(expr):
        m = _TOKEN.match(expr, pos)
        if not m:
            raise ValueError("unexpected text in a reference expression: %r" % expr[pos:pos + 30])
        if m.group(2) and m.group(2) not in names and m.group(2) not in _FUNCS:
            raise ValueError("unexpected name %r in a reference expression" % m.group(2))
        pos = m.end()


def _run(stmts, env):
    env = dict(env)
    res = {}
    for target, expr in stmts:
        _check_expr(expr, env)
        v = eval(expr, {"__builtins__": {}}, dict(_FUNCS, **env))   # arithmetic only: every token was checked above
        if "(" in target:
            res[target] = v
        else:
            env[target] = v
    return res, env


# ---------------------------------------------------------------------------------------------------------------------
# Straight-line scalar code (camera models).  A block of the reference's source is read as a sequence of
# `name = expr;`, `name += expr;`, `name *= expr;` statements over plain names after these renames: `_fu` -> `fu`,
# `p[0]` -> `p0`, `J(i, j)` -> `J_i_j`, `std::sqrt` -> `sqrt`, `double(1.0)` -> `1.0`, `(c) ? a : b` -> Python's
# conditional.  Every expression is token-checked (numbers, known names, + - * / ( ) , > and sqrt / atan / pow); a line
# of the block that is not such a statement must be named in `skip` (a regular expression) or the script stops.
_TOKEN2 = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][-+]?\d+)?)|([A-Za-z_]\w*)|([-+*/(),><]))")
_KEYWORDS = {"if", "else"}


def _rename(text, arrays):
    text = re.sub(r"//[^\n]*", "", text)
    text = text.replace("std::", "")
    text = re.sub(r"\bdouble\(\s*1\.0\s*\)", "1.0", text)
    for a in arrays:
        text = re.sub(r"\b%s\[(\d)\]" % re.escape(a), r"%s\1" % a, text)
    text = re.sub(r"\b(Jd|J)\(\s*(\d)\s*,\s*(\d)\s*\)", r"\1_\2_\3", text)
    text = re.sub(r"\b_([a-z]\w*)\b", r"\1", text)                 # members: _fu -> fu, _k1 -> k1
    text = re.sub(r"\(([^()?;]+)\)\s*\?\s*([^:;]+?)\s*:\s*([^;]+?);", r"(\2) if (\1) else (\3);", text)
    return text


def _program(path, lo, hi, arrays, skip=None):
    return _program_from_text(_lines(path, lo, hi), arrays, skip, "%s:%d-%d" % (path, lo, hi))


def _program_from_text(text, arrays, skip=None, where="text"):
    text = _rename(text, arrays)
    prog = []
    for stmt in text.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        if skip and re.search(skip, stmt):
            continue
        m = re.match(r"(?:const )?(?:double ?&? ?)?([A-Za-z_]\w*) ?(\+|\*)?= ?(.+)$", stmt)
        if m:
            prog.append((m.group(1), m.group(2) or "", m.group(3)))
            continue
        if re.match(r"(?:const )?double [A-Za-z_]\w*(?: ?, ?[A-Za-z_]\w*)*$", stmt):   # `double a, b, c`: declarations only
            continue
        raise ValueError("%s: not a scalar statement: %r" % (where, stmt))
    return prog


def _exec(prog, env):
    env = dict(env)
    for target, op, expr in prog:
        pos = 0
        while pos < len(expr):
            m = _TOKEN2.match(expr, pos)
            if not m:
                raise ValueError("unexpected text in a reference expression: %r" % expr[pos:pos + 30])
            if m.group(2) and m.group(2) not in env and m.group(2) not in _FUNCS and m.group(2) not in _KEYWORDS:
                raise ValueError("unexpected name %r in a reference expression" % m.group(2))
            pos = m.end()
        v = eval(expr, {"__builtins__": {}}, dict(_FUNCS, **env))   # arithmetic only: every token was checked above
        if op == "+":
            v = env[target] + v
        elif op == "*":
            v = env[target] * v
        env[target] = v
    return env


RT_FILE = "aslam_cv/aslam_cameras/include/aslam/cameras/implementation/RadialTangentialDistortion.hpp"
PH_FILE = "aslam_cv/aslam_cameras/include/aslam/cameras/implementation/PinholeProjection.hpp"
OM_FILE = "aslam_cv/aslam_cameras/include/aslam/cameras/implementation/OmniProjection.hpp"
DS_FILE = "aslam_cv/aslam_cameras/include/aslam/cameras/implementation/DoubleSphereProjection.hpp"


def _programs():
    P = {}
    P["radtan"] = _program(RT_FILE, 47, 63, ["y"])                       # distort(y, J): J first, then y in place
    P["radtan_k"] = _program(RT_FILE, 162, 180, ["imageY"], skip=r"MatrixBase|J\.derived|J\.setZero")
    P["equi"] = _program(EQ_FILE, 17, 28, ["y"])                        # distort(y) value: r .. y[1] *= scaling
    P["equi_J"] = [(t, "", e) for t, e in _statements(_lines(EQ_FILE, 54, 170),
                                                      [("y[0]", "y0"), ("y[1]", "y1")] + [("_k%d" % i, "k%d" % i) for i in range(1, 5)])
                   if t.startswith("J(")]
    P["equi_J"] = [(re.sub(r"J\((\d),(\d)\)", r"J_\1_\2", t), o, e) for t, o, e in P["equi_J"]]
    P["equi_k"] = [(re.sub(r"J\((\d),(\d)\)", r"J_\1_\2", t), "", e) for t, e in
                   _statements(_lines(EQ_FILE, 258, 273), [("y[0]", "y0"), ("y[1]", "y1")] + [("_k%d" % i, "k%d" % i) for i in range(1, 5)])
                   if t.startswith("J(")]
    P["pinhole_a"] = _program(PH_FILE, 123, 127, ["p", "outKeypoint"])   # rz, rz2, normalised point
    P["pinhole_b"] = _program(PH_FILE, 133, 141, ["p", "outKeypoint"])   # J through Jd, keypoint
    P["pinhole_i"] = _program(PH_FILE, 340, 351, ["p", "kp"], skip=r"keypoint_t kp|distortion\.distort")
    P["omni_a"] = _program(OM_FILE, 147, 159, ["p", "outKeypoint"])      # rz .. J(1, 2) before the distortion
    P["omni_b"] = _program(OM_FILE, 164, 177, ["p", "outKeypoint"])      # J through Jd, keypoint
    P["omni_i"] = _program(OM_FILE, 399, 405, ["p", "kp", "Jxi"], skip=r"Vector2d Jxi")
    P["omni_i2"] = _program(OM_FILE, 414, 418, ["p", "kp"])
    P["ds"] = _program(DS_FILE, 166, 217, ["p", "outKeypoint"], skip=r"fov_parameter")
    P["ds_i"] = _program(DS_FILE, 460, 501, ["p"], skip=r"J\.setZero")
    return P


def _distort(P, dist, y, k):
    """(distorted y, 2 x 2 Jacobian) through the reference's distort(y, J) of the named model"""
    env = dict(y0=float(y[0]), y1=float(y[1]), k1=float(k[0]), k2=float(k[1]))
    if dist == 0:
        return np.array(y, float), np.eye(2)
    if dist == 1:
        env.update(p1=float(k[2]), p2=float(k[3]))
        e = _exec(P["radtan"], env)
        J = np.array([[e["J_0_0"], e["J_0_1"]], [e["J_1_0"], e["J_1_1"]]])
        return np.array([e["y0"], e["y1"]]), J
    env.update(k3=float(k[2]), k4=float(k[3]))
    ej = _exec(P["equi_J"], env)                                      # the generated Jacobian reads the undistorted y
    e = _exec(P["equi"], env)
    J = np.array([[ej["J_0_0"], ej["J_0_1"]], [ej["J_1_0"], ej["J_1_1"]]])
    return np.array([e["y0"], e["y1"]]), J


def _distort_k(P, dist, y):
    env = dict(y0=float(y[0]), y1=float(y[1]), imageY0=float(y[0]), imageY1=float(y[1]))
    e = _exec(P["radtan_k"] if dist == 1 else P["equi_k"], env)
    return np.array([[e.get("J_%d_%d" % (i, j), 0.0) for j in range(4)] for i in range(2)])


def camera_vectors(rng, n=48):
    """Per (projection, distortion) pair of north_star: points in front of the camera, intrinsics near SURVEY.md 8d's, and
    the reference's keypoint, 2 x 3 point Jacobian, intrinsics Jacobian and distortion Jacobian."""
    P = _programs()
    out = {}
    radtan0, equi0 = np.array([-0.28, 0.07, 2e-4, 2e-5]), np.array([-0.01, 0.02, -0.01, 0.002])
    for model, dist, tag in ((0, 0, "pinhole_none"), (0, 1, "pinhole_radtan"), (0, 2, "pinhole_equi"), (1, 0, "omni_none"),
                             (1, 1, "omni_radtan"), (1, 2, "omni_equi"), (2, 0, "ds_none")):
        pd = 4 + model
        PR, DK, PT = np.zeros((n, pd)), np.zeros((n, 4)), np.zeros((n, 3))
        KP, JP, JI, JD = np.zeros((n, 2)), np.zeros((n, 2, 3)), np.zeros((n, 2, pd)), np.zeros((n, 2, 4))
        for k in range(n):
            fu, fv, cu, cv = 460.0 * rng.uniform(0.8, 1.2), 460.0 * rng.uniform(0.8, 1.2), 376.0 + rng.uniform(-20, 20), 240.0 + rng.uniform(-20, 20)
            pt = np.array([rng.uniform(-1.0, 1.0), rng.uniform(-0.7, 0.7), rng.uniform(0.8, 4.0)])
            if k % 8 == 7:
                pt[:2] *= 0.02                                           # near the optical axis
            dk = (radtan0 if dist == 1 else equi0) * rng.uniform(0.5, 1.5, 4) if dist else np.zeros(4)
            env = dict(fu=fu, fv=fv, cu=cu, cv=cv, p0=float(pt[0]), p1=float(pt[1]), p2=float(pt[2]))
            if model == 0:
                PR[k] = [fu, fv, cu, cv]
                e = _exec(P["pinhole_a"], env)
                yd, Jd = _distort(P, dist, [e["outKeypoint0"], e["outKeypoint1"]], dk)
                e.update(outKeypoint0=float(yd[0]), outKeypoint1=float(yd[1]))
                e.update({"Jd_%d_%d" % (i, j): float(Jd[i, j]) for i in range(2) for j in range(2)})
                e = _exec(P["pinhole_b"], e)
                # intrinsics: kp = distorted normalised point (`_distortion.distort(kp)` between the two halves of the block)
                ei = _exec(P["pinhole_i"][:3], env)
                ydi, _ = _distort(P, dist, [ei["kp0"], ei["kp1"]], dk)
                ei.update(kp0=float(ydi[0]), kp1=float(ydi[1]))
                ei = _exec(P["pinhole_i"][3:], ei)
                ynorm = [pt[0] / pt[2], pt[1] / pt[2]]
            elif model == 1:
                xi = 1.2 * rng.uniform(0.7, 1.2)
                PR[k] = [xi, fu, fv, cu, cv]
                env.update(xi=xi, d=math.sqrt(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]))     # `double d = p.norm();`
                e = _exec(P["omni_a"], env)
                yd, Jd = _distort(P, dist, [e["outKeypoint0"], e["outKeypoint1"]], dk)
                ynorm = [e["outKeypoint0"], e["outKeypoint1"]]
                e.update(outKeypoint0=float(yd[0]), outKeypoint1=float(yd[1]))
                e.update({"Jd_%d_%d" % (i, j): float(Jd[i, j]) for i in range(2) for j in range(2)})
                e = _exec(P["omni_b"], e)
                ei = _exec(P["omni_i"], env)
                ydi, Jdi = _distort(P, dist, [ei["kp0"], ei["kp1"]], dk)
                # `Jd.row(0) *= _fu; Jd.row(1) *= _fv; J.col(0) = Jd * Jxi;` (OmniProjection.hpp:410-412): Eigen row scaling and a
                # 2 x 2 by 2 x 1 product, written out
                jx = np.array([ei["Jxi0"], ei["Jxi1"]])
                col0 = np.array([fu * (Jdi[0, 0] * jx[0] + Jdi[0, 1] * jx[1]), fv * (Jdi[1, 0] * jx[0] + Jdi[1, 1] * jx[1])])
                ei.update(kp0=float(ydi[0]), kp1=float(ydi[1]))
                ei = _exec(P["omni_i2"], ei)
                ei["J_0_0"], ei["J_1_0"] = float(col0[0]), float(col0[1])
            else:
                xi, alpha = -0.2 * rng.uniform(0.5, 1.5), 0.6 * rng.uniform(0.8, 1.2)
                PR[k] = [xi, alpha, fu, fv, cu, cv]
                env.update(xi=xi, alpha=alpha)
                e = _exec(P["ds"], env)
                ei = _exec(P["ds_i"], env)
                ynorm = None
            DK[k], PT[k] = dk, pt
            KP[k] = [e["outKeypoint0"], e["outKeypoint1"]]
            JP[k] = [[e["J_%d_%d" % (i, j)] for j in range(3)] for i in range(2)]
            JI[k] = [[ei.get("J_%d_%d" % (i, j), 0.0) for j in range(pd)] for i in range(2)]
            if dist:
                # euclideanToKeypointDistortionJacobian: distortParameterJacobian at the normalised point, rows scaled by fu, fv
                # (`J.row(0) *= _fu; J.row(1) *= _fv;` PinholeProjection.hpp:371-372, OmniProjection.hpp:441-442)
                Jk = _distort_k(P, dist, ynorm)
                JD[k] = Jk * np.array([[fu], [fv]])
        out.update({"cam_%s_proj" % tag: PR, "cam_%s_dist" % tag: DK, "cam_%s_p" % tag: PT, "cam_%s_kp" % tag: KP,
                    "cam_%s_Jp" % tag: JP, "cam_%s_Ji" % tag: JI, "cam_%s_Jd" % tag: JD})
    # the radial-tangential distortion on its own (row E2), like the equidistant block above
    n2 = 64
    Y, K = rng.uniform(-1.0, 1.0, (n2, 2)), radtan0 * rng.uniform(0.5, 1.5, (n2, 4))
    YD, JY, JK = np.zeros((n2, 2)), np.zeros((n2, 2, 2)), np.zeros((n2, 2, 4))
    for k in range(n2):
        YD[k], JY[k] = _distort(P, 1, Y[k], K[k])
        JK[k] = _distort_k(P, 1, Y[k])
    out.update(rt_y=Y, rt_k=K, rt_yd=YD, rt_Jy=JY, rt_Jk=JK)
    return out


def rotation_vector_vectors(rng, n=96):
    text = _lines(RV_FILE, 126, 176)
    subst = [("pdot[%d]" % i, "pdot%d" % i) for i in range(3)] + [("p[%d]" % i, "p%d" % i) for i in range(3)]
    stmts = [(t, e) for t, e in _statements(text, subst) if re.match(r"(d?t\d+|J\(\d,\d\))$", t)]
    assert sum(1 for t, _ in stmts if t.startswith("J(")) == 9, [t for t, _ in stmts]
    P, V, D = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3, 3))
    for k in range(n):
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(0.3, 3.0) if k < n - 16 else 10.0 ** rng.uniform(-2.0, -0.6)
        P[k], V[k] = ax * ang, rng.standard_normal(3)
        res, _ = _run(stmts, {"p%d" % i: float(P[k, i]) for i in range(3)} | {"pdot%d" % i: float(V[k, i]) for i in range(3)})
        for i in range(3):
            for j in range(3):
                D[k, i, j] = res["J(%d,%d)" % (i, j)]
    return dict(rv_p=P, rv_pdot=V, rv_dSv_dp=D)


def equidistant_vectors(rng, n=96):
    subst = [("y[0]", "y0"), ("y[1]", "y1")] + [("_k%d" % i, "k%d" % i) for i in range(1, 5)]
    jy = [(t, e) for t, e in _statements(_lines(EQ_FILE, 54, 170), subst) if t.startswith("J(")]
    jk = [(t, e) for t, e in _statements(_lines(EQ_FILE, 258, 273), subst) if t.startswith("J(")]
    assert len(jy) == 4 and len(jk) == 8, (len(jy), len(jk))
    Y, K = np.zeros((n, 2)), np.zeros((n, 4))
    Jy, Jk = np.zeros((n, 2, 2)), np.zeros((n, 2, 4))
    for k in range(n):
        Y[k] = rng.uniform(-1.2, 1.2, 2) if k < n - 16 else rng.uniform(-0.05, 0.05, 2)
        K[k] = np.array([-0.01, 0.02, -0.01, 0.002]) * rng.uniform(0.5, 1.5, 4)
        env = dict(y0=float(Y[k, 0]), y1=float(Y[k, 1]), k1=float(K[k, 0]), k2=float(K[k, 1]), k3=float(K[k, 2]), k4=float(K[k, 3]))
        r1, _ = _run(jy, env)
        r2, _ = _run(jk, env)
        for i in range(2):
            for j in range(2):
                Jy[k, i, j] = r1["J(%d,%d)" % (i, j)]
            for j in range(4):
                Jk[k, i, j] = r2["J(%d,%d)" % (i, j)]
    return dict(eq_y=Y, eq_k=K, eq_Jy=Jy, eq_Jk=Jk)


QA_FILE = "Schweizer-Messer/sm_kinematics/src/quaternion_algebra.cpp"


def kinematics_vectors(rng, n=32):
    """quat2r (quaternion_algebra.cpp:90-98, nine `R(i,j) = ...` statements) and the Rodrigues matrix of
    RotationVector::parametersToRotationMatrix (RotationVector.cpp:31-43: scalar statements, then the nine expressions of the
    `C << ...` comma initialiser, row by row)."""
    text = re.sub(r"\bR\(\s*(\d)\s*,\s*(\d)\s*\)", r"R_\1_\2", _lines(QA_FILE, 90, 98))
    q2r = _program_from_text(text, ["q"])
    body = _lines(RV_FILE, 31, 43)
    head, init = body.split("C <<")
    rv = _program_from_text(head, ["axis"])
    cells = [c.strip() for c in init.replace(";", "").replace("\n", " ").split(",")]
    assert len(cells) == 9, cells
    Q, R, Pm, Cm = np.zeros((n, 4)), np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 3, 3))
    for k in range(n):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        e = _exec(q2r, {"q%d" % i: float(q[i]) for i in range(4)})
        Q[k], R[k] = q, [[e["R_%d_%d" % (i, j)] for j in range(3)] for i in range(3)]
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(0.05, 3.1)
        prm = ax * ang
        # `double angle = parameters.norm(); ... axis = parameters * recip_angle;` (RotationVector.cpp:14, 27): Eigen calls, written out
        angle = math.sqrt(prm[0] * prm[0] + prm[1] * prm[1] + prm[2] * prm[2])
        recip = 1.0 / angle
        env = dict(angle=angle, axis0=float(prm[0] * recip), axis1=float(prm[1] * recip), axis2=float(prm[2] * recip))
        env = _exec(rv, env)
        vals = _exec([("C%d" % i, "", c) for i, c in enumerate(cells)], env)
        Pm[k], Cm[k] = prm, np.array([vals["C%d" % i] for i in range(9)]).reshape(3, 3)
    return dict(kin_q=Q, kin_quat2r=R, kin_rv=Pm, kin_rv_C=Cm)


ME_FILE = "aslam_optimizer/aslam_backend/src/MEstimatorPolicies.cpp"


def mestimator_vectors(rng, n=40):
    """MEstimator::getWeight of the five robust kernels (MEstimatorPolicies.cpp:30-33, 46-49, 62-64, 77-79, 114-116): the
    bodies read as `w = expr` (`return` renamed), members `_sigma2`, `_k`, `_k2`, `_epsilon` as inputs"""
    bodies = dict(geman_mcclure=(31, 32), cauchy=(47, 48), huber=(63, 63), bz=(78, 78), blake_zisserman=(115, 115))
    out = {}
    e2 = np.concatenate([[0.0, 1e-12], np.exp(rng.uniform(-6, 6, n - 2))])
    a = np.exp(rng.uniform(-2, 2, n))
    out["mest_e2"], out["mest_a"] = e2, a
    for name, (lo, hi) in bodies.items():
        text = _lines(ME_FILE, lo, hi).replace("return ", "w = ")
        text = re.sub(r"w = ([^?;()]+?)\s*\?", r"w = (\1) ?", text)          # Huber's condition carries no parentheses
        prog = _program_from_text(text, [], None, "%s:%d-%d" % (ME_FILE, lo, hi))
        W = np.zeros((n, n))
        for i in range(n):
            for j in range(n):
                env = dict(error=float(e2[i]), mahalanobis2=float(e2[i]), sigma2=float(a[j]), k=float(a[j]), k2=float(a[j]) * float(a[j]),
                           epsilon=float(a[j]))
                W[i, j] = _exec(prog, env)["w"]
        out["mest_w_" + name] = W
    return out


def update_quat_vectors(rng, n=32):
    """sm::kinematics::updateQuat (quaternion_algebra.cpp:302-318) through axisAngle2quat's regular branch (:216, :219):
    na and ct as the source states them, the axis product `a * na` component-wise, then the four retq statements"""
    aa = _program_from_text(_lines(QA_FILE, 216, 216) + _lines(QA_FILE, 219, 219), [], None, QA_FILE + ":216,219")
    uq = _program_from_text(_lines(QA_FILE, 310, 310) + _lines(QA_FILE, 312, 315), ["q", "dq3", "retq"], None, QA_FILE + ":310-315")
    Q, D, R = np.zeros((n, 4)), np.zeros((n, 3)), np.zeros((n, 4))
    for k in range(n):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        dq = rng.standard_normal(3) * (10.0 ** rng.uniform(-3, 0.3))     # above the small-angle branch (theta > eps^(1/4) = 1.2e-4)
        theta = float(np.linalg.norm(dq))
        e = _exec(aa, dict(theta=theta))
        env = {"q%d" % i: float(q[i]) for i in range(4)}
        env.update({"dq3%d" % i: float(dq[i]) * e["na"] for i in range(3)})
        env["dq33"] = e["ct"]
        r = _exec(uq, env)
        Q[k], D[k], R[k] = q, dq, [r["retq%d" % i] for i in range(4)]
    return dict(uq_q=Q, uq_dq=D, uq_out=R)


def generate():
    rng = np.random.Generator(np.random.PCG64(20221023))
    out = {}
    out.update(rotation_vector_vectors(rng))
    out.update(equidistant_vectors(rng))
    out.update(camera_vectors(rng))
    out.update(kinematics_vectors(rng))
    out.update(mestimator_vectors(rng))      # appended in this order: the vectors above keep their random stream
    out.update(update_quat_vectors(rng))
    return out


def main():
    data = generate()
    if "--check" in sys.argv:
        old = np.load(OUT)
        for k, v in data.items():
            assert np.array_equal(old[k], v), k
        print("stored vectors are what the reference's expressions give")
        return
    np.savez(OUT, **data)
    print("wrote %s: %s" % (OUT, {k: v.shape for k, v in data.items()}))


if __name__ == "__main__":
    main()
