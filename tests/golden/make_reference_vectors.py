"""Golden vectors from the reference's OWN machine-generated expressions, evaluated where they lie.

Two pieces of the hot path exist in the reference only as symbolic-algebra dumps:

  * RotationVector::angularVelocityAndJacobian, the 3 x 3 d(S(p) pdot)/dp
    (Schweizer-Messer/sm_kinematics/src/RotationVector.cpp:126-176: `double t5 = ...; ... J(2,2) = ...;`)
  * EquidistantDistortion::distort's "MATLAB generated Jacobian" and distortParameterJacobian
    (aslam_cv/aslam_cameras/include/aslam/cameras/implementation/EquidistantDistortion.hpp:54-170, 258-273)

The oracle and the product restate them in closed analytic form.  This script READS those statement blocks as text out of
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

_FUNCS = {"pow": math.pow, "sqrt": math.sqrt, "sin": math.sin, "cos": math.cos, "atan": math.atan}
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


def generate():
    rng = np.random.Generator(np.random.PCG64(20221023))
    out = {}
    out.update(rotation_vector_vectors(rng))
    out.update(equidistant_vectors(rng))
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
