"""The caller side of the hot path: a Python mirror of the `aslam_backend` / `aslam_splines` /
`kalibr_errorterms` / `incremental_calibration` names that Multical's front-end uses to build its problem
(kalibr_sensor_calibration/sensors_and_targets.py:274-305, 542-630, 1103-1246; calibrator.py:94-123), and the LOWERING
PASS that turns the expression trees those calls create into the typed SoA batches of the C ABI.

The objects here are symbolic: `T_c_b * T_w_b.inverse() * T_w_p` builds a small tree, `problem.addErrorTerm(err)` stores
it, and `lower(problem)` pattern-matches every error term against the six shapes Multical creates (F1 reprojection,
F3 accelerometer, F4 gyroscope, F5 LiDAR point-to-plane, F7 motion; F2 lives in kalibr_camera_calibration) and emits a
`ProblemSpec` with the design variables in problem order and consecutive same-shaped terms merged into batches.  A term
that matches none of the shapes raises `UnsupportedErrorTerm` -- this library has no CPU fallback; an integrator keeps
such terms on the reference's CPU path.

Names and argument meaning follow the boost.python exports (aslam_backend_python/src/BackendExpressions.cpp:83-245,
python/aslam_backend/__init__.py:10-36, aslam_splines_python, kalibr_errorterms).  Numeric evaluation (`toScalar`,
`toTransformationMatrix`) is implemented for what the front-end calls while building the problem (time expressions and
the LiDAR sign check); it uses SciPy's B-spline evaluation and is not on the hot path.
"""
import numpy as np

from . import spec as S

CALIBRATION_GROUP_ID, HELPER_GROUP_ID = S.CALIBRATION_GROUP_ID, S.HELPER_GROUP_ID


class UnsupportedErrorTerm(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ helpers
def _quat2r(q):
    """quat2r of sm_kinematics (quaternion_algebra.cpp): q = [x y z w]."""
    x, y, z, w = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y + w * z), 2 * (x * z - w * y)],
                     [2 * (x * y - w * z), w * w - x * x + y * y - z * z, 2 * (y * z + w * x)],
                     [2 * (x * z + w * y), 2 * (y * z - w * x), w * w - x * x - y * y + z * z]])


def _rotvec2r(th):
    from .synth import rotvec_to_matrix
    return rotvec_to_matrix(np.asarray(th, dtype=np.float64))


class Transformation(object):
    """sm.Transformation: rotation quaternion [x y z w] + translation (T = [C(q) t; 0 1])."""

    def __init__(self, q_or_T=None, t=None):
        if q_or_T is None:
            self._q, self._t = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
        elif t is None:
            from .synth import quat_from_matrix
            T = np.asarray(q_or_T, dtype=np.float64)
            self._q, self._t = quat_from_matrix(T[:3, :3]), T[:3, 3].copy()
        else:
            self._q, self._t = np.asarray(q_or_T, dtype=np.float64).copy(), np.asarray(t, dtype=np.float64).copy()

    def q(self):
        return self._q

    def t(self):
        return self._t

    def T(self):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = _quat2r(self._q), self._t
        return M


# ------------------------------------------------------------------------------------------------ design variables
class DesignVariable(object):
    kind = None

    def __init__(self, params):
        self.params = np.asarray(params, dtype=np.float64).ravel().copy()
        self._active = False
        self.scaling = 1.0

    def setActive(self, a):
        self._active = bool(a)

    def isActive(self):
        return self._active


class RotationQuaternionDv(DesignVariable):
    kind = S.DV_ROTATION_QUATERNION

    def toExpression(self):
        return RotationExpression(("r_leaf", self))


class EuclideanPointDv(DesignVariable):
    kind = S.DV_EUCLIDEAN_POINT

    def toExpression(self):
        return EuclideanExpression(("e_leaf", self))


class EuclideanDirection(DesignVariable):
    kind = S.DV_EUCLIDEAN_DIRECTION

    def toExpression(self):
        return EuclideanExpression(("e_leaf", self))


class MatrixBasicDv(DesignVariable):
    """aopt.MatrixBasicDv(A, updatePattern) (aslam_backend_expressions/src/MatrixBasic.cpp:10-47; used by the
    scale-misalignment IMU models, sensors_and_targets.py:1435-1447, 1598-1606)."""
    kind = S.DV_MATRIX_BASIC

    def __init__(self, A, updatePattern=None):
        DesignVariable.__init__(self, np.asarray(A, dtype=np.float64).reshape(3, 3))
        self.pattern = (np.ones((3, 3), int) if updatePattern is None else np.asarray(updatePattern, dtype=int).reshape(3, 3)).ravel()

    def toExpression(self):
        return MatrixExpression(("m_leaf", self))

    def toMatrix3x3(self):
        return self.params.reshape(3, 3)


class Scalar(DesignVariable):
    kind = S.DV_SCALAR

    def __init__(self, v):
        DesignVariable.__init__(self, [v])

    def toExpression(self):
        return ScalarExpression(("s_leaf", self))

    def toScalar(self):
        return float(self.params[0])

    toValue = toScalar


class TransformationDv(object):
    """python/aslam_backend/__init__.py:10-36."""

    def __init__(self, transformation, rotationActive=True, translationActive=True):
        if not isinstance(transformation, Transformation):
            raise RuntimeError("The transformation must be an sm transformation")
        self.q = RotationQuaternionDv(transformation.q())
        self.q.setActive(rotationActive)
        self.t = EuclideanPointDv(transformation.t())
        self.t.setActive(translationActive)
        self.expression = TransformationExpression(("t_leaf", self.q, self.t))

    def toExpression(self):
        return self.expression

    def numDesignVariables(self):
        return 2

    def getDesignVariable(self, i):
        if i == 0:
            return self.q
        if i == 1:
            return self.t
        raise RuntimeError("Index out of bounds: %d >= 2" % i)

    designVariable = getDesignVariable

    def T(self):
        return self.expression.toTransformationMatrix()


# ------------------------------------------------------------------------------------------------ splines
class BSpline(object):
    """bsplines.BSpline (value spline) -- knots, coefficients (dim x n), order; enough of the interface for the front-end."""

    def __init__(self, order, knots=None, coeffs=None):
        self.order = int(order)
        self.knots = None if knots is None else np.asarray(knots, dtype=np.float64).copy()
        self.coeffs = None if coeffs is None else np.asarray(coeffs, dtype=np.float64).copy()

    def splineOrder(self):
        return self.order

    def initConstantSpline(self, t_min, t_max, num_segments, constant):
        """BSpline::initConstantSpline (bsplines/src/BSpline.cpp): uniform knots, every coefficient = constant."""
        k = self.order
        dt = (t_max - t_min) / num_segments
        self.knots = t_min + (np.arange(num_segments + 2 * k - 1) - (k - 1)) * dt
        c = np.asarray(constant, dtype=np.float64).ravel()
        self.coeffs = np.repeat(c[:, None], num_segments + k - 1, axis=1)

    def t_min(self):
        return float(self.knots[self.order - 1])

    def t_max(self):
        return float(self.knots[-self.order])

    def eval(self, t, derivative=0):
        from scipy.interpolate import BSpline as _B
        b = _B(self.knots, self.coeffs.T, self.order - 1, extrapolate=False)
        t = min(max(t, self.t_min()), np.nextafter(self.t_max(), -np.inf))
        return np.asarray(b.derivative(derivative)(t) if derivative else b(t), dtype=np.float64)


class BSplinePose(BSpline):
    """bsplines.BSplinePose with the rotation-vector parametrisation (curve value = [r; theta])."""

    def transformation(self, t):
        v = self.eval(t)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = _rotvec2r(v[3:]), v[:3]
        return T


class _SplineDv(object):
    def __init__(self, spline):
        self._spline = spline
        self.active = True
        self.group = HELPER_GROUP_ID

    def spline(self):
        return self._spline

    def numDesignVariables(self):
        return self._spline.coeffs.shape[1]


class BSplinePoseDesignVariable(_SplineDv):
    """aslam_splines BSplinePoseDesignVariable (ASP/src/BSplinePoseDesignVariable.cpp)."""

    def transformationAtTime(self, time_expression, bufferTmin, bufferTmax):
        if not isinstance(time_expression, ScalarExpression):
            raise TypeError("transformationAtTime takes a ScalarExpression")
        return TransformationExpression(("t_spline_at", self, time_expression, float(bufferTmin), float(bufferTmax)))

    def transformation(self, t):
        return TransformationExpression(("t_spline_at", self, ScalarExpression(float(t)), 0.0, 0.0))

    def orientation(self, t):
        return RotationExpression(("r_spline", self, float(t)))

    def linearAcceleration(self, t):
        return EuclideanExpression(("e_spline", self, float(t), "lin_acc"))

    def angularVelocityBodyFrame(self, t):
        return EuclideanExpression(("e_spline", self, float(t), "ang_vel_b"))

    def angularAccelerationBodyFrame(self, t):
        return EuclideanExpression(("e_spline", self, float(t), "ang_acc_b"))


class EuclideanBSplineDesignVariable(_SplineDv):
    def toEuclideanExpression(self, t, derivative_order):
        return EuclideanExpression(("e_bias_spline", self, float(t), int(derivative_order)))


def addSplineDesignVariables(problem, dvc, setActive=True, group_id=HELPER_GROUP_ID):
    """incremental_calibration addSplineDesignVariables: every coefficient becomes a design variable of the group."""
    dvc.active, dvc.group = bool(setActive), group_id
    problem._add_spline(dvc)


# ------------------------------------------------------------------------------------------------ expressions
def _const3(v):
    return EuclideanExpression(("e_const", np.asarray(v, dtype=np.float64).ravel().copy()))


class ScalarExpression(object):
    def __init__(self, node):
        self.node = ("s_const", float(node)) if isinstance(node, (int, float, np.floating)) else node

    def _wrap(self, o):
        return o if isinstance(o, ScalarExpression) else ScalarExpression(float(o))

    def __add__(self, o):
        return ScalarExpression(("s_add", self, self._wrap(o)))

    def __sub__(self, o):
        return ScalarExpression(("s_sub", self, self._wrap(o)))

    def __mul__(self, o):
        return ScalarExpression(("s_mul", self, self._wrap(o)))

    def __truediv__(self, o):
        return ScalarExpression(("s_div", self, self._wrap(o)))

    __div__ = __truediv__

    def toScalar(self):
        n = self.node
        k = n[0]
        if k == "s_const":
            return n[1]
        if k == "s_leaf":
            return n[1].toScalar()
        if k == "s_dot":
            return float(np.dot(n[1].toEuclidean(), n[2].toEuclidean()))
        a, b = n[1].toScalar(), n[2].toScalar()
        return {"s_add": a + b, "s_sub": a - b, "s_mul": a * b, "s_div": a / b if k == "s_div" else 0.0}[k]

    toValue = toScalar

    def affine(self):
        """(design variable or None, constant) if the expression is dv + constants, else None."""
        n = self.node
        if n[0] == "s_const":
            return None, n[1]
        if n[0] == "s_leaf":
            return n[1], 0.0
        if n[0] in ("s_add", "s_sub"):
            a, b = n[1].affine(), n[2].affine()
            if a is None or b is None or (a[0] is not None and b[0] is not None) or (n[0] == "s_sub" and b[0] is not None):
                return None
            return (a[0] if a[0] is not None else b[0]), (a[1] + b[1] if n[0] == "s_add" else a[1] - b[1])
        return None


class EuclideanExpression(object):
    def __init__(self, node):
        self.node = ("e_const", np.asarray(node, dtype=np.float64).ravel().copy()) if isinstance(node, (np.ndarray, list)) else node

    def __add__(self, o):
        return EuclideanExpression(("e_add", self, o))

    def __sub__(self, o):
        return EuclideanExpression(("e_sub", self, o))

    def cross(self, o):
        return EuclideanExpression(("e_cross", self, o))

    def dot(self, o):
        return ScalarExpression(("s_dot", self, o))

    def toEuclidean(self):
        n = self.node
        k = n[0]
        if k == "e_const":
            return n[1]
        if k == "e_leaf":
            if n[1].kind == S.DV_EUCLIDEAN_DIRECTION:
                return n[1].params.copy()
            return n[1].params
        if k == "e_from_t":
            return n[1].toTransformationMatrix()[:3, 3]
        if k == "e_rot_mul":
            return n[1].toRotationMatrix() @ n[2].toEuclidean()
        if k == "e_add":
            return n[1].toEuclidean() + n[2].toEuclidean()
        if k == "e_sub":
            return n[1].toEuclidean() - n[2].toEuclidean()
        if k == "e_cross":
            return np.cross(n[1].toEuclidean(), n[2].toEuclidean())
        if k == "e_bias_spline":
            return n[1].spline().eval(n[2], n[3])
        if k == "e_mat_mul":
            return n[1].toMatrix3x3() @ n[2].toEuclidean()
        raise NotImplementedError("numeric value of %s is not needed to build the problem" % k)


class MatrixExpression(object):
    def __init__(self, node):
        self.node = node

    def __mul__(self, o):
        return EuclideanExpression(("e_mat_mul", self, o))

    def toMatrix3x3(self):
        return self.node[1].toMatrix3x3()


class HomogeneousExpression(object):
    def __init__(self, node):
        self.node = ("h_const", np.asarray(node, dtype=np.float64).ravel().copy()) if isinstance(node, (np.ndarray, list)) else node

    def toHomogeneous(self):
        n = self.node
        if n[0] == "h_const":
            return n[1]
        return n[1].toTransformationMatrix() @ n[2].toHomogeneous()


class RotationExpression(object):
    def __init__(self, node):
        self.node = node

    def inverse(self):
        return RotationExpression(("r_inv", self))

    def __mul__(self, o):
        if isinstance(o, RotationExpression):
            return RotationExpression(("r_mul", self, o))
        if not isinstance(o, EuclideanExpression):
            o = _const3(o)
        return EuclideanExpression(("e_rot_mul", self, o))

    def toRotationMatrix(self):
        n = self.node
        k = n[0]
        if k == "r_leaf":
            return _quat2r(n[1].params)
        if k == "r_inv":
            return n[1].toRotationMatrix().T
        if k == "r_mul":
            return n[1].toRotationMatrix() @ n[2].toRotationMatrix()
        if k == "r_from_t":
            return n[1].toTransformationMatrix()[:3, :3]
        if k == "r_spline":
            return _rotvec2r(n[1].spline().eval(n[2])[3:])
        raise NotImplementedError(k)


class TransformationExpression(object):
    def __init__(self, node):
        self.node = ("t_const", np.asarray(node, dtype=np.float64)) if isinstance(node, np.ndarray) else node

    def inverse(self):
        return TransformationExpression(("t_inv", self))

    def __mul__(self, o):
        if isinstance(o, TransformationExpression):
            return TransformationExpression(("t_mul", self, o))
        if isinstance(o, HomogeneousExpression):
            return HomogeneousExpression(("h_transformed", self, o))
        raise TypeError("TransformationExpression * %s" % type(o).__name__)

    def toRotationExpression(self):
        return RotationExpression(("r_from_t", self))

    def toEuclideanExpression(self):
        return EuclideanExpression(("e_from_t", self))

    def toTransformationMatrix(self):
        n = self.node
        k = n[0]
        if k == "t_const":
            return n[1]
        if k == "t_leaf":
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = _quat2r(n[1].params), n[2].params
            return T
        if k == "t_inv":
            T = n[1].toTransformationMatrix()
            Ti = np.eye(4)
            Ti[:3, :3], Ti[:3, 3] = T[:3, :3].T, -T[:3, :3].T @ T[:3, 3]
            return Ti
        if k == "t_mul":
            return n[1].toTransformationMatrix() @ n[2].toTransformationMatrix()
        if k == "t_spline_at":
            return n[1].spline().transformation(n[2].toScalar())
        raise NotImplementedError(k)


# ------------------------------------------------------------------------------------------------ M-estimators
class _MEst(object):
    kind, param = S.MEST_NONE, 0.0


class NoMEstimator(_MEst):
    pass


class HuberMEstimator(_MEst):
    kind = S.MEST_HUBER

    def __init__(self, k):
        self.param = float(k)


class CauchyMEstimator(_MEst):
    kind = S.MEST_CAUCHY

    def __init__(self, sigma2):
        self.param = float(sigma2)


class GemanMcClureMEstimator(_MEst):
    kind = S.MEST_GEMAN_MCCLURE

    def __init__(self, sigma2):
        self.param = float(sigma2)


class BlakeZissermanMEstimator(_MEst):
    kind = S.MEST_BLAKE_ZISSERMAN

    def __init__(self, df, pCut=0.999, wCut=0.1):
        self.param = float(S.blake_zisserman_epsilon(int(df), pCut, wCut))


# ------------------------------------------------------------------------------------------------ cameras / error terms
class CameraGeometry(object):
    """aslam_cv camera geometry: projection model + distortion with their parameter vectors (update() order)."""

    def __init__(self, model, dist, projection, distortion=None):
        self.model, self.dist = model, dist
        self.projection = np.asarray(projection, dtype=np.float64).copy()
        self.distortion = None if distortion is None else np.asarray(distortion, dtype=np.float64).copy()


class Keypoint(object):
    def setMeasurement(self, y):
        self.y = np.asarray(y, dtype=np.float64).ravel().copy()

    def setInverseMeasurementCovariance(self, invR):
        self.invR = np.asarray(invR, dtype=np.float64).copy()


class Frame(object):
    def __init__(self):
        self.geometry, self.keypoints = None, []

    def setGeometry(self, g):
        self.geometry = g

    def addKeypoint(self, k):
        self.keypoints.append(k)


class ErrorTerm(object):
    def __init__(self):
        self.mest = NoMEstimator()

    def setMEstimatorPolicy(self, m):
        self.mest = m


class SimpleReprojectionError(ErrorTerm):
    """aslam_cv_error_terms SimpleReprojectionError(frame, keypointIndex, point)."""

    def __init__(self, frame, keypoint_index, point):
        ErrorTerm.__init__(self)
        self.frame, self.index, self.point = frame, int(keypoint_index), point


class EuclideanError(ErrorTerm):
    """kalibr_errorterms EuclideanError(measurement, invR, predicted)."""

    def __init__(self, measurement, invR, predicted):
        ErrorTerm.__init__(self)
        self.meas = np.asarray(measurement, dtype=np.float64).ravel().copy()
        self.invR, self.predicted = np.asarray(invR, dtype=np.float64), predicted


class GyroscopeError(EuclideanError):
    """kalibr_errorterms GyroscopeError(measurement, invR, angularVelocity, bias): a EuclideanError on
    angularVelocity + bias (kalibr/src/GyroscopeError.cpp:5-17)."""

    def __init__(self, measurement, invR, angularVelocity, bias):
        EuclideanError.__init__(self, measurement, invR, angularVelocity + bias)


class ErrorTermTransformation(ErrorTerm):
    """aopt.ErrorTermTransformation(T, prior, weightRotation, weightTranslation)
    (aslam_backend_expressions/src/ErrorTermTransformation.cpp:19-33)."""

    def __init__(self, T, prior, weightRotation, weightTranslation):
        ErrorTerm.__init__(self)
        if not isinstance(prior, Transformation):
            raise RuntimeError("The prior must be an sm transformation")
        self.T, self.prior = T, prior
        self.weightRotation, self.weightTranslation = float(weightRotation), float(weightTranslation)


class ScalarError(ErrorTerm):
    """kalibr_errorterms ScalarError(measurement, invR, predicted)."""

    def __init__(self, measurement, invR, predicted):
        ErrorTerm.__init__(self)
        self.meas, self.invR, self.predicted = float(measurement), float(np.asarray(invR).ravel()[0]), predicted

    def error(self):
        return self.predicted.toScalar() - self.meas


class BSplineMotionError(ErrorTerm):
    def __init__(self, spline_dv, W, error_term_order):
        ErrorTerm.__init__(self)
        self.spline_dv, self.W, self.order = spline_dv, np.asarray(W, dtype=np.float64), int(error_term_order)


BSplineEuclideanMotionError = BSplineMotionError


def addMotionErrorTerms(problem, spline_dv, W, error_term_order):
    problem.addErrorTerm(BSplineMotionError(spline_dv, W, error_term_order))


class OptimizationProblem(object):
    """incremental_calibration OptimizationProblem: design variables live in groups (first-insertion order of the groups,
    insertion order inside a group, OptimizationProblem.cpp:123-137, 272-292), error terms in insertion order."""

    def __init__(self):
        self.dvs = []          # (DesignVariable | _SplineDv, group)
        self.terms = []

    def addDesignVariable(self, dv, group_id=HELPER_GROUP_ID):
        if any(d is dv for d, _ in self.dvs):
            raise RuntimeError("design variable already included")
        self.dvs.append((dv, group_id))

    def _add_spline(self, sdv):
        self.dvs.append((sdv, sdv.group))

    def addErrorTerm(self, e):
        self.terms.append(e)

    def removeErrorTerm(self, e):
        self.terms = [t for t in self.terms if t is not e]


# ------------------------------------------------------------------------------------------------ lowering pass
def _isotropic(invR, what):
    """The scalar s with sqrtInvR = s I, or -- for a general inverse covariance -- the matrix itself (it then goes to
    the batch through ProblemSpec.set_inv_r, i.e. mc_set_last_batch_sqrt_inv_r)."""
    invR = np.atleast_2d(np.asarray(invR, dtype=np.float64))
    d = invR[0, 0]
    if not np.allclose(invR, d * np.eye(invR.shape[0]), rtol=1e-12, atol=0.0):
        if invR.shape[0] != invR.shape[1] or not np.allclose(invR, invR.T, rtol=1e-12, atol=0.0):
            raise UnsupportedErrorTerm("%s: the inverse covariance is not a symmetric matrix" % what)
        return invR.copy()
    return float(np.sqrt(d))


def _weights(entries, dim):
    """(scalar array for the batch call, None or the [n, dim, dim] inverse covariances) from a run's per-term entries"""
    if all(isinstance(e, float) for e in entries):
        return np.array(entries), None
    invR = np.stack([e * e * np.eye(dim) if isinstance(e, float) else e for e in entries])
    return np.ones(len(entries)), invR


def _match_t_leaf(T):
    return (T.node[1], T.node[2]) if T.node[0] == "t_leaf" else None


def _match_spline_at(T):
    if T.node[0] != "t_spline_at":
        return None
    _, sdv, time, lo, hi = T.node
    aff = time.affine()
    if aff is None or lo != hi:
        return None
    return sdv, aff[0], aff[1], lo


def _match_reprojection(term):
    """p = (T_c_b * inverse(T_w_b(t + d)) * inverse(T_p_w)) * const   (sensors_and_targets.py:571-611)"""
    p = term.point.node
    if p[0] != "h_transformed" or p[2].node[0] != "h_const":
        return None
    T = p[1].node
    if T[0] != "t_mul" or T[1].node[0] != "t_mul":
        return None
    Tcb = _match_t_leaf(T[1].node[1])
    inv_wb, inv_pw = T[1].node[2].node, T[2].node
    if Tcb is None or inv_wb[0] != "t_inv" or inv_pw[0] != "t_inv":
        return None
    sp = _match_spline_at(inv_wb[1])
    Tpw = _match_t_leaf(inv_pw[1])
    if sp is None or Tpw is None:
        return None
    pt = p[2].node[1]
    if pt.size != 4 or pt[3] != 1.0:
        return None
    kp = term.frame.keypoints[term.index]
    return dict(geometry=term.frame.geometry, spline=sp[0], toff=sp[1], time=sp[2], pad=sp[3], q_cb=Tcb[0], t_cb=Tcb[1],
                q_pw=Tpw[0], t_pw=Tpw[1], p=pt[:3], y=kp.y, sinv=_isotropic(kp.invR, "reprojection error"))


def _match_lidar(term):
    """-(n . t(T_p_l)) / (n . C(T_p_l) dir),  T_p_l = T_p_w * (T_w_b(t + d) * inverse(T_l_b)),  n = e_z
    (sensors_and_targets.py:274-305)"""
    n = term.predicted.node
    if n[0] != "s_mul" or n[2].node != ("s_const", -1.0) or n[1].node[0] != "s_div":
        return None
    d, th = n[1].node[1].node, n[1].node[2].node
    if d[0] != "s_dot" or th[0] != "s_dot":
        return None
    ez = np.array([0.0, 0.0, 1.0])
    for v in (d[1], th[1]):
        if v.node[0] != "e_const" or not np.array_equal(v.node[1], ez):
            return None
    if d[2].node[0] != "e_from_t" or th[2].node[0] != "e_rot_mul":
        return None
    Tpl = d[2].node[1]
    rot, direction = th[2].node[1].node, th[2].node[2].node
    if rot[0] != "r_from_t" or rot[1] is not Tpl or direction[0] != "e_const":
        return None
    T = Tpl.node
    if T[0] != "t_mul" or T[2].node[0] != "t_mul":
        return None
    Tpw = _match_t_leaf(T[1])
    sp = _match_spline_at(T[2].node[1])
    inv_lb = T[2].node[2].node
    if Tpw is None or sp is None or inv_lb[0] != "t_inv":
        return None
    Tlb = _match_t_leaf(inv_lb[1])
    if Tlb is None:
        return None
    return dict(spline=sp[0], toff=sp[1], time=sp[2], pad=sp[3], q_lb=Tlb[0], t_lb=Tlb[1], q_pw=Tpw[0], t_pw=Tpw[1],
                dir=direction[1], range=term.meas, sinv=float(np.sqrt(term.invR)))


def _match_bias(e):
    n = e.node
    if n[0] == "e_bias_spline" and n[3] == 0:
        return n[1], None, n[2]
    if n[0] == "e_leaf" and n[1].kind == S.DV_EUCLIDEAN_POINT:
        return None, n[1], None
    return None


def _spline_e(e, what):
    n = e.node
    return (n[1], n[2]) if n[0] == "e_spline" and n[3] == what else None


def _match_gyro(term):
    """C_i_b * w_b(t) + b_g(t)   (sensors_and_targets.py:1198-1212)"""
    n = term.predicted.node
    if n[0] != "e_add":
        return None
    rm, bias = n[1].node, _match_bias(n[2])
    if bias is None or rm[0] != "e_rot_mul" or rm[1].node[0] != "r_leaf":
        return None
    w = _spline_e(rm[2], "ang_vel_b")
    if w is None or (bias[2] is not None and bias[2] != w[1]):
        return None
    return dict(spline=w[0], t=w[1], q_ib=rm[1].node[1], bias_spline=bias[0], bias_dv=bias[1], meas=term.meas,
                sinv=_isotropic(term.invR, "gyroscope error"))


def _match_gyro_prior(term):
    """R_i_c * const + bias   (findExtrinsicPriorSensorsToCamera, sensors_and_targets.py:871-879)"""
    n = term.predicted.node
    if n[0] != "e_add":
        return None
    rm, b = n[1].node, n[2].node
    if rm[0] != "e_rot_mul" or rm[1].node[0] != "r_leaf" or rm[2].node[0] != "e_const":
        return None
    if b[0] != "e_leaf" or b[1].kind != S.DV_EUCLIDEAN_POINT:
        return None
    return dict(q=rm[1].node[1], bias=b[1], w_pred=rm[2].node[1], meas=term.meas, sinv=_isotropic(term.invR, "gyroscope error"))


def _match_transformation(term):
    """T_a or T_a * T_b.inverse() over TransformationDv leaves (sensors_and_targets.py:812-817, 886-919)"""
    a = _match_t_leaf(term.T)
    if a is not None:
        return dict(a=a, b=(None, None))
    n = term.T.node
    if n[0] == "t_mul" and n[2].node[0] == "t_inv":
        a, b = _match_t_leaf(n[1]), _match_t_leaf(n[2].node[1])
        if a is not None and b is not None:
            return dict(a=a, b=b)
    return None


def _match_gravity_part(e):
    """C_b_w(t) * (a_w(t) - g_w) -> (spline dv, t, gravity dv)"""
    n = e.node
    if n[0] != "e_rot_mul" or n[1].node[0] != "r_inv" or n[1].node[1].node[0] != "r_spline":
        return None
    sdv, t = n[1].node[1].node[1], n[1].node[1].node[2]
    diff = n[2].node
    if diff[0] != "e_sub" or _spline_e(diff[1], "lin_acc") != (sdv, t) or diff[2].node[0] != "e_leaf":
        return None
    return sdv, t, diff[2].node[1]


def _match_lever(lever, centripetal, sdv, t, arm):
    """w_dot_b x r  and  w_b x (w_b x r) over the same r; `arm(r_expression)` identifies r. Returns arm's result."""
    lever, centripetal = lever.node, centripetal.node
    if lever[0] != "e_cross" or _spline_e(lever[1], "ang_acc_b") != (sdv, t):
        return None
    if centripetal[0] != "e_cross" or _spline_e(centripetal[1], "ang_vel_b") != (sdv, t):
        return None
    c2 = centripetal[2].node
    if c2[0] != "e_cross" or _spline_e(c2[1], "ang_vel_b") != (sdv, t):
        return None
    r1, r2 = arm(lever[2]), arm(c2[2])
    if r1 is None or r2 is None or any(a is not b for a, b in zip(r1, r2)):
        return None
    return r1


def _leaf_arm(e):
    return (e.node[1],) if e.node[0] == "e_leaf" else None


def _match_specific_force(e):
    """C_b_w (a_w - g_w) + w_dot_b x r_b + w_b x (w_b x r_b) -> (spline dv, t, g, r_b)"""
    inner = e.node
    if inner[0] != "e_add" or inner[1].node[0] != "e_add":
        return None
    gp = _match_gravity_part(inner[1].node[1])
    if gp is None:
        return None
    r = _match_lever(inner[1].node[2], inner[2], gp[0], gp[1], _leaf_arm)
    if r is None:
        return None
    return gp[0], gp[1], gp[2], r[0]


def _match_accel(term):
    """C_i_b * (C_b_w (a_w - g_w) + w_dot_b x r_b + w_b x (w_b x r_b)) + b_a(t)   (sensors_and_targets.py:1152-1170)"""
    n = term.predicted.node
    if n[0] != "e_add":
        return None
    rm, bias = n[1].node, _match_bias(n[2])
    if bias is None or rm[0] != "e_rot_mul" or rm[1].node[0] != "r_leaf":
        return None
    sf = _match_specific_force(rm[2])
    if sf is None or (bias[2] is not None and bias[2] != sf[1]):
        return None
    return dict(spline=sf[0], t=sf[1], q_ib=rm[1].node[1], r_b=sf[3], g=sf[2], bias_spline=bias[0], bias_dv=bias[1], meas=term.meas,
                sinv=_isotropic(term.invR, "accelerometer error"))


def _mat_leaf(m):
    return m.node[1] if isinstance(m, MatrixExpression) and m.node[0] == "m_leaf" else None


def _match_accel_sm(term):
    """ScaledMisalignedImu:            M * (C_i_b * (specific force at r_b)) + b_a           (sensors_and_targets.py:1482-1485)
    ScaledMisalignedSizeEffectImu:  M * (C_i_b * (C_b_w (a_w - g_w)) + Ix * (C_i_b * lever(rx_b)) + Iy * (...) + Iz * (...)) + b_a,
                                    rx_b = r_b + C_i_b.inverse() * rx_i                      (:1637-1649)"""
    n = term.predicted.node
    if n[0] != "e_add":
        return None
    mm, bias = n[1].node, _match_bias(n[2])
    if bias is None or mm[0] != "e_mat_mul" or _mat_leaf(mm[1]) is None:
        return None
    M, body = _mat_leaf(mm[1]), mm[2].node
    out = dict(M=M, bias_spline=bias[0], bias_dv=bias[1], meas=term.meas, sinv=_isotropic(term.invR, "accelerometer error"))
    if body[0] == "e_rot_mul" and body[1].node[0] == "r_leaf":
        sf = _match_specific_force(body[2])
        if sf is None or (bias[2] is not None and bias[2] != sf[1]):
            return None
        out.update(spline=sf[0], t=sf[1], g=sf[2], r_b=sf[3], q_ib=body[1].node[1], size_effect=None)
        return out
    # size effect: ((grav + X) + Y) + Z
    try:
        z, y, x, grav = body[2].node, body[1].node[2].node, body[1].node[1].node[2].node, body[1].node[1].node[1].node
    except (IndexError, AttributeError, TypeError):
        return None
    if body[0] != "e_add" or grav[0] != "e_rot_mul" or grav[1].node[0] != "r_leaf":
        return None
    q_ib = grav[1].node[1]
    gp = _match_gravity_part(grav[2])
    if gp is None or (bias[2] is not None and bias[2] != gp[1]):
        return None
    sdv, t, g = gp

    def arm(e):      # r_b + C_i_b.inverse() * rx_i
        a = e.node
        if a[0] != "e_add" or a[1].node[0] != "e_leaf" or a[2].node[0] != "e_rot_mul":
            return None
        rot, leaf = a[2].node[1].node, a[2].node[2].node
        if rot[0] != "r_inv" or rot[1].node[0] != "r_leaf" or rot[1].node[1] is not q_ib or leaf[0] != "e_leaf":
            return None
        return a[1].node[1], leaf[1]

    arms, sels, r_b = [], [], None
    for part in (x, y, z):
        if part[0] != "e_mat_mul" or _mat_leaf(part[1]) is None:
            return None
        rot = part[2].node
        if rot[0] != "e_rot_mul" or rot[1].node[0] != "r_leaf" or rot[1].node[1] is not q_ib or rot[2].node[0] != "e_add":
            return None
        r = _match_lever(rot[2].node[1], rot[2].node[2], sdv, t, arm)
        if r is None or (r_b is not None and r[0] is not r_b):
            return None
        r_b = r[0]
        arms.append(r[1]); sels.append(_mat_leaf(part[1]))
    out.update(spline=sdv, t=t, g=g, r_b=r_b, q_ib=q_ib, size_effect=arms + sels)
    return out


def _match_gyro_sm(term):
    """M * (C_gyro_b * w_b) + Ma * (C_gyro_b * a_b) + b_g,  C_gyro_b = C_gyro_i * C_i_b   (sensors_and_targets.py:1519-1538)"""
    n = term.predicted.node
    if n[0] != "e_add" or n[1].node[0] != "e_add":
        return None
    bias = _match_bias(n[2])
    p1, p2 = n[1].node[1].node, n[1].node[2].node
    if bias is None or p1[0] != "e_mat_mul" or p2[0] != "e_mat_mul" or _mat_leaf(p1[1]) is None or _mat_leaf(p2[1]) is None:
        return None

    def rot_chain(e):
        r = e.node
        if r[0] != "e_rot_mul" or r[1].node[0] != "r_mul":
            return None
        a, b = r[1].node[1].node, r[1].node[2].node
        return (a[1], b[1], r[2]) if a[0] == "r_leaf" and b[0] == "r_leaf" else None

    c1, c2 = rot_chain(p1[2]), rot_chain(p2[2])
    if c1 is None or c2 is None or c1[0] is not c2[0] or c1[1] is not c2[1]:
        return None
    sf = _match_specific_force(c2[2])
    if sf is None or _spline_e(c1[2], "ang_vel_b") != (sf[0], sf[1]) or (bias[2] is not None and bias[2] != sf[1]):
        return None
    return dict(spline=sf[0], t=sf[1], g=sf[2], r_b=sf[3], q_gi=c1[0], q_ib=c1[1], M=_mat_leaf(p1[1]), Ma=_mat_leaf(p2[1]),
                bias_spline=bias[0], bias_dv=bias[1], meas=term.meas, sinv=_isotropic(term.invR, "gyroscope error"))


def lower(problem):
    """OptimizationProblem -> (ProblemSpec, handles).  `handles[id(obj)]` is the spec handle of a design variable,
    spline design variable or camera geometry."""
    sp = S.ProblemSpec()
    handles = {}
    for dv, group in problem.dvs:
        if isinstance(dv, _SplineDv):
            s = dv.spline()
            handles[id(dv)] = sp.add_spline(s.order, s.knots, s.coeffs, dv.active, group)
        elif isinstance(dv, MatrixBasicDv):
            handles[id(dv)] = sp.add_matrix_dv(dv.params, dv.pattern, dv.isActive(), dv.scaling, group)
        else:
            handles[id(dv)] = sp.add_dv(dv.kind, dv.params, dv.isActive(), dv.scaling, group)

    def h(obj):
        if obj is None:
            return -1
        if id(obj) not in handles:
            # a design variable the front-end never added to the problem: constant (inactive)
            if isinstance(obj, _SplineDv):
                s = obj.spline()
                handles[id(obj)] = sp.add_spline(s.order, s.knots, s.coeffs, False, HELPER_GROUP_ID)
            elif isinstance(obj, MatrixBasicDv):
                handles[id(obj)] = sp.add_matrix_dv(obj.params, obj.pattern, False, obj.scaling, HELPER_GROUP_ID)
            else:
                handles[id(obj)] = sp.add_dv(obj.kind, obj.params, False, obj.scaling, HELPER_GROUP_ID)
        return handles[id(obj)]

    def camera(geom):
        if id(geom) not in handles:
            hp = sp.add_dv(S.DV_PROJECTION, geom.projection, False, 1.0, CALIBRATION_GROUP_ID)
            hk = sp.add_dv(S.DV_DISTORTION, geom.distortion, False, 1.0, CALIBRATION_GROUP_ID) if geom.distortion is not None else -1
            handles[id(geom)] = sp.add_camera(geom.model, geom.dist, hp, hk)
        return handles[id(geom)]

    run_key, run = None, None

    def flush():
        nonlocal run_key, run
        if run is None:
            return
        kind = run_key[0]
        invR = None
        if "sinv" in run:
            run["sinv"], invR = _weights(run["sinv"], 2 if kind == "reproj" else 1 if kind == "lidar" else 3)
        if kind == "reproj":
            first = np.concatenate([[0], np.cumsum(run["counts"])]).astype(np.int32)
            op = sp.add_reprojection_batch(run["camera"], run["spline"], run["q_cb"], run["t_cb"], run["toff"], run["obs_time"],
                                      run["obs_q"], run["obs_t"], first, np.array(run["y"]), np.array(run["p"]),
                                      np.array(run["sinv"]), run["pad"], run["mest"].kind, run["mest"].param)
        elif kind == "lidar":
            op = sp.add_lidar_batch(run["spline"], run["toff"], run["q_lb"], run["t_lb"], np.array(run["t"]), np.array(run["dir"]),
                               np.array(run["range"]), np.array(run["sinv"]), run["q_pw"], run["t_pw"], run["pad"],
                               run["mest"].kind, run["mest"].param)
        elif kind == "accel":
            op = sp.add_accelerometer_batch(run["spline"], run["bias_spline"], run["bias_dv"], run["g"], run["q_ib"], run["r_b"],
                                       np.array(run["t"]), np.array(run["meas"]), np.array(run["sinv"]), run["mest"].kind,
                                       run["mest"].param)
        elif kind == "gyro":
            op = sp.add_gyroscope_batch(run["spline"], run["bias_spline"], run["bias_dv"], run["q_ib"], np.array(run["t"]),
                                   np.array(run["meas"]), np.array(run["sinv"]), run["mest"].kind, run["mest"].param)
        elif kind == "accel_sm":
            op = sp.add_accelerometer_batch_sm(run["spline"], run["bias_spline"], run["bias_dv"], run["g"], run["q_ib"], run["r_b"],
                                          run["M"], run["size_effect"], np.array(run["t"]), np.array(run["meas"]),
                                          np.array(run["sinv"]), run["mest"].kind, run["mest"].param)
        elif kind == "gyro_prior":
            op = sp.add_gyroscope_prior_batch(run["q"], run["bias"], np.array(run["w_pred"]), np.array(run["meas"]), np.array(run["sinv"]),
                                         run["mest"].kind, run["mest"].param)
        elif kind == "transformation":
            sp.add_transformation_batch(run["qa"], run["ta"], run["qb"], run["tb"], np.array(run["prior_q"]), np.array(run["prior_t"]),
                                        run["wr"], run["wt"], run["mest"].kind, run["mest"].param)
        elif kind == "gyro_sm":
            op = sp.add_gyroscope_batch_sm(run["spline"], run["bias_spline"], run["bias_dv"], run["q_ib"], run["q_gi"], run["M"],
                                      run["Ma"], run["g"], run["r_b"], np.array(run["t"]), np.array(run["meas"]),
                                      np.array(run["sinv"]), run["mest"].kind, run["mest"].param)
        if invR is not None:
            sp.set_inv_r(op, invR)
        run_key, run = None, None

    def start(key, **fields):
        nonlocal run_key, run
        if key != run_key:
            flush()
            run_key, run = key, dict(fields)
            return True
        return False

    for term in problem.terms:
        mk = (term.mest.kind, term.mest.param)
        if isinstance(term, BSplineMotionError):
            flush()
            sp.add_motion_term(h(term.spline_dv), term.W, term.order)
            continue
        if isinstance(term, SimpleReprojectionError):
            m = _match_reprojection(term)
            if m is None:
                raise UnsupportedErrorTerm("reprojection error over an expression that is not T_c_b * T_w_b(t+d)^-1 * T_p_w^-1 * p")
            key = ("reproj", id(m["geometry"]), id(m["spline"]), id(m["q_cb"]), id(m["t_cb"]), id(m["toff"]), m["pad"], mk)
            if start(key, camera=camera(m["geometry"]), spline=h(m["spline"]), q_cb=h(m["q_cb"]), t_cb=h(m["t_cb"]),
                     toff=h(m["toff"]), pad=m["pad"], mest=term.mest, obs_time=[], obs_q=[], obs_t=[], counts=[], y=[], p=[],
                     sinv=[], last=None):
                pass
            obs_id = (m["time"], id(m["q_pw"]), id(m["t_pw"]))
            if run["last"] != obs_id:        # a new image observation: same time, same target
                run["obs_time"].append(m["time"]); run["obs_q"].append(h(m["q_pw"])); run["obs_t"].append(h(m["t_pw"]))
                run["counts"].append(0); run["last"] = obs_id
            run["counts"][-1] += 1
            run["y"].append(m["y"]); run["p"].append(m["p"]); run["sinv"].append(m["sinv"])
        elif isinstance(term, ScalarError):
            m = _match_lidar(term)
            if m is None:
                raise UnsupportedErrorTerm("scalar error that is not Multical's point-to-plane form")
            key = ("lidar", id(m["spline"]), id(m["toff"]), id(m["q_lb"]), id(m["t_lb"]), id(m["q_pw"]), id(m["t_pw"]), m["pad"], mk)
            start(key, spline=h(m["spline"]), toff=h(m["toff"]), q_lb=h(m["q_lb"]), t_lb=h(m["t_lb"]), pad=m["pad"],
                  mest=term.mest, t=[], dir=[], range=[], sinv=[], q_pw=[], t_pw=[])
            run["t"].append(m["time"]); run["dir"].append(m["dir"]); run["range"].append(m["range"]); run["sinv"].append(m["sinv"])
            run["q_pw"].append(h(m["q_pw"])); run["t_pw"].append(h(m["t_pw"]))
        elif isinstance(term, ErrorTermTransformation):
            m = _match_transformation(term)
            if m is None:
                raise UnsupportedErrorTerm("transformation error over an expression that is not T_a or T_a * T_b^-1 of TransformationDvs")
            key = ("transformation", id(m["a"][0]), id(m["a"][1]), id(m["b"][0]), id(m["b"][1]), term.weightRotation,
                   term.weightTranslation, mk)
            start(key, qa=h(m["a"][0]), ta=h(m["a"][1]), qb=h(m["b"][0]), tb=h(m["b"][1]), wr=term.weightRotation,
                  wt=term.weightTranslation, mest=term.mest, prior_q=[], prior_t=[])
            run["prior_q"].append(term.prior.q()); run["prior_t"].append(term.prior.t())
        elif isinstance(term, EuclideanError) and _match_gyro_prior(term) is not None:
            m = _match_gyro_prior(term)
            key = ("gyro_prior", id(m["q"]), id(m["bias"]), mk)
            start(key, q=h(m["q"]), bias=h(m["bias"]), mest=term.mest, w_pred=[], meas=[], sinv=[])
            run["w_pred"].append(m["w_pred"]); run["meas"].append(m["meas"]); run["sinv"].append(m["sinv"])
        elif isinstance(term, EuclideanError):
            m = None
            for kind, matcher in (("accel", _match_accel), ("gyro", _match_gyro), ("accel_sm", _match_accel_sm),
                                  ("gyro_sm", _match_gyro_sm)):
                m = matcher(term)
                if m is not None:
                    break
            if m is None:
                raise UnsupportedErrorTerm("euclidean error that is neither an accelerometer nor a gyroscope form of the three IMU models")
            se = m.get("size_effect")
            key = (kind, id(m["spline"]), id(m["q_ib"]), id(m["bias_spline"]), id(m["bias_dv"]), id(m.get("g")), id(m.get("r_b")),
                   id(m.get("M")), id(m.get("Ma")), id(m.get("q_gi")), tuple(id(o) for o in se) if se else None, mk)
            start(key, spline=h(m["spline"]), bias_spline=h(m["bias_spline"]), bias_dv=h(m["bias_dv"]), q_ib=h(m["q_ib"]),
                  g=h(m.get("g")), r_b=h(m.get("r_b")), M=h(m.get("M")), Ma=h(m.get("Ma")), q_gi=h(m.get("q_gi")),
                  size_effect=[h(o) for o in se] if se else None, mest=term.mest, t=[], meas=[], sinv=[])
            run["t"].append(m["t"]); run["meas"].append(m["meas"]); run["sinv"].append(m["sinv"])
        else:
            raise UnsupportedErrorTerm("error term type %s is not lowered" % type(term).__name__)
    flush()
    return sp, handles
