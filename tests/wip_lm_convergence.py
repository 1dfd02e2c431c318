"""The BASELINE.json configurations the first round never ran on the GPU, each to the parity bar or -- at full size --
through the oracle-free properties of tests/gpu_props.py:

  configs[1]  camera-IMU, 1 camera + 1 IMU, no LiDAR (reference-IMU extrinsics inactive): a 3 s cut against the oracle at
              1e-9, the full 60 s through gradient / curvature / PSD
  configs[2]  multi-camera static calibration at full 4 cameras x 6 boards x 500 views: gradient / curvature / PSD /
              sqrtInvR scaling law
  configs[4]  one rank (3 of 8) of the 600 s stress problem: dimensions, private entries have one writer, gradient of the
              shard's own cost
  B8          the dim-6 order-2 BSplineMotionError pose prior (calibrator.py:125-130) in a parity problem
"""
import numpy as np
import pytest

from gpu_props import check_gradient_and_curvature, check_psd, scaled_sinv_spec

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _hip(spec, rank=None, world=None):
    from multical_amd.backend import HipApi
    g = HipApi(0)
    if world:
        g.set_shard(rank, world)
    ids = spec.emit(g)
    g.finalize()
    return g, ids


def _oracle(spec):
    from oracle_api import OracleApi
    o = OracleApi()
    ids = spec.emit(o)
    o.finalize()
    return o, ids




@pytest.mark.timeout(900)
def test_levenberg_marquardt_to_convergence_matches_oracle():
    """Optimizer2 + LM to CONVERGENCE (convergenceDeltaX 1e-5, convergenceJDescentRatioThreshold 1e-6 as calibrator.py:27-58
    sets them, lambda0 = 10) on a 10 s problem: mc_optimize on the GPU against the same loop over the oracle
    (tests/oracle_lm.py).  Same accepted / rejected sequence, final estimates within 1e-7."""
    from multical_amd import synth
    from multical_amd.backend import HipApi
    from oracle_lm import optimize_lm
    cfg = synth.SynthConfig(duration=10.0, n_cams=2, n_imus=2, n_lidars=1, lidar_points_per_frame=40, corner_stride=4,
                            cam_models=[(0, 1), (1, 1)], huber_accel=1.0, huber_gyro=1.0, blake_zisserman_cam=2,
                            motion_speed=3.0)
    spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
    g = HipApi(0)
    ids = spec.emit(g)
    o, oids = _oracle(spec)
    assert o.j_cols > 6000
    r = g.optimize(maxIterations=30, convergenceDeltaX=1e-5, convergenceJDescentRatioThreshold=1e-6,
                   levenbergMarquardtLambdaInit=10.0)
    ro = optimize_lm(o, 30, 10.0, 1e-5, 1e-6)
    assert r["iterations"] < 30, "did not converge"
    assert (r["iterations"], r["failedIterations"]) == (ro["iterations"], ro["failedIterations"]), (r, ro)
    assert abs(r["JFinal"] - ro["JFinal"]) <= 1e-9 * ro["JFinal"]
    assert r["JFinal"] < 0.05 * r["JStart"]
    worst = 0.0
    for h, v in ids.items():
        if isinstance(v, int) and spec.ops[h][0] == "dv":
            worst = max(worst, np.abs(g.dv_params(v) - o.dv_params(oids[h])).max())
    for key in ("pose",):
        first, ofirst = g.spline_first_dv(ids[H[key]][1]), o.spline_first_dv(oids[H[key]][1])
        for i in range(0, W["pose_coeffs"].shape[1], 5):
            worst = max(worst, np.abs(g.dv_params(first + i) - o.dv_params(ofirst + i)).max())
    assert worst <= 1e-7, worst
