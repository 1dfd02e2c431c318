"""The BASELINE.json configurations the first round never ran on the GPU, each to the parity bar or -- at full size --
through the oracle-free properties of tests/gpu_props.py:

  configs[1]  camera-IMU, 1 camera + 1 IMU, no LiDAR (reference-IMU extrinsics inactive): a 3 s cut against the oracle at
              1e-9, the full 60 s through gradient / curvature / PSD
  configs[2]  multi-camera static calibration at full 4 cameras x 6 boards x 500 views: gradient / curvature / PSD /
              sqrtInvR scaling law
  configs[4]  one rank (3 of 8) of the 600 s stress problem: dimensions, private entries have one writer, gradient of the
              shard's own cost
  LM          Optimizer2 + Levenberg-Marquardt run to convergence on a 10 s problem against the oracle-side driver
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


def _parity(spec, use_mest=True):
    g, _ = _hip(spec)
    o, _ = _oracle(spec)
    assert (g.j_rows, g.j_cols, g.n_blocks, g.n_terms) == (o.j_rows, o.j_cols, o.n_blocks, o.n_terms)
    Jo, eo = o.evaluate_error(use_mest)
    Jg, eg = g.evaluate_error(use_mest)
    assert abs(Jg - Jo) <= RTOL * abs(Jo)
    assert np.abs(eg - eo).max() <= RTOL * np.abs(eo).max()
    co, cg = o.term_chi2(), g.term_chi2()
    assert np.abs(co - cg).max() <= RTOL * np.abs(co).max()
    Cp, Ci, Cx, rhs = o.build_system(use_mest)
    Cpg, Cig = g.pattern()
    assert np.array_equal(Cp, Cpg) and np.array_equal(Ci, Cig)
    Cxg, rhsg = g.build_system(use_mest)
    assert np.abs(Cxg - Cx).max() <= RTOL * np.abs(Cx).max()
    assert np.linalg.norm(Cxg - Cx) <= RTOL * np.linalg.norm(Cx)
    assert np.abs(rhsg - rhs).max() <= RTOL * np.abs(rhs).max()
    assert np.all(Cxg[Cx == 0.0] == 0.0)
    return g, o


@pytest.mark.parametrize("use_mest", [True, False])
def test_config1_cut_matches_oracle(use_mest):
    """configs[1] cut to 3 s, every corner kept: the activity pattern of the camera-IMU problem (one IMU = the reference
    IMU, its q_i_b / r_b inactive; no LiDAR blocks; no board besides the fixed one active ...)."""
    from multical_amd import synth
    from multical_amd.backend import HipApi
    cfg = synth.baseline_config(1, duration=3.0, huber_accel=1.0, blake_zisserman_cam=2)
    spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
    g, o = _parity(spec, use_mest)
    assert g.n_terms > 5000


def test_config1_full_size_properties():
    from multical_amd import synth
    from multical_amd.backend import HipApi
    cfg = synth.baseline_config(1, bias_motion_terms=False)
    W = synth.make_world(cfg)
    meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
    spec, H = synth.build_spec(W, synth.initial_state(W), meas, with_mest=False)
    g, _ = _hip(spec)
    assert g.n_terms > 100_000 and g.j_cols > 36_000
    Cx, rhs = check_gradient_and_curvature(g)
    check_psd(g, Cx)
    g2, _ = _hip(scaled_sinv_spec(spec, 2.0))
    a2, r2 = g2.build_system(False)
    assert np.abs(a2 - 4.0 * Cx).max() <= 1e-12 * np.abs(a2).max()
    assert np.abs(r2 - 4.0 * rhs).max() <= 1e-12 * np.abs(r2).max()


def test_config2_full_size_properties():
    """configs[2] at 4 cameras x 6 boards x 500 views (the first round covered its shape at 3 x 2 x 8 only)."""
    from multical_amd import synth_static as SS
    from multical_amd.backend import HipApi
    cfg = SS.baseline_static_config(2)
    W = SS.make_static_world(cfg)
    meas = SS.synthesize_static(W, lambda: HipApi(0), noise=True)
    spec, H = SS.build_static_spec(W, SS.static_initial(W), meas, with_mest=False)
    g, _ = _hip(spec)
    assert g.n_terms > 500_000 and g.n_blocks > 2 * 500
    # parameters live on very different scales (focal lengths ~ 460, distortion ~ 0.1, poses ~ 1): probe along a
    # direction scaled to the state, so every block contributes to the differences
    Cx, rhs = check_gradient_and_curvature(g, h=1e-6, rtol=2e-6)
    check_psd(g, Cx)
    g2, _ = _hip(scaled_sinv_spec(spec, 2.0))
    a2, r2 = g2.build_system(False)
    assert np.abs(a2 - 4.0 * Cx).max() <= 1e-12 * np.abs(a2).max()
    assert np.abs(r2 - 4.0 * rhs).max() <= 1e-12 * np.abs(r2).max()
    # with the Blake-Zisserman weights of the real problem: two builds agree, the cost equals sum w chi2
    spec_m, _ = SS.build_static_spec(W, SS.static_initial(W), meas, with_mest=True)
    gm, _ = _hip(spec_m)
    J, _ = gm.evaluate_error(True, want_e=False)
    chi2 = gm.term_chi2()
    eps = spec_m.batch_ops()[0][2]["mest_param"]
    w = np.exp(-chi2) / (np.exp(-chi2) + eps)
    assert abs(J - (w * chi2).sum()) <= 1e-10 * J
    a, b = gm.build_system(True)
    a1, b1 = gm.build_system(True)
    assert np.abs(a - a1).max() <= 1e-12 * np.abs(a).max() and np.abs(b - b1).max() <= 1e-12 * np.abs(b).max()


def test_pose_motion_prior_matches_oracle():
    """Row B8 at dim 6 / order 2: asp.addMotionErrorTerms on the pose spline (calibrator.py:125-130,
    BSplineMotionError.hpp:93-166) together with the dim-3 order-1 bias priors."""
    from multical_amd import synth
    from multical_amd.backend import HipApi
    cfg = synth.SynthConfig(duration=2.0, n_cams=2, n_imus=2, n_lidars=1, lidar_points_per_frame=20, corner_stride=9,
                            pose_motion_terms=True, huber_accel=1.0)
    spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
    kinds = [k for _, k, _ in spec.batch_ops()]
    assert kinds.count("motion") == 2 * 2 + 1
    g, o = _parity(spec, True)
    # the prior alone: every other batch dropped, so its band is not hidden under the measurement terms
    import copy
    s2 = copy.copy(spec)
    s2.ops = [(k, d) for k, d in spec.ops if k in ("dv", "spline", "camera", "motion")]
    _parity(s2, True)


@pytest.mark.timeout(900)
def test_config4_one_rank_of_eight():
    """configs[4] (16 cameras + 8 IMUs + 4 LiDARs, 600 s, ~10^7 terms over 8 GPUs): rank 3 of 8 on this GPU."""
    import torch
    from multical_amd import synth
    from multical_amd.backend import HipApi
    cfg = synth.baseline_config(4, bias_motion_terms=False)
    W = synth.make_world(cfg)
    meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
    spec, H = synth.build_spec(W, synth.initial_state(W), meas, with_mest=False)
    world, rank = 8, 3
    g, _ = _hip(spec, rank, world)
    assert g.n_terms > 9_000_000 and g.j_cols > 380_000
    sidx = g.shared_indices()
    assert 0 < sidx.size < 0.05 * g.nnz
    # this rank's partial system: gradient / curvature of ITS terms' cost (evaluateError sums this rank's terms only)
    Cx, rhs = check_gradient_and_curvature(g, trials=1)
    # it writes only its own window: non-zeros outside the shared set lie in one contiguous band of columns
    private = np.ones(g.nnz, bool)
    private[sidx] = False
    Cp, _ = g.pattern()
    cols = np.searchsorted(Cp, np.nonzero((Cx != 0.0) & private)[0], side="right") - 1
    pose_cols = cols[cols < 6 * W["pose_coeffs"].shape[1]]
    span = (pose_cols.max() - pose_cols.min()) / (6.0 * W["pose_coeffs"].shape[1])
    assert span < 1.3 / world, span
    # a neighbouring rank writes a disjoint private set
    g2, _ = _hip(spec, rank + 1, world)
    Cx2, _ = g2.build_system(False)
    assert not np.any((Cx != 0.0) & (Cx2 != 0.0) & private)
    del g2
    torch.cuda.empty_cache()


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
                            motion_speed=8.0)
    spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
    g = HipApi(0)
    ids = spec.emit(g)
    o, oids = _oracle(spec)
    assert o.j_cols > 6000
    r = g.optimize(maxIterations=40, convergenceDeltaX=1e-5, convergenceJDescentRatioThreshold=1e-6,
                   levenbergMarquardtLambdaInit=10.0)
    ro = optimize_lm(o, 40, 10.0, 1e-5, 1e-6)
    assert r["iterations"] < 40, "did not converge"
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


@pytest.mark.timeout(1200)
def test_levenberg_marquardt_along_a_weakly_observable_direction_matches_oracle():
    """The trajectory the convergence test above avoids: with motion_speed = 3.0 the same problem has a weakly observable
    direction -- from iteration ~10 the cost creeps by ~1 per step while lambda falls to its floor -- and does not reach the
    convergence thresholds in 40 iterations.  That creeping stretch is where GPU and oracle iterates would part first, so
    the run is compared for a FIXED 40 iterations with no convergence assertion: same accepted / rejected sequence, JFinal
    within 1e-9, estimates within 1e-7."""
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
    iters = 40
    r = g.optimize(maxIterations=iters, convergenceDeltaX=1e-5, convergenceJDescentRatioThreshold=1e-6,
                   levenbergMarquardtLambdaInit=10.0)
    ro = optimize_lm(o, iters, 10.0, 1e-5, 1e-6)
    assert (r["iterations"], r["failedIterations"]) == (ro["iterations"], ro["failedIterations"]), (r, ro)
    assert abs(r["JFinal"] - ro["JFinal"]) <= 1e-9 * ro["JFinal"], (r["JFinal"], ro["JFinal"])
    worst = 0.0
    for h, v in ids.items():
        if isinstance(v, int) and spec.ops[h][0] == "dv":
            worst = max(worst, np.abs(g.dv_params(v) - o.dv_params(oids[h])).max())
    first, ofirst = g.spline_first_dv(ids[H["pose"]][1]), o.spline_first_dv(oids[H["pose"]][1])
    for i in range(0, W["pose_coeffs"].shape[1], 5):
        worst = max(worst, np.abs(g.dv_params(first + i) - o.dv_params(ofirst + i)).max())
    assert worst <= 1e-7, worst
