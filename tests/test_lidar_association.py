"""LiDAR.findPointsOnTarget (kalibr_sensor_calibration/sensors_and_targets.py:199-235) on the GPU against the oracle's
per-reading restatement: same kept readings, same per-target inlier index lists, same world-frame points.  The readings
are the small problem's LiDAR points (which lie on the boards up to noise and the perturbed initial guess) mixed with
clutter spread through the room and with readings whose time stamps fall outside the pose spline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(small_problem, k, rng):
    W, meas = small_problem["W"], small_problem["meas"]
    lid = W["lidars"][k]
    on_board = np.column_stack([lid["dir"] * meas["lidar"][k][:, None], lid["t"]])
    t0, t1 = W["pose_knots"][5], W["pose_knots"][-6]
    n_clutter = 3 * on_board.shape[0]
    clutter = np.column_stack([rng.uniform(-4.0, 4.0, (n_clutter, 3)), rng.uniform(t0 - 0.05, t1 + 0.05, n_clutter)])
    pts = np.vstack([on_board, clutter])
    return pts[rng.permutation(pts.shape[0])], on_board.shape[0]


@pytest.mark.parametrize("threshold", [0.3, 0.1])
def test_association_matches_the_oracle(small_problem, small_oracle, threshold):
    from multical_amd.backend import HipApi
    spec, H, W = small_problem["spec"], small_problem["H"], small_problem["W"]
    o, oids = small_oracle
    g = HipApi(0)
    ids = spec.emit(g)
    g.finalize()
    rng = np.random.default_rng(3)
    pts_b = W["board_points"]
    lo = np.tile(pts_b.min(0), (len(H["boards"]), 1))
    hi = np.tile(pts_b.max(0), (len(H["boards"]), 1))
    for k, h in enumerate(H["lidars"]):
        pts, n_on = _cloud(small_problem, k, rng)
        args = lambda m: (m[H["pose"]][1], m[h["toff"]], m[h["q"]], m[h["t"]], [m[b[0]] for b in H["boards"]],
                          [m[b[1]] for b in H["boards"]], lo, hi, threshold, pts)
        mask, world, margin = o.lidar_find_points_on_targets(*args(oids))
        kept, inliers, gworld = g.lidar_find_points_on_targets(*args(ids), want_world=True)
        assert np.array_equal(kept, np.nonzero(mask >> 31)[0])
        assert 0 < kept.size < pts.shape[0]                     # some readings fall outside the spline and are dropped
        assert np.abs(gworld - world[kept]).max() <= 1e-12 * np.abs(world).max()
        safe = margin[kept] > 1e-9                               # a reading within 1e-9 of a face may legitimately flip
        found = 0
        for p, idx in enumerate(inliers):
            want = np.nonzero((mask[kept] >> p) & 1)[0]
            assert np.array_equal(idx[safe[idx]], want[safe[want]])
            assert np.all(np.diff(idx) > 0)                      # ascending, like np.where in _onPlane
            found += idx.size
        assert found >= 0.8 * n_on                               # the on-board readings are recovered


def test_association_refuses_bad_arguments(small_problem):
    from multical_amd.backend import BackendError, HipApi
    spec, H = small_problem["spec"], small_problem["H"]
    g = HipApi(0)
    ids = spec.emit(g)
    h = H["lidars"][0]
    pts = np.zeros((4, 4))
    call = lambda: g.lidar_find_points_on_targets(ids[H["pose"]][1], ids[h["toff"]], ids[h["q"]], ids[h["t"]], [ids[H["boards"][0][0]]],
                                                  [ids[H["boards"][0][1]]], np.zeros((1, 3)), np.ones((1, 3)), 0.1, pts)
    with pytest.raises(BackendError, match="mc_finalize"):
        call()
    g.finalize()
    kept, inl, _ = call()
    assert kept.size == 0 or kept.size == 4
    with pytest.raises(BackendError, match="wrong kind"):
        g.lidar_find_points_on_targets(ids[H["pose"]][1], ids[h["toff"]], ids[h["t"]], ids[h["q"]], [], [], np.zeros((0, 3)),
                                       np.zeros((0, 3)), 0.1, pts)
