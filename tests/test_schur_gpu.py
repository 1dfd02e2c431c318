"""The factorisation's GPU stage on the device (mc_schur_*): same solution as the host factorisation of the whole system."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(**kw):
    import torch
    from multical_amd import synth
    from multical_amd.backend import HipApi
    cfg = synth.SynthConfig(**kw)
    spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
    g = HipApi(0)
    spec.emit(g)
    g.finalize()
    g.evaluate_error(True)
    dCx = torch.zeros(g.nnz, dtype=torch.float64, device="cuda:0")
    drhs = torch.zeros(g.j_cols, dtype=torch.float64, device="cuda:0")
    g.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g.check_status()
    return g, dCx, drhs


def _host_solution(g, Cx, rhs, shift):
    from multical_amd.backend import HostSparseCholesky
    Cp, Ci = g.pattern()
    Cx = Cx.copy()
    Cx[Cp[1:] - 1] += shift          # the diagonal entry is the last of its column
    ch = HostSparseCholesky()
    ch.analyze(g.j_cols, Cp, Ci, g.block_bases())
    ch.set_block_coordinates(g.block_times(), 4)
    x = ch.solve(g.j_cols, Cp, Ci, Cx, rhs, g.block_bases())
    assert x is not None
    return x


@pytest.mark.parametrize("parts", [1, 2, 5, 0])
def test_device_elimination_matches_the_host_factorisation(parts):
    import torch
    from multical_amd.backend import DeviceSchurSolver
    g, dCx, drhs = _problem(duration=6.0, n_cams=2, n_imus=2, n_lidars=1, lidar_points_per_frame=12, corner_stride=9,
                            huber_accel=1.0, blake_zisserman_cam=2, lidar_mest=(2, 4.0))
    s = DeviceSchurSolver(g, dCx.data_ptr(), parts)
    assert s.planned, s.why
    Cx, rhs = dCx.cpu().numpy(), drhs.cpu().numpy()
    for shift in (100.0, 1.0):
        rc, x = s.solve(dCx.data_ptr(), drhs.data_ptr(), shift, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        want = _host_solution(g, Cx, rhs, shift)
        assert np.abs(x - want).max() <= 1e-8 * np.abs(want).max(), (parts, shift)
    info = s.info()
    assert info["r"] == 5 and info["block_dim"] == 6 and info["reduced_columns"] < g.j_cols
    if parts:
        assert info["interiors"] == parts


def test_not_positive_definite_and_grown_structure_are_reported():
    import torch
    from multical_amd.backend import DeviceSchurSolver
    g, dCx, drhs = _problem(duration=3.0, n_cams=1, n_imus=1, n_lidars=1, lidar_points_per_frame=10, corner_stride=9)
    st = torch.cuda.current_stream().cuda_stream
    s = DeviceSchurSolver(g, dCx.data_ptr(), 2)
    assert s.planned, s.why
    rc, _ = s.solve(dCx.data_ptr(), drhs.data_ptr(), -1e12, st)       # H - 1e12 I is not positive definite
    assert rc == 1
    # a structurally present block of a pose-coefficient column that is zero in this state becomes non-zero
    Cp, Ci = g.pattern()
    col = int(g.block_bases()[40])
    Cx = dCx.cpu().numpy()
    zero = np.nonzero(Cx[Cp[col]:Cp[col + 1]] == 0.0)[0]
    assert zero.size > 0
    d2 = dCx.clone()
    d2[int(Cp[col] + zero[0])] = 1e-3
    rc, _ = s.solve(d2.data_ptr(), drhs.data_ptr(), 1.0, st)
    assert rc == 2
    rc, x = s.solve(dCx.data_ptr(), drhs.data_ptr(), 1.0, st)         # the handle is still good for the planned structure
    assert rc == 0 and np.all(np.isfinite(x))


def test_blocks_wider_than_eight_columns(small_problem):
    """The scale-misalignment IMU model brings 9-column design-variable blocks (MatrixBasicDv): the gather kernels take a lane
    per column of a block pair, up to 16."""
    import torch
    from multical_amd.backend import HipApi, DeviceSchurSolver
    from test_variants import make_variant
    sp = make_variant(small_problem["spec"], "scale_misalignment")
    g = HipApi(0)
    sp.emit(g)
    g.finalize()
    assert int(np.diff(g.block_bases()).max()) > 8
    g.evaluate_error(True)
    dCx = torch.zeros(g.nnz, dtype=torch.float64, device="cuda:0")
    drhs = torch.zeros(g.j_cols, dtype=torch.float64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), st)
    torch.cuda.synchronize()
    s = DeviceSchurSolver(g, dCx.data_ptr(), 0)
    assert s.planned, s.why
    rc, x = s.solve(dCx.data_ptr(), drhs.data_ptr(), 100.0, st)
    assert rc == 0
    want = _host_solution(g, dCx.cpu().numpy(), drhs.cpu().numpy(), 100.0)
    assert np.abs(x - want).max() <= 1e-6 * np.abs(want).max()       # an ill-conditioned system; LinearSolverTests.cpp:55-57's bar
