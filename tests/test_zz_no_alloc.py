"""mc_build_system_device_noalloc (the entry point that allocates nothing: its replica scratch is a file-scope device
array of the library) against mc_build_system_device, through the C ABI, on two calls in a row -- the second on a smaller
problem, so stale contents of the shared scratch array would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(cfg):
    from multical_amd import synth
    from multical_amd.backend import HipApi
    spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
    g = HipApi(0)
    spec.emit(g)
    g.finalize()
    return g


def _both(g, use_mest=True):
    import torch
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for fn in (g.build_system_device, g.build_system_device_noalloc):
        dCx = torch.full((g.nnz,), 7.0, dtype=torch.float64, device="cuda:0")
        drhs = torch.full((g.j_cols,), 7.0, dtype=torch.float64, device="cuda:0")
        fn(use_mest, dCx.data_ptr(), drhs.data_ptr(), st)
        torch.cuda.synchronize()
        g.check_status()
        out.append((dCx.cpu().numpy(), drhs.cpu().numpy()))
    return out


def test_noalloc_entry_point_matches_build_system_device_twice_in_a_row():
    from multical_amd import synth
    big = synth.SynthConfig(duration=3.0, n_cams=3, n_imus=2, n_lidars=1, lidar_points_per_frame=30, corner_stride=7,
                            cam_models=[(0, 1), (1, 1), (2, 0)], huber_accel=1.0, huber_gyro=1.0, blake_zisserman_cam=2,
                            lidar_mest=(2, 4.0))
    small = synth.SynthConfig(duration=1.5, n_cams=1, n_imus=1, n_lidars=1, lidar_points_per_frame=20, corner_stride=9)
    for cfg in (big, small):
        g = _ctx(cfg)
        g.evaluate_error(True)
        (a, ra), (b, rb) = _both(g)
        # bit-exact for the pattern of zeros, 1e-12 for the values (fp64 atomics: order-dependent last bits); bar of the parity
        # suite is 1e-9
        assert np.abs(a).max() > 0
        assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
        assert np.abs(ra - rb).max() <= 1e-12 * np.abs(ra).max()
        assert np.array_equal(a == 0.0, b == 0.0)
        # and a second call on the same context reproduces itself (the call clears the scratch it uses)
        (_, _), (b2, rb2) = _both(g)
        assert np.abs(b2 - b).max() <= 1e-12 * np.abs(b).max() and np.abs(rb2 - rb).max() <= 1e-12 * np.abs(rb).max()


def test_noalloc_entry_point_allocates_no_device_memory():
    """hipMemGetInfo before and after the call (after a first call has loaded the kernels' code objects): the free device
    memory does not move, while the allocating entry point is free to change it."""
    import torch
    from multical_amd import synth
    g = _ctx(synth.SynthConfig(duration=2.0, n_cams=2, n_imus=1, n_lidars=1, lidar_points_per_frame=20, corner_stride=7))
    st = torch.cuda.current_stream().cuda_stream
    dCx = torch.zeros(g.nnz, dtype=torch.float64, device="cuda:0")
    drhs = torch.zeros(g.j_cols, dtype=torch.float64, device="cuda:0")
    g.build_system_device_noalloc(True, dCx.data_ptr(), drhs.data_ptr(), st)   # loads the code objects
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(3):
        g.build_system_device_noalloc(True, dCx.data_ptr(), drhs.data_ptr(), st)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 == free0, (free0, free1)
    g.check_status()


def test_noalloc_entry_point_rejects_missing_outputs():
    from multical_amd import synth
    from multical_amd.backend import BackendError
    g = _ctx(synth.SynthConfig(duration=1.5, n_cams=1, n_imus=1, n_lidars=0, corner_stride=9))
    with pytest.raises(BackendError, match="required"):
        g.build_system_device_noalloc(True, None, None, None)


def test_noalloc_entry_point_rejects_sizes_beyond_its_scratch():
    """308 scalar columns of global design variables (50 views x 6 + intrinsics) need more replicas than the static array
    holds: rejected with a message, nothing launched.  608 columns: the library uses no replicas at all (more than 512),
    so the call is accepted and equals mc_build_system_device."""
    from multical_amd import synth_static as SS
    from multical_amd.backend import HipApi, BackendError
    for views, ok in ((50, False), (100, True)):
        cfg = SS.StaticConfig(n_cams=1, n_boards=1, n_views=views, corner_stride=5, cam_models=[(0, 1)])
        spec, H, W, meas = SS.make_static_problem(cfg, lambda: HipApi(0))
        g = HipApi(0)
        spec.emit(g)
        g.finalize()
        g.evaluate_error(True)
        if ok:
            (a, ra), (b, rb) = _both(g)
            assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max() and np.abs(ra - rb).max() <= 1e-12 * np.abs(ra).max()
        else:
            import torch
            dCx = torch.zeros(g.nnz, dtype=torch.float64, device="cuda:0")
            drhs = torch.zeros(g.j_cols, dtype=torch.float64, device="cuda:0")
            with pytest.raises(BackendError, match="at most 256"):
                g.build_system_device_noalloc(True, dCx.data_ptr(), drhs.data_ptr(), torch.cuda.current_stream().cuda_stream)
