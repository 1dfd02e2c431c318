import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_problem():
    """2 s, 3 cameras (pinhole-radtan, omni-radtan, double sphere), 2 IMUs, 1 LiDAR, every M-estimator in use."""
    from multical_amd import synth
    from oracle_api import OracleApi
    cfg = synth.SynthConfig(duration=2.0, n_cams=3, n_imus=2, n_lidars=1, lidar_points_per_frame=20, corner_stride=9,
                            cam_models=[(0, 1), (1, 1), (2, 0)], huber_accel=1.0, huber_gyro=1.0, blake_zisserman_cam=2,
                            lidar_mest=(2, 4.0))
    spec, H, W, meas = synth.make_problem(cfg, OracleApi)
    return dict(cfg=cfg, spec=spec, H=H, W=W, meas=meas)


@pytest.fixture(scope="session")
def small_oracle(small_problem):
    from oracle_api import OracleApi
    o = OracleApi()
    ids = small_problem["spec"].emit(o)
    o.finalize()
    return o, ids
