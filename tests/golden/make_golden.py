"""Generates tests/golden/small_problem.npz from the CPU oracle (run from the repo root:
python tests/golden/make_golden.py).  The reference cannot be built or imported here, so these vectors
come from the restatement itself: they guard against drift of the oracle, they do not pin it to the reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from multical_amd import synth  # noqa: E402
from oracle_api import OracleApi  # noqa: E402

cfg = synth.SynthConfig(duration=2.0, n_cams=3, n_imus=2, n_lidars=1, lidar_points_per_frame=20, corner_stride=9,
                        cam_models=[(0, 1), (1, 1), (2, 0)], huber_accel=1.0, huber_gyro=1.0, blake_zisserman_cam=2,
                        lidar_mest=(2, 4.0))
spec, H, W, meas = synth.make_problem(cfg, OracleApi)
o = OracleApi()
spec.emit(o)
o.finalize()
J, e = o.evaluate_error(True)
Cp, Ci, Cx, rhs = o.build_system(True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "small_problem.npz"), J=J, e=e, Cp=Cp, Ci=Ci, Cx=Cx, rhs=rhs)
print("wrote golden: J=%.9g nnz=%d" % (J, Cx.size))
