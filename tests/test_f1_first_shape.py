"""The first shape of the F1 buildSystem kernel (a wave per observation, MC_F1_WAVE_PER_OBSERVATION=1) is kept for
comparison with the corner-shared one; the switch is read once per process, so the comparison needs a second process."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from multical_amd import synth
from multical_amd.backend import HipApi
cfg = synth.SynthConfig(duration=2.0, n_cams=3, n_imus=1, n_lidars=1, lidar_points_per_frame=10, corner_stride=5,
                        cam_models=[(0, 1), (1, 1), (2, 0)], blake_zisserman_cam=2)
spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
g = HipApi(0); spec.emit(g); g.finalize()
J, e = g.evaluate_error(True)
Cx, rhs = g.build_system(True)
np.savez(sys.argv[1], J=J, e=e, Cx=Cx, rhs=rhs)
"""


@pytest.mark.gpu
def test_both_shapes_of_the_f1_build_kernel_agree(tmp_path):
    out = {}
    for name, extra in (("corner_shared", {}), ("wave_per_observation", {"MC_F1_WAVE_PER_OBSERVATION": "1"})):
        path = str(tmp_path / (name + ".npz"))
        env = dict(os.environ)
        env.pop("MC_F1_WAVE_PER_OBSERVATION", None)
        env.update(extra)
        subprocess.run([sys.executable, "-c", SNIPPET % dict(root=ROOT), path], check=True, env=env, timeout=300)
        out[name] = np.load(path)
    a, b = out["corner_shared"], out["wave_per_observation"]
    assert a["Cx"].shape == b["Cx"].shape and np.abs(a["Cx"]).max() > 0
    # same arithmetic per corner; the Gram sums and the fp64 atomics associate differently
    assert np.abs(a["Cx"] - b["Cx"]).max() <= 1e-11 * np.abs(a["Cx"]).max()
    assert np.abs(a["rhs"] - b["rhs"]).max() <= 1e-11 * np.abs(a["rhs"]).max()
    assert np.array_equal(a["e"], b["e"]) and abs(float(a["J"]) - float(b["J"])) <= 1e-13 * abs(float(a["J"]))
