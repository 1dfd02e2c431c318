"""Exploration aid: LM trace of mc_optimize (verbose) and of the oracle-side driver on the same problem."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from multical_amd import synth
from multical_amd.backend import HipApi
from oracle_api import OracleApi
from oracle_lm import optimize_lm
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with_oracle = (sys.argv[3] == "1") if len(sys.argv) > 3 else True
cfg = synth.SynthConfig(duration=dur, n_cams=2, n_imus=2, n_lidars=1, lidar_points_per_frame=40, corner_stride=4,
                        cam_models=[(0, 1), (1, 1)], huber_accel=1.0, huber_gyro=1.0, blake_zisserman_cam=2, motion_speed=3.0)
spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
g = HipApi(0)
ids = spec.emit(g)
r = g.optimize(maxIterations=iters, convergenceDeltaX=1e-5, convergenceJDescentRatioThreshold=1e-6, levenbergMarquardtLambdaInit=10.0, verbose=1)
print(r)
if with_oracle:
    o = OracleApi(); oids = spec.emit(o); o.finalize()
    ro = optimize_lm(o, iters, 10.0, 1e-5, 1e-6, log=print)
    print({k: v for k, v in ro.items() if k != "trace"})
