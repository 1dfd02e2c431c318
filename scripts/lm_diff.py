import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from multical_amd import synth, backend as B
from oracle_api import OracleApi
from oracle_lm import optimize_lm
cfg = synth.SynthConfig(duration=2.0, n_cams=3, n_imus=2, n_lidars=1, lidar_points_per_frame=20, corner_stride=9,
                        cam_models=[(0, 1), (1, 1), (2, 0)], huber_accel=1.0, huber_gyro=1.0, blake_zisserman_cam=2,
                        lidar_mest=(2, 4.0), motion_speed=5.0)
spec, H, W, meas = synth.make_problem(cfg, lambda: B.HipApi(0))
for iters in (6, 12, 25):
    g = B.HipApi(0); ids = spec.emit(g)
    o = OracleApi(); oids = spec.emit(o); o.finalize()
    o2 = OracleApi(); oids2 = spec.emit(o2); o2.finalize()
    r = g.optimize(maxIterations=iters, convergenceDeltaX=1e-5, convergenceJDescentRatioThreshold=1e-6, levenbergMarquardtLambdaInit=10.0)
    ro = o.optimize(iters, 10.0, 1e-5, 1e-6, 1)
    ro2 = optimize_lm(o2, iters, 10.0, 1e-5, 1e-6)
    d = []; d2 = []
    for h, v in ids.items():
        if isinstance(v, int) and spec.ops[h][0] == "dv":
            d.append((np.abs(g.dv_params(v) - o.dv_params(oids[h])).max(), h, spec.ops[h][1]["kind"]))
            d2.append((np.abs(o2.dv_params(oids2[h]) - o.dv_params(oids[h])).max(), h))
    d.sort(reverse=True); d2.sort(reverse=True)
    print(iters, r["iterations"], r["failedIterations"], ro["iterations"], ro["failedIterations"], "gpu-vs-oracle(dense):", d[:3], " oracle(sparse LU)-vs-oracle(dense):", d2[:2], flush=True)
