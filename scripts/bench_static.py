#!/usr/bin/env python
"""evaluateError + buildSystem timing of the static camera-calibration problems (BASELINE configs[0] and configs[2],
ReprojectionError with intrinsics: kernel k_static) and of the J^T backend the reference's default solver
uses on them.  Not the bench line (bench.py measures configs[3]); numbers go to DESIGN.md.

    python scripts/bench_static.py 2
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multical_amd import synth_static as SS
from multical_amd.backend import HipApi

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = SS.baseline_static_config(idx)
t0 = time.time()
spec, H, W, meas = SS.make_static_problem(cfg, lambda: HipApi(0))
api = HipApi(0)
spec.emit(api)
api.finalize()
setup = time.time() - t0
for _ in range(3):
    api.build_system_timed(True, True)
b = np.mean([api.build_system_timed(True, True) for _ in range(10)], axis=0)
e = np.mean([api.build_system_timed(True, False) for _ in range(10)], axis=0)
out = dict(config=idx, terms=api.n_terms, j_cols=api.j_cols, nnz=api.nnz, n_blocks=api.n_blocks, setup_s=round(setup, 2),
           build_ms=round(float(b.sum()), 4), evaluate_ms=round(float(e.sum()), 4),
           terms_per_s=float(api.n_terms / ((b.sum() + e.sum()) * 1e-3)))
# J^T backend (SparseCholeskyLinearSystemSolver, the default the camera calibrators run)
nnz = api.jt_nnz()
d = torch.zeros(nnz, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    api._chk(api.lib.mc_build_jacobian_transpose_device(api.ctx, 1, d.data_ptr(), st))
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(10):
    api._chk(api.lib.mc_build_jacobian_transpose_device(api.ctx, 1, d.data_ptr(), st))
torch.cuda.synchronize()
out["jt_nnz"] = nnz
out["jt_build_ms"] = round((time.perf_counter() - t1) / 10 * 1e3, 4)
r = api.optimize(maxIterations=1)
out["host_cholesky"] = dict(solve_s=round(r["seconds_solve_system"], 4), factor_nnz=r["factor_nnz"], factor_flops=r["factor_flops"])
print(json.dumps(out))
