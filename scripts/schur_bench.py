#!/usr/bin/env python
"""Factor + solve of the normal equations of BASELINE configs[3]: host factorisation of the whole system against the GPU
stage (interiors eliminated on the device, reduced system on the host).  Not the bench line; numbers go to DESIGN.md.

    python scripts/schur_bench.py [parts]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multical_amd import synth
from multical_amd.backend import HipApi, HostSparseCholesky, DeviceSchurSolver, host_threads

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 0
idx = int(os.environ.get("MC_SCHUR_BENCH_CONFIG", "3"))
cfg = synth.baseline_config(idx)
spec, H, W, meas = synth.make_problem(cfg, lambda: HipApi(0))
g = HipApi(0)
spec.emit(g)
g.finalize()
st = torch.cuda.current_stream().cuda_stream
dCx = torch.zeros(g.nnz, dtype=torch.float64, device="cuda:0")
drhs = torch.zeros(g.j_cols, dtype=torch.float64, device="cuda:0")
g.evaluate_error(True)
g.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), st)
torch.cuda.synchronize()
shift = 100.0
out = dict(config=idx, terms=g.n_terms, j_cols=g.j_cols, nnz=g.nnz, threads=host_threads())
# host: read-back + factor + solve
Cp, Ci = g.pattern()
ch = HostSparseCholesky()
ch.analyze(g.j_cols, Cp, Ci, g.block_bases())
ch.set_block_coordinates(g.block_times())
host = []
for it in range(4):
    t0 = time.perf_counter()
    Cx = dCx.cpu().numpy()
    rhs = drhs.cpu().numpy()
    t1 = time.perf_counter()
    Cx[Cp[1:] - 1] += shift
    x_host = ch.solve(g.j_cols, Cp, Ci, Cx, rhs, g.block_bases())
    t2 = time.perf_counter()
    host.append((t1 - t0, t2 - t1))
out["host"] = dict(read_back_s=round(min(h[0] for h in host[1:]), 5), factor_solve_s=round(min(h[1] for h in host[1:]), 5), **ch.stats())
# device stage
t0 = time.perf_counter()
s = DeviceSchurSolver(g, dCx.data_ptr(), parts)
out["plan_s"] = round(time.perf_counter() - t0, 3)
if not s.planned:
    out["device"] = s.why
else:
    dev = []
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, x = s.solve(dCx.data_ptr(), drhs.data_ptr(), shift, st)
        dev.append(time.perf_counter() - t0)
        assert rc == 0, rc
    info = s.info()
    info["seconds"] = {k: round(v, 5) for k, v in info["seconds"].items()}
    out["device"] = dict(solve_s=round(min(dev[1:]), 5), first_solve_s=round(dev[0], 4), **info)
    out["max_rel_diff_vs_host"] = float(np.abs(x - x_host).max() / np.abs(x_host).max())
print(json.dumps(out))
