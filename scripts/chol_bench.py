#!/usr/bin/env python
"""Host factorisation + solve on the bench workload's pattern, without a GPU: the structure comes from the product's own
finalize (tests/hostprobe), the values are a synthetic positive definite fill of that pattern.
   python scripts/chol_bench.py [config] [duration_s] [threads]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from multical_amd import synth
from multical_amd.backend import HostSparseCholesky as HostCholesky
from probe_api import ProbeApi

cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dur = float(sys.argv[2]) if len(sys.argv) > 2 else None
if len(sys.argv) > 3:
    os.environ["MC_CHOLESKY_THREADS"] = sys.argv[3]
cfg = synth.baseline_config(cfgi) if dur is None else synth.baseline_config(cfgi, duration=dur)
W = synth.make_world(cfg)
spec, H = synth.build_spec(W, synth.initial_state(W), synth.zero_measurements(W))
p = ProbeApi(); spec.emit(p); p.finalize()
Cp, Ci = p.pattern()
bb = p.block_bases()
n = p.j_cols
rng = np.random.RandomState(0)
t0 = time.perf_counter()
_, Cx, rhs = p.cpu_optimised(True, True, os.cpu_count())     # the real normal equations (zero measurements): real zeros
Cx[Cp[1:] - 1] += 100.0                                       # Levenberg-Marquardt damping, lambda = 10
print("assembled on the host in %.1f s; %.1f %% of the stored values are zero" % (time.perf_counter() - t0, 100.0 * np.mean(Cx == 0.0)))
b = rng.standard_normal(n)
if os.environ.get('MC_CHOL_DUMP'):      # raw arrays for a stand-alone C++ driver of host/block_cholesky.hpp
    d = os.environ['MC_CHOL_DUMP']
    for name, arr in (('Cp', Cp), ('Ci', Ci), ('Cx', Cx), ('bb', bb), ('b', b), ('times', p.block_times())):
        np.ascontiguousarray(arr).tofile(os.path.join(d, name + '.bin'))
    sys.exit(0)
h = HostCholesky()
from multical_amd.backend import host_threads
threads = host_threads()
# block times without a device context: the same rule as mc_get_block_times, from the spec
times = p.block_times()
t0 = time.perf_counter(); h.analyze(n, Cp, Ci, bb); h.set_block_coordinates(times, threads); t1 = time.perf_counter()
print("n %d  nnz %d  blocks %d  threads %d  analyze %.3f s" % (n, Ci.size, bb.size - 1, threads, t1 - t0))
for rep in range(3):
    t0 = time.perf_counter()
    x = h.solve(n, Cp, Ci, Cx, b, bb)
    t1 = time.perf_counter()
    print("factor + solve %.4f s   %s" % (t1 - t0, h.stats() if rep == 0 else ""))
# residual of the upper-triangular CCS system
import scipy.sparse as sp
A = sp.csc_matrix((Cx, Ci, Cp), shape=(n, n)); A = A + sp.triu(A, 1).T
print("relative residual %.2e" % (np.linalg.norm(A @ x - b) / np.linalg.norm(b)))
