#!/usr/bin/env python
"""CPU-only timing aid: the host factorisation of the reduced system of BASELINE configs[idx] (what is left once the
pose-band interiors are eliminated), with the normal equations from the host twin of the kernels.

    MC_CHOLESKY_TIMING=1 python scripts/schur_host_profile.py [config] [threads] [parts]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from multical_amd import synth
from probe_api import ProbeApi, _d

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dur = float(os.environ.get("MC_PROFILE_SECONDS", "0")) or None
cfg = synth.baseline_config(idx, duration=dur) if dur else synth.baseline_config(idx)
t0 = time.perf_counter()
W = synth.make_world(cfg)
meas = synth.zero_measurements(W)     # values do not matter for the timing: the system is shifted to positive definite
spec, _ = synth.build_spec(W, synth.initial_state(W), meas)
p = ProbeApi()
spec.emit(p)
p.finalize()
print("problem: %d terms, %d columns, set-up %.1f s" % (p.n_terms, p.j_cols, time.perf_counter() - t0), flush=True)
J, Cx, rhs = p.cpu_optimised(True, True, threads)
if parts <= 0:
    parts = max(1, min(512, int(os.environ.get("MC_BAND_BLOCKS", "0")) // 90)) or 1
out = np.zeros(4)
p.L.probe_schur_reduced_bench.restype = C.c_int
rc = p.L.probe_schur_reduced_bench(p.h, _d(Cx), _d(rhs), C.c_double(100.0), C.c_int(parts), C.c_int(threads), C.c_int(6), _d(out))
print("rc", rc, "factorize %.2f ms, solve %.2f ms, reduced columns %d" % (out[0], out[1], out[2]))
