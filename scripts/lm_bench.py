#!/usr/bin/env python
"""Wall time of whole Levenberg-Marquardt iterations (mc_optimize: build on the GPU, read-back, host factorisation, update,
trial evaluation) on the bench workload.   python scripts/lm_bench.py [config] [iterations]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from multical_amd import synth
from multical_amd.backend import HipApi
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = synth.baseline_config(cfgi)
W = synth.make_world(cfg)
meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
spec, H = synth.build_spec(W, synth.initial_state(W), meas)
for n in (1, iters):      # the first call pays the symbolic phase of the factorisation
    g = HipApi(0); spec.emit(g)
    t0 = time.perf_counter()
    r = g.optimize(maxIterations=n, convergenceDeltaX=0.0, convergenceJDescentRatioThreshold=0.0, levenbergMarquardtLambdaInit=10.0, verbose=0)
    t1 = time.perf_counter()
    print("maxIterations %d: %.3f s  %s" % (n, t1 - t0, {k: r[k] for k in r if k in ("iterations", "failedIterations", "JFinal", "linearSolverFailure")}))
