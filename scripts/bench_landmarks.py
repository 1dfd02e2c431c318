#!/usr/bin/env python
"""Static camera-calibration problem with and without estimateLandmarks (target corners as HomogeneousPoint design
variables): evaluateError + buildSystem time of k_static.  VERDICT round 2, item 7: the landmark problem within 2x of the
plain one.

    python scripts/bench_landmarks.py [config index, default 0]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multical_amd import synth_static as SS
from multical_amd.backend import HipApi

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = SS.baseline_static_config(idx)
base = SS.make_static_problem(cfg, lambda: HipApi(0))[0]
out = dict(config=idx)
for name, spec in (("plain", base), ("landmarks", SS.with_landmarks(base, inactive_every=0, constant_every=0))):
    api = HipApi(0)
    spec.emit(api)
    api.finalize()
    for _ in range(3):
        api.build_system_timed(True, True)
    b = np.mean([api.build_system_timed(True, True) for _ in range(10)], axis=0)
    e = np.mean([api.build_system_timed(True, False) for _ in range(10)], axis=0)
    out[name] = dict(terms=api.n_terms, j_cols=api.j_cols, nnz=api.nnz, n_blocks=api.n_blocks,
                     build_ms=round(float(b.sum()), 4), evaluate_ms=round(float(e.sum()), 4))
out["build_ratio"] = round(out["landmarks"]["build_ms"] / out["plain"]["build_ms"], 2)
print(json.dumps(out))
