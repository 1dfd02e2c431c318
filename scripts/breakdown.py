#!/usr/bin/env python
"""Per-kernel phase breakdown on the bench workload: build modes 0 (evaluate), 3 (Jacobians only), 2 (+Gram, no atomics), 1 (full)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multical_amd import synth
from multical_amd.backend import HipApi
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dur = float(sys.argv[2]) if len(sys.argv) > 2 else None
cfg = synth.baseline_config(cfgi) if dur is None else synth.baseline_config(cfgi, duration=dur)
W = synth.make_world(cfg)
meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
spec, H = synth.build_spec(W, synth.initial_state(W), meas)
api = HipApi(0); spec.emit(api); api.finalize()
info = api.batch_info()
names = {0: "reproj", 1: "static", 2: "accel", 3: "gyro", 4: "lidar", 5: "motion"}
res = {}
for mode in (0, 3, 2, 1):
    for _ in range(2): api.build_system_timed(True, mode)
    ms = np.mean([api.build_system_timed(True, mode) for _ in range(5)], axis=0)
    res[mode] = ms
agg = {}
for i, (kind, nt, ng) in enumerate(info):
    a = agg.setdefault(names[kind], dict(n=0, terms=0, groups=0, ms={m: 0.0 for m in res}))
    a["n"] += 1; a["terms"] += nt; a["groups"] += ng
    for m in res: a["ms"][m] += float(res[m][i])
for k, a in agg.items():
    print("%-7s batches %3d terms %8d groups %7d (%.1f terms/group) | eval %.3f  jac %.3f  +gram %.3f  full %.3f ms" %
          (k, a["n"], a["terms"], a["groups"], a["terms"] / max(1, a["groups"]), a["ms"][0], a["ms"][3], a["ms"][2], a["ms"][1]))
print("total: " + "  ".join("mode%d %.3f" % (m, float(res[m].sum())) for m in res))
