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
info = api.launch_info()
res = {}
MODES = tuple(int(m) for m in os.environ.get('MC_MODES', '0,3,2,1').split(','))
for mode in MODES:
    for _ in range(2): api.build_system_timed(True, mode)
    res[mode] = np.mean([api.build_system_timed(True, mode) for _ in range(5)], axis=0)
for i, (name, nbatch, nt, nwg, fd, fl, rb) in enumerate(info):
    print("%-22s batches %3d terms %8d workgroups %7d (%.1f terms/wg) | " % (name, nbatch, nt, nwg, nt / max(1, nwg)) +
          "  ".join("%s %.3f" % ({0: "eval", 3: "jac", 2: "+gram", 1: "full", 4: "m4", 5: "m5"}[m], res[m][i]) for m in MODES) + " ms")
print("total: " + "  ".join("mode%d %.3f" % (m, float(res[m].sum())) for m in res))
