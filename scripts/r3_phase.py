#!/usr/bin/env python
"""In-kernel phase marks of the split F1 assembly kernel (diagnostic build: scripts/build_timing_lib.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multical_amd import synth
from multical_amd.backend import HipApi
cfg = synth.baseline_config(3)
W = synth.make_world(cfg)
meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
spec, H = synth.build_spec(W, synth.initial_state(W), meas)
api = HipApi(0); spec.emit(api); api.finalize()
for _ in range(2):
    api.build_system_timed(True, 1)
ph = api.phase_cycles(True, 1)
ntile = int(os.environ.get("NT", "1502"))
print("assemble phases, cycles per workgroup (wave 0): " + "  ".join("%d: %.0f" % (i, ph[i] / ntile) for i in range(8)))
print("stream slots: " + "  ".join("%d: %.3g" % (i, ph[i]) for i in range(8, 24)))
