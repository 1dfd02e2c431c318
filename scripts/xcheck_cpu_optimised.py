"""Cross-check at a realistic size (configs[3], 12 s: 367 k terms, chunks of 30 coefficient blocks): the GPU system against the
host build of the same formulation (tests/hostprobe cpu_optimised, 16 threads).  Not the oracle: a check of the kernels' reductions."""
import os, sys, time
root = "/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from multical_amd import synth
from multical_amd.backend import HipApi
from probe_api import ProbeApi
cfg = synth.baseline_config(3, duration=12.0)
W = synth.make_world(cfg)
meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
spec, H = synth.build_spec(W, synth.initial_state(W), meas)
g = HipApi(0); spec.emit(g); g.finalize()
p = ProbeApi(); spec.emit(p); p.finalize()
Jg, eg = g.evaluate_error(True)
Cxg, rhsg = g.build_system(True)
Jp, Cxp, rhsp = p.cpu_optimised(True, True, 16)
print("terms", g.n_terms, "nnz", Cxg.size)
print("J rel", abs(Jg - Jp) / abs(Jp))
print("Cx max rel (to max)", np.abs(Cxg - Cxp).max() / np.abs(Cxp).max())
print("rhs max rel", np.abs(rhsg - rhsp).max() / np.abs(rhsp).max())
# per-column-scale check like the parity tests
print("zeros agree", np.array_equal(Cxg == 0.0, Cxp == 0.0))
