#!/usr/bin/env python
"""Per-kernel phase breakdown on the bench workload: build modes 0 (evaluate), 3 (Jacobians only), 2 (+Gram, no atomics), 1 (full).
Modes 2 / 3 and the in-kernel phase marks need the diagnostic build: scripts/build_timing_lib.sh, then
MULTICAL_HIP_LIB=$PWD/multical_amd/libmulticalhip_timing.so python scripts/breakdown.py (with the product library modes 2 / 3 run as 1)."""
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

if 'timing' not in os.environ.get('MULTICAL_HIP_LIB', ''):
    sys.exit(0)
names = {0: "stage record", 1: "project+z", 2: "wait waves", 3: "mfma gram", 4: "G->T->M", 5: "board flush", 6: "image flush",
         8: "lidar eval", 9: "lidar wait", 10: "lidar runs", 11: "lidar band", 12: "lidar arrow", 13: "lidar oo",
         16: "imu eval", 17: "imu wait", 18: "imu runs", 19: "imu band", 20: "imu arrow", 21: "imu oo"}
nwg = {0: info[0][3], 8: info[3][3], 16: info[2][3]}
for pmode in (int(m) for m in os.environ.get('MC_PHASE_MODES', '1').split(',')):
    ph = api.phase_cycles(True, pmode)
    for base in (0, 8, 16):
        tot = ph[base:base + 8].sum()
        print("mode %d phases (%s), cycles per workgroup: " % (pmode, "k_reproj" if base == 0 else names[base].split()[0]) +
              "  ".join("%s %.0f (%.0f%%)" % (names[i], ph[i] / max(1, nwg[base]), 100 * ph[i] / max(1.0, tot)) for i in range(base, base + 8) if i in names))
