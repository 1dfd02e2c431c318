#!/bin/bash
# usage (through gpurun): scripts/r3_check.sh <tag> [pytest -k expression]
# quick parity subset + serial-launch kernel trace of the bench command -> gpurun_out/<tag>_*
TAG=${1:-r3x}
KEXPR=${2:-"test_gpu_parity or test_camera_models or test_f1_first_shape or test_zfull or test_zz"}
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread -k "$KEXPR" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
MC_SERIAL_LAUNCH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG/trace_serial -o trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solve > gpurun_out/prof_$TAG/bench_serial.log 2>&1 || exit 1
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/prof_$TAG/trace_serial/trace_kernel_stats.csv')):
    n = r['Name']
    if 'mc::' not in n: continue
    print("%-60s calls %4s avg %9.1f us" % (n.split('(')[0][-58:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
timeout -k 10 280 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-solve > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print('ms_per_step', round(d['ms_per_step'],4), 'build_ms', d['config']['build_ms'], d['roofline']['kernels_ms'])"
