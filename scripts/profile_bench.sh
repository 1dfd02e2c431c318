#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + two PMC passes of the default bench command.
# usage: scripts/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ARGS > $OUT/bench_trace.log 2>&1
# the product launches the term families on separate streams, so their kernels overlap and rocprofv3's per-kernel durations
# stretch; a second trace with the launches serialised gives durations comparable with bench.py's HIP-event figures
export MC_SERIAL_LAUNCH=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o trace --output-format csv -- python3 $ARGS > $OUT/bench_trace_serial.log 2>&1
unset MC_SERIAL_LAUNCH
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 $ARGS > $OUT/bench_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- python3 $ARGS > $OUT/bench_pmc_write.log 2>&1
find $OUT -name '*.csv' | head -20
