#!/bin/bash
# usage (through gpurun): scripts/r3_variants.sh <kernel substring> <tag> [<tag> ...]: rocprofv3 average of one kernel per experimental library
KPAT=$1; shift
export TMPDIR=/tmp
for tag in "$@"; do
  lib=$PWD/multical_amd/libmulticalhip_$tag.so
  [ "$tag" = base ] && lib=$PWD/multical_amd/libmulticalhip.so
  out=gpurun_out/var_$tag
  rm -rf $out; mkdir -p $out
  MULTICAL_HIP_LIB=$lib MC_SERIAL_LAUNCH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solve > $out/log.txt 2>&1 || { echo "$tag failed"; tail -3 $out/log.txt; exit 1; }
  python3 - <<PY
import csv
for r in csv.DictReader(open('$out/trace_kernel_stats.csv')):
    if '$KPAT' in r['Name']:
        print("$tag %-50s avg %9.1f us" % (r['Name'].split('(')[0][-48:], float(r['AverageNs']) / 1e3))
PY
done
