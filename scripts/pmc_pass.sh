#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counters...>   -> gpurun_out/pmc_<tag>/
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" -d $OUT -o pmc --output-format csv -- python3 scripts/breakdown.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open("$OUT/pmc_counter_collection.csv")):
    k = r["Kernel_Name"][:60]
    a = agg[k][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    if "mc::" not in k: continue
    print(k, " ".join("%s=%.4g" % (c, v[1] / v[0]) for c, v in d.items()), "launches", list(d.values())[0][0])
PY
