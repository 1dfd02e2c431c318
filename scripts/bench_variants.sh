#!/bin/bash
# Per-kernel timings (eval / full build) of the bench workload for every experimental library given by tag.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for tag in "$@"; do
  lib=$PWD/multical_amd/libmulticalhip_$tag.so
  [ "$tag" = base ] && lib=$PWD/multical_amd/libmulticalhip.so
  echo "== $tag" >> gpurun_out/variants.log
  MC_MODES=1 MULTICAL_HIP_LIB=$lib timeout -k 10 120 python scripts/breakdown.py 2>&1 | grep -E "k_reproj|k_stream<Imu>|k_stream<Lidar>|total" >> gpurun_out/variants.log || exit 1
done
cat gpurun_out/variants.log
