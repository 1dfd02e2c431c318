#!/bin/bash
# Chunk-size sweep of the stream kernels on the bench workload (MC_CHUNK_TERMS_* are tuning aids read by Problem::finalize).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for cfg in "$@"; do
  l=${cfg%%:*}; i=${cfg##*:}
  echo "== lidar $l imu $i" >> gpurun_out/sweep.log
  MC_CHUNK_TERMS_LIDAR=$l MC_CHUNK_TERMS_IMU=$i MC_MODES=1 timeout -k 10 120 python scripts/breakdown.py 2>&1 | grep -E "k_stream<Imu>|k_stream<Lidar>|total" >> gpurun_out/sweep.log || exit 1
done
cat gpurun_out/sweep.log
