#!/bin/bash
# Host factorisation on the GPU box's CPU cores: panel-merging settings (tuning aids MC_CHOLESKY_RELAX / MC_CHOLESKY_MAX_PANEL).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/chol_sweep.log
for cfg in "$@"; do
  r=${cfg%%:*}; m=${cfg##*:}
  echo "== relax $r maxp $m" >> gpurun_out/chol_sweep.log
  MC_CHOLESKY_RELAX=$r MC_CHOLESKY_MAX_PANEL=$m MC_CHOLESKY_TIMING=1 timeout -k 10 200 python scripts/chol_bench.py 3 2>&1 | grep -E "^cholesky|^factor|^n " | cut -c1-200 >> gpurun_out/chol_sweep.log || exit 1
done
cat gpurun_out/chol_sweep.log
