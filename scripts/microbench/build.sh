#!/bin/sh
# builds the fp64-atomics micro-benchmark next to its source (the binary is not tracked)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomics_f64 atomics_f64.hip
