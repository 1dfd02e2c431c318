#!/bin/bash
# Diagnostic build with in-kernel phase marks (mc_phase_cycles).  Use: MULTICAL_HIP_LIB=$PWD/multical_amd/libmulticalhip_timing.so
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMC_PHASE_TIMING -fPIC -shared -o multical_amd/libmulticalhip_timing.so multical_amd/csrc/mc_kernels.hip
