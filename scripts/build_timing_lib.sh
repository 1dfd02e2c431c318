#!/bin/bash
# Diagnostic build with in-kernel phase marks (mc_phase_cycles) and the by-removal modes of the stream kernels.
# Use: MULTICAL_HIP_LIB=$PWD/multical_amd/libmulticalhip_timing.so.  Only mc_kernels.hip is rebuilt (with -DMC_PHASE_TIMING); the other
# objects are the product's (python __graft_entry__.py first).
set -e
cd "$(dirname "$0")/.."
B=multical_amd/csrc/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Xarch_host -march=x86-64-v3 -DMC_PHASE_TIMING -fPIC -c -o $B/mc_kernels_timing.o multical_amd/csrc/mc_kernels.hip
hipcc -shared -fPIC -o multical_amd/libmulticalhip_timing.so $B/mc_kernels_timing.o $B/mc_driver.o $B/mc_host.o $B/mc_schur.o
