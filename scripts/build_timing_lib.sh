#!/bin/bash
# Diagnostic build with in-kernel phase marks (mc_phase_cycles).  Use: MULTICAL_HIP_LIB=$PWD/multical_amd/libmulticalhip_timing.so
set -e
cd "$(dirname "$0")/.."
mkdir -p multical_amd/csrc/build
g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -c -o multical_amd/csrc/build/mc_host.o multical_amd/csrc/mc_host.cpp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Xarch_host -march=x86-64-v3 -DMC_PHASE_TIMING -fPIC -c -o multical_amd/csrc/build/mc_kernels_timing.o multical_amd/csrc/mc_kernels.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xarch_host -march=x86-64-v3 -fPIC -c -o multical_amd/csrc/build/mc_driver.o multical_amd/csrc/mc_driver.hip
hipcc -shared -fPIC -o multical_amd/libmulticalhip_timing.so multical_amd/csrc/build/mc_kernels_timing.o multical_amd/csrc/build/mc_driver.o multical_amd/csrc/build/mc_host.o
