#!/bin/bash
# Experimental builds of the device library with extra -D flags: scripts/build_variant.sh <tag> [-DFOO ...]
# Use: MULTICAL_HIP_LIB=$PWD/multical_amd/libmulticalhip_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p multical_amd/csrc/build
[ -f multical_amd/csrc/build/mc_host.o ] || g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -c -o multical_amd/csrc/build/mc_host.o multical_amd/csrc/mc_host.cpp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Xarch_host -march=x86-64-v3 "$@" -fPIC -c -o multical_amd/csrc/build/mc_kernels_$tag.o multical_amd/csrc/mc_kernels.hip
[ -f multical_amd/csrc/build/mc_driver.o ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xarch_host -march=x86-64-v3 -fPIC -c -o multical_amd/csrc/build/mc_driver.o multical_amd/csrc/mc_driver.hip
hipcc -shared -fPIC -o multical_amd/libmulticalhip_$tag.so multical_amd/csrc/build/mc_kernels_$tag.o multical_amd/csrc/build/mc_driver.o multical_amd/csrc/build/mc_host.o multical_amd/csrc/build/mc_schur.o
