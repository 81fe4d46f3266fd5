#!/bin/bash
# Phase-stub variants of the library for the legged core kernel (legged_kernel.h LEG_STOP); time with
#   scripts/ab_libs.sh digit build/phases/lib_leg*.so raptor_amd/libraptor_hip.so
set -e
mkdir -p build/phases
for k in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLEG_STOP=$k -o build/phases/lib_leg$k.so raptor_amd/csrc/raptor_hip.hip &
done
wait
