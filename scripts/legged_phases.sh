#!/bin/bash
# Phase-stub variants (1: phase 0, 2: A, 3: B + C, 4: E, 5: D, 6: F1 loop, 7: F1 outputs; the full kernel adds F2)
# Phase-stub variants of the library for the legged core kernel (legged_kernel.h LEG_STOP); time with
#   scripts/ab_libs.sh digit build/r3/lib_leg*.so raptor_amd/libraptor_hip.so
set -e
mkdir -p build/r3
for k in 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLEG_STOP=$k -o build/r3/lib_leg$k.so raptor_amd/csrc/raptor_hip.hip &
done
wait
