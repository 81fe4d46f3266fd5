#!/bin/bash
# Builds phase-stub variants of the library (kinova_kernel.h KIN_STOP) for the phase table of DESIGN.md 3.2; run locally
# (cross-compiles), then time them on the GPU box with: python scripts/ab_bench.py build/phases/lib_stop*.so raptor_amd/libraptor_hip.so
set -e
mkdir -p build/phases
for k in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKIN_STOP=$k -o build/phases/lib_stop$k.so raptor_amd/csrc/raptor_hip.hip &
done
wait
ls -la build/phases
