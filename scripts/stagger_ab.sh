#!/bin/bash
# Staggered start of the fused core kernel: a units per wave slot, b units per SIMD (units of ~3.5 us), Digit 14 x 4096
for ab in 8,2 12,2 12,0 12,3 16,2 6,2 8,0 10,1 0,3; do
  RAPTOR_STAGGER_AB=$ab timeout -k 10 200 python bench.py --workload digit --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('digit ab=$ab', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done
