#!/bin/bash
# Staggered start of the fused core kernel, G1 at the sweep size (8192 problems: 14 rounds of wavefronts)
for u in 0 1 2 3 4 6; do
  RAPTOR_STAGGER_UNITS=$u timeout -k 10 200 python bench.py --workload g1 --batch 8192 --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('g1 8192 units=$u', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done
