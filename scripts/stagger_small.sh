#!/bin/bash
# Staggered start of the fused core kernel at the multi-start shard size (G1, 1024 problems: 1.75 rounds of wavefronts) and for Digit 1024
for W in g1 digit; do for u in 0 1 2 3; do
  RAPTOR_STAGGER_UNITS=$u timeout -k 10 200 python bench.py --workload $W --batch 1024 --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$W units=$u', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done; done
