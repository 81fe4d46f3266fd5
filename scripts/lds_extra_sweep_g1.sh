#!/bin/bash
# Fewer resident wavefronts per CU for the fused core kernel, G1 / Talos: 8 / 7 / 6 per CU at 1024, 2048, 4096 and 8192 problems
for B in 1024 2048 4096 8192; do for x in 0 2900 6800; do
  RAPTOR_LEG_LDS_EXTRA=$x timeout -k 10 200 python bench.py --workload g1 --batch $B --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('g1 B=$B extra=$x', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done; done
