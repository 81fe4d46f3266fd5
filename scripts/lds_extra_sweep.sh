#!/bin/bash
# Fewer resident wavefronts per CU for the fused core kernel (extra dynamic LDS per workgroup): 8 / 7 / 6 / 5 / 4 per CU
for x in 0 2900 6800 12288 20480; do
  RAPTOR_LEG_LDS_EXTRA=$x timeout -k 10 200 python bench.py --workload digit --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('extra=$x', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done
