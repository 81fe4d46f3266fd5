#!/bin/bash
# G1 at the shard size (1024 problems) and the sweep size (8192) for each library given (probe builds leave garbage rows: --no-check)
for lib in "$@"; do for B in 1024 8192; do
  RAPTOR_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload g1 --batch $B --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib B=$B', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done; done
