#!/bin/bash
# G1 at the multi-start shard size (1024 problems) and at the full sweep size (8192), fused core kernel against the two-kernel form
for B in 1024 8192; do
  for nf in 0 1; do
    if [ $nf = 1 ]; then export RAPTOR_NO_FUSE=1; else unset RAPTOR_NO_FUSE; fi
    timeout -k 10 200 python bench.py --workload g1 --batch $B --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$B nofuse=$nf', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
  done
done
