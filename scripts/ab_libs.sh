#!/bin/bash
# A/B of library builds on one workload: scripts/ab_libs.sh <workload> lib1.so lib2.so ...  (kernel times per library)
W=$1; shift
for lib in "$@"; do
  RAPTOR_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '%.4g' % d['value'], ['%.3f' % v for v in d['kernels_ms'].values()])" || exit 1
done
