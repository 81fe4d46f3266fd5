#!/bin/bash
# GPU tests (optionally filtered) + short benches; prints one line per workload.  usage: quick_bench.sh "<pytest -k expr>" w1 w2 ...
K=$1; shift
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$K" > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for w in "$@"; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-secondary > gpurun_out/b_$w.json 2>gpurun_out/b_$w.err || { tail -5 gpurun_out/b_$w.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/b_$w.json'));print('$w', '%.4g se/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], ['%.3f' % v for v in d['kernels_ms'].values()], 'frac %.3f' % d['roofline']['frac'])"
done
