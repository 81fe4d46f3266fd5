for w in digit g1; do for b in 1 4 512 1024; do
  a=$(RAPTOR_NO_FORK=1 timeout -k 10 100 python bench.py --workload $w --batch $b --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
  f=$(RAPTOR_FORK_MAX=1000000 timeout -k 10 100 python bench.py --workload $w --batch $b --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
  echo "$w B=$b (E=$((b*14))): one stream $a ms, forked $f ms"
done; done
