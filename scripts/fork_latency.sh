for w in g1 digit; do for mode in default nofork; do
  if [ $mode = nofork ]; then export RAPTOR_NO_FORK=1; else unset RAPTOR_NO_FORK; fi
  timeout -k 10 150 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-multistart --no-eval-h 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w $mode', 'HBM %.1f us' % (1e3*d['single_problem_eval_jac_g_latency_ms']), 'host %.1f us' % (1e3*d['single_problem_eval_jac_g_latency_ms_host']))"
done; done
