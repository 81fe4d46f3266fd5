run() { "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('secondary',{}).get('digit',d)
print('%.1f us HBM, %.1f us host' % (1e3*s['single_problem_eval_jac_g_latency_ms'], 1e3*s['single_problem_eval_jac_g_latency_ms_host']))"; }
echo -n "kinova+secondary: "; run timeout -k 10 200 python bench.py --no-multistart --no-eval-h --no-cpu-baseline
echo -n "kinova+secondary, no fork: "; RAPTOR_NO_FORK=1 run timeout -k 10 200 python bench.py --no-multistart --no-eval-h --no-cpu-baseline
echo -n "digit alone: "; run timeout -k 10 200 python bench.py --workload digit --no-multistart --no-eval-h --no-cpu-baseline --no-secondary
echo -n "kinova+secondary with cpu baseline: "; run timeout -k 10 300 python bench.py --no-multistart --no-eval-h --cpu-seconds 3
