#!/bin/bash
# Write-request mix at the L2's memory side for the G1 evaluation at 8192 problems (fused form)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_wrreq_g1
mkdir -p $OUT
ARGS="$R/bench.py --workload g1 --batch 8192 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h --no-spinup"
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/fused -o w -- python3 $ARGS > $OUT/fused.json 2> $OUT/fused.err || { tail -5 $OUT/fused.err; exit 1; }
cd $R
python3 scripts/pmc_summary.py $(find $OUT/fused -name "*counter_collection.csv") | grep legged
