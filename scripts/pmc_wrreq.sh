#!/bin/bash
# Write-request mix at the L2's memory side for the Digit step, fused form against the two-kernel form:
# TCC_EA0_WRREQ (all) and TCC_EA0_WRREQ_64B (full 64-byte requests): the difference is partial (32-byte) writes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_wrreq
mkdir -p $OUT
ARGS="$R/bench.py --workload digit --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h --no-spinup"
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/fused -o w -- python3 $ARGS > $OUT/fused.json 2> $OUT/fused.err || { tail -5 $OUT/fused.err; exit 1; }
export RAPTOR_NO_FUSE=1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/two -o w -- python3 $ARGS > $OUT/two.json 2> $OUT/two.err || { tail -5 $OUT/two.err; exit 1; }
cd $R
python3 scripts/pmc_summary.py $(find $OUT/fused -name "*counter_collection.csv") > $OUT/fused_summary.txt
python3 scripts/pmc_summary.py $(find $OUT/two -name "*counter_collection.csv") > $OUT/two_summary.txt
cat $OUT/fused_summary.txt $OUT/two_summary.txt
