#!/bin/bash
# Kernel trace + PMC passes for one bench workload.  usage: scripts/profile_round.sh <workload> <tag>
# Each rocprofv3 pass is its own run (counters are never combined with traces), program directly after `--`.
set -o pipefail
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="$R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ARGS --no-spinup > $OUT/w.json 2> $OUT/w.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ARGS --no-spinup > $OUT/f.json 2> $OUT/f.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq -o s -- python3 $ARGS --no-spinup > $OUT/s.json 2> $OUT/s.err || exit 1
cd $R
python3 scripts/pmc_summary.py $(find $OUT -name "*counter_collection.csv") > $OUT/pmc_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv") $OUT/kernel_stats.csv
