#!/bin/bash
# One PMC pass with the pipeline-activity counters (which unit a kernel keeps busy) for one bench workload.
# usage: scripts/pmc_pipes.sh <workload> <tag>      -> gpurun_out/pipes_<tag>.txt
set -o pipefail
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pipes_$TAG
mkdir -p $OUT
ARGS="$R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-latency --no-multistart --no-eval-h --no-spinup --no-check"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o a -- python3 $ARGS > $OUT/a.json 2> $OUT/a.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o b -- python3 $ARGS > $OUT/b.json 2> $OUT/b.err || exit 1
cd $R
python3 scripts/pmc_summary.py $(find $OUT -name "*counter_collection.csv") > $R/gpurun_out/pipes_$TAG.txt
