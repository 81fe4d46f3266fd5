#!/bin/bash
# Pipeline-activity counters of the eval_h kernels (scripts/ab_hess.py on one library).  usage: scripts/pmc_hess.sh <lib.so> <tag>
set -o pipefail
LIB=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/hess_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o a -- python3 $R/scripts/ab_hess.py $R/$LIB > $OUT/a.log 2> $OUT/a.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o b -- python3 $R/scripts/ab_hess.py $R/$LIB > $OUT/b.log 2> $OUT/b.err || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $R/scripts/ab_hess.py $R/$LIB > $OUT/t.log 2> $OUT/t.err || exit 1
cd $R
python3 scripts/pmc_summary.py $(find $OUT -name "*counter_collection.csv") > $R/gpurun_out/hess_$TAG.txt
cp $(find $OUT/t -name "*kernel_stats.csv") $R/gpurun_out/hess_${TAG}_kernel_stats.csv
