#!/bin/bash
# Builds a variant of libraptor_hip.so for A/B runs: scripts/build_variant.sh <out.so> [git-rev|WORK] [extra hipcc flags...]
# WORK (default) = the working tree's sources; a git revision = that revision's raptor_amd/csrc + include (exported to build/src_<rev>).
set -e
OUT=$1; REV=${2:-WORK}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$REV" = "WORK" ]; then
  SRC=$ROOT/raptor_amd/csrc
else
  D=$ROOT/build/src_$REV
  rm -rf $D && mkdir -p $D/raptor_amd $D/include
  git -C $ROOT archive $REV raptor_amd/csrc include | tar -x -C $D
  SRC=$D/raptor_amd/csrc
fi
cd $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o $ROOT/$OUT raptor_hip.hip
echo built $OUT
