#!/bin/bash
# usage (GPU box): tools/trace_stage.sh <tag> <stage> <marker> <per-step occurrences>  -> gpurun_out/<tag>_stage<stage>_kernels.txt
set -u
cd $GRAFT_REPO_ROOT
TAG=$1; ST=$2; MARK=$3; PER=${4:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_stage${ST}
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_full.py --stage $ST --steps 4 > $OUT/run.log 2>&1 )
python3 tools/step_kernels.py $OUT/trace "$MARK" $PER seq > gpurun_out/${TAG}_stage${ST}_kernels.txt 2>&1
rm -rf $OUT/trace
head -50 gpurun_out/${TAG}_stage${ST}_kernels.txt
