#!/bin/bash
# per-launch kernel trace of a few steps: scripts/trace_step.sh <workload> <dtype>  ->  gpurun_out/trace_<wl>_<dt>/
WL=${1:-c5}; DT=${2:-fp32}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_${WL}_${DT}
rm -rf $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload $WL --dtype $DT --steps 4 --warmup 3 --timed-only > $OUT.log 2>&1
find $OUT -name "*kernel_trace.csv" | head -1
