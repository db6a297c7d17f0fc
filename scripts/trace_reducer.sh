cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_c5_reducer
rm -rf $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload c5 --steps 4 --warmup 3 --timed-only --force-reducer > $OUT.log 2>&1
python scripts/trace_print.py "$OUT/**/*kernel_trace.csv" > gpurun_out/r3_trace_reducer.txt 2>&1
tail -1 gpurun_out/r3_trace_reducer.txt
