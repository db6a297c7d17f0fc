#!/bin/bash
# HBM traffic of the attention kernels per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (counters
# only, never with a trace domain; MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots") over the timed-only bench command,
# for the headline (c5 fp32) and for p4_1600 bf16.  Writes gpurun_out/r03_pmc_traffic.json (copied to profiles/ by hand).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "c5 fp32" "p4_1600 bf16"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$1_$2_$c -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --dtype $2 --steps 4 --warmup 2 --timed-only > $GRAFT_REPO_ROOT/gpurun_out/pmc_$1_$2_$c.log 2>&1)
    echo "pass $1 $2 $c done"
  done
done
python3 scripts/pmc_traffic_parse.py r03
