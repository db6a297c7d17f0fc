#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the step's kernels, separate passes (MI355X_MICROARCH.md HBM section; never with sys-trace)
set -e
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d gpurun_out/pmc_$c -o run --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "pass $c done"
done
