#!/bin/bash
# timed-region-only bench lines for the four workloads, fp32 and bf16 (same box): scripts/quick_bench.sh [steps]
S=${1:-30}
for wl in c5 p4_1408 p4_1600 v2_800; do
  for dt in fp32 bf16; do
    python bench.py --workload $wl --dtype $dt --steps $S --warmup 8 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $dt', d['ms_per_step'], 'ms/step')"
  done
done
