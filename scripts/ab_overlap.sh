#!/bin/bash
# same-box A/B of the K/V-projection schedules (env toggles of head.hip), two rounds each
for round in 1 2; do
for cfg in "1 1" "0 0" "1 0" "0 1"; do
  set -- $cfg
  for wl in c5 p4_1600; do for dt in fp32 bf16; do
    PETR_KV_FWD_SPLIT=$1 PETR_KV_BWD_OVERLAP=$2 python bench.py --workload $wl --dtype $dt --steps 40 --warmup 8 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd_split=$1 bwd_overlap=$2 $wl $dt', d['ms_per_step'])"
  done; done
done; done
