#!/bin/bash
# fp32 attention backward alone under explicit query-split plans (tiles per split, largest first): scripts/bwd32_plan_sweep.sh <shape> plan...
SH=$1; shift
for pl in "$@"; do
  echo -n "$SH plan $pl: "
  SHAPE=$SH PETR_MHA_BWD_PLAN=$pl python scripts/bwd32_time.py 2>&1 | grep "Q=900" | awk '{print $4, $5}'
done
