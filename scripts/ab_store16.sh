#!/bin/bash
# same-box A/B of bf16 storage of dK/dV (PETR_DKV16) and of the position-embedding hiddens (PETR_HID16), bf16 step
for round in 1 2; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v; for wl in p4_1600 v2_800 c5; do
  PETR_DKV16=$1 PETR_HID16=$2 python bench.py --workload $wl --dtype bf16 --steps 40 --warmup 8 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DKV16=$1 HID16=$2 $wl bf16', d['ms_per_step'])"
done; done; done
