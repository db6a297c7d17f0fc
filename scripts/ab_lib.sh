#!/bin/bash
# same-box A/B of two builds of the library: scripts/ab_lib.sh <other.so> ["wl:dtype ..."]   (the in-tree build is "new")
OTHER=$1; CFGS=${2:-"c5:fp32 p4_1600:bf16"}
for round in 1 2; do for lib in prev new; do for cfg in $CFGS; do
  wl=${cfg%%:*}; dt=${cfg##*:}
  if [ $lib = prev ]; then export PETR_HIP_LIB=$OTHER; else unset PETR_HIP_LIB; fi
  python bench.py --workload $wl --dtype $dt --steps 40 --warmup 8 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $wl $dt', d['ms_per_step'])"
done; done; done
