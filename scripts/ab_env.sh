#!/bin/bash
# same-box A/B of one environment toggle over the timed-only bench: scripts/ab_env.sh VAR "v1 v2" ["wl:dtype ..."]
VAR=$1; VALS=$2; CFGS=${3:-"c5:fp32 p4_1600:bf16"}
for round in 1 2; do for v in $VALS; do for cfg in $CFGS; do
  wl=${cfg%%:*}; dt=${cfg##*:}
  env $VAR=$v python bench.py --workload $wl --dtype $dt --steps 40 --warmup 8 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v $wl $dt', d['ms_per_step'])"
done; done; done
