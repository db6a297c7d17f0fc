#!/bin/bash
# diagnostic: step time vs the number of side streams
for n in 1 2 3 4; do
  PETR_AMD_SIDE_STREAMS=$n timeout -k 10 100 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/side_$n.json || exit 1
  python3 -c "import json; d=json.load(open('/tmp/side_$n.json')); print('side streams $n: ms/step', d['ms_per_step'], 'fwd_ms', d['fwd_ms'])"
done
