#!/bin/bash
# diagnostic: time the step under different values of one environment variable on ONE box:  ab_env.sh VAR v1 v2 ...
var=$1; shift
for round in 1 2; do
for v in "$@"; do
  env $var=$v timeout -k 10 100 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/abe.json || exit 1
  python3 -c "import json; d=json.load(open('/tmp/abe.json')); k=d['kernels']; print('$var=$v', 'ms/step', d['ms_per_step'], 'fwd_ms', d['fwd_ms'], 'bwd_cross', k['mha_bwd_cross']['mean_us'], 'bwd_self', k['mha_bwd_self']['mean_us'])"
done
done
