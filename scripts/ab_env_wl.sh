#!/bin/bash
# diagnostic: ab_env.sh for another workload:  ab_env_wl.sh <workload> VAR v1 v2 ...
wl=$1; var=$2; shift; shift
for round in 1 2; do
for v in "$@"; do
  env $var=$v timeout -k 10 200 python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > /tmp/abe.json || exit 1
  python3 -c "import json; d=json.load(open('/tmp/abe.json')); k=d['kernels']; print('$wl $var=$v', 'ms/step', d['ms_per_step'], 'bwd_cross', k['mha_bwd_cross']['mean_us'], 'bwd_self', k['mha_bwd_self']['mean_us'])"
done
done
