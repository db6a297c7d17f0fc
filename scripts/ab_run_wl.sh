#!/bin/bash
# diagnostic: like ab_run.sh for another workload:  ab_run_wl.sh <workload> <steps> tag...
wl=$1; steps=$2; shift; shift
for round in 1 2; do
for tag in "$@"; do
  PETR_HIP_LIB=$PWD/petr_amd/lib/libpetr_hip_$tag.so timeout -k 10 200 python3 bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null > /tmp/ab_$tag.json || exit 1
  python3 -c "import json; d=json.load(open('/tmp/ab_$tag.json')); k=d['kernels']; print('$wl $tag', 'ms/step', d['ms_per_step'], 'fwd_ms', d['fwd_ms'], 'fwd_cross', k['mha_fwd_cross']['mean_us'], 'bwd_cross', k['mha_bwd_cross']['mean_us'])"
done
done
