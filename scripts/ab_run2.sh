#!/bin/bash
# like ab_run.sh, also printing the self-attention forward kernel
for round in 1 2; do
for tag in "$@"; do
  PETR_HIP_LIB=$PWD/petr_amd/lib/libpetr_hip_$tag.so timeout -k 10 100 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/ab_$tag.json || exit 1
  python3 -c "import json; d=json.load(open('/tmp/ab_$tag.json')); k=d['kernels']; print('$tag', 'ms/step', d['ms_per_step'], 'fwd_ms', d['fwd_ms'], 'fwd_self', k['mha_fwd_self']['mean_us'], 'fwd_cross', k['mha_fwd_cross']['mean_us'])"
done
done
