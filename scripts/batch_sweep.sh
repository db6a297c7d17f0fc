#!/bin/bash
# diagnostic: per-GPU batch size vs throughput (the reference trains with samples_per_gpu = 1)
for b in 1 2 4 8; do
  timeout -k 10 200 python3 bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > /tmp/bs_$b.json || exit 1
  python3 -c "import json; d=json.load(open('/tmp/bs_$b.json')); print('batch $b: samples/s', d['value'], 'ms/step', d['ms_per_step'], 'fwd samples/s', d['fwd_samples_per_s'], 'roofline frac', d['roofline']['frac'])"
done
