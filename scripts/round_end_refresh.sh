#!/bin/bash
# round-end evidence: GPU tests, the default bench line, its rocprofv3 kernel stats, and the other workloads' lines
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { tail -20 gpurun_out/final_gpu_tests.log; exit 1; }
tail -1 gpurun_out/final_gpu_tests.log
timeout -k 10 300 python3 bench.py > gpurun_out/final_bench_c5.json 2> gpurun_out/final_bench_c5.err || exit 2
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/final_stats -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline > /tmp/final_stats.log 2>&1) || exit 3
cp $(find /tmp/final_stats -name '*kernel_stats.csv' | head -1) gpurun_out/final_kernel_stats.csv
timeout -k 10 200 python3 bench.py --workload p4_1408 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final_bench_p4_1408.json 2>/dev/null || exit 4
timeout -k 10 200 python3 bench.py --workload v2_800 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final_bench_v2_800.json 2>/dev/null || exit 5
python3 -c "
import json
for n in ('c5','p4_1408','v2_800'):
    d=json.load(open('gpurun_out/final_bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['fwd_ms'], d['roofline']['frac'], (d.get('bf16_attention_inference') or {}).get('fwd_ms'))
"
