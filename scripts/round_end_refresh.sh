#!/bin/bash
# round-end evidence: GPU tests, the default bench line, its rocprofv3 kernel stats, the p4_1600 bf16 step's kernel stats
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { tail -20 gpurun_out/final_gpu_tests.log; exit 1; }
tail -1 gpurun_out/final_gpu_tests.log
timeout -k 10 400 python3 bench.py > gpurun_out/final_bench_c5.json 2> gpurun_out/final_bench_c5.err || exit 2
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/final_stats -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline > /tmp/final_stats.log 2>&1) || exit 3
cp $(find /tmp/final_stats -name '*kernel_stats.csv' | head -1) gpurun_out/final_kernel_stats.csv
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/final_stats16 -o p --output-format csv -- python3 $R/bench.py --workload p4_1600 --dtype bf16 --steps 20 --warmup 5 --timed-only > /tmp/final_stats16.log 2>&1) || exit 4
cp $(find /tmp/final_stats16 -name '*kernel_stats.csv' | head -1) gpurun_out/final_kernel_stats_p4_1600_bf16.csv
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/final_stats32 -o p --output-format csv -- python3 $R/bench.py --workload c5 --steps 40 --warmup 10 --timed-only > /tmp/final_stats32.log 2>&1) || exit 5
cp $(find /tmp/final_stats32 -name '*kernel_stats.csv' | head -1) gpurun_out/final_kernel_stats_c5_timed_only.csv
python3 -c "
import json
d=json.load(open('gpurun_out/final_bench_c5.json'))
print('c5', d['value'], d['ms_per_step'], d['fwd_ms'], d['roofline']['frac'], d['bf16']['ms_per_step'])
for n,w in d['workloads'].items(): print(n, w['fp32']['ms_per_step'], w['bf16']['ms_per_step'], w['bf16']['mha_bwd_cross'])
"
# one step of the headline and of the largest bf16 workload as per-queue timelines
bash scripts/trace_step.sh c5 fp32 > /dev/null && python3 scripts/trace_print.py "gpurun_out/trace_c5_fp32/*/*kernel_trace.csv" > gpurun_out/final_trace_c5_fp32_step.txt
bash scripts/trace_step.sh p4_1600 bf16 > /dev/null && python3 scripts/trace_print.py "gpurun_out/trace_p4_1600_bf16/*/*kernel_trace.csv" > gpurun_out/final_trace_p4_1600_bf16_step.txt
tail -1 gpurun_out/final_trace_c5_fp32_step.txt; tail -1 gpurun_out/final_trace_p4_1600_bf16_step.txt
