#!/bin/bash
# kernel durations of the batched K/V projection: fp32, fp32 + bf16 store, bf16 route (rocprofv3 kernel stats)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for L in 4224 24000; do
  rm -rf /tmp/gb
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/gb -o p --output-format csv -- python3 scripts/gemm_bf16_time.py $L > /tmp/gb.log 2>&1
  f=$(find /tmp/gb -name '*kernel_stats.csv' 2>/dev/null | head -1)
  echo "L=$L"
  if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'gemm' in r['Name']: print(f\"{float(r['AverageNs'])/1e3:8.1f} us  x{r['Calls']:>4s}  {r['Name'][:90]}\")
"; else tail -5 /tmp/gb.log; fi
done
