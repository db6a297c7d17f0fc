#!/bin/bash
# diagnostic: PMC passes over the bf16 cross-attention forward kernel (scripts/attn_one_bf16.py B L ns); prints per-counter sums
export TMPDIR=/tmp
ARGS="${@:-1 24000 0}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmcbf$i -o run --output-format csv -- python3 scripts/attn_one_bf16.py $ARGS > /tmp/pmcbf$i.log 2>&1
  f=$(find /tmp/pmcbf$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'mha_fwd_bf16' in r['Kernel_Name']:
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    print(f'{k:32s} per-launch {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
done
