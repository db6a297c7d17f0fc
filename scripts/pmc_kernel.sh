#!/bin/bash
# PMC passes (counters only, no tracing) over one kernel: bash scripts/pmc_kernel.sh <kernel-substring> <python script> [args...]
# prints per-launch means of each counter for the launches whose kernel name contains the substring
export TMPDIR=/tmp
pat=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmck$i
  rocprofv3 --pmc $set -d /tmp/pmck$i -o run --output-format csv -- python3 "$@" > /tmp/pmck$i.log 2>&1
  f=$(find /tmp/pmck$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    print(f'{k:32s} {sum(v)/len(v):16.0f}  (mean of {len(v)} launches)')
PY
done
