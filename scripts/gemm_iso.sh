#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace -d /tmp/gemmiso -o p --output-format csv -- python3 scripts/gemm_iso.py > /tmp/gemmiso.log 2>&1
f=$(find /tmp/gemmiso -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    d[(name[:75], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Z']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (k, gx, gz), v in sorted(d.items()):
    v2 = sorted(v)
    print(f'n={len(v):4d} med={v2[len(v2)//2]/1e3:7.1f}us min={v2[0]/1e3:6.1f} grid=({gx},{gz}) {k}')
PY
