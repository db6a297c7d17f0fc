#!/bin/bash
# kernel-only durations (rocprofv3 kernel trace) of scripts/ffn_fused_time.py: the fused FFN forward per grid size against the
# two contractions.  PETR_FFN_* environment toggles pass through.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ffnprof
rocprofv3 --kernel-trace -d /tmp/ffnprof -o p --output-format csv -- python3 $R/scripts/ffn_fused_time.py > /dev/null 2>&1
f=$(find /tmp/ffnprof -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'][:50], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (k, wgs, gy), v in sorted(d.items()):
    if 'ffn' in k or 'gemm' in k:
        v = sorted(v)
        print(f'{k:50s} grid=({wgs},{gy}) n={len(v):5d} median={v[len(v)//2]/1e3:7.1f} us')
PY
