#!/bin/bash
# per-(kernel, grid) durations inside the bench's training steps (rocprofv3 kernel trace): bash scripts/step_breakdown.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/stepbd -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 "$@" > /tmp/stepbd.log 2>&1
f=$(find /tmp/stepbd -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    d[(name[:70], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in d.values())
for (k, gx, gy, gz), v in rows[:45]:
    v2 = sorted(v)
    print(f'{100*sum(v)/tot:5.1f}% n={len(v):5d} med={v2[len(v2)//2]/1e3:7.1f}us  grid=({gx},{gy},{gz}) {k}')
PY
