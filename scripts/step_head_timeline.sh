#!/bin/bash
# timeline of the first ~450 us of one training step's forward (all queues): bash scripts/step_head_timeline.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sh1
rocprofv3 --kernel-trace -d /tmp/sh1 -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 "$@" > /tmp/sh1.log 2>&1
f=$(find /tmp/sh1 -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
copies = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('__amd_rocclr_copyBuffer')]
marks = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('posemb3d_kernel')]
m = marks[20]
i0 = max(i for i in copies if i < m)
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = max(int(r['End_Timestamp']) for r in rows[max(0, i0 - 40):i0])
print(f'previous step: last kernel ended {(prev_end - t0)/1e3:8.1f} us relative to the upload')
for r in rows[i0:i0 + 70]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if s > 450000: break
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    print(f"q{r['Queue_Id']}  {s/1e3:7.1f} -> {e/1e3:7.1f} us  ({(e-s)/1e3:5.1f})  {name}:{int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1)}x{r['Grid_Size_Z']}")
PY
