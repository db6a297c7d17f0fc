#!/bin/bash
# per-queue busy time over 10 timed training steps of bench.py (rocprofv3 kernel trace): which stream is the critical path,
# how full it is, and where its idle gaps are
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/su
rocprofv3 --kernel-trace -d /tmp/su -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 "$@" > /tmp/su.log 2>&1
f=$(find /tmp/su -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
marks = [int(r['Start_Timestamp']) for r in rows if r['Kernel_Name'].startswith('posemb3d_kernel')]   # one per forward
t0, t1 = marks[15], marks[25]          # ten of the thirty timed steps (ten warm-up steps come first)
rows = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
span = t1 - t0
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
q = collections.defaultdict(list)
for r in rows:
    q[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
print(f'10 steps = {span/1e6:.2f} ms, {len(rows)} launches ({len(rows)/10:.0f} per step)')
for k, iv in sorted(q.items(), key=lambda kv: -union(kv[1])):
    print(f'queue {k}: launches/step {len(iv)/10:5.0f}  busy {100*union(iv)/span:5.1f}%')
print(f'any queue busy {100*union([x for iv in q.values() for x in iv])/span:5.1f}%')
k0 = max(q, key=lambda k: union(q[k])); iv = sorted(q[k0])
names = {int(r['Start_Timestamp']): r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:34] + ':' + str(int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) for r in rows if r['Queue_Id'] == k0}
gaps = [(iv[i+1][0] - iv[i][1], names[iv[i][0]], names[iv[i+1][0]]) for i in range(len(iv)-1) if iv[i+1][0] > iv[i][1]]
print(f'busiest queue {k0}: idle {sum(g for g,_,_ in gaps)/1e7:.3f} ms/step in {len(gaps)/10:.0f} gaps/step; gaps > 10 us: {sum(g for g,_,_ in gaps if g > 10000)/1e7:.3f} ms/step')
pairs = collections.defaultdict(list)
for g, a, b in gaps:
    if g > 10000: pairs[(a, b)].append(g)
for (a, b), v in sorted(pairs.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f'  {sum(v)/1e7:6.3f} ms/step  n/step={len(v)/10:4.1f} mean {sum(v)/len(v)/1e3:6.1f} us   {a}  ->  {b}')
PY
