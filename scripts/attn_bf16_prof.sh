#!/bin/bash
# kernel-only durations of the bf16 attention per workload and split count (rocprofv3 kernel trace):
#   bash scripts/attn_bf16_prof.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in ${WLS:-c5 p4_1600}; do
  WL=$wl SPLITS=${SPLITS:-1,2,3,4,5,6,8,12} rocprofv3 --kernel-trace --stats -d /tmp/bf16prof_$wl -o p --output-format csv -- python3 $R/scripts/attn_bf16_time.py > /dev/null 2>&1
  f=$(find /tmp/bf16prof_$wl -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $wl <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'][:58], int(r['Grid_Size_X']) // 512)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (k, wgs), v in sorted(d.items()):
    if 'bf16_kernel' in k or 'combine' in k or 'mha_fwd_kernel' in k:
        v = sorted(v)
        print(f'{sys.argv[2]:8s} {k:58s} wgs={wgs:6d} n={len(v):5d} median={v[len(v)//2]/1e3:7.1f} us')
PY
done
