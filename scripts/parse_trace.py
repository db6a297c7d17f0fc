"""print mean GPU duration per consecutive run of identical (kernel, grid) launches in a rocprofv3 kernel trace"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
runs = []
for r in rows:
    key = (r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60],
           int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Workgroup_Size_X']), int(r['Grid_Size_Z']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append((key, [d]))
for key, ds in runs:
    if len(ds) >= 10:
        ds2 = sorted(ds)[2:-2]
        print(f'{sum(ds2)/len(ds2):8.1f} us  n={len(ds):3d}  wg={key[1]:5d} x{key[2]:4d} z={key[3]:3d}  {key[0]}')
