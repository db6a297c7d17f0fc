"""Print one step of a rocprofv3 kernel trace as a per-queue timeline: scripts/trace_print.py <kernel_trace.csv> [first_us last_us]"""
import csv, re, sys, glob
f = sys.argv[1]
if '*' in f: f = glob.glob(f)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'coords3d_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
# a step = from the gradient zero-fill (torch's elementwise fill in front of the forward) to the next one; without such a
# kernel in the trace (inference loops) fall back to a fixed offset in front of coords3d
fills = [i for i, r in enumerate(rows) if 'vectorized_elementwise_kernel' in r['Kernel_Name'] and 'Fill' in r['Kernel_Name']]
fa = [i for i in fills if i < a and a - i < 12]
fb = [i for i in fills if i < b and b - i < 12]
step = rows[fa[-1]:fb[-1]] if fa and fb else rows[a - 3:b - 3]
t0 = int(step[0]['Start_Timestamp'])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -1e9
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(petr_gemm_args.*', '', n); n = re.sub(r'\(Mha\w+\)', '', n)
    return n[:62]
qs = sorted({r['Queue_Id'] for r in step})
busy = {q: 0 for q in qs}
for r in step:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    busy[r['Queue_Id']] += e - s
    if s < lo or s > hi: continue
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Z']))
    col = qs.index(r['Queue_Id'])
    print(f"{s:8.1f} {e - s:7.1f} {'    ' * col}q{r['Queue_Id']} {str(g):12s} {short(r['Kernel_Name'])}")
print('span', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'busy', busy, 'launches', len(step))
