"""Host enqueue time vs GPU time of the training step (is the step launch-bound on the host?).
   python scripts/host_time.py [workload] [dtype]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'c5'
dt = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
dev = torch.device('cuda:0')
head, metas, feats, g_cls, g_box = bench.build_workload(wl, 1, 900, dev, 0, need_grad=True)
head.train(True)
head.attn_dtype = dt
torch.manual_seed(1000)

def step(tm=None):
    t0 = time.perf_counter()
    head.zero_grad_flat()
    feats.grad = None
    t1 = time.perf_counter()
    out = head([feats], metas)
    t2 = time.perf_counter()
    torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls, g_box])
    t3 = time.perf_counter()
    if tm is not None:
        tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2

for _ in range(10):
    step()
torch.cuda.synchronize()
N = 40
tm = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(N):
    step(tm)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print(f'{wl} {dt}: host enqueue {th / N * 1e3:.3f} ms/step, with sync {tg / N * 1e3:.3f} ms/step; '
      f'zero_grad {tm[0] / N * 1e3:.3f}  forward {tm[1] / N * 1e3:.3f}  backward {tm[2] / N * 1e3:.3f}')
# host-only cost with the GPU idle in between (no back-pressure): sync after every step
tm = [0.0, 0.0, 0.0]
for _ in range(N):
    step(tm)
    torch.cuda.synchronize()
print(f'  per-call host time with a sync after each step: zero_grad {tm[0] / N * 1e3:.3f}  forward {tm[1] / N * 1e3:.3f}  backward {tm[2] / N * 1e3:.3f}')

# where does the forward's host time go when the device is kept busy?  wrap the phases
import types
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w
head._prepare = timed('_prepare', head._prepare)
head._upload = timed('  _upload', head._upload)
head._launch_forward_impl = timed('_launch_forward_impl', head._launch_forward_impl)
head._launch_backward_impl = timed('_launch_backward_impl', head._launch_backward_impl)
from petr_amd import _C
L = _C.lib()
class LW:
    def __init__(s, L): s.L = L
    def __getattr__(s, n):
        f = getattr(s.L, n)
        return timed('C.' + n, f) if n in ('petr_head_fwd', 'petr_head_bwd') else f
_lw = LW(L)
_C.lib = lambda: _lw
import petr_amd.petr_head as ph
ph._C.lib = _C.lib
torch.cuda.synchronize()
acc.clear()
for _ in range(N):
    step()
torch.cuda.synchronize()
print('  pipelined, per step (ms):', {k: round(v / N * 1e3, 3) for k, v in acc.items()})
