"""petr_branch_fwd alone: the class / box branch over the 5 400 rows of the c5 head (6 levels x 900 queries)."""
import sys, torch
sys.path.insert(0, '.')
from petr_amd import ops
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
x = r(1, 5400, 256)
w1, w2, w3 = r(1, 256, 256) / 16, r(1, 256, 256) / 16, r(1, 10, 256) / 16
b1, b2, b3 = r(1, 256), r(1, 256), r(1, 10)
ln = (1 + 0.1 * r(1, 256), 0.1 * r(1, 256))
for name, l, save in (('class (LayerNorm)', ln, True), ('box (ReLU)', None, True), ('class, nothing saved', ln, False), ('box, nothing saved', None, False)):
    res = ops.branch_fwd(x, w1, b1, w2, b2, w3, b3, ln1=l, ln2=l, save=save)
    f = res['_launch']
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f'{name}: {us:.1f} us per launch, {2 * 5400 * 256 * 256 * 2 / us / 1e6:.1f} TFLOP/s')
