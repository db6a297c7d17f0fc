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
for name, l in (('class (LayerNorm)', ln), ('box (ReLU)', None)):
    f = lambda: ops.branch_fwd(x, w1, b1, w2, b2, w3, b3, ln1=l, ln2=l)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the wrapper transposes the weights per call: time the launch itself through a captured argument block
    from petr_amd import _C
    import ctypes as C
    res = f()
    n = 200
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call (wrapper: 2 transposes + cat + launch)')
