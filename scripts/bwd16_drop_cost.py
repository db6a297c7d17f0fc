"""petr_mha_bwd_bf16 / petr_mha_fwd_bf16 with and without the dropout mask: what regenerating the mask costs."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for L in (4224, 24000):
    Q = 900
    mk = lambda n: torch.randn(1, n, 256, generator=g).cuda().view(1, n, 8, 32).permute(0, 2, 1, 3)
    q, do, k, v = mk(Q), mk(Q), mk(L), mk(L)
    kb, vb = ops.cast_bf16(k.permute(0, 2, 1, 3).contiguous().view(1, L, 256)).view(1, L, 8, 32).permute(0, 2, 1, 3), \
             ops.cast_bf16(v.permute(0, 2, 1, 3).contiguous().view(1, L, 256)).view(1, L, 8, 32).permute(0, 2, 1, 3)
    for drop in (None, (1234, 3, 0.1)):
        o, lse = ops.mha_fwd_bf16(q, kb, vb, drop=drop)
        o = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
        tf = t(lambda: ops.mha_fwd_bf16(q, kb, vb, drop=drop))
        tb = t(lambda: ops.mha_bwd_bf16(q, kb, vb, o, do, lse, drop=drop, overwrite=True))
        print(f'L={L} drop={drop is not None}: fwd {tf:.1f} us, bwd (incl. 3 fills) {tb:.1f} us', flush=True)
