"""petr_mha_bwd_bf16 time against the number of query tiles at fixed L (separates the per-tile cost from the fixed cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4224
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
k = torch.randn(1, L, 256, generator=g).cuda(); v = torch.randn(1, L, 256, generator=g).cuda()
kb, vb = ops.cast_bf16(k).view(1, L, 8, 32).permute(0, 2, 1, 3), ops.cast_bf16(v).view(1, L, 8, 32).permute(0, 2, 1, 3)
for Q in (32, 64, 128, 256, 448, 900, 1792):
    q = torch.randn(1, Q, 256, generator=g).cuda().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    do = torch.randn(1, Q, 256, generator=g).cuda().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    o, lse = ops.mha_fwd_bf16(q, kb, vb)
    o = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    os.environ['PETR_MHA_BWD16_QSPLITS'] = '1'
    print(f'L={L} Q={Q:5d} tiles {(Q+31)//32:3d}: {t(lambda: ops.mha_bwd_bf16(q, kb, vb, o, do, lse)):7.1f} us')
