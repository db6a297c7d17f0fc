"""Times of the bf16 cross-attention forward / backward (and the fp32 ones) at one shape, with and without the
probability dropout: python scripts/attn16_time.py L [Q]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 900
g = torch.Generator().manual_seed(0)
q = torch.randn(1, Q, 256, generator=g).cuda().view(1, Q, 8, 32).permute(0, 2, 1, 3)
k = torch.randn(1, L, 256, generator=g).cuda()
v = torch.randn(1, L, 256, generator=g).cuda()
do = torch.randn(1, Q, 256, generator=g).cuda().view(1, Q, 8, 32).permute(0, 2, 1, 3)
kb, vb = ops.cast_bf16(k).view(1, L, 8, 32).permute(0, 2, 1, 3), ops.cast_bf16(v).view(1, L, 8, 32).permute(0, 2, 1, 3)
kf, vf = k.view(1, L, 8, 32).permute(0, 2, 1, 3), v.view(1, L, 8, 32).permute(0, 2, 1, 3)


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for drop in (None, (1, 2, 0.1)):
    o, lse = ops.mha_fwd_bf16(q, kb, vb, drop=drop)
    o = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    print(f'L={L} drop={drop is not None}: fwd16 {t(lambda: ops.mha_fwd_bf16(q, kb, vb, drop=drop)):.1f} us (incl. alloc)', end='')
    print(f'  bwd16 {t(lambda: ops.mha_bwd_bf16(q, kb, vb, o, do, lse, drop=drop)):.1f}', end='')
    print(f'  fwd32 {t(lambda: ops.mha_fwd(q, kf, vf, drop=drop)):.1f}  bwd32 ', end='')
    o32, lse32 = ops.mha_fwd(q, kf, vf, drop=drop)
    print(f'{t(lambda: ops.mha_bwd(q, kf, vf, o32, do, lse32, drop=drop)):.1f} us')
