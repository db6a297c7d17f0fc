"""Attention kernels: forward leaving the packed dropout mask + backward reading it, against hashing in both (and no dropout)."""
import os, sys
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
drop = (1234, 3, 0.1)
for L in (4224, 24000):
    Q = 900
    mk = lambda n: torch.randn(1, n, 256, generator=g).cuda().view(1, n, 8, 32).permute(0, 2, 1, 3)
    q, do, k, v = mk(Q), mk(Q), mk(L), mk(L)
    kb = ops.cast_bf16(k.permute(0, 2, 1, 3).contiguous().view(1, L, 256)).view(1, L, 8, 32).permute(0, 2, 1, 3)
    vb = ops.cast_bf16(v.permute(0, 2, 1, 3).contiguous().view(1, L, 256)).view(1, L, 8, 32).permute(0, 2, 1, 3)
    bq, bk = ops.dropout_bits(drop, 8, Q, L)
    print(f'L={L}: dropout_bits kernel {t(lambda: ops.dropout_bits(drop, 8, Q, L)):.1f} us (incl. two allocations)')
    bout = torch.zeros_like(bk)
    for name, dr, fb, bb in (('no dropout', None, None, None), ('hash', drop, None, None), ('bits', drop, bout, bout)):
        o, lse = ops.mha_fwd(q, k, v, drop=dr, drop_bits=fb)
        of = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
        f32 = t(lambda: ops.mha_fwd(q, k, v, drop=dr, drop_bits=fb))
        b32 = t(lambda: ops.mha_bwd(q, k, v, of, do, lse, drop=dr, drop_bits=bb))
        o, lse = ops.mha_fwd_bf16(q, kb, vb, drop=dr, drop_bits=fb)
        of = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
        f16 = t(lambda: ops.mha_fwd_bf16(q, kb, vb, drop=dr, drop_bits=fb))
        b16 = t(lambda: ops.mha_bwd_bf16(q, kb, vb, of, do, lse, drop=dr, drop_bits=bb, overwrite=True))
        print(f'  {name:11s}: fp32 fwd {f32:6.1f} bwd {b32:6.1f} | bf16 fwd {f16:6.1f} bwd {b16:6.1f} us   (wrappers allocate / fill outputs)', flush=True)
