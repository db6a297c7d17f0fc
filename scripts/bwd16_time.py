"""petr_mha_bwd_bf16 alone at the cross-attention shapes of the bench workloads, as the executor calls it (stored bf16 dK / dV,
packed dropout bits): python scripts/bwd16_time.py [drop_p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
g = torch.Generator().manual_seed(0)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Q, L in (('c5', 900, 4224), ('v2_800', 900, 12000), ('p4_1408', 900, 16896), ('p4_1600', 900, 24000)):
    mk = lambda n: torch.randn(1, n, 256, generator=g).cuda().view(1, n, 8, 32).permute(0, 2, 1, 3)
    q, do = mk(Q), mk(Q)
    kb = ops.cast_bf16(torch.randn(1, L, 256, generator=g).cuda()).view(1, L, 8, 32).permute(0, 2, 1, 3)
    vb = ops.cast_bf16(torch.randn(1, L, 256, generator=g).cuda()).view(1, L, 8, 32).permute(0, 2, 1, 3)
    drop = (1234, 3, p) if p > 0 else None
    bits = torch.zeros_like(ops.dropout_bits(drop, 8, Q, L)[1]) if drop else None
    o, lse = ops.mha_fwd_bf16(q, kb, vb, drop=drop, drop_bits=bits)
    o = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    us = t(lambda: ops.mha_bwd_bf16(q, kb, vb, o, do, lse, drop=drop, overwrite=True, dkv_bf16=True, drop_bits=bits))
    fl = 10.0 * Q * L * 256
    print(f'{name:8s} L={L:6d}: {us:8.1f} us (incl. the wrapper\'s allocations)  {fl / us * 1e-6:7.1f} TFLOP/s', flush=True)
