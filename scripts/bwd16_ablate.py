"""Timing-only ablations of petr_mha_bwd_bf16 (libs built with -DPETR_BWD16_DIAG=<bit>, selected through PETR_HIP_LIB):
python scripts/bwd16_ablate.py L   -> one line: time without dropout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4224
g = torch.Generator().manual_seed(0)
q = torch.randn(1, 900, 256, generator=g).cuda().view(1, 900, 8, 32).permute(0, 2, 1, 3)
k = torch.randn(1, L, 256, generator=g).cuda(); v = torch.randn(1, L, 256, generator=g).cuda()
do = torch.randn(1, 900, 256, generator=g).cuda().view(1, 900, 8, 32).permute(0, 2, 1, 3)
kb, vb = ops.cast_bf16(k).view(1, L, 8, 32).permute(0, 2, 1, 3), ops.cast_bf16(v).view(1, L, 8, 32).permute(0, 2, 1, 3)
o, lse = ops.mha_fwd_bf16(q, kb, vb)
o = o.permute(0, 2, 1, 3).contiguous().view(1, 900, 8, 32).permute(0, 2, 1, 3)
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(os.environ.get('PETR_HIP_LIB', 'full').split('_')[-1], f'L={L}', f'{t(lambda: ops.mha_bwd_bf16(q, kb, vb, o, do, lse)):.1f} us',
      f'drop {t(lambda: ops.mha_bwd_bf16(q, kb, vb, o, do, lse, drop=(1, 2, 0.1))):.1f} us')
