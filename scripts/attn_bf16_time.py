"""Times the bf16-K/V cross-attention forward next to the fp32 one (same box, same inputs): python3 scripts/attn_bf16_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petr_amd import ops  # noqa: E402


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


WL = os.environ.get('WL')
for name, L in (('c5', 4224), ('p4_1408', 16896), ('p4_1600', 24000), ('v2_800', 12000)):
    if WL and name != WL:
        continue
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(1, 8, n, 32, generator=g).cuda() for n in (900, L, L))
    kb, vb = ops.cast_bf16(k), ops.cast_bf16(v)
    flops = 4 * 900 * L * 256
    t32 = timeit(lambda: ops.mha_fwd(q, k, v))
    line = f'{name:8s} L={L:6d} fp32 {t32:7.1f} us ({flops / t32 / 1e6:6.1f} TF)'
    for ns in (() if WL else (0,)) + tuple(int(s) for s in os.environ.get('SPLITS', '4,8,12,16').split(',')):
        tb = timeit(lambda: ops.mha_fwd_bf16(q, kb, vb, n_split=ns))
        line += f' | bf16 ns={ns}: {tb:6.1f} us ({flops / tb / 1e6:6.1f} TF)'
    tc = timeit(lambda: ops.cast_bf16(k))
    print(line + f' | cast {tc:5.1f} us', flush=True)
