"""diagnostic: cross-attention forward kernel time vs L, n_split, batch (events, back-to-back launches)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops

def bench(B, L, ns, iters=30):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, 8, 900, 32, generator=g).cuda(); k = torch.randn(B, 8, L, 32, generator=g).cuda(); v = torch.randn(B, 8, L, 32, generator=g).cuda()
    for _ in range(3): ops.mha_fwd(q, k, v, n_split=ns)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.mha_fwd(q, k, v, n_split=ns)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tf = 4.0 * B * 900 * L * 256 / (us * 1e-6) / 1e12
    return us, tf

for B, L in [(1, 4224), (4, 4224), (1, 24000), (8, 4224)]:
    for ns in [1, 2, 4, 6, 8, 11, 16, 22, 33]:
        if ns > L // 64: continue
        us, tf = bench(B, L, ns)
        print(f'B={B} L={L} n_split={ns:2d}: {us:8.1f} us/call (incl. combine + python)  {tf:6.1f} TFLOP/s', flush=True)
