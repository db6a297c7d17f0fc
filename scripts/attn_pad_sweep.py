"""diagnostic: cross-attention forward time vs the occupancy cap (dynamic-LDS pad) and n_split.
Run one (pad) per process: PETR_MHA_FWD_LDS_PAD=<bytes> python3 scripts/attn_pad_sweep.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops

def bench(B, L, ns, iters=40):
    global DYN
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, 8, 900, 32, generator=g).cuda(); k = torch.randn(B, 8, L, 32, generator=g).cuda(); v = torch.randn(B, 8, L, 32, generator=g).cuda()
    for _ in range(5): ops.mha_fwd(q, k, v, n_split=ns, dynamic=DYN)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.mha_fwd(q, k, v, n_split=ns, dynamic=DYN)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

DYN = os.environ.get('DYN', '1') == '1'
pad = os.environ.get('PETR_MHA_FWD_LDS_PAD', '0')
for B, L in [(1, 4224), (1, 24000)]:
    for ns in [6, 8, 11, 16, 22]:
        us = bench(B, L, ns)
        tf = 4.0 * B * 900 * L * 256 / (us * 1e-6) / 1e12
        print(f'dyn={DYN} pad={pad} B={B} L={L} n_split={ns:2d}: {us:8.1f} us/call (incl. combine)  {tf:6.1f} TFLOP/s', flush=True)
