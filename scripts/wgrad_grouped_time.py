"""petr_wgrad_grouped alone: one decoder layer's parameter gradients at 900 rows (the item list of a c5 layer stage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 900
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
shapes = [(256, 2048), (2048, 256)] + [(256, 256)] * 5 + [(512, 256)]          # (M, N) of dw
items = [(r(K, m), r(K, n), torch.zeros(m, n).cuda(), torch.zeros(m).cuda(), 1) for m, n in shapes]
fl = sum(2.0 * K * m * n for m, n in shapes)
for _ in range(5): ops.wgrad_grouped(items)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.wgrad_grouped(items)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f'wgrad_grouped K={K}: {us:.1f} us per launch, {fl / us * 1e-6:.1f} TFLOP/s ({sum(m * n for m, n in shapes) // 4096} tiles)')
