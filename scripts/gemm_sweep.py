"""diagnostic: GEMM kernel time vs shape (events, back-to-back launches)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops

def bench(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

g = torch.Generator().manual_seed(0)
print('linear y = x @ w.T  (A,B K-contiguous)')
for M, N, K in [(900, 256, 32), (900, 256, 64), (900, 256, 128), (900, 256, 256), (900, 256, 1024), (900, 256, 2048),
                (900, 768, 256), (900, 2048, 256), (5400, 256, 256), (4224, 256, 256), (4224, 1024, 192), (4224, 256, 1024),
                (4224, 1024, 384), (25344, 256, 256), (16384, 4096, 1024)]:
    x = torch.randn(M, K, generator=g).cuda(); w = torch.randn(N, K, generator=g).cuda(); b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N).cuda()
    us = bench(lambda: ops.linear(x, w, b, out=out))
    print(f'  M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s', flush=True)
print('empty-ish kernel launch floor (fill 1K floats)')
t = torch.empty(1024).cuda()
import ctypes as C
from petr_amd import _C
L = _C.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f'  petr_fill: {bench(lambda: L.petr_fill(C.c_void_p(t.data_ptr()), 0.0, 1024, s)):.2f} us')
