"""petr_mha_bwd (fp32) kernel time at the cross-attention shapes of the bench workloads, buffers reused (no fills in the loop).
   python scripts/bwd32_time.py [drop_p]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops, _C
from petr_amd.ops import _ptr, _bhsd, _stream
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
Lb = _C.lib()
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
SHAPES = (('self', 900, 900), ('c5', 900, 4224), ('p4_1408', 900, 16896), ('v2_800', 900, 12000), ('p4_1600', 900, 24000))
ONLY = os.environ.get('SHAPE')          # SHAPE=c5: that shape alone (PMC passes)
for name, Q, L in SHAPES:
    if ONLY and name != ONLY: continue
    mk = lambda n: torch.randn(1, n, 256, generator=g).cuda().view(1, n, 8, 32).permute(0, 2, 1, 3)
    q, do, k, v = mk(Q), mk(Q), mk(L), mk(L)
    drop = (1234, 3, p) if p > 0 else None
    o, lse = ops.mha_fwd(q, k, v, drop=drop)
    o = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    dq = torch.zeros(1, Q, 256, device='cuda').view(1, Q, 8, 32).permute(0, 2, 1, 3)
    dk = torch.zeros(1, L, 256, device='cuda').view(1, L, 8, 32).permute(0, 2, 1, 3)
    dv = torch.zeros(1, L, 256, device='cuda').view(1, L, 8, 32).permute(0, 2, 1, 3)
    nbytes = Lb.petr_mha_bwd_workspace_bytes(1, 8, Q, L)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device='cuda')
    bits = ops.dropout_bits(drop, 8, Q, L)[1] if drop is not None and os.environ.get('BITS', '1') != '0' else None   # key-major words
    a = _C.MhaBwdArgs(_ptr(q), *_bhsd(q), _ptr(k), *_bhsd(k), _ptr(v), *_bhsd(v), _ptr(o), *_bhsd(o), _ptr(do), *_bhsd(do),
                      _ptr(lse), None, _ptr(dq), *_bhsd(dq), _ptr(dk), *_bhsd(dk), _ptr(dv), *_bhsd(dv), 1, 8, Q, L,
                      32 ** -0.5, _ptr(ws), nbytes, _C.dropout(drop), _ptr(bits) if bits is not None else None)
    us = t(lambda: _C.check(Lb.petr_mha_bwd(C.byref(a), _stream()), 'petr_mha_bwd'))
    fl = 5 * 2.0 * Q * L * 256
    print(f'{name:8s} Q={Q} L={L:6d}: {us:8.1f} us  {fl / us * 1e-6:6.1f} TFLOP/s ({fl / us * 1e-6 / 157.3:.3f} of the fp32 MFMA peak)', flush=True)
