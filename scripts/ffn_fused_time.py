"""Fused FFN forward (petr_ffn_fwd) against the two contractions it replaces, alone on the device (900 x 256 -> 2048 -> 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops, _C
g = torch.Generator().manual_seed(0)
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in [int(v) for v in os.environ.get("MS", "900,1800").split(",")]:
    C, F = 256, 2048
    x = torch.randn(M, C, generator=g).cuda(); w1 = (torch.randn(F, C, generator=g) * 0.06).cuda(); b1 = torch.randn(F, generator=g).cuda()
    w2 = (torch.randn(C, F, generator=g) * 0.03).cuda()
    hid = torch.empty(M, F, device='cuda'); part = torch.empty(8, M, C, device='cuda')
    drop = (1, 2, 0.1)
    wt = (w1.t().contiguous(), w2.t().contiguous())
    def two(dr):
        ops.gemm_raw(a=x, lda=C, a_kcontig=1, b=w1, ldb=C, b_kcontig=1, c=hid, ldc=F, bias=b1, M=M, N=F, K=C, nb0=1, nb1=1, flags=_C.GEMM_RELU, alpha=1.0, drop=dr)
        ops.gemm_raw(a=hid, lda=F, a_kcontig=1, b=w2, ldb=F, b_kcontig=1, c=part, ldc=C, M=M, N=C, K=F, nb0=1, nb1=1, split_k=4, c_split_stride=M * C, alpha=1.0)
    print(f'M={M}: two contractions            {t(lambda: two(None)):.1f} us   with dropout {t(lambda: two(drop)):.1f} us')
    for ns in (2, 4, 8):
        for sh in (True, False):
            a = t(lambda: ops.ffn_fwd(x, w1, b1, w2, n_split=ns, store_hidden=sh, transposed=wt))
            b = t(lambda: ops.ffn_fwd(x, w1, b1, w2, n_split=ns, drop=drop, store_hidden=sh, transposed=wt))
            print(f'M={M}: fused n_split={ns} hidden={int(sh)}   {a:.1f} us   with dropout {b:.1f} us   (includes two torch.empty)')
