"""900-row FFN contractions: fp32 kernels vs the bf16 route (weights bf16), alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops, _C
g = torch.Generator().manual_seed(0)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, C, F = 900, 256, 2048
x = torch.randn(M, C, generator=g).cuda(); w1 = torch.randn(F, C, generator=g).cuda(); b1 = torch.randn(F, generator=g).cuda()
w2 = torch.randn(C, F, generator=g).cuda(); h = torch.randn(M, F, generator=g).cuda()
w1h, w2h = w1.bfloat16(), w2.bfloat16()
hid = torch.empty(M, F, device='cuda'); part = torch.empty(16, M, C, device='cuda')
BF = _C.GEMM_BF16
print('ffn1 fp32      ', t(lambda: ops.gemm_raw(a=x, lda=C, a_kcontig=1, b=w1, ldb=C, b_kcontig=1, c=hid, ldc=F, bias=b1, M=M, N=F, K=C, nb0=1, nb1=1, flags=_C.GEMM_RELU, alpha=1.0)))
print('ffn1 bf16 (B16)', t(lambda: ops.gemm_raw(a=x, lda=C, a_kcontig=1, b=w1h, ldb=C, b_kcontig=1, c=hid, ldc=F, bias=b1, M=M, N=F, K=C, nb0=1, nb1=1, flags=_C.GEMM_RELU | BF | _C.GEMM_B_BF16, alpha=1.0)))
for sk in (4, 8, 16):
    print(f'ffn2 fp32 split {sk}', t(lambda: ops.gemm_raw(a=h, lda=F, a_kcontig=1, b=w2, ldb=F, b_kcontig=1, c=part, ldc=C, M=M, N=C, K=F, nb0=1, nb1=1, split_k=sk, c_split_stride=M * C, alpha=1.0)))
    print(f'ffn2 bf16 split {sk}', t(lambda: ops.gemm_raw(a=h, lda=F, a_kcontig=1, b=w2h, ldb=F, b_kcontig=1, c=part, ldc=C, M=M, N=C, K=F, nb0=1, nb1=1, split_k=sk, c_split_stride=M * C, flags=BF | _C.GEMM_B_BF16, alpha=1.0)))
# dgrads: d_h = d_f2 @ W2 (K = 256, N = 2048, relu mask), d_x2 = d_h @ W1 (K = 2048, N = 256, slabs)
df2 = torch.randn(M, C, generator=g).cuda(); dh = torch.empty(M, F, device='cuda')
w2t, w1t = w2.t().contiguous(), w1.t().contiguous()       # transposed copies [F, C] / [C, F]
print('ffn2 dgrad fp32 (transposed W)', t(lambda: ops.gemm_raw(a=df2, lda=C, a_kcontig=1, b=w2t, ldb=C, b_kcontig=1, c=dh, ldc=F, r=h, ldr=F, M=M, N=F, K=C, nb0=1, nb1=1, flags=_C.GEMM_RELU_MASK, alpha=1.0)))
print('ffn2 dgrad bf16 (W bf16 K-major)', t(lambda: ops.gemm_raw(a=df2, lda=C, a_kcontig=1, b=w2h, ldb=F, b_kcontig=0, c=dh, ldc=F, r=h, ldr=F, M=M, N=F, K=C, nb0=1, nb1=1, flags=_C.GEMM_RELU_MASK | BF | _C.GEMM_B_BF16, alpha=1.0)))
print('ffn2 dgrad bf16 (transposed bf16 W)', t(lambda: ops.gemm_raw(a=df2, lda=C, a_kcontig=1, b=w2t.bfloat16(), ldb=C, b_kcontig=1, c=dh, ldc=F, r=h, ldr=F, M=M, N=F, K=C, nb0=1, nb1=1, flags=_C.GEMM_RELU_MASK | BF | _C.GEMM_B_BF16, alpha=1.0)))
w1th = w1t.bfloat16()
for sk in (4, 8):
    print(f'ffn1 dgrad fp32 split {sk}', t(lambda: ops.gemm_raw(a=dh, lda=F, a_kcontig=1, b=w1t, ldb=F, b_kcontig=1, c=part, ldc=C, M=M, N=C, K=F, nb0=1, nb1=1, split_k=sk, c_split_stride=M * C, alpha=1.0)))
    print(f'ffn1 dgrad bf16 split {sk}', t(lambda: ops.gemm_raw(a=dh, lda=F, a_kcontig=1, b=w1th, ldb=F, b_kcontig=1, c=part, ldc=C, M=M, N=C, K=F, nb0=1, nb1=1, split_k=sk, c_split_stride=M * C, flags=BF | _C.GEMM_B_BF16, alpha=1.0)))
# 900 x 256 x 256 projections
o = torch.empty(M, C, device='cuda'); wq = torch.randn(C, C, generator=g).cuda()
print('proj 900x256x256 fp32 (skinny)', t(lambda: ops.gemm_raw(a=x, lda=C, a_kcontig=1, b=wq, ldb=C, b_kcontig=1, c=o, ldc=C, M=M, N=C, K=C, nb0=1, nb1=1, alpha=1.0)))
print('proj 900x256x256 bf16', t(lambda: ops.gemm_raw(a=x, lda=C, a_kcontig=1, b=wq.bfloat16(), ldb=C, b_kcontig=1, c=o, ldc=C, M=M, N=C, K=C, nb0=1, nb1=1, flags=BF | _C.GEMM_B_BF16, alpha=1.0)))
