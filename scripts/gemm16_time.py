"""Times of the token-sized bf16 contractions of the p4-1600 training step (one process per kernel variant:
PETR_GEMM16_PC=0/1 selects the general kernel / the producer-consumer kernel of gemm_bf16.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops, _C

L, C, NL = int(sys.argv[1]) if len(sys.argv) > 1 else 24000, 256, 6
g = torch.Generator().manual_seed(0)
dev = 'cuda'


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dkv = torch.randn(NL, L, C, generator=g).to(dev)
W = torch.randn(NL, C, C, generator=g).to(dev)
src = torch.randn(L, C, generator=g).to(dev)
out = torch.zeros(L, C, device=dev)
gf = lambda fl: 2.0 * fl / 1e9
# (1) d_src = sum_l dKV_l W_l : K-contiguous A in 6 segments, K-major B
f1 = lambda: ops.gemm_raw(a=dkv, lda=C, a_kcontig=1, b=W, ldb=C, b_kcontig=0, c=out, ldc=C, M=L, N=C, K=NL * C, nb0=1, nb1=1,
                          k_seg=C, a_seg_stride=L * C, b_seg_stride=C * C, flags=_C.GEMM_BF16, alpha=1.0)
us = t(f1); print(f'dgrad d_src   M={L} N=256 K=1536 : {us:7.1f} us  {gf(L*C*NL*C)/us*1e3:6.1f} TFLOP/s')
# (2) dW_l = dKV_l^T src (6 layers batched): both K-major, split-K atomics
dw = torch.zeros(NL, C, C, device=dev); db = torch.zeros(NL, C, device=dev)
for sk in (21, 64):
    f2 = lambda: ops.gemm_raw(a=dkv, lda=C, a_kcontig=0, a_bs0=L * C, b=src, ldb=C, b_kcontig=0, c=dw, ldc=C, c_bs0=C * C,
                              a_colsum=db, cs_bs0=C, M=C, N=C, K=L, nb0=NL, nb1=1, split_k=sk, flags=_C.GEMM_BF16 | _C.GEMM_ATOMIC,
                              alpha=1.0)
    us = t(f2); print(f'wgrad dW_kv   6x(256x256) K={L} split {sk}: {us:7.1f} us  {gf(NL*C*C*L)/us*1e3:6.1f} TFLOP/s')
# (3) pe2 dgrad with ReLU mask: dy [L,256] @ w2 [256,1024] -> [L,1024]
w2 = torch.randn(C, 4 * C, generator=g).to(dev); hid = torch.randn(L, 4 * C, generator=g).to(dev); dh = torch.zeros(L, 4 * C, device=dev)
f3 = lambda: ops.gemm_raw(a=src, lda=C, a_kcontig=1, b=w2, ldb=4 * C, b_kcontig=0, c=dh, ldc=4 * C, r=hid, ldr=4 * C, M=L, N=4 * C,
                          K=C, nb0=1, nb1=1, flags=_C.GEMM_BF16 | _C.GEMM_RELU_MASK, alpha=1.0)
us = t(f3); print(f'dgrad pe2     M={L} N=1024 K=256 : {us:7.1f} us  {gf(L*4*C*C)/us*1e3:6.1f} TFLOP/s')
# (4) pe2 wgrad: dW2[256,1024] = dy^T hid
dw2 = torch.zeros(C, 4 * C, device=dev)
f4 = lambda: ops.gemm_raw(a=src, lda=C, a_kcontig=0, b=hid, ldb=4 * C, b_kcontig=0, c=dw2, ldc=4 * C, M=C, N=4 * C, K=L, nb0=1,
                          nb1=1, split_k=32, flags=_C.GEMM_BF16 | _C.GEMM_ATOMIC, alpha=1.0)
us = t(f4); print(f'wgrad pe2     256x1024 K={L} split 32 : {us:7.1f} us  {gf(C*4*C*L)/us*1e3:6.1f} TFLOP/s')
# (5) forward K projection, 6 layers batched, bf16 store (gemm.hip's staged kernel unless forced general via accumulate)
k16 = torch.zeros(NL, L, C, dtype=torch.bfloat16, device=dev); bias = torch.randn(NL, C, generator=g).to(dev)
f5 = lambda: ops.gemm_raw(a=src, lda=C, a_kcontig=1, b=W, ldb=C, b_kcontig=1, b_bs1=C * C, c=k16, ldc=C, c_bs1=L * C, bias=bias,
                          bias_bs1=C, M=L, N=C, K=C, nb0=1, nb1=NL, flags=_C.GEMM_BF16 | _C.GEMM_STORE_BF16, alpha=1.0)
us = t(f5); print(f'fwd   K proj  6x({L}x256) K=256 (staged kernel of gemm.hip): {us:7.1f} us  {gf(NL*L*C*C)/us*1e3:6.1f} TFLOP/s')
kf = torch.zeros(NL, L, C, device=dev)
f6 = lambda: ops.gemm_raw(a=src, lda=C, a_kcontig=1, b=W, ldb=C, b_kcontig=1, b_bs1=C * C, c=kf, ldc=C, c_bs1=L * C, bias=bias,
                          bias_bs1=C, M=L, N=C, K=C, nb0=1, nb1=NL, flags=_C.GEMM_BF16 | _C.GEMM_ACCUMULATE, alpha=1.0)
us = t(f6); print(f'fwd   K proj  same through gemm_bf16.hip (accumulate epilogue): {us:7.1f} us  {gf(NL*L*C*C)/us*1e3:6.1f} TFLOP/s')
