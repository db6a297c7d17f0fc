"""Every token-sized bf16 contraction of the bf16 training step, timed alone, with its algorithmic HBM bytes:
   python scripts/token_gemm_time.py [views H W D]     (default: p4-1600 = 6 x 40 x 100, D = 64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops, _C

V, H, Wd, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (6, 40, 100, 64)
HW, C, NL, Cin = H * Wd, 256, 6, 256
L = V * HW
dev = 'cuda'
g = torch.Generator().manual_seed(0)
F, BF, ST, ACC, AT, MASK, A16, B16, R16 = (_C.GEMM_BF16, _C.GEMM_BF16, _C.GEMM_STORE_BF16, _C.GEMM_ACCUMULATE, _C.GEMM_ATOMIC,
                                          _C.GEMM_RELU_MASK, _C.GEMM_A_BF16, _C.GEMM_B_BF16, _C.GEMM_R_BF16)
RELU = _C.GEMM_RELU


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


def rnd(*s, dt=torch.float32):
    return torch.randn(*s, generator=g).to(dev).to(dt)


tot = [0.0, 0.0]
def row(name, fn, flop, nbytes):
    us = t(fn)
    tot[0] += us; tot[1] += nbytes / 8e6
    print(f'{name:34s} {us:8.1f} us  {2 * flop / us * 1e-6:7.1f} TFLOP/s  {nbytes / us * 1e-3:7.1f} GB/s   (HBM floor {nbytes / 8e6:6.1f} us)', flush=True)


bf = torch.bfloat16
feats = rnd(V, Cin, HW); vol = rnd(V, 3 * D, HW); sine = rnd(V, 384, HW)
w_in, b_in = rnd(C, Cin), rnd(C)
w_pe1, w_ad1, b4 = rnd(4 * C, 3 * D), rnd(4 * C, 384), rnd(4 * C)
w_pe2, b1 = rnd(C, 4 * C), rnd(C)
Wkv, bkv = rnd(NL, C, C), rnd(NL, C)
mem = torch.empty(L, C, device=dev); h16 = torch.empty(L, 4 * C, dtype=bf, device=dev); pos = torch.empty(L, C, device=dev)
k16 = torch.empty(NL, L, C, dtype=bf, device=dev)
print(f'tokens L = {L}')
# ---- forward ----
row('fwd input_proj (NCHW A)', lambda: ops.gemm_raw(a=feats, lda=HW, a_kcontig=0, a_bs0=Cin * HW, b=w_in, ldb=Cin, b_kcontig=1, c=mem, ldc=C,
    c_bs0=HW * C, bias=b_in, M=HW, N=C, K=Cin, nb0=V, nb1=1, flags=BF, alpha=1.0), L * C * Cin, L * Cin * 4 + L * C * 4)
row('fwd pe1 (NCHW A, relu, bf16 out)', lambda: ops.gemm_raw(a=vol, lda=HW, a_kcontig=0, a_bs0=3 * D * HW, b=w_pe1, ldb=3 * D, b_kcontig=1, c=h16,
    ldc=4 * C, c_bs0=HW * 4 * C, bias=b4, M=HW, N=4 * C, K=3 * D, nb0=V, nb1=1, flags=BF | ST | RELU, alpha=1.0), L * 4 * C * 3 * D, L * 3 * D * 4 + L * 4 * C * 2)
row('fwd ad1 (NCHW A, relu, bf16 out)', lambda: ops.gemm_raw(a=sine, lda=HW, a_kcontig=0, a_bs0=384 * HW, b=w_ad1, ldb=384, b_kcontig=1, c=h16,
    ldc=4 * C, c_bs0=HW * 4 * C, bias=b4, M=HW, N=4 * C, K=384, nb0=V, nb1=1, flags=BF | ST | RELU, alpha=1.0), L * 4 * C * 384, L * 384 * 4 + L * 4 * C * 2)
row('fwd pe2 (bf16 A)', lambda: ops.gemm_raw(a=h16, lda=4 * C, a_kcontig=1, b=w_pe2, ldb=4 * C, b_kcontig=1, c=pos, ldc=C, bias=b1, M=L, N=C,
    K=4 * C, nb0=1, nb1=1, flags=BF | A16, alpha=1.0), L * C * 4 * C, L * 4 * C * 2 + L * C * 4)
row('fwd ad2 (bf16 A, + residual)', lambda: ops.gemm_raw(a=h16, lda=4 * C, a_kcontig=1, b=w_pe2, ldb=4 * C, b_kcontig=1, c=pos, ldc=C, bias=b1, r=mem,
    ldr=C, M=L, N=C, K=4 * C, nb0=1, nb1=1, flags=BF | A16, alpha=1.0), L * C * 4 * C, L * 4 * C * 2 + 2 * L * C * 4)
row('fwd K proj (6 layers, bf16 out)', lambda: ops.gemm_raw(a=mem, lda=C, a_kcontig=1, b=Wkv, ldb=C, b_kcontig=1, b_bs1=C * C, c=k16, ldc=C,
    c_bs1=L * C, bias=bkv, bias_bs1=C, M=L, N=C, K=C, nb0=1, nb1=NL, flags=BF | ST, alpha=1.0), NL * L * C * C, L * C * 4 + NL * L * C * 2)
# ---- backward ----
dkv = rnd(NL, L, C, dt=bf); dsrc = torch.empty(L, C, device=dev); src = rnd(L, C)
row('bwd d_src (bf16 dKV, 6 segments)', lambda: ops.gemm_raw(a=dkv, lda=C, a_kcontig=1, b=Wkv, ldb=C, b_kcontig=0, c=dsrc, ldc=C, M=L, N=C, K=NL * C,
    nb0=1, nb1=1, k_seg=C, a_seg_stride=L * C, b_seg_stride=C * C, flags=BF | A16, alpha=1.0), L * C * NL * C, NL * L * C * 2 + L * C * 4)
dw = torch.zeros(NL, C, C, device=dev); db = torch.zeros(NL, C, device=dev)
row('bwd dW_kv (bf16 dKV, atomics)', lambda: ops.gemm_raw(a=dkv, lda=C, a_kcontig=0, a_bs0=L * C, b=src, ldb=C, b_kcontig=0, c=dw, ldc=C, c_bs0=C * C,
    a_colsum=db, cs_bs0=C, M=C, N=C, K=L, nb0=NL, nb1=1, split_k=max(1, min(128, 512 // (4 * NL))), flags=BF | AT | A16, alpha=1.0),
    NL * C * C * L, NL * L * C * 2 + L * C * 4)
dy = rnd(L, C); dh16 = torch.empty(L, 4 * C, dtype=bf, device=dev); hid16 = rnd(L, 4 * C, dt=bf)
row('bwd pe2 dgrad (mask, bf16 out)', lambda: ops.gemm_raw(a=dy, lda=C, a_kcontig=1, b=w_pe2, ldb=4 * C, b_kcontig=0, c=dh16, ldc=4 * C, r=hid16,
    ldr=4 * C, M=L, N=4 * C, K=C, nb0=1, nb1=1, flags=BF | MASK | R16 | ST, alpha=1.0), L * 4 * C * C, L * C * 4 + 2 * L * 4 * C * 2)
dw2 = torch.zeros(C, 4 * C, device=dev); db2 = torch.zeros(C, device=dev)
row('bwd pe2 wgrad (bf16 hidden)', lambda: ops.gemm_raw(a=dy, lda=C, a_kcontig=0, b=hid16, ldb=4 * C, b_kcontig=0, c=dw2, ldc=4 * C, a_colsum=db2,
    M=C, N=4 * C, K=L, nb0=1, nb1=1, split_k=32, flags=BF | AT | B16, alpha=1.0), C * 4 * C * L, L * C * 4 + L * 4 * C * 2)
dw1 = torch.zeros(4 * C, 3 * D, device=dev); db1 = torch.zeros(4 * C, device=dev)
row('bwd pe1 wgrad (bf16 d_h, NCHW B)', lambda: ops.gemm_raw(a=dh16, lda=4 * C, a_kcontig=0, b=vol, ldb=HW, b_kcontig=1, c=dw1, ldc=3 * D, a_colsum=db1,
    M=4 * C, N=3 * D, K=L, nb0=1, nb1=1, k_seg=HW, a_seg_stride=HW * 4 * C, b_seg_stride=3 * D * HW, split_k=max(1, min(128, 512 // (8 * 2))),
    flags=BF | AT | A16, alpha=1.0), 4 * C * 3 * D * L, L * 4 * C * 2 + L * 3 * D * 4)
dwa = torch.zeros(4 * C, 384, device=dev)
row('bwd ad1 wgrad (bf16 d_h, NCHW B)', lambda: ops.gemm_raw(a=dh16, lda=4 * C, a_kcontig=0, b=sine, ldb=HW, b_kcontig=1, c=dwa, ldc=384, a_colsum=db1,
    M=4 * C, N=384, K=L, nb0=1, nb1=1, k_seg=HW, a_seg_stride=HW * 4 * C, b_seg_stride=384 * HW, split_k=max(1, min(128, 512 // (8 * 3))),
    flags=BF | AT | A16, alpha=1.0), 4 * C * 384 * L, L * 4 * C * 2 + L * 384 * 4)
dwi = torch.zeros(C, Cin, device=dev); dbi = torch.zeros(C, device=dev)
row('bwd input_proj dW (NCHW B)', lambda: ops.gemm_raw(a=dy, lda=C, a_kcontig=0, b=feats, ldb=HW, b_kcontig=1, c=dwi, ldc=Cin, a_colsum=dbi, M=C,
    N=Cin, K=L, nb0=1, nb1=1, k_seg=HW, a_seg_stride=HW * C, b_seg_stride=Cin * HW, split_k=max(1, min(128, 512 // 4)), flags=BF | AT,
    alpha=1.0), C * Cin * L, L * C * 4 + L * Cin * 4)
dfe = torch.empty(V, Cin, HW, device=dev)
row('bwd d_feats (NCHW out)', lambda: ops.gemm_raw(a=w_in, lda=Cin, a_kcontig=0, b=dy, ldb=C, b_kcontig=1, b_bs0=HW * C, c=dfe, ldc=HW,
    c_bs0=Cin * HW, M=Cin, N=HW, K=C, nb0=V, nb1=1, flags=BF, alpha=1.0), L * C * Cin, L * C * 4 + L * Cin * 4)
print(f'sum (dW_kv, d_src, pe2 dgrad/wgrad counted once): {tot[0]:.1f} us; HBM floors sum {tot[1]:.1f} us')
