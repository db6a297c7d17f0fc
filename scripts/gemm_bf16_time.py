"""diagnostic: the K/V projection of one forward (6 layers batched) in fp32 and through the bf16 route (run under rocprofv3)"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops, _C
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
g = torch.Generator().manual_seed(0)
x = torch.randn(L, 256, generator=g).cuda(); pos = torch.randn(L, 256, generator=g).cuda()
w = torch.randn(6, 256, 256, generator=g).cuda(); b = torch.randn(6, 256, generator=g).cuda()
o32 = torch.empty(6, L, 256).cuda(); o16 = torch.empty(6, L, 256, dtype=torch.bfloat16).cuda()
for _ in range(20):
    for flags, out in ((0, o32), (_C.GEMM_STORE_BF16, o16), (_C.GEMM_BF16 | _C.GEMM_STORE_BF16, o16)):
        ops.gemm_raw(a=x, lda=256, a_kcontig=1, a2=pos, a2_rows=0, a2_ncols=0, b=w, ldb=256, b_kcontig=1, b_bs1=256 * 256, c=out, ldc=256,
                     c_bs1=L * 256, bias=b, bias_bs1=256, M=L, N=256, K=256, nb0=1, nb1=6, flags=flags, alpha=1.0)
torch.cuda.synchronize()
