"""diagnostic: the step's small contractions in isolation (run under rocprofv3 --kernel-trace; scripts/gemm_iso.sh)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops
g = torch.Generator().manual_seed(0)
M, N, K = 900, 256, 256
x = torch.randn(M, K, generator=g).cuda(); w = torch.randn(N, K, generator=g).cuda(); b = torch.randn(N, generator=g).cuda()
dy = torch.randn(M, N, generator=g).cuda()
out = torch.empty(M, N).cuda(); dx = torch.empty(M, K).cuda(); dw = torch.zeros(N, K).cuda()
for _ in range(30):
    ops.linear(x, w, b, out=out)                                                       # forward: both K-contiguous
    # dgrad dX[m,k] = sum_n dY[m,n] W[n,k]: A = dY (K-contig over n), B(k_out, n) = W[n, k_out] -> not K-contiguous
    ops.gemm_raw(a=dy, lda=N, a_kcontig=1, b=w, ldb=K, b_kcontig=0, c=dx, ldc=K, M=M, N=K, K=N, flags=0, alpha=1.0)
    # wgrad dW[n,k] = sum_m dY[m,n] X[m,k]: A(n, m) = dY[m, n], B(k, m) = X[m, k]: neither K-contiguous
    ops.gemm_raw(a=dy, lda=N, a_kcontig=0, b=x, ldb=K, b_kcontig=0, c=dw, ldc=K, M=N, N=K, K=M, flags=0, alpha=1.0)
torch.cuda.synchronize()
