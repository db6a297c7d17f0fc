"""diagnostic: run the bf16 cross-attention forward kernel a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = torch.Generator().manual_seed(0)
q = torch.randn(B, 8, 900, 32, generator=g).cuda()
k = ops.cast_bf16(torch.randn(B, 8, L, 32, generator=g).cuda())
v = ops.cast_bf16(torch.randn(B, 8, L, 32, generator=g).cuda())
for _ in range(5):
    ops.mha_fwd_bf16(q, k, v, n_split=ns)
torch.cuda.synchronize()
