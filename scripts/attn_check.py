"""diagnostic: cross-attention forward vs a torch reference, per mode / n_split, with an error map."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops
B, H, Q, L = 1, 8, 900, int(sys.argv[1]) if len(sys.argv) > 1 else 4224
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Q, 32, generator=g).cuda(); k = torch.randn(B, H, L, 32, generator=g).cuda(); v = torch.randn(B, H, L, 32, generator=g).cuda()
ref = torch.softmax((q.double() @ k.double().transpose(-1, -2)) * 32 ** -0.5, -1) @ v.double()
for dyn in (False, True):
    for ns in (1, 2, 8):
        for rep in range(2):
            o, lse = ops.mha_fwd(q, k, v, n_split=ns, dynamic=dyn)
            torch.cuda.synchronize()
            err = (o.double() - ref).abs()
            print(f'dyn={dyn} ns={ns} rep={rep}: max err {err.max().item():.3e}  nan {torch.isnan(o).sum().item()}', flush=True)
            if err.max() > 1e-3 or torch.isnan(o).any():
                e = err.amax(-1)[0]           # [H, Q]
                bad = (e > 1e-3) | torch.isnan(e)
                print('  bad rows per head:', bad.sum(1).tolist())
                print('  bad rows per 32-query group (head 0):', bad[0].float().view(-1)[:896].view(28, 32).sum(1).int().tolist())
                print('  sample o/ref:', o[0, 0, 0, :4].tolist(), ref[0, 0, 0, :4].tolist())
