"""diagnostic: which 64-key tiles does the forward kernel visit?  q = 0 (uniform attention), V one-hot in (tile mod 32)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from petr_amd import ops
B, H, Q, L = 1, 8, 900, 4224
q = torch.zeros(B, H, Q, 32).cuda(); k = torch.randn(B, H, L, 32).cuda()
v = torch.zeros(B, H, L, 32)
tile = torch.arange(L) // 64
v[:, :, torch.arange(L), tile % 32] = 1.0
v = v.cuda()
for dyn, ns in ((False, 2), (True, 2), (True, 8)):
    o, _ = ops.mha_fwd(q, k, v, n_split=ns, dynamic=dyn)
    torch.cuda.synchronize()
    cnt = (o * (L / 64.0))          # tiles visited per residue class
    print(f'dyn={dyn} ns={ns}: head0 q0 tiles per class:', [round(x, 2) for x in cnt[0, 0, 0].tolist()])
    print(f'   head3 q500:', [round(x, 2) for x in cnt[0, 3, 500].tolist()], ' sum', round(cnt[0, 3, 500].sum().item(), 2))
