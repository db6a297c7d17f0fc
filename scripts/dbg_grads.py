"""debug: per-tensor gradient error of the HIP head vs the oracle in fp64 and in fp32 (c5 shape)."""
import sys, os, copy
sys.path.insert(0, os.getcwd())
import torch
import petr_amd
from oracle import petr_oracle as O

Q, B, N, H, W, pad = 900, 1, 6, 16, 44, (512, 1408)
if len(sys.argv) > 1 and sys.argv[1] == 'toy':
    Q, B, N, H, W, pad = 16, 2, 2, 4, 6, (128, 192)
oracle = O.seeded_head(5, 77, num_query=Q)
head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=Q)); head.load_state_dict(oracle.state_dict()); head = head.cuda().eval()
metas = O.synthetic_img_metas(B, N, pad, seed=5)
g = torch.Generator().manual_seed(5)
feats = torch.randn(B, N, 256, H, W, generator=g)
g_cls, g_box = torch.randn(6, B, Q, 10, generator=g), torch.randn(6, B, Q, 10, generator=g)

def run_oracle(dtype):
    o = copy.deepcopy(oracle).to(dtype)
    f = feats.to(dtype).requires_grad_(True)
    # oracle helper tensors are created in fp32; run under default dtype switch
    torch.set_default_dtype(dtype)
    try:
        out = o([f], metas)
        (out['all_cls_scores'] * g_cls.to(dtype)).sum().add((out['all_bbox_preds'] * g_box.to(dtype)).sum()).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return {k: p.grad for k, p in o.named_parameters() if p.grad is not None}, f.grad

g64, f64 = run_oracle(torch.float64)
g32, f32 = run_oracle(torch.float32)
fg = feats.cuda().requires_grad_(True)
out = head([fg], metas)
torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
gmax = max(v.abs().max().item() for v in g64.values())
print('global max grad', gmax)
rows = []
for k, p in head.named_parameters():
    if p.grad is None or k not in g64: continue
    w = g64[k]; sc = max(w.abs().max().item(), 1e-30)
    e_gpu = (p.grad.double().cpu() - w).abs().max().item()
    e_cpu = (g32[k].double() - w).abs().max().item()
    rows.append((e_gpu / sc, e_cpu / sc, sc, k))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f'gpu_rel {r[0]:.3e}  cpu32_rel {r[1]:.3e}  |want|max {r[2]:.3e}  {r[3]}')
print('d_feats gpu', ((fg.grad.double().cpu() - f64).abs().max() / f64.abs().max()).item(), 'cpu32', ((f32.double() - f64).abs().max() / f64.abs().max()).item())
