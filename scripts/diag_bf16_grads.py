"""Where does the bf16 training step deviate from the float64 oracle?  Per-tensor gradient error in fp32 and bf16 mode,
hidden-state (outs_dec) error relative to its spread, and a near-uniform-attention operator check."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import petr_amd
from petr_amd import ops
from oracle import petr_oracle as O

torch.set_num_threads(16)
shape = sys.argv[1] if len(sys.argv) > 1 else 'c5'
N, H, W, pad = {'toy': (2, 4, 6, (128, 192)), 'c5': (6, 16, 44, (512, 1408)), 'p4': (6, 32, 88, (512, 1408))}[shape]
Q = 16 if shape == 'toy' else 900
oracle = O.seeded_head(0, None, num_query=Q)
head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=Q))
head.load_state_dict(oracle.state_dict())
head = head.cuda().eval()
metas = O.synthetic_img_metas(1, N, pad, seed=2)
g = torch.Generator().manual_seed(2)
feats = torch.randn(1, N, 256, H, W, generator=g)
g_cls, g_box = torch.randn(6, 1, Q, 10, generator=g), torch.randn(6, 1, Q, 10, generator=g)
o = copy.deepcopy(oracle).double()
f = feats.double().requires_grad_(True)
torch.set_default_dtype(torch.float64)
want = o([f], metas, return_intermediates=True)
(want['all_cls_scores'] * g_cls.double()).sum().add((want['all_bbox_preds'] * g_box.double()).sum()).backward()
torch.set_default_dtype(torch.float32)
wg = {k: p.grad for k, p in o.named_parameters() if p.grad is not None}
res = {}
for mode in ('fp32', 'bf16'):
    head.attn_dtype = mode
    head.zero_grad_flat()
    fg = feats.cuda().requires_grad_(True)
    got = head([fg], metas)
    outs = head.workspace_view('outs_dec').view(6, 1, Q, 256).double().cpu().clone()
    torch.autograd.backward([got['all_cls_scores'], got['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    wo = want['_outs_dec'].double()
    e_h = [((outs[l] - wo[l]).norm() / (wo[l] - wo[l].mean(1, keepdim=True)).norm()).item() for l in range(6)]
    e_c = ((got['all_cls_scores'].double().cpu() - want['all_cls_scores']).norm() /
           (want['all_cls_scores'] - want['all_cls_scores'].mean()).norm()).item()
    print(mode, 'outs_dec error / spread over queries per level:', [f'{x:.1e}' for x in e_h], 'cls err/spread', f'{e_c:.1e}')
    res[mode] = {k: p.grad.double().cpu().clone() for k, p in head.named_parameters() if p.requires_grad}
    res[mode]['d_feats'] = fg.grad.double().cpu().clone()
wg['d_feats'] = f.grad
names = ['cls_branches.0.6.weight', 'cls_branches.0.0.weight', 'reg_branches.0.0.weight', 'transformer.decoder.post_norm.weight']
for l in (5, 3, 0):
    pre = f'transformer.decoder.layers.{l}.'
    names += [pre + 'ffns.0.layers.1.weight', pre + 'ffns.0.layers.0.0.weight', pre + 'attentions.1.attn.out_proj.weight',
              pre + 'attentions.1.attn.in_proj_weight', pre + 'attentions.0.attn.in_proj_weight']
names += ['query_embedding.2.weight', 'reference_points.weight', 'position_encoder.2.weight', 'position_encoder.0.weight',
          'adapt_pos3d.0.weight', 'input_proj.weight', 'd_feats']
for n in names:
    b = wg[n].double()
    def e(a):
        return ((a - b).norm() / b.norm()).item()
    cos = torch.nn.functional.cosine_similarity(res['bf16'][n].flatten(), b.flatten(), dim=0).item()
    print(f'{n:70s} fp32 {e(res["fp32"][n]):.1e}  bf16 {e(res["bf16"][n]):.1e}  cos(bf16) {cos:.5f}')

# operator check in the head's regime: small logits (near-uniform attention over many keys)
for qs in (1.0, 0.1):
    B, Hh, L = 1, 8, 16896
    gq = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, Hh, n, 32, generator=gq) for n in (900, L, L))
    q = q * qs
    v = v + 2.0         # a common component, as the projected memory has
    k = k + 1.0
    do = torch.randn(B, Hh, 900, 32, generator=gq)
    kb, vb = ops.cast_bf16(k.cuda()), ops.cast_bf16(v.cuda())
    oo, lse = ops.mha_fwd_bf16(q.cuda(), kb, vb)
    dq, dk, dv = ops.mha_bwd_bf16(q.cuda(), kb, vb, oo, do.cuda(), lse)
    qd = q.cuda().double().requires_grad_(True); kd = kb.double().requires_grad_(True); vd = vb.double().requires_grad_(True)
    s = torch.einsum('bhqd,bhkd->bhqk', qd, kd) * 32 ** -0.5
    wnt = torch.einsum('bhqk,bhkd->bhqd', torch.softmax(s, -1), vd)
    wnt.backward(do.cuda().double())
    r = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
    print(f'op regime q*{qs}: o {r(oo, wnt):.1e} dq {r(dq, qd.grad):.1e} dk {r(dk, kd.grad):.1e} dv {r(dv, vd.grad):.1e} (L2 rel)')
