"""CPFPN neck boundary: the HIP path against the fixtures written from the reference's own CPFPN
(oracle/make_golden_neck.py) and, at the BASELINE p4 shapes, against the CPU restatement those fixtures pin."""
import os

import numpy as np
import pytest
import torch

from oracle import neck_oracle as NO  # checker only

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    fx = np.load(os.path.join(GOLDEN, name + '.npz'))
    state = {k[6:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('param:')}
    n = sum(1 for k in fx.files if k.startswith('in'))
    return fx, state, [torch.from_numpy(fx[f'in{i}']) for i in range(n)], [torch.from_numpy(fx[f'out{i}']) for i in range(n)]


@pytest.mark.parametrize('name', ['cpfpn_toy', 'cpfpn_odd3'])
def test_neck_oracle_matches_reference_fixture(name):
    """CPU: the restatement reproduces the reference CPFPN's outputs bit for bit, and the product module has the
    reference's state_dict keys."""
    fx, state, inputs, want = _load(name)
    got = NO.cpfpn_forward(state, inputs)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    import petr_amd
    neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=[x.shape[1] for x in inputs], out_channels=64, num_outs=len(inputs)))
    assert sorted(neck.state_dict().keys()) == list(fx['keys'])
    neck.load_state_dict(state)
    with pytest.raises(Exception, match='extra levels'):
        petr_amd.build_neck(dict(type='CPFPN', in_channels=[8, 8], out_channels=64, num_outs=3))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cpfpn_toy', 'cpfpn_odd3'])
def test_cpfpn_golden_gpu(name):
    import petr_amd
    fx, state, inputs, want = _load(name)
    neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=[x.shape[1] for x in inputs], out_channels=64, num_outs=len(inputs)))
    neck.load_state_dict(state)
    neck = neck.cuda()
    got = neck([x.cuda() for x in inputs])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.shape == b.shape
        err = (a.cpu().double() - b.double()).abs().max().item() / b.abs().max().item()
        assert err < 1e-5, err                  # fp32, different summation order (MFMA k order vs torch's conv)
    with pytest.raises(RuntimeError, match='GPU only'):
        neck(inputs)


@pytest.mark.gpu
@pytest.mark.parametrize('chans,V,sizes', [([1024, 2048], 6, [(32, 88), (16, 44)]), ([768, 1024], 2, [(40, 100), (20, 50)]),
                                           ([96, 128], 12, [(20, 50), (10, 25)])])
def test_cpfpn_baseline_shapes_gpu(chans, V, sizes):
    """petr_r50dcn_gridmask_p4 (1024/2048 -> 256, 32x88), vovnet p4 1600x640 (768/1024, 40x100, two views to bound the CPU
    oracle's time) and the 800x320 map of PETRv2 (W = 50: not a multiple of 4, the scalar-load path), then straight into
    the head: neck output -> [B, N, 256, H, W] view -> PETRHead.forward."""
    import petr_amd
    torch.manual_seed(3)
    neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=chans, out_channels=256, num_outs=2))
    neck.init_weights()
    g = torch.Generator().manual_seed(9)
    inputs = [torch.randn(V, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
    with torch.no_grad():
        want = NO.cpfpn_forward(neck.state_dict(), inputs)
    neck = neck.cuda()
    got = neck([x.cuda() for x in inputs])
    for a, b in zip(got, want):
        err = (a.cpu().double() - b.double()).abs().max().item() / b.abs().max().item()
        assert err < 2e-5, err
    if V == 6:      # the caller contract of petr3d.py:95-99: [B*N, C, H, W] -> [B, N, C, H, W] -> head
        from oracle import petr_oracle as O
        head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=64)).cuda().eval()
        metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=1)
        with torch.no_grad():
            out = head(petr_amd.glue.reshape_backbone_feats(list(got), 1), metas)
        assert torch.isfinite(out['all_cls_scores']).all()


def _neck_grad_case(neck_cfg, inputs, seed, state=None, only_level0=False):
    """forward + backward of petr_amd.CPFPN against torch autograd through the pinned CPU restatement (float64)."""
    import petr_amd
    neck = petr_amd.build_neck(neck_cfg)
    if state is None:
        torch.manual_seed(seed)
        neck.init_weights()
        with torch.no_grad():
            for p in neck.parameters():
                if p.dim() == 1:
                    p.uniform_(-0.5, 0.5)          # non-zero biases
    else:
        neck.load_state_dict(state)
    st64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in neck.state_dict().items()}
    xin = [x.double().clone().requires_grad_(True) for x in inputs]
    want = NO.cpfpn_forward(st64, xin)
    g = torch.Generator().manual_seed(seed + 1)
    gouts = [torch.randn(w.shape, generator=g) for w in want]
    loss = sum((w * go.double()).sum() for w, go in (list(zip(want, gouts))[:1] if only_level0 else zip(want, gouts)))
    loss.backward()
    neck = neck.cuda().train()
    xg = [x.cuda().requires_grad_(True) for x in inputs]
    got = neck(xg)
    assert all(o.requires_grad for o in got)
    if only_level0:          # what the detector does: only the position_level-0 map feeds the head (petr_head.py:381)
        got[0].backward(gouts[0].cuda())
    else:
        torch.autograd.backward(list(got), [go.cuda() for go in gouts])

    def err(a, b):
        return ((a.detach().double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    for a, b in zip(got, want):
        assert err(a, b.detach()) < 2e-5
    for i, (a, b) in enumerate(zip(xg, xin)):
        assert a.grad is not None and err(a.grad, b.grad) < 2e-5, f'd_input[{i}]'
    for name, p in neck.named_parameters():
        assert p.grad is not None and err(p.grad, st64[name].grad) < 2e-5, name
    return neck, xg


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cpfpn_toy', 'cpfpn_odd3'])
def test_cpfpn_backward_golden_shapes_gpu(name):
    """the reference fixtures' weights and inputs (two / three levels, odd sizes): every input and parameter gradient."""
    fx, state, inputs, _ = _load(name)
    cfg = dict(type='CPFPN', in_channels=[x.shape[1] for x in inputs], out_channels=64, num_outs=len(inputs))
    _neck_grad_case(cfg, inputs, seed=5, state=state)
    _neck_grad_case(cfg, inputs, seed=6, state=state, only_level0=True)


@pytest.mark.gpu
@pytest.mark.parametrize('chans,V,sizes', [([256, 512], 6, [(32, 88), (16, 44)]), ([96, 128], 12, [(20, 50), (10, 25)])])
def test_cpfpn_backward_baseline_shapes_gpu(chans, V, sizes):
    """p4-sized maps (32x88 / 16x44, six views; channel counts reduced to bound the float64 CPU oracle) and the 800x320 map
    of PETRv2 (W = 50: the scalar-load path), gradient accumulation (+=) on a second backward, and the refusal of CPU maps."""
    g = torch.Generator().manual_seed(11)
    inputs = [torch.randn(V, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
    neck, xg = _neck_grad_case(dict(type='CPFPN', in_channels=chans, out_channels=256, num_outs=2), inputs, seed=3, only_level0=True)
    g1 = {n: p.grad.clone() for n, p in neck.named_parameters()}
    out = neck(xg)
    out[0].backward(torch.ones_like(out[0]))
    out = neck(xg)
    out[0].backward(torch.ones_like(out[0]))          # autograd accumulates: .grad = g1 + 2 * (gradient of the all-ones upstream)
    for n, p in neck.named_parameters():
        assert torch.isfinite(p.grad).all() and not torch.equal(p.grad, g1[n])


@pytest.mark.gpu
def test_cpfpn_into_head_training_step_gpu():
    """neck -> [B, N, C, H, W] view -> PETRHead (train mode) -> backward: the head's d_feats reaches the neck's parameters and
    the backbone maps (the producer of the 256-channel map inside the fwd + bwd step)."""
    import petr_amd
    from oracle import petr_oracle as O
    torch.manual_seed(1)
    neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=[64, 128], out_channels=256, num_outs=2))
    neck.init_weights()
    neck = neck.cuda().train()
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=64)).cuda().train()
    metas = O.synthetic_img_metas(1, 6, (256, 704), seed=1)
    g = torch.Generator().manual_seed(2)
    xs = [torch.randn(6, 64, 16, 44, generator=g).cuda().requires_grad_(True), torch.randn(6, 128, 8, 22, generator=g).cuda().requires_grad_(True)]
    feats = petr_amd.glue.reshape_backbone_feats(list(neck(xs)), 1)
    out = head(feats, metas)
    (out['all_cls_scores'].sum() + out['all_bbox_preds'].sum()).backward()
    for x in xs:
        assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().max().item() > 0
    for n, p in neck.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max().item() > 0, n


@pytest.mark.gpu
@pytest.mark.parametrize('chans,sizes,pad_hw,attn_dtype', [([768, 1024], [(40, 100), (20, 50)], (640, 1600), 'fp32'),
                                                           ([768, 1024], [(40, 100), (20, 50)], (640, 1600), 'bf16'),
                                                           ([1024, 2048], [(32, 88), (16, 44)], (512, 1408), 'fp32'),
                                                           ([96, 128], [(20, 50), (10, 25)], (320, 800), 'fp32')])
def test_cpfpn_folded_input_proj_gpu(chans, sizes, pad_hw, attn_dtype):
    """SURVEY 8(f)-4's fold (cp_fpn.py:190-192 followed by petr_head.py:390 as ONE 3x3 conv, token-major output): the
    projected memory equals what neck -> head.input_proj produces, and so do the head's outputs; the oracle composition
    (float64 conv3x3 then conv1x1) checks the memory independently."""
    import petr_amd
    from oracle import petr_oracle as O
    torch.manual_seed(5)
    neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=chans, out_channels=256, num_outs=2))
    neck.init_weights()
    with torch.no_grad():
        for m in neck.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.bias.uniform_(-0.1, 0.1)                     # the fold must carry W_proj b_3 + b_proj
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=100))
    with torch.no_grad():
        head.input_proj.bias.uniform_(-0.1, 0.1)
    g = torch.Generator().manual_seed(11)
    V = 6
    inputs = [torch.randn(V, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
    with torch.no_grad():       # float64 composition through the pinned restatement
        sd64 = {k: v.double() for k, v in neck.state_dict().items()}
        lvl0 = NO.cpfpn_forward(sd64, [x.double() for x in inputs])[0]
        want_mem = torch.nn.functional.conv2d(lvl0, head.input_proj.weight.double(), head.input_proj.bias.double()).permute(0, 2, 3, 1)
    neck, head = neck.cuda().eval(), head.cuda().eval()
    head.attn_dtype = attn_dtype
    metas = O.synthetic_img_metas(1, V, pad_hw, seed=4)
    xs = [x.cuda() for x in inputs]
    mem = neck.forward_folded(xs, head)
    H, W = sizes[0]
    assert mem.shape == (V, H, W, 256)
    err = (mem.cpu().double() - want_mem).abs().max().item() / want_mem.abs().max().item()
    assert err < 2e-5, err
    with torch.no_grad():
        ref = head(petr_amd.glue.reshape_backbone_feats(list(neck(xs)), 1), metas)
        ref_mem = head.workspace_view('memory').clone()
        got = head.forward_projected(mem.view(1, V, H, W, 256), metas)
        got_mem = head.workspace_view('memory').clone()
    if attn_dtype == 'fp32':
        assert (got_mem - ref_mem).abs().max().item() <= 2e-5 * ref_mem.abs().max().item()
    tol = 2e-4 if attn_dtype == 'fp32' else 3e-2           # bf16 mode: memory is rounded to bf16 on both routes, from values 1e-6 apart
    for k in ('all_cls_scores', 'all_bbox_preds'):
        d = (got[k] - ref[k]).abs().max().item()
        assert d <= tol * max(1.0, ref[k].abs().max().item()), (k, d)
    # a changed weight re-folds
    with torch.no_grad():
        head.input_proj.weight.mul_(0.5)
        head._ensure_flat()
    mem2 = neck.forward_folded(xs, head)
    with torch.no_grad():
        want2 = torch.nn.functional.conv2d(lvl0, head.input_proj.weight.detach().cpu().double(), head.input_proj.bias.detach().cpu().double()).permute(0, 2, 3, 1)
    assert (mem2.cpu().double() - want2).abs().max().item() / want2.abs().max().item() < 2e-5
    # inference only: refused in train mode, and no gradient is recorded
    assert not got['all_cls_scores'].requires_grad
    head.train()
    with pytest.raises(RuntimeError, match='inference path'):
        head.forward_projected(mem.view(1, V, H, W, 256), metas)
