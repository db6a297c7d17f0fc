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
