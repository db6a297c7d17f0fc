"""Size-independent properties at BASELINE.json's FULL sizes (p4-1600: L = 6 x 40 x 100 = 24 000 key tokens, 900 queries;
PETRv2 800x320: 12 views, L = 12 000), where the CPU oracle is too slow to be the checker: rows of a softmax sum to one,
the attention is linear in V and invariant under a permutation of the key axis and under a common shift of all keys,
masking the tail of the key axis equals truncating it, the backward obeys the matching conservation laws, and the whole
head is batch-consistent, deterministic and has a backward that is linear in the upstream gradient.  Through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import petr_oracle as O  # noqa: E402  (checker only: weights and synthetic img_metas)


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from petr_amd import ops as _ops
    return _ops


def relerr(got, want):
    got, want = got.detach().double(), want.detach().double()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def _qkv(L, seed, Q=900, B=1):
    g = torch.Generator().manual_seed(seed)
    return tuple(torch.randn(B, 8, n, 32, generator=g).cuda() for n in (Q, L, L))


@pytest.mark.parametrize('L', [24000, 12000])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_attention_forward_properties_full_size(ops, L, dtype):
    q, k, v = _qkv(L, seed=L)
    if dtype == 'fp32':
        fwd = lambda kk, vv, **kw: ops.mha_fwd(q, kk, vv, **kw)[0]                                   # noqa: E731
        tol_exact, tol_lin, tol_perm, tol_shift = 2e-6, 1e-5, 5e-6, 2e-4
    else:
        fwd = lambda kk, vv, **kw: ops.mha_fwd_bf16(q, ops.cast_bf16(kk), ops.cast_bf16(vv), **kw)[0]   # noqa: E731
        # V and P are rounded to bf16 inside: linearity / shift invariance hold to bf16 accuracy of the averaged output; a
        # permutation changes the tile-wise reference maxima and with them the bf16 rounding of every probability
        # (measured 3e-3 of the largest output at L = 24 000, where the outputs themselves are averages of size ~1e-2)
        tol_exact, tol_lin, tol_perm, tol_shift = 2e-6, 1.5e-2, 8e-3, 1.5e-2
    o = fwd(k, v)
    # (1) softmax rows sum to one: a constant V comes back exactly (the constants are bf16-representable)
    const = torch.tensor([0.5, -2.0, 1.0, 3.0] * 8).cuda()
    oc = fwd(k, const.expand(1, 8, L, 32).contiguous())
    assert (oc - const).abs().max().item() < tol_exact * 3.0
    # (2) linear in V
    v2 = torch.randn(1, 8, L, 32, generator=torch.Generator().manual_seed(1)).cuda()
    assert relerr(fwd(k, 0.75 * v - 1.5 * v2), 0.75 * o - 1.5 * fwd(k, v2)) < tol_lin
    # (3) the keys are a set: permuting (K, V) rows together changes only the order of summation
    perm = torch.randperm(L, generator=torch.Generator().manual_seed(2)).cuda()
    assert relerr(fwd(k[:, :, perm].contiguous(), v[:, :, perm].contiguous()), o) < tol_perm
    # (4) a common shift of all keys moves every score of a row by the same amount: softmax unchanged
    t = 0.25 * torch.randn(1, 8, 1, 32, generator=torch.Generator().manual_seed(3)).cuda()
    assert relerr(fwd(k + t, v), o) < tol_shift
    # (5) masking the tail of the key axis == truncating it
    cut = L - 1777
    kpm = torch.zeros(1, L, dtype=torch.bool, device='cuda')
    kpm[:, cut:] = True
    assert relerr(fwd(k, v, key_padding_mask=kpm), fwd(k[:, :, :cut].contiguous(), v[:, :, :cut].contiguous())) < tol_perm


def test_attention_backward_conservation_laws_full_size(ops):
    """dV: every probability row sums to one, so sum_k dV[k, :] = sum_q dO[q, :];  dK: the scores of a row can be shifted
    together for free, so sum_k dK[k, :] = 0;  dQ is linear in dO.  (fp32, L = 24 000)"""
    L = 24000
    q, k, v = _qkv(L, seed=7)
    do = torch.randn(1, 8, 900, 32, generator=torch.Generator().manual_seed(8)).cuda()
    o, lse = ops.mha_fwd(q, k, v)
    dq, dk, dv = ops.mha_bwd(q, k, v, o, do, lse)
    assert relerr(dv.double().sum(2), do.double().sum(2)) < 1e-5
    assert dk.double().sum(2).abs().max().item() < 1e-4 * dk.abs().max().item() * 30       # ~ sqrt(L) rounding noise
    do2 = torch.randn(1, 8, 900, 32, generator=torch.Generator().manual_seed(9)).cuda()
    dq2, dk2, dv2 = ops.mha_bwd(q, k, v, o, do2, lse)
    dq3, dk3, dv3 = ops.mha_bwd(q, k, v, o, 2.0 * do - 0.5 * do2, lse)
    # (o is held fixed, so delta = rowsum(dO * O) is linear in dO as well)
    assert relerr(dq3, 2.0 * dq - 0.5 * dq2) < 2e-5 and relerr(dk3, 2.0 * dk - 0.5 * dk2) < 2e-5
    assert relerr(dv3, 2.0 * dv - 0.5 * dv2) < 2e-5


@pytest.fixture(scope='module')
def head_p4():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import petr_amd
    oracle = O.seeded_head(0, None, num_query=900)
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=900))
    head.load_state_dict(oracle.state_dict())
    return head.cuda().eval()


def test_head_full_size_batch_consistency_and_determinism(head_p4):
    """BASELINE configs[3] shape (6 x 256 x 40 x 100, 900 queries): a batch of two identical samples gives each of them
    the single-sample result; the same call twice is bit-identical (no atomics on the forward path)."""
    head = head_p4
    metas = O.synthetic_img_metas(1, 6, (640, 1600), seed=4)
    feats = torch.randn(1, 6, 256, 40, 100, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        a = {k_: v_.clone() for k_, v_ in head([feats], metas).items() if v_ is not None}
        b = head([feats], metas)
        assert torch.equal(a['all_cls_scores'], b['all_cls_scores']) and torch.equal(a['all_bbox_preds'], b['all_bbox_preds'])
        two = head([torch.cat([feats, feats], 0)], metas + metas)
    for key in ('all_cls_scores', 'all_bbox_preds'):
        assert relerr(two[key][:, 0], a[key][:, 0]) < 1e-5 and relerr(two[key][:, 1], a[key][:, 0]) < 1e-5
    assert torch.isfinite(a['all_cls_scores']).all() and torch.isfinite(a['all_bbox_preds']).all()


def test_head_full_size_backward_is_linear_in_the_upstream_gradient(head_p4):
    """d(params), d(feats) for the upstream gradient 2 g1 - g2 equal 2 x those for g1 minus those for g2 (eval mode, so the
    forward is the same function each time; float atomics in the weight gradients bound the tolerance)."""
    head = head_p4
    metas = O.synthetic_img_metas(1, 6, (640, 1600), seed=5)
    x = torch.randn(1, 6, 256, 40, 100, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
    g = torch.Generator().manual_seed(6)
    g1c, g1b, g2c, g2b = (torch.randn(6, 1, 900, 10, generator=g).cuda() for _ in range(4))

    def grads(gc, gb):
        head.zero_grad_flat()
        x.grad = None
        out = head([x], metas)
        torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [gc, gb])
        return head.flat_gradients().clone(), x.grad.clone()

    p1, f1 = grads(g1c, g1b)
    p2, f2 = grads(g2c, g2b)
    p3, f3 = grads(2.0 * g1c - g2c, 2.0 * g1b - g2b)
    assert relerr(f3, 2.0 * f1 - f2) < 1e-4
    assert relerr(p3, 2.0 * p1 - p2) < 1e-4
