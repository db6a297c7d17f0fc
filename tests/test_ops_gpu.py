"""GPU parity of every C-ABI operator against the CPU oracle / plain fp32 torch on the CPU."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import petr_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from petr_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def relerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


# ------------------------------------------------------------------ K1 coords3d
@pytest.mark.parametrize('name', ['coords3d_toy', 'coords3d_toy_masked'])
def test_coords3d_golden(ops, golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + '.npz'))
    N, H, W = [int(v) for v in fx['shape']]
    pad_h, pad_w = [int(v) for v in fx['pad_hw']]
    metas = [{'lidar2img': list(fx['lidar2img']), 'pad_shape': [(pad_h, pad_w, 3)] * N}]
    i2l = O.img2lidar_matrices(metas).reshape(-1, 16)
    depth = O.depth_bins(64, 1, [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], True)
    vol, cmask = ops.coords3d(dev(i2l), dev(depth), 1, N, H, W, pad_h, pad_w, [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                              want_mask=True)
    want = torch.from_numpy(fx['volume'])
    assert vol.shape == want.shape                       # layout [B*N, 3D, H, W], channel = 3d+axis
    # index space exact: compare in the normalised (pre-logit) space where the tolerance is meaningful
    got_n = torch.sigmoid(vol.cpu().double())
    want_n = torch.sigmoid(want.double())
    assert (got_n - want_n).abs().max().item() < 1e-5
    # logit values: tight away from the clamps
    inner = (want.abs() < 8)
    assert (vol.cpu()[inner] - want[inner]).abs().max().item() < 2e-3
    geo = torch.from_numpy(fx['coords_mask']) & ~torch.from_numpy(fx['masks']) | torch.from_numpy(fx['coords_mask'])
    got_mask = cmask.cpu() | torch.from_numpy(fx['masks'])
    # index space: the kernel reproduces the reference's fp32 operation order (k-ordered sum of rounded products, then
    # (x - lo) / (hi - lo), then the > 1 / < 0 tests), so the mask equals the reference's with ZERO flips
    assert torch.equal(got_mask, torch.from_numpy(fx['coords_mask']))
    del geo


def test_coords3d_c5_full(ops):
    metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=0)
    want, wmask, wnorm = O.coords3d_volume(1, 6, 16, 44, metas)
    i2l = O.img2lidar_matrices(metas).reshape(-1, 16)
    depth = O.depth_bins(64, 1, [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], True)
    vol, cmask = ops.coords3d(dev(i2l), dev(depth), 1, 6, 16, 44, 512, 1408, [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                              want_mask=True)
    assert (torch.sigmoid(vol.cpu().double()) - torch.sigmoid(want.double())).abs().max().item() < 1e-5
    assert torch.equal(cmask.cpu(), wmask)               # zero flips (same fp32 operation order as the reference)
    # determinism: same input twice -> bit-identical
    vol2, _ = ops.coords3d(dev(i2l), dev(depth), 1, 6, 16, 44, 512, 1408, [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0])
    assert torch.equal(vol, vol2)


@pytest.mark.parametrize('N,H,W,pad,batch', [(6, 40, 100, (640, 1600), 2), (12, 20, 50, (320, 800), 1), (6, 32, 88, (512, 1408), 1)])
def test_coords3d_full_size_configs(ops, N, H, W, pad, batch):
    """The coordinate volume at the shapes of BASELINE configs 3-5 (p4-1600 with batch 2, PETRv2 12 views, p4-1408) against
    the oracle, which is fast enough here: grid / mask / layout in index space, values in the normalised space."""
    rng = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]
    metas = O.synthetic_img_metas(batch, N, pad, seed=3)
    want, wmask, _ = O.coords3d_volume(batch, N, H, W, metas)
    i2l = O.img2lidar_matrices(metas).reshape(-1, 16)
    depth = O.depth_bins(64, 1, rng, True)
    vol, cmask = ops.coords3d(dev(i2l), dev(depth), batch, N, H, W, pad[0], pad[1], rng, want_mask=True)
    assert vol.shape == want.shape
    assert (torch.sigmoid(vol.cpu().double()) - torch.sigmoid(want.double())).abs().max().item() < 1e-5
    assert torch.equal(cmask.cpu(), wmask)                         # index work: bit-exact, no flips


# ------------------------------------------------------------------ K3 sine / posemb
def test_sine3d(ops, golden_dir):
    fx = np.load(os.path.join(golden_dir, 'sine3d.npz'))
    mask = torch.from_numpy(fx['mask'])
    dim_t = O.sine_dim_t(128)
    got = ops.sine3d(dev(mask), dev(dim_t), *mask.shape)
    assert (got.cpu() - torch.from_numpy(fx['pos'])).abs().max().item() < 2e-6
    got0 = ops.sine3d(None, dev(dim_t), 1, 6, 16, 44)
    want0 = O.sine_positional_encoding_3d(torch.zeros(1, 6, 16, 44, dtype=torch.bool), 128, normalize=True)
    assert (got0.cpu() - want0).abs().max().item() < 2e-6
    # the per-row kernel of the unmasked case returns the bits of the general (masked) kernel run on an all-zero mask
    for shape in ((1, 6, 16, 44), (2, 3, 40, 100), (1, 2, 5, 7)):
        zero = torch.zeros(shape, dtype=torch.bool)
        assert torch.equal(ops.sine3d(None, dev(dim_t), *shape), ops.sine3d(dev(zero), dev(dim_t), *shape))


def test_posemb3d_fwd_bwd(ops, golden_dir):
    fx = np.load(os.path.join(golden_dir, 'pos2posemb3d.npz'))
    pos = torch.from_numpy(fx['pos'])
    dim_t = O.sine_dim_t(128)
    got = ops.posemb3d(dev(pos), dev(dim_t))
    assert (got.cpu() - torch.from_numpy(fx['emb'])).abs().max().item() < 2e-6
    p = pos.clone().double().requires_grad_(True)
    g = torch.randn(64, 384, generator=torch.Generator().manual_seed(0))
    O.pos2posemb3d(p).backward(g.double())
    dpos = ops.posemb3d_bwd(dev(pos), dev(dim_t), dev(g))
    assert relerr(dpos, p.grad) < 1e-4


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K', [(900, 256, 256), (4224, 1024, 192), (5400, 10, 256), (37, 70, 44), (900, 256, 2048)])
def test_linear(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    want = torch.relu(x.double() @ w.double().T + b.double() + res.double())
    got = ops.linear(dev(x), dev(w), dev(b), relu=True, residual=dev(res))
    assert relerr(got, want) < 2e-6


def test_linear_addend_splitk_accumulate(ops):
    g = torch.Generator().manual_seed(7)
    x, e = torch.randn(2 * 90, 256, generator=g), torch.randn(90, 256, generator=g)
    w, b = torch.randn(768, 256, generator=g) / 16, torch.randn(768, generator=g)
    xe = x + e.repeat(2, 1)
    want = torch.cat([xe.double() @ w[:512].double().T, x.double() @ w[512:].double().T], 1) + b.double()
    got = ops.linear(dev(x), dev(w), dev(b), a2=dev(e), a2_rows=90, a2_ncols=512)
    assert relerr(got, want) < 2e-6
    # split-K partial slabs reduced by LayerNorm's prologue
    x2, w2 = torch.randn(900, 2048, generator=g), torch.randn(256, 2048, generator=g) / 45
    parts = ops.linear(dev(x2), dev(w2), split_k=4)
    assert relerr(parts.sum(0), x2.double() @ w2.double().T) < 2e-6
    acc = dev(torch.ones(900, 256))
    ops.linear(dev(x2), dev(w2), out=acc, accumulate=True)
    assert relerr(acc, x2.double() @ w2.double().T + 1) < 2e-6


def test_conv1x1_and_transposed_layouts(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 256, 704, generator=g)             # [views, C_in, HW]
    w, b = torch.randn(256, 256, generator=g) / 16, torch.randn(256, generator=g)
    want = torch.einsum('vch,oc->vho', x.double(), w.double()).reshape(-1, 256) + b.double()
    got = ops.conv1x1(dev(x), dev(w), dev(b))
    assert relerr(got, want) < 2e-6
    # backward forms: dA = dC @ W (B not K-contiguous), dW = dC^T @ A (neither K-contiguous)
    dC, A, W = torch.randn(900, 256, generator=g), torch.randn(900, 192, generator=g), torch.randn(256, 192, generator=g)
    dA = torch.empty(900, 192).cuda()
    ops.gemm_raw(a=dev(dC), lda=256, a_kcontig=1, b=dev(W), ldb=192, b_kcontig=0, c=dA, ldc=192, M=900, N=192, K=256, alpha=1.0)
    assert relerr(dA, dC.double() @ W.double()) < 2e-6
    dW = torch.empty(256, 192).cuda()
    ops.gemm_raw(a=dev(dC), lda=256, a_kcontig=0, b=dev(A), ldb=192, b_kcontig=0, c=dW, ldc=192, M=256, N=192, K=900, alpha=1.0)
    assert relerr(dW, dC.double().T @ A.double()) < 2e-6


def test_colsum(ops):
    x = torch.randn(5400, 300, generator=torch.Generator().manual_seed(1))
    assert relerr(ops.colsum(dev(x)), x.double().sum(0)) < 1e-6


# ------------------------------------------------------------------ LayerNorm
def test_layernorm_fwd_bwd(ops):
    g = torch.Generator().manual_seed(5)
    x, res, bias = torch.randn(900, 256, generator=g) * 3, torch.randn(900, 256, generator=g), torch.randn(256, generator=g)
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    zt = (x + bias + res).double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    want = torch.relu(torch.nn.functional.layer_norm(zt, (256,), gd, bd, 1e-5))
    y, z, mean, rstd = ops.layernorm(dev(x), dev(gamma), dev(beta), bias=dev(bias), residual=dev(res), relu=True,
                                     save_stats=True)
    assert relerr(y, want) < 2e-6 and relerr(z, zt) < 1e-6
    dy = torch.randn(900, 256, generator=g)
    want.backward(dy.double())
    dz, dgm, dbt = ops.layernorm_bwd(z, mean, rstd, dev(gamma), dev(dy), y=y, relu=True)
    assert relerr(dz, zt.grad) < 1e-5 and relerr(dgm, gd.grad) < 1e-5 and relerr(dbt, bd.grad) < 1e-5
    # nan_to_num epilogue (petr_head.py:435)
    xn = x.clone()
    xn[3, :] = float('nan')
    yn = ops.layernorm(dev(xn), dev(gamma), dev(beta), nan_to_num=True)
    assert torch.isfinite(yn).all() and (yn[3] == 0).all()


@pytest.mark.parametrize('G,rows,n_out,ln', [(1, 5400, 10, True), (1, 5400, 10, False), (6, 900, 10, True), (6, 900, 0, False),
                                             (2, 37, 7, True), (1, 16, 16, False)])
def test_branch_fwd(ops, G, rows, n_out, ln):
    """petr_branch_fwd = the class branch (Linear, LayerNorm, ReLU) x 2 + Linear / the box branch (Linear, ReLU) x 2 + Linear of
    reference petr_head.py:226-247 in one launch, against float64 torch; the rows the backward reads are checked too.  G = 6,
    rows = 900: PETRv2's deep-copied branches (900 % 32 != 0: ragged last block per group); n_out = 0: trunk only."""
    g = torch.Generator().manual_seed(100 + rows + n_out)
    r = lambda *s: torch.randn(*s, generator=g)
    x = r(G, rows, 256)
    w1, w2 = r(G, 256, 256) / 16, r(G, 256, 256) / 16
    b1, b2 = r(G, 256) * 0.1, r(G, 256) * 0.1
    w3, b3 = (r(G, n_out, 256) / 16, r(G, n_out) * 0.1) if n_out else (None, None)
    ln1 = (1 + 0.1 * r(G, 256), 0.1 * r(G, 256)) if ln else None
    ln2 = (1 + 0.1 * r(G, 256), 0.1 * r(G, 256)) if ln else None
    dv = lambda t: None if t is None else (tuple(u.cuda() for u in t) if isinstance(t, tuple) else t.cuda())
    res = ops.branch_fwd(dv(x), dv(w1), dv(b1), dv(w2), dv(b2), dv(w3), dv(b3), ln1=dv(ln1), ln2=dv(ln2))
    F = torch.nn.functional
    for gi in range(G):
        d = lambda t: t[gi].double()
        h1 = d(x) @ d(w1).T + d(b1)
        y1 = F.relu(F.layer_norm(h1, (256,), d(ln1[0]), d(ln1[1]), 1e-5)) if ln else F.relu(h1)
        h2 = y1 @ d(w2).T + d(b2)
        y2 = F.relu(F.layer_norm(h2, (256,), d(ln2[0]), d(ln2[1]), 1e-5)) if ln else F.relu(h2)
        want = {'y1': y1, 'y2': y2}
        if ln:
            want.update(h1=h1, h2=h2, mean1=h1.mean(-1), mean2=h2.mean(-1),
                        rstd1=1 / torch.sqrt(h1.var(-1, unbiased=False) + 1e-5), rstd2=1 / torch.sqrt(h2.var(-1, unbiased=False) + 1e-5))
        if n_out:
            want['out'] = y2 @ d(w3).T + d(b3)
        for k, v in want.items():
            got = res[k][gi].cpu().double()
            assert got.shape == v.shape, k
            err = (got - v).abs().max().item() / max(v.abs().max().item(), 1e-6)
            assert err < 5e-6, (gi, k, err)
    assert ('out' in res) == bool(n_out)


@pytest.mark.parametrize('G,rows,n_out,ln', [(1, 5400, 10, True), (1, 5400, 10, False), (6, 900, 10, True), (6, 900, 0, False),
                                             (2, 37, 7, True)])
def test_branch_bwd(ops, G, rows, n_out, ln):
    """petr_branch_bwd against float64 autograd through the same (Linear, [LayerNorm,] ReLU) x 2 [+ Linear] chain: the input
    gradient, the two pre-activation gradients the weight gradients are formed from, and the LayerNorm parameter gradients.
    n_out = 0: the gradient of the second activation is given (PETRv2: the RegLayer heads produce it)."""
    g = torch.Generator().manual_seed(200 + rows + n_out)
    r = lambda *s: torch.randn(*s, generator=g)
    x = r(G, rows, 256)
    w1, w2 = r(G, 256, 256) / 16, r(G, 256, 256) / 16
    b1, b2 = r(G, 256) * 0.1, r(G, 256) * 0.1
    w3, b3 = (r(G, n_out, 256) / 16, r(G, n_out) * 0.1) if n_out else (None, None)
    ln1 = (1 + 0.1 * r(G, 256), 0.1 * r(G, 256)) if ln else None
    ln2 = (1 + 0.1 * r(G, 256), 0.1 * r(G, 256)) if ln else None
    d_top = r(G, rows, n_out) if n_out else r(G, rows, 256)
    dv = lambda t: None if t is None else (tuple(u.cuda() for u in t) if isinstance(t, tuple) else t.cuda())
    fwd = ops.branch_fwd(dv(x), dv(w1), dv(b1), dv(w2), dv(b2), dv(w3), dv(b3), ln1=dv(ln1), ln2=dv(ln2))
    res = ops.branch_bwd(fwd, dv(w1), dv(w2), dv(w3), d_out=dv(d_top) if n_out else None, d_y2=None if n_out else dv(d_top),
                         ln1=dv(ln1), ln2=dv(ln2))
    F = torch.nn.functional
    for gi in range(G):
        d = lambda t: t[gi].double()
        xd = d(x).requires_grad_(True)
        gam1, gam2 = (d(ln1[0]).requires_grad_(True), d(ln2[0]).requires_grad_(True)) if ln else (None, None)
        bet1, bet2 = (d(ln1[1]).requires_grad_(True), d(ln2[1]).requires_grad_(True)) if ln else (None, None)
        # ReLU as a multiplication by the mask the KERNEL saw (y > 0 of its own fp32 activations): a pre-activation within an
        # ulp of zero may fall on the other side in float64, and one flipped mask element changes its whole LayerNorm row
        m1, m2 = (fwd['y1'][gi] > 0).cpu().double(), (fwd['y2'][gi] > 0).cpu().double()
        h1 = xd @ d(w1).T + d(b1)
        h1.retain_grad()
        y1 = (F.layer_norm(h1, (256,), gam1, bet1, 1e-5) if ln else h1) * m1
        h2 = y1 @ d(w2).T + d(b2)
        h2.retain_grad()
        y2 = (F.layer_norm(h2, (256,), gam2, bet2, 1e-5) if ln else h2) * m2
        w3d, b3d = (d(w3).requires_grad_(True), d(b3).requires_grad_(True)) if n_out else (None, None)
        top = (y2 @ w3d.T + b3d) if n_out else y2
        top.backward(d(d_top))
        want = {'d_x': xd.grad, 'd_h1': h1.grad, 'd_h2': h2.grad}
        if n_out:
            want.update(dw3=w3d.grad, db3=b3d.grad)
        if ln:
            want.update(dg1=gam1.grad, dbe1=bet1.grad, dg2=gam2.grad, dbe2=bet2.grad)
        for k, v in want.items():
            got = res[k][gi].cpu().double()
            assert got.shape == v.shape, k
            err = (got - v).abs().max().item() / max(v.abs().max().item(), 1e-6)
            assert err < (2e-5 if k.startswith('d_') else 1e-4), (gi, k, err)      # parameter sums: 5 400 float atomics


@pytest.mark.parametrize('G,rows', [(6, 900), (1, 70), (2, 33)])
def test_task_heads_fwd_bwd(ops, G, rows):
    """petr_task_heads_fwd / _bwd: the 1-3 column Linear that ends each of PETRv2's five RegLayer heads (petrv2_head.py:81-95), all
    heads and groups in one launch, against float64: outputs concatenated at columns (0, 2, 3, 6, 8); input gradient with the ReLU
    in front folded in; weight / bias gradients."""
    from petr_amd import _C
    import ctypes as C
    dims, cols, H = [2, 1, 3, 2, 2], [0, 2, 3, 6, 8], 5
    g = torch.Generator().manual_seed(40 + rows)
    h = torch.relu(torch.randn(G, H, rows, 256, generator=g))            # hidden activations (post-ReLU: zeros included)
    slot = 3 * 256 + 8                                                    # per-head parameter slot: [3 x 256] weights, then the bias
    params = torch.zeros(G, H, slot)
    for t in range(H):
        params[:, t, :dims[t] * 256] = torch.randn(G, dims[t] * 256, generator=g) / 16
        params[:, t, 768:768 + dims[t]] = torch.randn(G, dims[t], generator=g)
    d_out = torch.randn(G, rows, 10, generator=g)
    hd, pd, dd = h.cuda(), params.cuda(), d_out.cuda()
    out = torch.full((G, rows, 10), float('nan'), device='cuda')
    fa = _C.TaskHeadsFwdArgs(hd.data_ptr(), pd.data_ptr(), pd.data_ptr() + 4 * 768, H * slot, slot, out.data_ptr(), 10, rows, G, H,
                             (C.c_int * 8)(*dims, 0, 0, 0), (C.c_int * 8)(*cols, 0, 0, 0))
    _C.check(_C.lib().petr_task_heads_fwd(C.byref(fa), ops._stream()), 'petr_task_heads_fwd')
    d_h = torch.empty_like(hd)
    gp = torch.zeros_like(pd)
    ba = _C.TaskHeadsBwdArgs(dd.data_ptr(), 10, hd.data_ptr(), pd.data_ptr(), H * slot, slot, d_h.data_ptr(), gp.data_ptr(),
                             gp.data_ptr() + 4 * 768, rows, G, H, (C.c_int * 8)(*dims, 0, 0, 0), (C.c_int * 8)(*cols, 0, 0, 0))
    _C.check(_C.lib().petr_task_heads_bwd(C.byref(ba), ops._stream()), 'petr_task_heads_bwd')
    for gi in range(G):
        for t in range(H):
            w = params[gi, t, :dims[t] * 256].view(dims[t], 256).double()
            b = params[gi, t, 768:768 + dims[t]].double()
            hh = h[gi, t].double()
            want = hh @ w.T + b
            assert relerr(out[gi, :, cols[t]:cols[t] + dims[t]], want) < 1e-5
            dy = d_out[gi, :, cols[t]:cols[t] + dims[t]].double()
            assert relerr(d_h[gi, t], (dy @ w) * (hh > 0)) < 1e-5
            assert relerr(gp[gi, t, :dims[t] * 256].view(dims[t], 256), dy.T @ hh) < 2e-5
            assert relerr(gp[gi, t, 768:768 + dims[t]], dy.sum(0)) < 2e-5
    assert not torch.isnan(out).any()


def test_wgrad_grouped(ops):
    """petr_wgrad_grouped: the weight / bias gradients of one decoder layer's linear maps in one launch (strided operand
    views as the executor passes them, ragged K = 900, a K-split item with atomics) against fp64, accumulating (+=)."""
    g = torch.Generator().manual_seed(21)
    K = 900
    qkv = torch.randn(K, 768, generator=g)
    xs = [torch.randn(K, n, generator=g) for n in (256, 2048, 256)]
    dys = [qkv[:, :512], torch.randn(K, 256, generator=g), torch.randn(K, 2048, generator=g), qkv[:, 512:]]
    xin = [xs[0], xs[1], xs[2], xs[0]]
    big_dy, big_x = torch.randn(4224, 256, generator=g), torch.randn(4224, 192, generator=g)
    items, want = [], []
    qkv_d = dev(qkv)
    dyd = [qkv_d[:, :512], dev(dys[1]), dev(dys[2]), qkv_d[:, 512:]]
    xd = [dev(x) for x in xs]
    xind = [xd[0], xd[1], xd[2], xd[0]]
    for dy, x, dy_d, x_d, with_b in zip(dys, xin, dyd, xind, (True, True, False, True)):
        dw0 = torch.randn(dy.shape[1], x.shape[1], generator=g)
        db0 = torch.randn(dy.shape[1], generator=g)
        dw, db = dev(dw0), dev(db0)
        items.append((dy_d, x_d, dw, db if with_b else None, 1))
        want.append((dw0.double() + dy.double().t() @ x.double(), db0.double() + dy.double().sum(0), with_b))
    dwb, dbb = torch.zeros(256, 192).cuda(), torch.zeros(256).cuda()
    items.append((dev(big_dy), dev(big_x), dwb, dbb, 8))
    want.append((big_dy.double().t() @ big_x.double(), big_dy.double().sum(0), True))
    ops.wgrad_grouped(items)
    for (dy_d, x_d, dw, db, _), (w, b, with_b) in zip(items, want):
        assert relerr(dw, w) < 2e-6
        if with_b:
            assert relerr(db, b) < 2e-6
    with pytest.raises(RuntimeError, match='multiples of 64'):
        ops.wgrad_grouped([(dev(torch.randn(64, 10)), dev(torch.randn(64, 64)), torch.zeros(10, 64).cuda(), None, 1)])


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, kpm, scale):
    s = torch.einsum('bhqd,bhkd->bhqk', q.double(), k.double()) * scale
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    p = torch.softmax(s, -1)
    return torch.einsum('bhqk,bhkd->bhqd', p, v.double()), torch.logsumexp(s, -1)


@pytest.mark.parametrize('B,H,Q,L,split,masked', [(1, 8, 900, 4224, 0, False), (2, 8, 900, 900, 0, False),
                                                   (1, 2, 70, 333, 3, True), (1, 8, 900, 1000, 1, True),
                                                   (1, 1, 16, 40, 0, True)])
def test_mha_fwd(ops, B, H, Q, L, split, masked):
    g = torch.Generator().manual_seed(Q + L)
    q, k, v = (torch.randn(B, H, n, 32, generator=g) for n in (Q, L, L))
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - L // 4:] = True
        kpm[0, 3] = True
    want, wlse = _attn_ref(q, k, v, kpm, 32 ** -0.5)
    o, lse = ops.mha_fwd(dev(q), dev(k), dev(v), dev(kpm) if masked else None, n_split=split)
    assert relerr(o, want) < 2e-6
    assert (lse.cpu().double() - wlse).abs().max().item() < 1e-5


def test_mha_fwd_strided_views_and_all_masked(ops):
    """q from a [B,Q,768] projection buffer, k/v head-split: the layouts the executor uses."""
    g = torch.Generator().manual_seed(11)
    B, H, Q, L = 2, 8, 100, 200
    qkv = torch.randn(B, Q, 768, generator=g)
    kk, vv = torch.randn(B, H, L, 32, generator=g), torch.randn(B, H, L, 32, generator=g)
    qd = dev(qkv)
    qv = qd.view(B, Q, 24, 32)[:, :, :8].permute(0, 2, 1, 3)
    kpm = torch.zeros(B, L, dtype=torch.bool)
    kpm[1, :] = True                                   # every key of sample 1 masked -> NaN like torch
    want, _ = _attn_ref(qkv.view(B, Q, 24, 32)[:, :, :8].permute(0, 2, 1, 3), kk, vv, kpm, 32 ** -0.5)
    o, _ = ops.mha_fwd(qv, dev(kk), dev(vv), dev(kpm))
    assert relerr(o[0], want[0]) < 2e-6
    assert torch.isnan(o[1]).all() and torch.isnan(want[1]).all()


# Shapes: c5 / self-attention / ragged, plus one per branch of the query-split planner of the atomic form and the
# (batch 2, p4-1408, v2-800, p4-1600) key counts the executor runs: every BASELINE config's backward is compared with fp64.
@pytest.mark.parametrize('B,H,Q,L,masked', [(1, 8, 900, 4224, False), (1, 2, 70, 333, True), (2, 8, 900, 900, False),
                                            (2, 8, 900, 4224, False), (1, 8, 900, 12000, False), (1, 8, 900, 16896, True),
                                            (1, 8, 900, 24000, False), (1, 8, 300, 4224, False)])
def test_mha_bwd(ops, B, H, Q, L, masked):
    g = torch.Generator().manual_seed(Q * 3 + L)
    q, k, v = (torch.randn(B, H, n, 32, generator=g).double().requires_grad_(True) for n in (Q, L, L))
    do = torch.randn(B, H, Q, 32, generator=g)
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - L // 4:] = True
    want, _ = _attn_ref(q, k, v, kpm, 32 ** -0.5)
    want.backward(do.double())
    qf, kf, vf = (dev(t.detach().float()) for t in (q, k, v))
    o, lse = ops.mha_fwd(qf, kf, vf, dev(kpm) if masked else None)
    dq, dk, dv = ops.mha_bwd(qf, kf, vf, o, dev(do), lse, dev(kpm) if masked else None)
    assert relerr(dq, q.grad) < 1e-5 and relerr(dk, k.grad) < 1e-5 and relerr(dv, v.grad) < 1e-5


# ------------------------------------------------------------------ box epilogue
def test_bbox_epilogue(ops):
    g = torch.Generator().manual_seed(9)
    Q, rows = 50, 6 * 2 * 50
    reg = torch.randn(rows, 10, generator=g).double().requires_grad_(True)
    ref = torch.rand(Q, 3, generator=g).double().requires_grad_(True)
    pc = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    t = reg.clone().view(12, Q, 10)
    r = O.inverse_sigmoid(ref)
    xy = torch.sigmoid(t[..., 0:2] + r[..., 0:2])
    z = torch.sigmoid(t[..., 4:5] + r[..., 2:3])
    want = torch.cat([xy[..., 0:1] * (pc[3] - pc[0]) + pc[0], xy[..., 1:2] * (pc[4] - pc[1]) + pc[1], t[..., 2:4],
                      z * (pc[5] - pc[2]) + pc[2], t[..., 5:]], -1).view(rows, 10)
    got = ops.bbox_epilogue(dev(reg.detach().float()), dev(ref.detach().float()), Q, pc)
    assert relerr(got, want) < 2e-6
    dout = torch.randn(rows, 10, generator=g)
    want.backward(dout.double())
    dreg, dref = ops.bbox_epilogue_bwd(got, dev(ref.detach().float()), dev(dout), Q, pc)
    assert relerr(dreg, reg.grad) < 1e-5 and relerr(dref, ref.grad) < 1e-4


# ------------------------------------------------------------------ dropout (training mode)
# The kernels draw their masks from a counter-based hash of (seed, site, row, col) (include/petr_hip.h "Dropout"),
# which cannot equal torch's Philox stream: every test exports the mask with petr_dropout_mask and gives the SAME
# mask to the fp64 torch restatement.  Scale = 1/(1-p) with p rounded to a 32-bit threshold (relative 1e-9).
def test_dropout_mask_is_a_fair_deterministic_coin(ops):
    p = 0.1
    m0 = ops.dropout_mask((1234, 3, p), 900, 4224)
    m0b = ops.dropout_mask((1234, 3, p), 900, 4224)
    m1 = ops.dropout_mask((1234, 4, p), 900, 4224)       # another site
    m2 = ops.dropout_mask((1235, 3, p), 900, 4224)       # another seed
    assert torch.equal(m0, m0b)
    n = m0.numel()
    sigma = math.sqrt(p * (1 - p) / n)
    for m in (m0, m1, m2):
        assert abs((1 - m.float().mean().item()) - p) < 5 * sigma
    # independent masks agree on a fraction p^2 + (1-p)^2 of the elements
    for a, b in ((m0, m1), (m0, m2)):
        agree = (a == b).float().mean().item()
        assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.002
    # no structure along rows / columns (every row and column drops about p)
    assert (1 - m0.float().mean(1)).sub(p).abs().max().item() < 6 * math.sqrt(p * (1 - p) / 4224)
    assert (1 - m0.float().mean(0)).sub(p).abs().max().item() < 6 * math.sqrt(p * (1 - p) / 900)
    assert ops.dropout_mask((7, 0, 0.0), 10, 10).all()
    # neighbours are uncorrelated: the two columns that share one hash, adjacent pairs, adjacent rows, and the
    # (row+1, col+1) diagonal all agree on p^2 + (1-p)^2 of the elements
    f = m0.float()
    want = p * p + (1 - p) * (1 - p)
    for a, b in ((f[:, 0::2], f[:, 1::2]), (f[:, 1:-1:2], f[:, 2::2]), (f[:-1], f[1:]), (f[:-1, :-1], f[1:, 1:]),
                 (f[:, :-64], f[:, 64:]), (f[:-32], f[32:])):
        assert abs((a == b).float().mean().item() - want) < 0.002
    # realised rate: round(p * 2^16) / 2^16
    assert abs((1 - f.mean().item()) - 6554 / 65536) < 5 * sigma


def test_linear_relu_dropout(ops):
    g = torch.Generator().manual_seed(11)
    x, w, b = torch.randn(900, 256, generator=g), torch.randn(2048, 256, generator=g) * 0.1, torch.randn(2048, generator=g)
    drop = (99, 44, 0.1)
    keep = ops.dropout_mask(drop, 900, 2048).cpu()
    want = torch.relu(x.double() @ w.double().t() + b.double()) * keep / (1 - 6554 / 65536)
    got = ops.linear(dev(x), dev(w), dev(b), relu=True, drop=drop)
    assert relerr(got, want) < 2e-6
    assert ((got == 0).cpu() | keep).all()


def test_layernorm_dropout_fwd_bwd(ops):
    g = torch.Generator().manual_seed(6)
    parts = torch.randn(4, 900, 256, generator=g)
    res, bias = torch.randn(900, 256, generator=g), torch.randn(256, generator=g)
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    drop = (5, 13, 0.1)
    keep = ops.dropout_mask(drop, 900, 256).cpu().double() / (1 - 6554 / 65536)
    f = (parts.double().sum(0) + bias.double()).requires_grad_(True)     # sub-layer output
    r = res.double().requires_grad_(True)                                 # identity
    zt = f * keep + r
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    want = torch.nn.functional.layer_norm(zt, (256,), gd, bd, 1e-5)
    y, z, mean, rstd = ops.layernorm(dev(parts), dev(gamma), dev(beta), bias=dev(bias), residual=dev(res), save_stats=True,
                                     drop=drop)
    assert relerr(y, want) < 2e-6 and relerr(z, zt) < 1e-6
    dy = torch.randn(900, 256, generator=g)
    want.backward(dy.double())
    dz, dgm, dbt, dzd = ops.layernorm_bwd(z, mean, rstd, dev(gamma), dev(dy), drop=drop)
    assert relerr(dz, r.grad) < 1e-5 and relerr(dzd, f.grad) < 1e-5
    assert relerr(dgm, gd.grad) < 1e-5 and relerr(dbt, bd.grad) < 1e-5


@pytest.mark.parametrize('B,H,Q,L,split,masked', [(1, 8, 900, 1000, 0, False), (2, 2, 70, 333, 3, True), (1, 8, 130, 4224, 8, False),
                                                   (1, 8, 900, 12000, 0, False), (1, 4, 900, 16896, 0, True)])
def test_mha_dropout_fwd_bwd(ops, B, H, Q, L, split, masked):
    g = torch.Generator().manual_seed(Q * 7 + L)
    q, k, v = (torch.randn(B, H, n, 32, generator=g).double().requires_grad_(True) for n in (Q, L, L))
    do = torch.randn(B, H, Q, 32, generator=g)
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - L // 4:] = True
    p = 6554 / 65536          # how the kernels realise 0.1
    drop = (2024, 2, 0.1)
    keep = ops.dropout_mask(drop, B * H * Q, L).cpu().view(B, H, Q, L)
    s = torch.einsum('bhqd,bhkd->bhqk', q, k) * 32 ** -0.5
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    want = torch.einsum('bhqk,bhkd->bhqd', torch.softmax(s, -1) * keep / (1 - p), v)
    want.backward(do.double())
    qf, kf, vf = (dev(t.detach().float()) for t in (q, k, v))
    o, lse = ops.mha_fwd(qf, kf, vf, dev(kpm) if masked else None, n_split=split, drop=drop)
    assert relerr(o, want) < 2e-6
    assert relerr(lse, torch.logsumexp(s, -1)) < 2e-6                     # the LSE is the undropped softmax's
    dq, dk, dv = ops.mha_bwd(qf, kf, vf, o, dev(do), lse, dev(kpm) if masked else None, drop=drop)
    assert relerr(dq, q.grad) < 1e-5 and relerr(dk, k.grad) < 1e-5 and relerr(dv, v.grad) < 1e-5
    # ... and with the packed keep bits the forward leaves (what the executor's backward reads)
    bits = torch.zeros_like(ops.dropout_bits(drop, B * H, Q, L)[1])
    ops.mha_fwd(qf, kf, vf, dev(kpm) if masked else None, n_split=split, drop=drop, drop_bits=bits)
    dq, dk, dv = ops.mha_bwd(qf, kf, vf, o, dev(do), lse, dev(kpm) if masked else None, drop=drop, drop_bits=bits)
    assert relerr(dq, q.grad) < 1e-5 and relerr(dk, k.grad) < 1e-5 and relerr(dv, v.grad) < 1e-5
    # p = 0 is the dropout-free kernel
    o0, _ = ops.mha_fwd(qf, kf, vf, dev(kpm) if masked else None, n_split=split, drop=(2024, 2, 0.0))
    o1, _ = ops.mha_fwd(qf, kf, vf, dev(kpm) if masked else None, n_split=split)
    assert torch.equal(o0, o1)


# ------------------------------------------------------------------ attention: ragged / tiny shapes
@pytest.mark.parametrize('B,H,Q,L,split', [(1, 1, 1, 1, 0), (2, 3, 31, 31, 0), (1, 2, 33, 65, 2), (3, 1, 129, 129, 0),
                                            (1, 4, 5, 513, 8), (2, 2, 257, 63, 0), (1, 8, 64, 4224, 11), (1, 1, 200, 97, 2)])
@pytest.mark.parametrize('masked,drop', [(False, None), (True, (77, 9, 0.3))])
def test_mha_ragged_shapes(ops, B, H, Q, L, split, masked, drop):
    """Every combination of ragged query blocks, ragged / single key tiles, idle waves, more splits than tiles,
    key-padding masks and probability dropout, forward and backward, against fp64 torch."""
    g = torch.Generator().manual_seed(B * 1000 + Q * 7 + L)
    q, k, v = (torch.randn(B, H, n, 32, generator=g).double().requires_grad_(True) for n in (Q, L, L))
    do = torch.randn(B, H, Q, 32, generator=g)
    kpm = None
    if masked and L > 1:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - max(1, L // 3):] = True
        kpm[0, 0] = True
    s = torch.einsum('bhqd,bhkd->bhqk', q, k) * 32 ** -0.5
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    p = torch.softmax(s, -1)
    if drop is not None:
        keep = ops.dropout_mask(drop, B * H * Q, L).cpu().view(B, H, Q, L)
        p = p * keep / (1 - round(drop[2] * 65536) / 65536)
    want = torch.einsum('bhqk,bhkd->bhqd', p, v)
    want.backward(do.double())
    qf, kf, vf = (dev(t.detach().float()) for t in (q, k, v))
    kw = dict(drop=drop) if drop is not None else {}
    o, lse = ops.mha_fwd(qf, kf, vf, dev(kpm) if kpm is not None else None, n_split=split, **kw)
    assert relerr(o, want) < 5e-6
    dq, dk, dv = ops.mha_bwd(qf, kf, vf, o, dev(do), lse, dev(kpm) if kpm is not None else None, **kw)

    def err(got, want_):     # a single key makes dq / dk identically zero: relative to the gradient scale, not to 0
        got, want_ = got.double().cpu(), want_.double()
        return ((got - want_).abs().max() / max(want_.abs().max().item(), 1e-2)).item()
    assert err(dq, q.grad) < 2e-5 and err(dk, k.grad) < 2e-5 and err(dv, v.grad) < 2e-5
    # the ticket scheduler is another assignment of the same tiles
    o2, _ = ops.mha_fwd(qf, kf, vf, dev(kpm) if kpm is not None else None, n_split=split, dynamic=True, **kw)
    assert relerr(o2, want) < 5e-6


# ------------------------------------------------------------------ attention, bf16 K/V (BASELINE configs 3-5)
# Tolerances ("tolerance vs fp32 oracle stated separately, expect ~1e-2 rel"):
#   BF16_TOL_SAME_INPUTS  against the fp64 attention of the SAME bf16-rounded K/V: what is left is the rounding of the
#                         pre-scaled Q and of the probabilities to bf16 (8 mantissa bits each), fp32 everywhere else
#   BF16_TOL_FP32_ORACLE  against the fp64 attention of the unrounded fp32 inputs (adds the K/V rounding itself)
# both as max |error| / max |reference|; the LSE is compared absolutely (it is a log).
BF16_TOL_SAME_INPUTS = 1e-2     # measured 4e-3 ... 7e-3 over the cases below
BF16_TOL_FP32_ORACLE = 1.5e-2
BF16_TOL_LSE_ABS = 3e-2


@pytest.mark.parametrize('B,H,Q,L,split,masked,drop', [
    (1, 8, 900, 4224, 0, False, None), (1, 8, 900, 8800, 0, False, None), (1, 8, 900, 4224, 0, True, (31, 2, 0.1)),
    (2, 8, 900, 900, 0, False, None), (1, 2, 70, 333, 3, True, None), (1, 1, 1, 1, 0, False, None),
    (2, 3, 31, 65, 2, False, (5, 1, 0.3)), (1, 4, 5, 513, 8, True, None), (3, 1, 129, 129, 0, False, None),
    (1, 8, 64, 4224, 11, False, None), (1, 1, 200, 97, 2, True, (9, 9, 0.1))])
def test_mha_fwd_bf16(ops, B, H, Q, L, split, masked, drop):
    g = torch.Generator().manual_seed(3 * Q + L)
    q, k, v = (torch.randn(B, H, n, 32, generator=g) for n in (Q, L, L))
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - L // 4:] = True
        kpm[0, min(3, L - 1)] = L > 4
    kb, vb = ops.cast_bf16(dev(k)), ops.cast_bf16(dev(v))
    assert torch.equal(kb.cpu(), k.to(torch.bfloat16)) and torch.equal(vb.cpu(), v.to(torch.bfloat16))   # RNE, bit-exact
    o, lse = ops.mha_fwd_bf16(dev(q), kb, vb, dev(kpm) if masked else None, n_split=split, drop=drop)
    scale = 32 ** -0.5
    keep = 1.0
    if drop is not None:
        p_real = round(drop[2] * 65536) / 65536
        keep = ops.dropout_mask(drop, B * H * Q, L).cpu().view(B, H, Q, L).double() / (1 - p_real)

    def ref(kk, vv):
        s = torch.einsum('bhqd,bhkd->bhqk', q.double(), kk.double()) * scale
        if kpm is not None:
            s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
        return torch.einsum('bhqk,bhkd->bhqd', torch.softmax(s, -1) * keep, vv.double()), torch.logsumexp(s, -1)

    same, same_lse = ref(k.to(torch.bfloat16), v.to(torch.bfloat16))
    full, _ = ref(k, v)
    assert relerr(o, same) < BF16_TOL_SAME_INPUTS, relerr(o, same)
    assert relerr(o, full) < BF16_TOL_FP32_ORACLE, relerr(o, full)
    assert (lse.cpu().double() - same_lse).abs().max().item() < BF16_TOL_LSE_ABS


def test_mha_fwd_bf16_head_split_views_and_errors(ops):
    """K/V as head-split views of one [B, L, 256] bf16 projection buffer (row stride 256), q from a [B,Q,768] buffer."""
    g = torch.Generator().manual_seed(17)
    B, H, Q, L = 2, 8, 100, 300
    qkv = torch.randn(B, Q, 768, generator=g)
    kbuf, vbuf = torch.randn(B, L, 256, generator=g), torch.randn(B, L, 256, generator=g)
    kb, vb = ops.cast_bf16(dev(kbuf)), ops.cast_bf16(dev(vbuf))
    qv = dev(qkv).view(B, Q, 24, 32)[:, :, :8].permute(0, 2, 1, 3)
    kv, vv = (t.view(B, L, 8, 32).permute(0, 2, 1, 3) for t in (kb, vb))
    o, _ = ops.mha_fwd_bf16(qv, kv, vv)
    want, _ = _attn_ref(qkv.view(B, Q, 24, 32)[:, :, :8].permute(0, 2, 1, 3),
                        kbuf.to(torch.bfloat16).view(B, L, 8, 32).permute(0, 2, 1, 3),
                        vbuf.to(torch.bfloat16).view(B, L, 8, 32).permute(0, 2, 1, 3), None, 32 ** -0.5)
    assert relerr(o, want) < BF16_TOL_SAME_INPUTS
    with pytest.raises(RuntimeError, match='16-byte aligned'):      # rows that are not 16-byte aligned are refused, loudly
        odd = torch.zeros(B, 8, L, 36, dtype=torch.bfloat16, device='cuda')[..., 4:]
        ops.mha_fwd_bf16(qv, odd, odd)


def test_gemm_store_bf16_epilogue(ops):
    """PETR_GEMM_STORE_BF16: the contraction's epilogue stores bf16 (round to nearest even) - what the K/V projections of
    the bf16 attention use.  Bit-exact against rounding the SAME kernel's fp32 output; head-split layout and a ragged
    shape; accumulate / split-K are refused."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(23)
    for M, N, K in ((4224, 256, 256), (333, 96, 70)):
        x, w, b = (torch.randn(s, generator=g).cuda() for s in ((M, K), (N, K), (N,)))
        ref = ops.linear(x, w, b)                                    # 4224 x 256 goes through the tiled kernel as well
        out32 = torch.empty(M, N, device='cuda')
        ops.gemm_raw(a=x, lda=K, a_kcontig=1, b=w, ldb=K, b_kcontig=1, c=out32, ldc=N, M=M, N=N, K=K, bias=b, flags=0,
                     alpha=1.0, nb0=1, nb1=1)
        out16 = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
        ops.gemm_raw(a=x, lda=K, a_kcontig=1, b=w, ldb=K, b_kcontig=1, c=out16, ldc=N, M=M, N=N, K=K, bias=b,
                     flags=_C.GEMM_STORE_BF16, alpha=1.0, nb0=1, nb1=1)
        assert relerr(out32, ref) < 1e-5
        assert torch.equal(out16, ref.to(torch.bfloat16)) or torch.equal(out16, out32.to(torch.bfloat16))
    with pytest.raises(RuntimeError, match='STORE_BF16'):
        ops.gemm_raw(a=x, lda=K, a_kcontig=1, b=w, ldb=K, b_kcontig=1, c=out16, ldc=N, M=M, N=N, K=K,
                     flags=_C.GEMM_STORE_BF16 | _C.GEMM_ACCUMULATE, alpha=1.0)


@pytest.mark.parametrize('M,N,K,a2,store16', [(4224, 256, 256, True, True), (900, 2048, 256, False, False),
                                              (333, 96, 80, False, False), (70, 256, 1024, True, False)])
def test_gemm_bf16_route(ops, M, N, K, a2, store16):
    """PETR_GEMM_BF16: operands rounded to bf16 on load, bf16 MFMA, fp32 accumulation (BASELINE configs 3-5).
    Checked tightly against the fp64 product of the SAME bf16-rounded operands (+ fp32 bias / residual / ReLU), loosely
    (bf16 accuracy) against the unrounded one; with the key_pos-style addend and the bf16 store."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(M + K)
    x, w, b, r, x2 = (torch.randn(s, generator=g) for s in ((M, K), (N, K), (N,), (M, N), (M, K)))
    flags = _C.GEMM_BF16 | _C.GEMM_RELU | (_C.GEMM_STORE_BF16 if store16 else 0)
    out = torch.zeros(M, N, dtype=torch.bfloat16 if store16 else torch.float32, device='cuda')
    kw = dict(a=x.cuda(), lda=K, a_kcontig=1, b=w.cuda(), ldb=K, b_kcontig=1, c=out, ldc=N, M=M, N=N, K=K, bias=b.cuda(),
              r=r.cuda(), ldr=N, flags=flags, alpha=1.0, nb0=1, nb1=1)
    if a2:
        kw.update(a2=x2.cuda(), a2_rows=0, a2_ncols=0)
    ops.gemm_raw(**kw)
    xa = (x + x2) if a2 else x
    same = torch.relu(xa.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().T + b.double() + r.double())
    full = torch.relu(xa.double() @ w.double().T + b.double() + r.double())
    tol = 4e-3 if store16 else 2e-5             # the bf16 store adds one more rounding (2^-9 relative)
    assert relerr(out.float(), same) < tol, relerr(out.float(), same)
    assert relerr(out.float(), full) < 1.5e-2
    with pytest.raises(RuntimeError, match='PETR_GEMM_BF16'):                     # K-contiguous rows that are not 16-byte aligned
        ops.gemm_raw(a=x.cuda()[:, :K - 6].contiguous(), lda=K - 6, a_kcontig=1, b=w.cuda()[:, :K - 6].contiguous(), ldb=K - 6,
                     b_kcontig=1, c=torch.zeros(M, N, device='cuda'), ldc=N, M=M, N=N, K=K - 6, flags=_C.GEMM_BF16, alpha=1.0)


@pytest.mark.parametrize('V,Cin,HW,N', [(6, 256, 704, 256), (2, 192, 4000, 1024), (3, 384, 1000, 1024)])
def test_gemm_bf16_route_channel_major_input(ops, V, Cin, HW, N):
    """PETR_GEMM_BF16 with a K-major A operand: a 1x1 convolution reading an NCHW map [V][Cin][HW] (input_proj, the first
    position-embedding convolutions), batched over the views, token-major output.  Shapes: c5 (HW = 704), p4-1600
    (HW = 4 000), PETRv2 (HW = 1 000, not a multiple of 128: ragged row tiles)."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(V * HW)
    x, w, b = torch.randn(V, Cin, HW, generator=g), torch.randn(N, Cin, generator=g), torch.randn(N, generator=g)
    out = torch.zeros(V, HW, N, device='cuda')
    ops.gemm_raw(a=x.cuda(), lda=HW, a_kcontig=0, a_bs0=Cin * HW, b=w.cuda(), ldb=Cin, b_kcontig=1, c=out, ldc=N, c_bs0=HW * N,
                 bias=b.cuda(), M=HW, N=N, K=Cin, nb0=V, nb1=1, flags=_C.GEMM_BF16 | _C.GEMM_RELU, alpha=1.0)
    same = torch.relu(torch.einsum('vkm,nk->vmn', x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()) + b.double())
    full = torch.relu(torch.einsum('vkm,nk->vmn', x.double(), w.double()) + b.double())
    assert relerr(out, same) < 2e-5, relerr(out, same)
    assert relerr(out, full) < 1.5e-2


# ------------------------------------------------------------------ general bf16 contraction (gradient shapes)
def _bf(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize('M,N,K,akc,bkc', [(1000, 256, 1536, True, False), (256, 1024, 4224, False, False),
                                           (1024, 192, 2112, False, True), (333, 70, 200, True, True),
                                           (130, 260, 100, False, False), (704, 256, 256, False, True)])
def test_gemm_bf16_general_layouts(ops, M, N, K, akc, bkc):
    """PETR_GEMM_BF16 through gemm_bf16.hip: every operand-layout combination (K-contiguous / K-major A and B), ragged
    row and K tiles, with the input-gradient epilogues (ReLU mask, accumulate).  Tight against the fp64 product of the
    SAME bf16-rounded operands, loose (bf16 accuracy) against the unrounded product."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A, Bm, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    a_dev = A.cuda() if akc else A.t().contiguous().cuda()
    b_dev = Bm.cuda() if bkc else Bm.t().contiguous().cuda()
    same = _bf(A) @ _bf(Bm).T
    full = A.double() @ Bm.double().T
    base = dict(a=a_dev, lda=K if akc else M, a_kcontig=int(akc), b=b_dev, ldb=K if bkc else N, b_kcontig=int(bkc), ldc=N,
                M=M, N=N, K=K, alpha=1.0, nb0=1, nb1=1)
    # ReLU mask epilogue (dgrad through a ReLU): out = (r > 0) * product ; forced onto the general kernel
    out = torch.full((M, N), 7.0, device='cuda')
    ops.gemm_raw(c=out, r=r.cuda(), ldr=N, flags=_C.GEMM_BF16 | _C.GEMM_RELU_MASK, **base)
    assert relerr(out, same * (r > 0)) < 2e-5, relerr(out, same * (r > 0))
    assert relerr(out, full * (r > 0)) < 1.5e-2
    # accumulate epilogue with a bias
    bias = torch.randn(N, generator=g)
    out2 = r.cuda().clone()
    ops.gemm_raw(c=out2, bias=bias.cuda(), flags=_C.GEMM_BF16 | _C.GEMM_ACCUMULATE, **base)
    assert relerr(out2, same + bias.double() + r.double()) < 2e-5


@pytest.mark.parametrize('Mo,No,rows,split,seg', [(256, 256, 4224, 8, 0), (256, 1024, 3000, 5, 1000), (10, 256, 900, 3, 0),
                                                  (1024, 192, 2816, 16, 704)])
def test_gemm_bf16_weight_gradient(ops, Mo, No, rows, split, seg):
    """dW[Mo,No] += dY^T X with both operands K-major, K = rows cut into segments and into slices over workgroups,
    float-atomic accumulation, and the bias gradient as exact fp32 column sums of dY (a_colsum)."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(Mo + rows)
    dy, x = torch.randn(rows, Mo, generator=g), torch.randn(rows, No, generator=g)
    dw = torch.zeros(Mo, No, device='cuda')
    db = torch.zeros(Mo, device='cuda')
    kw = dict(a=dy.cuda(), lda=Mo, a_kcontig=0, b=x.cuda(), ldb=No, b_kcontig=0, c=dw, ldc=No, M=Mo, N=No, K=rows,
              flags=_C.GEMM_BF16 | _C.GEMM_ATOMIC, split_k=split, a_colsum=db, alpha=1.0, nb0=1, nb1=1)
    if seg:   # segments at different base addresses: here simply consecutive blocks of `seg` rows
        kw.update(k_seg=seg, a_seg_stride=seg * Mo, b_seg_stride=seg * No)
    ops.gemm_raw(**kw)
    same = _bf(dy).T @ _bf(x)
    assert relerr(dw, same) < 3e-5, relerr(dw, same)
    assert relerr(dw, dy.double().T @ x.double()) < 1.5e-2
    assert relerr(db, dy.double().sum(0)) < 1e-5            # column sums are taken from the fp32 values, not the rounded ones
    # slabs instead of atomics
    slabs = torch.zeros(split, Mo, No, device='cuda')
    kw.update(c=slabs, flags=_C.GEMM_BF16, c_split_stride=Mo * No, a_colsum=None)
    kw.pop('a_colsum')
    ops.gemm_raw(**kw)
    assert relerr(slabs.sum(0), same) < 3e-5


def test_gemm_bf16_batched_segments_and_nchw_output(ops):
    """The two batched gradient shapes of the head: (i) d_src[b][t][c] = sum_{l,o} dKV[b][l][t][o] W_l[o][c] (K-contiguous A
    in segments, K-major B in segments, batch over b); (ii) d_x[view][ci][hw] = sum_o W[o][ci] d_mem[view*HW+hw][o]
    (K-major A, K-contiguous B, batched NCHW output)."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(5)
    B, NL, Lt, Cc = 2, 3, 300, 256
    dkv, W = torch.randn(B, NL, Lt, Cc, generator=g), torch.randn(NL, Cc, Cc, generator=g)
    out = torch.zeros(B, Lt, Cc, device='cuda')
    ops.gemm_raw(a=dkv.cuda(), lda=Cc, a_kcontig=1, a_bs0=NL * Lt * Cc, b=W.cuda(), ldb=Cc, b_kcontig=0, c=out, ldc=Cc,
                 c_bs0=Lt * Cc, M=Lt, N=Cc, K=NL * Cc, nb0=B, nb1=1, k_seg=Cc, a_seg_stride=Lt * Cc, b_seg_stride=Cc * Cc,
                 flags=_C.GEMM_BF16, alpha=1.0)
    want = torch.einsum('blto,loc->btc', _bf(dkv), _bf(W))
    assert relerr(out, want) < 2e-5
    V, Cin, HW = 3, 192, 1000
    Wi, dmem = torch.randn(Cc, Cin, generator=g), torch.randn(V * HW, Cc, generator=g)
    dx = torch.zeros(V, Cin, HW, device='cuda')
    ops.gemm_raw(a=Wi.cuda(), lda=Cin, a_kcontig=0, b=dmem.cuda(), ldb=Cc, b_kcontig=1, b_bs0=HW * Cc, c=dx, ldc=HW,
                 c_bs0=Cin * HW, M=Cin, N=HW, K=Cc, nb0=V, nb1=1, flags=_C.GEMM_BF16, alpha=1.0)
    want = torch.einsum('oc,vho->vch', _bf(Wi), _bf(dmem).view(V, HW, Cc))
    assert relerr(dx, want) < 2e-5


# ------------------------------------------------------------------ bf16 attention backward
BF16_BWD_TOL = 2e-2      # max-norm relative error of dq / dk / dv against fp64 autograd on the same bf16-rounded K / V


@pytest.mark.parametrize('B,H,Q,L,masked,drop', [
    (1, 8, 900, 4224, False, None), (1, 8, 900, 4224, False, (3, 2, 0.1)), (1, 2, 70, 333, True, None),
    (2, 3, 31, 65, False, (5, 1, 0.3)), (1, 1, 1, 1, False, None), (1, 4, 129, 1000, True, (8, 4, 0.1)),
    (1, 2, 200, 700, True, None), (1, 8, 900, 12000, False, None),
    (1, 8, 900, 24000, False, None), (1, 8, 900, 24000, False, (3, 2, 0.1)), (1, 8, 900, 16896, True, (9, 1, 0.1))])
def test_mha_bwd_bf16(ops, B, H, Q, L, masked, drop):
    """petr_mha_bwd_bf16 (gradient of petr_mha_fwd_bf16) vs fp64 autograd through softmax attention on the same
    bf16-rounded K / V (and the same exported dropout mask)."""
    kt = 1
    g = torch.Generator().manual_seed(Q * 3 + L + kt)
    q, k, v = (torch.randn(B, H, n, 32, generator=g) for n in (Q, L, L))
    do = torch.randn(B, H, Q, 32, generator=g)
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - L // 4:] = True
        kpm[0, min(3, L - 1)] = L > 4
    kb, vb = ops.cast_bf16(dev(k)), ops.cast_bf16(dev(v))
    o, lse = ops.mha_fwd_bf16(dev(q), kb, vb, dev(kpm) if masked else None, drop=drop)
    dq, dk, dv = ops.mha_bwd_bf16(dev(q), kb, vb, o, dev(do), lse, dev(kpm) if masked else None, drop=drop)
    # the store variant (dkv_overwrite: what the executor uses) into NaN-poisoned buffers gives the same dK / dV
    _, dk2, dv2 = ops.mha_bwd_bf16(dev(q), kb, vb, o, dev(do), lse, dev(kpm) if masked else None, drop=drop, overwrite=True)
    assert torch.isfinite(dk2).all() and (dk2 - dk).abs().max().item() <= 1e-6 * (1 + dk.abs().max().item())
    assert torch.isfinite(dv2).all() and (dv2 - dv).abs().max().item() <= 1e-6 * (1 + dv.abs().max().item())
    # ... and with bf16 stores (dkv_bf16): exactly the round-to-nearest-even image of the fp32 result
    _, dk3, dv3 = ops.mha_bwd_bf16(dev(q), kb, vb, o, dev(do), lse, dev(kpm) if masked else None, drop=drop, overwrite=True,
                                   dkv_bf16=True)
    assert dk3.dtype == torch.bfloat16 and torch.equal(dk3, dk2.to(torch.bfloat16)) and torch.equal(dv3, dv2.to(torch.bfloat16))
    keep = 1.0
    if drop is not None:
        p_real = round(drop[2] * 65536) / 65536
        keep = ops.dropout_mask(drop, B * H * Q, L).cpu().view(B, H, Q, L).double() / (1 - p_real)
    qd = q.double().requires_grad_(True)
    kd = k.to(torch.bfloat16).double().requires_grad_(True)
    vd = v.to(torch.bfloat16).double().requires_grad_(True)
    s = torch.einsum('bhqd,bhkd->bhqk', qd, kd) * 32 ** -0.5
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    want = torch.einsum('bhqk,bhkd->bhqd', torch.softmax(s, -1) * keep, vd)
    want.backward(do.double())
    def err(got, want):      # max-norm error relative to max(|want|, 0.05): with ONE key the true dq / dk are exactly zero
        return (got.double().cpu() - want).abs().max().item() / max(want.abs().max().item(), 0.05)
    e = [err(dq, qd.grad), err(dk, kd.grad), err(dv, vd.grad)]
    print(f'mha_bwd_bf16 B{B} H{H} Q{Q} L{L}: dq {e[0]:.2e} dk {e[1]:.2e} dv {e[2]:.2e}')
    assert max(e) < BF16_BWD_TOL, e
    # size-independent identities: sum_k dV = sum_q (P*keep)^T dO has column sums equal to ... checked in test_properties
    assert torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all()


@pytest.mark.parametrize('a16,b16', [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_bf16_kmajor_weight_gradient_batched(ops, a16, b16):
    """The transposed-LDS-read kernel (both operands K-major, 128 x 128 tiles, float atomics): three batched weight gradients
    (the per-layer dW_k of the K projections) over K = 2 segments of 1 500 rows (ragged last K step), cut into 7 K slices,
    fp32 / bf16 sources, with the column sums of A."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(31 + 2 * a16 + b16)
    NB, Mo, No, seg, nseg = 3, 256, 384, 1500, 2
    dy = torch.randn(nseg, NB, seg, Mo, generator=g)          # [segment][batch][k][m]
    x = torch.randn(nseg, seg, No, generator=g)               # [segment][k][n], shared by the batch
    dw = torch.zeros(NB, Mo, No, device='cuda')
    db = torch.zeros(NB, Mo, device='cuda')
    fl = _C.GEMM_BF16 | _C.GEMM_ATOMIC | (_C.GEMM_A_BF16 if a16 else 0) | (_C.GEMM_B_BF16 if b16 else 0)
    a_dev = dy.cuda().bfloat16() if a16 else dy.cuda()
    b_dev = x.cuda().bfloat16() if b16 else x.cuda()
    ops.gemm_raw(a=a_dev, lda=Mo, a_kcontig=0, a_bs0=seg * Mo, b=b_dev, ldb=No, b_kcontig=0, b_bs0=0, c=dw, ldc=No, c_bs0=Mo * No,
                 M=Mo, N=No, K=nseg * seg, k_seg=seg, a_seg_stride=NB * seg * Mo, b_seg_stride=seg * No, split_k=7, a_colsum=db,
                 cs_bs0=Mo, flags=fl, alpha=1.0, nb0=NB, nb1=1)
    for bi in range(NB):
        want = sum(_bf(dy[sg, bi]).T @ _bf(x[sg]) for sg in range(nseg))
        assert relerr(dw[bi], want) < 3e-5, (bi, relerr(dw[bi], want))
        col = (_bf(dy[:, bi]) if a16 else dy[:, bi].double()).sum((0, 1))
        assert relerr(db[bi], col) < 1e-5


@pytest.mark.parametrize('a16,Mo,Cin,V,HW', [(True, 1024, 192, 3, 1000), (False, 256, 256, 2, 704), (True, 1024, 384, 2, 4000),
                                             (False, 1024, 192, 2, 700)])
def test_gemm_bf16_weight_gradient_nchw_operand(ops, a16, Mo, Cin, V, HW):
    """dW[Mo, Cin] += sum_{view, hw} dY[view*HW + hw][o] X[view][c][hw]: A K-major (tokens x outputs, fp32 or bf16), B the NCHW
    planes read K-CONTIGUOUS (one K segment per view), float atomics, K slices, column sums - the first position-embedding
    convolutions' and input_proj's weight gradients (head.hip final stage).  Cin = 192: ragged N tile; HW = 700: ragged K step."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(Mo + Cin + HW)
    dy = torch.randn(V * HW, Mo, generator=g)
    x = torch.randn(V, Cin, HW, generator=g)
    dw = torch.zeros(Mo, Cin, device='cuda')
    db = torch.zeros(Mo, device='cuda')
    a_dev = dy.cuda().bfloat16() if a16 else dy.cuda()
    ops.gemm_raw(a=a_dev, lda=Mo, a_kcontig=0, b=x.cuda(), ldb=HW, b_kcontig=1, c=dw, ldc=Cin, a_colsum=db, M=Mo, N=Cin, K=V * HW,
                 k_seg=HW, a_seg_stride=HW * Mo, b_seg_stride=Cin * HW, split_k=8, alpha=1.0, nb0=1, nb1=1,
                 flags=_C.GEMM_BF16 | _C.GEMM_ATOMIC | (_C.GEMM_A_BF16 if a16 else 0))
    want = _bf(dy).T @ _bf(x.permute(0, 2, 1).reshape(V * HW, Cin))
    assert relerr(dw, want) < 3e-5, relerr(dw, want)
    assert relerr(db, (_bf(dy) if a16 else dy.double()).sum(0)) < 1e-5


def test_gemm_bf16_sources(ops):
    """PETR_GEMM_A_BF16 / _B_BF16 / _R_BF16: operands already stored as bf16 (the activations a producing epilogue wrote with
    PETR_GEMM_STORE_BF16, the bf16 dK / dV of the attention backward).  Bit-for-bit the same product as rounding the fp32
    values on load (same rounding, same accumulation order): compared with the fp32-source run of the same kernel."""
    from petr_amd import _C
    g = torch.Generator().manual_seed(77)
    M, N, K = 1000, 384, 520                      # ragged rows, K % 8 == 0 but not % 32
    A, Bm, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    for akc in (True, False):
        for bkc in (True, False):
            a32 = (A if akc else A.t().contiguous()).cuda()
            b32 = (Bm if bkc else Bm.t().contiguous()).cuda()
            base = dict(lda=K if akc else M, a_kcontig=int(akc), ldb=K if bkc else N, b_kcontig=int(bkc), ldc=N, M=M, N=N, K=K,
                        alpha=1.0, nb0=1, nb1=1, r=r.cuda(), ldr=N)
            ref = torch.empty(M, N, device='cuda')
            ops.gemm_raw(a=a32, b=b32, c=ref, flags=_C.GEMM_BF16 | _C.GEMM_RELU_MASK, **base)
            for fl, aa, bb in ((_C.GEMM_A_BF16, a32.bfloat16(), b32), (_C.GEMM_B_BF16, a32, b32.bfloat16()),
                               (_C.GEMM_A_BF16 | _C.GEMM_B_BF16, a32.bfloat16(), b32.bfloat16())):
                out = torch.empty(M, N, device='cuda')
                ops.gemm_raw(a=aa, b=bb, c=out, flags=_C.GEMM_BF16 | _C.GEMM_RELU_MASK | fl, **base)
                assert torch.equal(out, ref), (akc, bkc, fl)
    # bf16 mask operand + bf16 store + weight-gradient form with the column sums of a bf16 A
    base.pop('r'); base.pop('ldr')
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    ops.gemm_raw(a=a32, b=b32, c=out16, r=r.cuda().bfloat16(), ldr=N,
                 flags=_C.GEMM_BF16 | _C.GEMM_RELU_MASK | _C.GEMM_R_BF16 | _C.GEMM_STORE_BF16, **base)
    want = (_bf(A) @ _bf(Bm).T) * (r.bfloat16().double() > 0)
    assert relerr(out16.float(), want) < 4e-3
    dy, x = torch.randn(3000, 256, generator=g), torch.randn(3000, 192, generator=g)
    dw, db = torch.zeros(256, 192, device='cuda'), torch.zeros(256, device='cuda')
    ops.gemm_raw(a=dy.cuda().bfloat16(), lda=256, a_kcontig=0, b=x.cuda().bfloat16(), ldb=192, b_kcontig=0, c=dw, ldc=192, M=256,
                 N=192, K=3000, split_k=7, a_colsum=db, flags=_C.GEMM_BF16 | _C.GEMM_ATOMIC | _C.GEMM_A_BF16 | _C.GEMM_B_BF16,
                 alpha=1.0, nb0=1, nb1=1)
    assert relerr(dw, _bf(dy).T @ _bf(x)) < 3e-5
    assert relerr(db, _bf(dy).sum(0)) < 1e-5


@pytest.mark.parametrize('BH,Q,L', [(8, 900, 4224), (3, 70, 333), (2, 31, 65), (1, 1, 1), (4, 129, 513)])
def test_dropout_bits_are_the_packed_mask(ops, BH, Q, L):
    """petr_dropout_bits: both packed layouts hold exactly the mask of petr_dropout_mask (row = bh*Q + q, col = key)."""
    drop = (77, 5, 0.1)
    keep = ops.dropout_mask(drop, BH * Q, L).view(BH, Q, L).cpu()
    bq, bk = ops.dropout_bits(drop, BH, Q, L)
    nqt, nkb = (Q + 31) // 32, (L + 31) // 32
    sh = torch.arange(32, dtype=torch.int64)
    wq = bq.cpu().to(torch.int64).bitwise_and(0xFFFFFFFF).view(BH, nkb, 32 * nqt)        # [bh][kb][q]: bit j = key 32 kb + j
    got_q = ((wq[..., None] >> sh) & 1).permute(0, 2, 1, 3).reshape(BH, 32 * nqt, 32 * nkb)[:, :Q, :L].bool()
    assert torch.equal(got_q, keep)
    c = torch.arange(32)
    slot = 2 * ((c & 3) + 4 * (c >> 3)) + ((c >> 2) & 1)        # position of key c inside its 32-word block (petr_hip.h)
    wk = bk.cpu().to(torch.int64).bitwise_and(0xFFFFFFFF).view(BH, nqt, nkb, 32)[..., slot].reshape(BH, nqt, 32 * nkb)
    got_k = ((wk[..., None] >> sh) & 1).permute(0, 1, 3, 2).reshape(BH, 32 * nqt, 32 * nkb)[:, :Q, :L].bool()
    assert torch.equal(got_k, keep)


@pytest.mark.parametrize('B,H,Q,L,split,masked', [(1, 8, 900, 4224, 0, False), (2, 3, 70, 333, 3, True), (1, 2, 31, 65, 2, False),
                                                   (1, 8, 900, 900, 0, False), (1, 4, 129, 513, 8, True)])
def test_attention_backward_reads_the_mask_the_forward_left(ops, B, H, Q, L, split, masked):
    """The forward kernels (fp32 and bf16 K/V) leave the dropout mask they applied as packed key-major bits: exactly the
    words of petr_dropout_bits wherever a (query tile, key) exists; the backward that tests those bits returns what the
    re-hashing backward returns (same mask, same arithmetic; float atomics reorder the last bit)."""
    g = torch.Generator().manual_seed(Q * 7 + L)
    q, k, v, do = (dev(torch.randn(B, H, n, 32, generator=g)) for n in (Q, L, L, Q))
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        kpm[:, L - L // 4:] = True
        kpm = dev(kpm)
    drop = (4321, 9, 0.1)
    _, want_k = ops.dropout_bits(drop, B * H, Q, L)
    nqt, nkb = (Q + 31) // 32, (L + 31) // 32
    valid = torch.zeros(B * H, nqt, 32 * nkb, dtype=torch.bool)
    valid[:, :, :L] = True                                   # keys beyond L are never produced
    rowmask = torch.full((nqt,), -1, dtype=torch.int64)      # rows beyond Q: bits undefined
    if Q % 32:
        rowmask[-1] = (1 << (Q % 32)) - 1
    kb, vb = ops.cast_bf16(k), ops.cast_bf16(v)
    for fwd, bwd, kk, vv in ((ops.mha_fwd, ops.mha_bwd, k, v), (ops.mha_fwd_bf16, ops.mha_bwd_bf16, kb, vb)):
        bits = torch.zeros_like(want_k)
        o0, l0 = fwd(q, kk, vv, kpm, n_split=split, drop=drop)
        o1, l1 = fwd(q, kk, vv, kpm, n_split=split, drop=drop, drop_bits=bits)
        assert torch.equal(o0, o1) and torch.equal(l0, l1)
        got = bits.cpu().view(B * H, nqt, 32 * nkb).to(torch.int64) & rowmask[None, :, None]
        ref = want_k.cpu().view(B * H, nqt, 32 * nkb).to(torch.int64) & rowmask[None, :, None]
        assert torch.equal(got[valid], ref[valid])
        g0 = bwd(q, kk, vv, o0, do, l0, kpm, drop=drop)
        g1 = bwd(q, kk, vv, o0, do, l0, kpm, drop=drop, drop_bits=bits)
        for a, b in zip(g0, g1):
            assert (a - b).abs().max().item() <= 1e-5 * (1 + a.abs().max().item())


@pytest.mark.parametrize('B,Q,L,split,training', [(1, 900, 4224, 0, True), (2, 70, 333, 3, False), (1, 900, 900, 0, True),
                                                   (1, 37, 40, 1, True)])
def test_attn_out_ln_equals_its_four_separate_launches(ops, B, Q, L, split, training):
    """petr_attn_out_ln (merge of the L-split partials + out-projection + dropout + residual + LayerNorm + query_pos add in
    one launch) against the chain it replaces: mha_fwd (with its merge kernel) -> linear -> layernorm, to fp32 rounding
    (the 256-deep dot products are summed in a different order)."""
    g = torch.Generator().manual_seed(B * 1000 + Q + L)
    H, C = 8, 256
    q, k, v = (dev(torch.randn(B, n, C, generator=g)).view(B, n, H, 32).permute(0, 2, 1, 3) for n in (Q, L, L))
    w, bias, res = dev(torch.randn(C, C, generator=g) * 0.06), dev(torch.randn(C, generator=g)), dev(torch.randn(B * Q, C, generator=g))
    gamma, beta, pos = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g)), dev(torch.randn(Q, C, generator=g))
    pdrop = (77, 2, 0.1) if training else None           # attention probabilities
    odrop = (77, 3, 0.1) if training else None           # projection output
    o, lse = ops.mha_fwd(q, k, v, n_split=split, drop=pdrop)
    ao = o.permute(0, 2, 1, 3).reshape(B * Q, C).contiguous()
    lin = ops.linear(ao, w, bias)
    y_ref, z_ref, mean_ref, rstd_ref = ops.layernorm(lin, gamma, beta, residual=res, save_stats=True, drop=odrop)
    # fused
    parts, ns = ops.mha_fwd(q, k, v, n_split=split, drop=pdrop, defer_merge=True)
    a = torch.full((B * Q, C), float('nan'), device='cuda') if ns > 1 else ao.clone()
    scale = 1.0 / (1.0 - round(0.1 * 65536) / 65536) if training else 1.0
    y, y2, z, mean, rstd, lse2 = ops.attn_out_ln(a, w, bias, res, gamma, beta, partials=parts, n_split=ns, BHQ=(B, H, Q),
                                                 attn_scale=scale, drop=odrop, add2=pos, add2_rows=Q)
    if ns > 1:        # same partials, same summation order; the compiler may contract the multiply-adds differently
        assert relerr(a, ao) < 1e-6
        assert (lse2.view(B, H, Q) - lse).abs().max().item() < 1e-5
    assert relerr(z, z_ref) < 2e-6 and relerr(y, y_ref) < 5e-6
    assert relerr(mean, mean_ref) < 1e-5 and relerr(rstd, rstd_ref) < 1e-5
    assert relerr(y2, y_ref + pos.repeat(B, 1)) < 5e-6
    if training:          # the same elements are dropped
        assert torch.equal((z - res) == 0, (z_ref - res) == 0) or ((z - res == 0) != (z_ref - res == 0)).float().mean() < 1e-4
    # with the optional second projection (the next attention's query projection of y2)
    w2, b2 = dev(torch.randn(C, C, generator=g) * 0.06), dev(torch.randn(C, generator=g))
    a2 = torch.full((B * Q, C), float('nan'), device='cuda') if ns > 1 else ao.clone()
    out = ops.attn_out_ln(a2, w, bias, res, gamma, beta, partials=parts, n_split=ns, BHQ=(B, H, Q), attn_scale=scale, drop=odrop,
                          add2=pos, add2_rows=Q, w2=w2, bias2=b2)
    assert torch.equal(out[0], y) and torch.equal(out[1], y2)
    assert relerr(out[6], ops.linear(y2, w2, b2)) < 5e-6


@pytest.mark.parametrize('M,P,training', [(900, 4, True), (1800, 1, False), (37, 2, True)])
def test_ln_proj_equals_layernorm_then_in_projection(ops, M, P, training):
    """petr_ln_proj against the two launches it replaces: layernorm (split-K slabs + bias + dropout + residual, query_pos add)
    and the next layer's self-attention in-projection (q, k rows from y + query_pos, v rows from y)."""
    g = torch.Generator().manual_seed(M + P)
    C = 256
    x = dev(torch.randn(P, M, C, generator=g))
    bias, res = dev(torch.randn(C, generator=g)), dev(torch.randn(M, C, generator=g))
    gamma, beta = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g))
    Q = 900 if M % 900 == 0 else M
    pos = dev(torch.randn(Q, C, generator=g))
    w_in, b_in = dev(torch.randn(3 * C, C, generator=g) * 0.06), dev(torch.randn(3 * C, generator=g))
    drop = (5, 6, 0.1) if training else None
    y_ref, z_ref, mean_ref, rstd_ref = ops.layernorm(x, gamma, beta, bias=bias, residual=res, save_stats=True, drop=drop)
    y2_ref = y_ref + pos.repeat(M // Q, 1)
    qk = ops.linear(y2_ref, w_in[:2 * C], b_in[:2 * C])
    vv = ops.linear(y_ref, w_in[2 * C:], b_in[2 * C:])
    y, y2, z, mean, rstd, out2 = ops.ln_proj(x, gamma, beta, w_in, b_in, bias=bias, residual=res, drop=drop, add2=pos, add2_rows=Q,
                                             n2_pos=2)
    assert relerr(z, z_ref) < 2e-6 and relerr(y, y_ref) < 5e-6 and relerr(y2, y2_ref) < 5e-6
    assert relerr(mean, mean_ref) < 1e-5 and relerr(rstd, rstd_ref) < 1e-5
    assert relerr(out2[:, :2 * C], qk) < 5e-6 and relerr(out2[:, 2 * C:], vv) < 5e-6


@pytest.mark.parametrize('M,F,n_split,training', [(900, 2048, 4, True), (900, 2048, 8, False), (1800, 2048, 2, True), (37, 512, 1, True),
                                                   (37, 512, 2, False)])
def test_ffn_fwd_equals_the_two_contractions(ops, M, F, n_split, training):
    """petr_ffn_fwd against the two launches it replaces: Linear + ReLU + dropout (petr_gemm's epilogue: the same mask) and the
    split-K second contraction; the slabs' sum against a float64 product of the stored hidden."""
    g = torch.Generator().manual_seed(M + F + n_split)
    C = 256
    x = dev(torch.randn(M, C, generator=g))
    w1, b1 = dev(torch.randn(F, C, generator=g) * 0.06), dev(torch.randn(F, generator=g) * 0.1)
    w2 = dev(torch.randn(C, F, generator=g) * 0.03)
    drop = (7, 8, 0.1) if training else None
    h_ref = torch.relu(x.double() @ w1.double().t() + b1.double())
    if training:
        h_ref = h_ref * ops.dropout_mask(drop, M, F).double() / (1 - 6554 / 65536)
    hidden, part = ops.ffn_fwd(x, w1, b1, w2, n_split=n_split, drop=drop)
    assert relerr(hidden, h_ref) < 2e-6
    if training:
        assert torch.equal(hidden == 0, h_ref == 0)                       # ReLU zeros and dropped units: the same elements
    hb = F // n_split
    for s_ in range(n_split):
        ref = hidden[:, s_ * hb:(s_ + 1) * hb].double() @ w2[:, s_ * hb:(s_ + 1) * hb].double().t()
        assert relerr(part[s_], ref) < 2e-6, s_
    _, part2 = ops.ffn_fwd(x, w1, b1, w2, n_split=n_split, drop=drop, store_hidden=False)
    assert torch.equal(part, part2)


@pytest.mark.parametrize('M,training', [(900, True), (37, False)])
def test_ln_bwd_proj_leading_in_projection_gradient_only(ops, M, training):
    """petr_ln_bwd_proj with a K = 768 leading product and no trailing one: the upstream gradient of the LayerNorm backward is
    d_qkv W_in (a self-attention in-projection's input gradient) + one dy slab (the identity path) + dy_residual."""
    g = torch.Generator().manual_seed(M)
    C = 256
    x = dev(torch.randn(M, C, generator=g))
    gamma, beta = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g))
    _, z, mean, rstd = ops.layernorm(x, gamma, beta, save_stats=True)
    d_qkv = dev(torch.randn(M, 3 * C, generator=g))
    w_in = dev(torch.randn(3 * C, C, generator=g) * 0.06)
    d_id, d_br = dev(torch.randn(M, C, generator=g)), dev(torch.randn(M, C, generator=g))
    drop = (3, 4, 0.1) if training else None
    dy_sum = (d_qkv.double() @ w_in.double() + d_id.double() + d_br.double()).float()
    ref = ops.layernorm_bwd(z, mean, rstd, gamma, dy_sum, drop=drop)
    dz, dzd, dg, db, out = ops.ln_bwd_proj(z, mean, rstd, gamma, d_id, None, dy_residual=d_br, drop=drop, pre_a=d_qkv, pre_w=w_in)
    assert out is None
    assert relerr(dz, ref[0]) < 1e-5 and relerr(dg, ref[1]) < 1e-5 and relerr(db, ref[2]) < 1e-5
    if training:
        assert relerr(dzd, ref[3]) < 1e-5


@pytest.mark.parametrize('M,F,n_split,alpha', [(900, 2048, 8, 1.0 / (1 - 6554 / 65536)), (1800, 2048, 4, 1.0), (37, 512, 2, 1.25)])
def test_ffn_bwd_equals_the_two_input_gradients(ops, M, F, n_split, alpha):
    """petr_ffn_bwd against float64: d_hidden = alpha (dy W2) where the forward's hidden is positive, slabs = d_hidden W1 per slice."""
    g = torch.Generator().manual_seed(M + F + n_split)
    C = 256
    dy = dev(torch.randn(M, C, generator=g))
    w1, w2 = dev(torch.randn(F, C, generator=g) * 0.06), dev(torch.randn(C, F, generator=g) * 0.03)
    hidden = dev(torch.relu(torch.randn(M, F, generator=g)))            # zeros where the ReLU / the dropout blocked the unit
    dh, part = ops.ffn_bwd(dy, w1, w2, hidden, alpha=alpha, n_split=n_split)
    dh_ref = (dy.double() @ w2.double()) * alpha * (hidden > 0)
    assert relerr(dh, dh_ref) < 2e-6
    assert torch.equal(dh == 0, dh_ref == 0)
    hb = F // n_split
    for s_ in range(n_split):
        ref = dh[:, s_ * hb:(s_ + 1) * hb].double() @ w1[s_ * hb:(s_ + 1) * hb].double()
        assert relerr(part[s_], ref) < 2e-6, s_


@pytest.mark.parametrize('M,P,F,training', [(900, 1, 256, True), (900, 4, 256, False), (900, 1, 2048, True), (37, 2, 512, True)])
def test_ln_bwd_proj_equals_layernorm_bwd_then_input_gradient(ops, M, P, F, training):
    """petr_ln_bwd_proj against the two launches it replaces: layernorm_bwd (slab sum + identity path, dropped copy, dgamma /
    dbeta) and the input-gradient contraction dx = dz_drop @ W (with the FFN's ReLU mask and dropout scale when F > 256)."""
    g = torch.Generator().manual_seed(M + P + F)
    C = 256
    x = dev(torch.randn(M, C, generator=g))
    gamma, beta = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g))
    _, z, mean, rstd = ops.layernorm(x, gamma, beta, save_stats=True)
    dy = dev(torch.randn(P, M, C, generator=g))
    res = dev(torch.randn(M, C, generator=g)) if P > 1 else None
    w = dev(torch.randn(C, F, generator=g) * 0.06)                 # nn.Linear(F -> 256) weight [256, F]
    drop = (3, 4, 0.1) if training else None
    mask = dev(torch.randn(M, F, generator=g)) if F > 256 else None
    alpha = 1.25 if F > 256 else 1.0
    dy_sum = dy.sum(0) + (res if res is not None else 0)
    ref = ops.layernorm_bwd(z, mean, rstd, gamma, dy_sum, drop=drop)
    a_ref = ref[3] if training else ref[0]
    out_ref = (a_ref.double() @ w.double()) * alpha
    if mask is not None:
        out_ref = out_ref * (mask > 0)
    dz, dzd, dg, db, out = ops.ln_bwd_proj(z, mean, rstd, gamma, dy if P > 1 else dy[0], w, dy_residual=res, drop=drop, alpha=alpha,
                                           relu_mask=mask)
    assert relerr(dz, ref[0]) < 5e-6 and relerr(dg, ref[1]) < 2e-5 and relerr(db, ref[2]) < 2e-5
    if training:
        assert relerr(dzd, ref[3]) < 5e-6
    assert relerr(out, out_ref) < 1e-5


def test_ln_bwd_proj_with_leading_product(ops):
    """The upstream gradient formed inside the kernel: dy = pre_a @ pre_w + dy_residual (the query projection's input gradient
    plus the identity path), then LayerNorm backward and the out-projection input gradient."""
    g = torch.Generator().manual_seed(4)
    M, C = 900, 256
    x = dev(torch.randn(M, C, generator=g))
    gamma, beta = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g))
    _, z, mean, rstd = ops.layernorm(x, gamma, beta, save_stats=True)
    d_q, w_q = dev(torch.randn(M, C, generator=g)), dev(torch.randn(C, C, generator=g) * 0.06)
    res, w = dev(torch.randn(M, C, generator=g)), dev(torch.randn(C, C, generator=g) * 0.06)
    drop = (8, 1, 0.1)
    dy = (d_q.double() @ w_q.double()).float() + res
    ref = ops.layernorm_bwd(z, mean, rstd, gamma, dy, drop=drop)
    dz, dzd, dg, db, out = ops.ln_bwd_proj(z, mean, rstd, gamma, None, w, dy_residual=res, drop=drop, pre_a=d_q, pre_w=w_q)
    assert relerr(dz, ref[0]) < 1e-5 and relerr(dzd, ref[3]) < 1e-5 and relerr(dg, ref[1]) < 3e-5 and relerr(db, ref[2]) < 3e-5
    assert relerr(out, ref[3].double() @ w.double()) < 2e-5
