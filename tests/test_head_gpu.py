"""GPU parity of the whole hot path (petr_amd.PETRHead -> petr_head_fwd / petr_head_bwd) against the
CPU oracle and the golden vectors captured from the reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import petr_oracle as O  # noqa: E402  (checker only)

REL = 1e-4     # north_star: decoder outputs within 1e-4 rel of the reference (fp32 config)


@pytest.fixture(scope='module')
def pa():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import petr_amd
    return petr_amd


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def make_pair(pa, oracle, **cfg_kw):
    head = pa.build_head(pa.petr_head_cfg(**cfg_kw))
    head.load_state_dict(oracle.state_dict())
    return head.cuda().eval()


def metas_from(fx):
    N = fx['lidar2img'].shape[0]
    ph, pw = [int(v) for v in fx['pad_hw']]
    ih, iw = [int(v) for v in fx['img_hw']]
    return [{'pad_shape': [(ph, pw, 3)] * N, 'img_shape': [(ih, iw, 3)] * N, 'lidar2img': list(fx['lidar2img'])}]


@pytest.mark.parametrize('name', ['head_toy', 'head_toy_masked'])
def test_head_golden(pa, golden_dir, name):
    """fixture = outputs of the REFERENCE PETRHead.forward (oracle/make_golden.py)."""
    fx = np.load(os.path.join(golden_dir, name + '.npz'))
    oracle = O.seeded_head(2, 1234, num_query=16)
    wsum = sum(v.double().abs().sum().item() for v in oracle.state_dict().values())
    if abs(wsum - float(fx['weight_abs_sum'])) > 1e-9 * wsum:
        pytest.skip('torch RNG stream differs from the build container')
    head = make_pair(pa, oracle, num_query=16)
    with torch.no_grad():
        out = head([torch.from_numpy(fx['feats']).cuda()], metas_from(fx))
    assert out['all_cls_scores'].shape == (6, 1, 16, 10) and out['all_bbox_preds'].shape == (6, 1, 16, 10)
    assert out['enc_cls_scores'] is None and out['enc_bbox_preds'] is None
    assert rel(out['all_cls_scores'], torch.from_numpy(fx['all_cls_scores'])) < REL
    assert rel(out['all_bbox_preds'], torch.from_numpy(fx['all_bbox_preds'])) < REL


def test_headv2_golden(pa, golden_dir):
    """PETRv2Head (fpe + RegLayer + with_time, 12 views, deep-copied branches) vs the reference's own output."""
    fx = np.load(os.path.join(golden_dir, 'headv2_toy.npz'))
    oracle = O.seeded_head(3, 4321, num_query=16, v2=True, with_fpe=True, with_time=True, with_multi=True,
                           code_weights=[1.0] * 10)
    wsum = sum(v.double().abs().sum().item() for v in oracle.state_dict().values())
    if abs(wsum - float(fx['weight_abs_sum'])) > 1e-9 * wsum:
        pytest.skip('torch RNG stream differs from the build container')
    head = pa.build_head(pa.petrv2_head_cfg(num_query=16))
    head.load_state_dict(oracle.state_dict())
    head = head.cuda().eval()
    metas = metas_from(fx)
    metas[0]['timestamp'] = list(fx['timestamp'])
    with torch.no_grad():
        out = head([torch.from_numpy(fx['feats']).cuda()], metas)
    assert out['all_bbox_preds'].shape == (6, 1, 16, 10)
    assert rel(out['all_cls_scores'], torch.from_numpy(fx['all_cls_scores'])) < REL
    assert rel(out['all_bbox_preds'], torch.from_numpy(fx['all_bbox_preds'])) < REL


def test_headv2_backward(pa):
    kw = dict(num_query=16, v2=True, with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10)
    _grad_case(pa, 1, 12, 4, 5, (64, 80), (64, 80), 16, seed=6, oracle_kw=kw, with_time=True)


def test_head_c5_forward_and_intermediates(pa):
    """BASELINE configs[1]: c5 shape, 900 queries, 6 layers, fp32, vs configs[0] (the CPU path)."""
    oracle = O.seeded_head(0, None, num_query=900)
    head = make_pair(pa, oracle, num_query=900)
    metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=0)
    feats = torch.randn(1, 6, 256, 16, 44, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = oracle([feats], metas, return_intermediates=True)
        got = head([feats.cuda()], metas)
    L = 6 * 16 * 44
    mem = head.workspace_view('memory').view(1, 6, 16, 44, 256).permute(0, 1, 4, 2, 3)
    pos = head.workspace_view('pos_embed').view(1, 6, 16, 44, 256).permute(0, 1, 4, 2, 3)
    assert rel(mem, want['_memory']) < 1e-5
    assert rel(pos, want['_pos_embed']) < 1e-5
    assert rel(head.workspace_view('query_embed').view(900, 256), want['_query_embeds']) < 1e-5
    outs = head.workspace_view('outs_dec').view(6, 1, 900, 256)
    assert rel(outs, want['_outs_dec']) < REL
    assert rel(got['all_cls_scores'], want['all_cls_scores']) < REL
    assert rel(got['all_bbox_preds'], want['all_bbox_preds']) < REL
    # golden samples captured from the reference head at this exact configuration
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'head_c5_samples.npz'))
    wsum = sum(v.double().abs().sum().item() for v in oracle.state_dict().values())
    if abs(wsum - float(fx['weight_abs_sum'])) <= 1e-9 * wsum:
        idx = torch.from_numpy(fx['idx'])
        assert (got['all_cls_scores'].cpu().flatten()[idx] - torch.from_numpy(fx['cls'])).abs().max() < 1e-3
        assert (got['all_bbox_preds'].cpu().flatten()[idx] - torch.from_numpy(fx['bbox'])).abs().max() < 1e-3
    # same input twice -> bit-identical (no atomics on the forward path)
    with torch.no_grad():
        again = head([feats.cuda()], metas)
    assert torch.equal(again['all_cls_scores'], got['all_cls_scores'])
    assert torch.equal(again['all_bbox_preds'], got['all_bbox_preds'])
    del L


def _dropout_masks(seed, p, B, Q, L, n_layers=6, heads=8, C=256, F=2048):
    """The keep masks the kernels use in training mode (site = 8*layer + k, include/petr_hip.h "Dropout"), exported
    with petr_dropout_mask and laid out as oracle.DecoderLayer.forward(masks=...) wants them."""
    from petr_amd import ops
    shapes = {'sp': (B * heads * Q, Q), 'so': (B * Q, C), 'cp': (B * heads * Q, L), 'co': (B * Q, C), 'fh': (B * Q, F),
              'fo': (B * Q, C)}
    out = []
    for l in range(n_layers):
        m = {k: ops.dropout_mask((seed, 8 * l + i, p), *shp).cpu() for i, (k, shp) in enumerate(shapes.items())}
        m['scale'] = 1.0 / (1.0 - round(p * 65536) / 65536)    # p is realised on a 16-bit grid
        out.append(m)
    return out


def _oracle_grads(oracle, feats, metas, g_cls, g_box, dtype, dropout_masks=None):
    """fwd+bwd of the oracle in `dtype` (float64 = yardstick that separates fp32 noise from disagreement)."""
    import copy
    o = copy.deepcopy(oracle).to(dtype)
    f = feats.to(dtype).clone().requires_grad_(True)
    torch.set_default_dtype(dtype)
    try:
        out = o([f], metas, dropout_masks=dropout_masks)
        (out['all_cls_scores'] * g_cls.to(dtype)).sum().add((out['all_bbox_preds'] * g_box.to(dtype)).sum()).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return out, {k: p.grad for k, p in o.named_parameters() if p.grad is not None}, f.grad


def _grad_case(pa, B, N, H, W, pad_hw, img_hw, Q, seed, oracle_kw=None, with_time=False, dropout=None):
    """Gradients vs the float64 oracle.  Two metrics per tensor, both relative to max(|want|_inf, 1e-4*global):
    L2 (tight) and max-abs (loose): a ReLU whose pre-activation sits within 1e-7 of zero flips its mask
    between ANY two fp32 evaluations (the CPU fp32 oracle shows the same O(1e-2) single-element jumps against
    float64), so max-abs cannot be tight; tensors whose true gradient is identically zero (softmax shift
    invariance: position_encoder/adapt_pos3d last bias; layer-0 self-attention in_proj with query = 0) are
    checked absolutely."""
    if oracle_kw is None:
        oracle = O.seeded_head(seed, 77, num_query=Q)
        head = make_pair(pa, oracle, num_query=Q)
    else:
        oracle = O.seeded_head(seed, 77, **oracle_kw)
        head = pa.build_head(pa.petrv2_head_cfg(num_query=Q))
        head.load_state_dict(oracle.state_dict())
        head = head.cuda().eval()
    metas = O.synthetic_img_metas(B, N, pad_hw, img_hw, seed=seed, with_time=with_time)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, N, 256, H, W, generator=g)
    g_cls, g_box = torch.randn(6, B, Q, 10, generator=g), torch.randn(6, B, Q, 10, generator=g)
    masks = None
    if dropout is not None:      # training mode: the oracle gets the very masks the kernels will draw
        drop_seed, p = dropout
        head.train()
        head._dropout_seed_override = drop_seed
        masks = _dropout_masks(drop_seed, p, B, Q, N * H * W)
    want, wgrads, wfeat = _oracle_grads(oracle, feats, metas, g_cls, g_box, torch.float64, masks)
    fg = feats.cuda().requires_grad_(True)
    got = head([fg], metas)
    torch.autograd.backward([got['all_cls_scores'], got['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    assert rel(got['all_cls_scores'], want['all_cls_scores']) < REL
    assert rel(got['all_bbox_preds'], want['all_bbox_preds']) < REL
    gmax = max(v.abs().max().item() for v in wgrads.values())

    def errs(a, b):
        a, b = a.detach().double().cpu(), b.double()
        sc = max(b.abs().max().item(), 1e-4 * gmax)
        return (a - b).norm().item() / max(b.norm().item(), sc), (a - b).abs().max().item() / sc
    l2, mx = errs(fg.grad, wfeat)
    assert l2 < 5e-3 and mx < 5e-2, f'd_feats: l2 {l2:.2e} max {mx:.2e}'
    bad = {}
    for name, p in head.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        l2, mx = errs(p.grad, wgrads[name])
        if l2 > 5e-3 or mx > 5e-2:
            bad[name] = (round(l2, 5), round(mx, 5))
    assert not bad, f'gradient mismatch (l2, max): {sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]}'
    return head, metas, feats, (g_cls, g_box)


def test_head_backward_toy_batch2_masked(pa):
    _grad_case(pa, 2, 2, 4, 6, (128, 192), (100, 150), 16, seed=4)


def test_head_training_mode_dropout_toy_batch2_masked(pa):
    """train(): all six dropout layers of every decoder layer (p = 0.1, the reference's rate), forward AND backward,
    against the fp64 oracle running the same masks."""
    head, metas, feats, _ = _grad_case(pa, 2, 2, 4, 6, (128, 192), (100, 150), 16, seed=4, dropout=(987654321, 0.1))
    assert head._last_dropout == (987654321, 0.1)
    # another seed is another function; eval() is the dropout-free path again
    with torch.no_grad():
        a = head([feats.cuda()], metas)['all_cls_scores']
        head._dropout_seed_override = 5
        b = head([feats.cuda()], metas)['all_cls_scores']
        head._dropout_seed_override = None
        c = head([feats.cuda()], metas)['all_cls_scores']     # fresh seed from torch's generator
        head.eval()
        e1 = head([feats.cuda()], metas)['all_cls_scores']
        e2 = head([feats.cuda()], metas)['all_cls_scores']
    assert not torch.equal(a, b) and not torch.equal(b, c) and torch.equal(e1, e2)
    assert head._last_dropout == (0, 0.0)


def test_head_training_mode_dropout_900_queries(pa):
    """c5 query count and FFN width, L = 6*8*11 = 528 keys (every attention kernel path incl. L-splits)."""
    _grad_case(pa, 1, 6, 8, 11, (256, 352), (256, 352), 900, seed=6, dropout=(31337, 0.1))


def test_headv2_training_mode_dropout(pa):
    """PETRv2Head (fpe, RegLayer, with_time, 12 views, deep-copied branches) in train(): same decoder dropouts."""
    kw = dict(num_query=16, v2=True, with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10)
    _grad_case(pa, 1, 12, 4, 6, (128, 192), (128, 192), 16, seed=8, oracle_kw=kw, with_time=True, dropout=(424242, 0.1))


def test_dynamic_tile_tickets_head_parity(golden_dir):
    """PETR_MHA_DYNAMIC=1 (opt-in ticket scheduling of the attention K/V tiles) is read once per process: run the
    golden toy forward in a child process with it set."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np, torch\n"
        "sys.path.insert(0, os.getcwd())\n"
        "from oracle import petr_oracle as O\n"
        "import petr_amd\n"
        "o = O.seeded_head(2, 1234, num_query=900)\n"
        "h = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=900)); h.load_state_dict(o.state_dict()); h = h.cuda().eval()\n"
        "m = O.synthetic_img_metas(1, 6, (256, 352), seed=2)\n"
        "f = torch.randn(1, 6, 256, 8, 11, generator=torch.Generator().manual_seed(0))\n"
        "with torch.no_grad():\n"
        "    w = o([f], m); g = h([f.cuda()], m)\n"
        "e = max(((g[k].cpu().double() - w[k].double()).abs().max() / w[k].double().abs().max()).item() for k in ('all_cls_scores', 'all_bbox_preds'))\n"
        "print('REL', e); sys.exit(0 if e < 1e-4 else 1)\n")
    env = dict(os.environ, PETR_MHA_DYNAMIC='1')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]


def test_head_backward_c5(pa):
    head, metas, feats, (g_cls, g_box) = _grad_case(pa, 1, 6, 16, 44, (512, 1408), (512, 1408), 900, seed=5)
    # gradient accumulation semantics: a second backward adds onto .grad
    g1 = head.flat_gradients().clone()
    got = head([feats.cuda()], metas)
    torch.autograd.backward([got['all_cls_scores'], got['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    assert rel(head.flat_gradients(), 2 * g1) < 1e-4
    head.zero_grad(set_to_none=True)
    got = head([feats.cuda()], metas)
    torch.autograd.backward([got['all_cls_scores'], got['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    assert rel(head.flat_gradients(), g1) < 1e-4


def test_position_embeding_api(pa, golden_dir):
    """PETRHead.position_embeding keeps the reference signature and layout ([B,N,256,H,W], mask)."""
    fx = np.load(os.path.join(golden_dir, 'coords3d_toy_masked.npz'))
    oracle = O.seeded_head(2, 1234, num_query=16)
    head = make_pair(pa, oracle, num_query=16)
    N, H, W = [int(v) for v in fx['shape']]
    metas = metas_from(fx)
    masks = torch.from_numpy(fx['masks'])
    with torch.no_grad():
        want_pe, want_mask = oracle.position_embeding(1, N, H, W, metas, masks)
    pe, cmask = head.position_embeding([torch.zeros(1, N, 256, H, W).cuda()], metas, masks.cuda())
    assert pe.shape == (1, N, 256, H, W)
    assert rel(pe, want_pe) < 1e-5
    assert torch.equal(cmask.cpu(), want_mask)       # the coordinate kernel follows the reference's fp32 op order: zero flips


def test_transformer_module_api(pa):
    """PETRTransformer.forward(x, mask, query_embed, pos_embed) drop-in (inference) vs the oracle transformer."""
    oracle = O.seeded_head(2, 1234, num_query=16)
    head = make_pair(pa, oracle, num_query=16)
    g = torch.Generator().manual_seed(1)
    x, pos = torch.randn(2, 2, 256, 4, 6, generator=g), torch.randn(2, 2, 256, 4, 6, generator=g)
    qe = torch.randn(16, 256, generator=g)
    mask = torch.zeros(2, 2, 4, 6, dtype=torch.bool)
    mask[1, :, :, 4:] = True
    with torch.no_grad():
        want, _ = oracle.transformer(x, mask, qe, pos)
        got, mem = head.transformer(x.cuda(), mask.cuda(), qe.cuda(), pos.cuda())
    assert got.shape == (6, 2, 16, 256) and mem.shape == x.shape
    assert rel(got, want) < REL


def test_bucketed_allreduce_rccl_single_rank(pa):
    """RCCL path of petr_amd.dist on real hardware: one rank, collectives forced on, issued per backward stage on the
    side stream behind events; AVG over one rank must leave the gradients of a plain backward unchanged."""
    import socket
    import torch.distributed as dist
    from petr_amd.dist import BucketedGradAllReduce
    oracle = O.seeded_head(2, 1234, num_query=16)
    head = make_pair(pa, oracle, num_query=16)
    metas = O.synthetic_img_metas(1, 2, (128, 192), seed=1)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(1, 2, 256, 4, 6, generator=g).cuda()
    g_cls, g_box = torch.randn(6, 1, 16, 10, generator=g).cuda(), torch.randn(6, 1, 16, 10, generator=g).cuda()

    def run():
        head.zero_grad_flat()
        out = head([feats], metas)
        torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls, g_box])

    run()
    want = head.flat_gradients().clone()
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        red = BucketedGradAllReduce(head, merge=2, force=True)
        assert len(red.buckets) == 4
        run()
        red.finish()
        torch.cuda.synchronize()
        got = head.flat_gradients()
        assert rel(got, want) < 1e-5        # float atomics in the weight gradients: not bit-identical run to run
        red.detach()
    finally:
        dist.destroy_process_group()


def test_head_p4_1408_forward(pa):
    """BASELINE configs[2] shape (6x256x32x88, L = 16 896) in fp32: forward parity at scale."""
    oracle = O.seeded_head(0, None, num_query=900)
    head = make_pair(pa, oracle, num_query=900)
    metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=2)
    feats = torch.randn(1, 6, 256, 32, 88, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = oracle([feats], metas)
        got = head([feats.cuda()], metas)
    assert rel(got['all_cls_scores'], want['all_cls_scores']) < REL
    assert rel(got['all_bbox_preds'], want['all_bbox_preds']) < REL


# BASELINE configs[2] proper: bf16 K/V in the cross-attention, fp32 accumulate/softmax.  Tolerance against the fp32
# CPU oracle, stated separately from the fp32 bar (SURVEY 8(d) config 3 expected ~1e-2 rel; the head-level outputs turn
# out far tighter than the operator-level bound in tests/test_ops_gpu.py, so the bar here is 1e-3).
REL_BF16 = 1e-3      # measured 7e-5 (cls) / 5e-5 (bbox) at p4-1408: the rounding noise averages out over 16 896 keys
# the bf16 TRAINING step also runs the FFN contractions (2048-wide hidden) and the token-side activations / weights in bf16, as
# autocast does: measured 1.0e-3 (cls) at c5 with dropout, 3e-4 ... 6e-4 at the larger workloads; still 3x inside SURVEY 8(d)'s
# expected ~1e-2 for this configuration
REL_BF16_TRAIN = 3e-3


def test_head_p4_1408_forward_bf16_attention(pa):
    oracle = O.seeded_head(0, None, num_query=900)
    head = make_pair(pa, oracle, num_query=900)
    metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=2)
    feats = torch.randn(1, 6, 256, 32, 88, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = oracle([feats], metas)
        ref32 = head([feats.cuda()], metas)
        ref32 = {k: v.clone() for k, v in ref32.items() if v is not None}
        head.attn_dtype = 'bf16'
        got = head([feats.cuda()], metas)
    e_cls, e_box = rel(got['all_cls_scores'], want['all_cls_scores']), rel(got['all_bbox_preds'], want['all_bbox_preds'])
    print(f'bf16 attention vs fp32 oracle at p4-1408: cls {e_cls:.2e} bbox {e_box:.2e}')
    assert e_cls < REL_BF16 and e_box < REL_BF16
    assert not torch.equal(got['all_cls_scores'], ref32['all_cls_scores'])      # the bf16 path really ran


# bf16 TRAINING step (BASELINE configs 3-5 as stated: bf16 fwd+bwd).  Gradients against the float64 oracle on the same
# inputs (and, in training mode, the very dropout masks the kernels draw).  What bf16 can and cannot meet here, measured
# (scripts/diag_bf16_grads.py, DESIGN.md section 4 "bf16 training step"): every bf16 OPERATOR agrees with fp64 on the same
# rounded operands to 2e-5 (contractions) / 3e-3 L2 (attention backward), tests/test_ops_gpu.py.  At the HEAD level the
# random-init decoder attends almost uniformly, so the query-specific part of every hidden state is a ~1/sqrt(L) residual
# of large common components, and rounding P / ds / K / V to 8 bits perturbs that residual while the outputs move by
# 2e-4 ... 5e-4 of their range; the gradients inherit a 4-5 % L2 deviation of the whole flat gradient, 4-8 % per tensor
# (cosine >= 0.997), the same at c5, p4-1408, p4-1600 and v2-800 (round 3, all five cases below: flat 0.041 ... 0.050, worst
# tensor 0.076 / cosine 0.9971).  Round 3 built the "centre the keys before rounding" cure the round-2 notes proposed
# (key' = bf16(memory + pos - per-sample mean): exact for the softmax) and measured NO change (flat 0.048 -> 0.047,
# 0.041 -> 0.042, 0.045 -> 0.044, 0.049 -> 0.050, 0.049 -> 0.047), so the common key mean is not what bounds these
# numbers and the code was removed again; the rounding of the probabilities and of ds is inherent to bf16 matrix operands.
# Bars (tightened in round 3 to the measured level): outputs 3e-3 of the range (REL_BF16_TRAIN); per tensor L2 <= 0.10 and
# cosine >= 0.995; whole flat gradient L2 <= 0.06; tensors whose true gradient is identically zero (softmax shift
# invariance) or below 1e-3 of the largest gradient entry (PETRv2's per-task reg heads on a synthetic upstream gradient:
# entries 1e-6 of gmax, where an fp32 run differs from fp64 by as much): absolute error <= 1e-3 of that entry.
def _bf16_grad_case(pa, head, oracle, feats, metas, g_cls, g_box, masks=None, l2_bar=0.10, cos_bar=0.995, flat_bar=0.06,
                    out_bar=REL_BF16_TRAIN):
    want, wgrads, wfeat = _oracle_grads(oracle, feats, metas, g_cls, g_box, torch.float64, masks)
    head.attn_dtype = 'bf16'
    head.zero_grad_flat()
    fg = feats.cuda().requires_grad_(True)
    got = head([fg], metas)
    torch.autograd.backward([got['all_cls_scores'], got['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    e_cls, e_box = rel(got['all_cls_scores'], want['all_cls_scores']), rel(got['all_bbox_preds'], want['all_bbox_preds'])
    gmax = max(v.abs().max().item() for v in wgrads.values())
    pairs = {'d_feats': (fg.grad, wfeat)}
    for name, p in head.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
            pairs[name] = (p.grad, wgrads[name])
    stats, bad, num, den = {}, {}, 0.0, 0.0
    for name, (a, b) in pairs.items():
        a, b = a.detach().double().cpu().flatten(), b.double().flatten()
        if b.abs().max().item() < 1e-3 * gmax:          # (near-)zero gradient: absolute check against the largest entry
            if (a - b).abs().max().item() > 1e-3 * gmax:
                bad[name] = ('small-gradient tensor', (a - b).abs().max().item() / gmax)
            num += (a - b).pow(2).sum().item()
            den += b.pow(2).sum().item()
            continue
        l2 = ((a - b).norm() / b.norm()).item()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        stats[name] = (l2, cos)
        num += (a - b).pow(2).sum().item()
        den += b.pow(2).sum().item()
        if l2 > l2_bar or cos < cos_bar:
            bad[name] = (round(l2, 4), round(cos, 5))
    flat = (num / den) ** 0.5
    top = sorted(stats.items(), key=lambda kv: -kv[1][0])[:4]
    print(f'bf16 training step: cls {e_cls:.2e} bbox {e_box:.2e}; flat gradient L2 {flat:.3f}; worst tensors (l2, cos): '
          f'{[(k, round(a, 4), round(b, 5)) for k, (a, b) in top]}')
    assert e_cls < out_bar and e_box < out_bar
    assert not bad, f'bf16 gradient mismatch: {sorted(bad.items(), key=lambda kv: str(kv[1]))[:8]}'
    assert flat < flat_bar, flat
    head.attn_dtype = 'fp32'


def test_head_bf16_training_toy_batch2_masked(pa):
    """toy shape, two samples, a padding mask, dropout 0.1 active (48 keys: no averaging, so the bars are the operator-level ones)."""
    oracle = O.seeded_head(4, 77, num_query=16)
    head = make_pair(pa, oracle, num_query=16).train()
    head._dropout_seed_override = 991
    metas = O.synthetic_img_metas(2, 2, (128, 192), (100, 150), seed=4)
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(2, 2, 256, 4, 6, generator=g)
    g_cls, g_box = torch.randn(6, 2, 16, 10, generator=g), torch.randn(6, 2, 16, 10, generator=g)
    masks = _dropout_masks(991, 0.1, 2, 16, 2 * 4 * 6)
    _bf16_grad_case(pa, head, oracle, feats, metas, g_cls, g_box, masks, out_bar=2e-2)


@pytest.mark.parametrize('attn_dtype', ['fp32', 'bf16'])
def test_forward_projected_equals_forward_batch2_masked(pa, attn_dtype):
    """PETRHead.forward_projected (memory_in of the C ABI: the producer already applied input_proj) on the memory the ordinary
    forward made: two samples, a padding mask, c5-sized maps - everything behind petr_head.py:390 is the same launch sequence on the
    same values, so the outputs must be EQUAL (fp32; bf16 mode rounds the same memory to the same bf16 image)."""
    oracle = O.seeded_head(5, 78, num_query=64)
    head = make_pair(pa, oracle, num_query=64).eval()
    head.attn_dtype = attn_dtype
    metas = O.synthetic_img_metas(2, 6, (512, 1408), (400, 1200), seed=6)
    g = torch.Generator().manual_seed(6)
    feats = torch.randn(2, 6, 256, 16, 44, generator=g).cuda()
    with torch.no_grad():
        ref = head([feats], metas)
        if attn_dtype == 'fp32':
            mem = head.workspace_view('memory').clone().view(2, 6, 16, 44, 256)
        else:       # the workspace holds the bf16 image in bf16 mode: recompute the fp32 memory the producer would hand over
            w = head.input_proj.weight.view(256, 256)
            mem = (torch.einsum('bnchw,oc->bnhwo', feats, w) + head.input_proj.bias).contiguous()
        got = head.forward_projected(mem, metas)
    for k in ('all_cls_scores', 'all_bbox_preds'):
        if attn_dtype == 'fp32':
            assert torch.equal(got[k], ref[k]), k
        else:
            d = (got[k] - ref[k]).abs().max().item()
            assert d <= 3e-2 * max(1.0, ref[k].abs().max().item()), (k, d)
    head.attn_dtype = 'fp32'


def test_head_p4_1408_bf16_training_step(pa):
    """BASELINE configs[2] as stated: 6x256x32x88 (L = 16 896), 900 queries, bf16, forward AND backward, eval-mode
    (dropout off) so that the float64 oracle needs no 16 896-wide mask tensors."""
    oracle = O.seeded_head(0, None, num_query=900)
    head = make_pair(pa, oracle, num_query=900)
    metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=2)
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(1, 6, 256, 32, 88, generator=g)
    g_cls, g_box = torch.randn(6, 1, 900, 10, generator=g), torch.randn(6, 1, 900, 10, generator=g)
    _bf16_grad_case(pa, head, oracle, feats, metas, g_cls, g_box)


def test_head_p4_1600_bf16_training_step(pa):
    """BASELINE configs[3] as stated (reference petr_vovnet_gridmask_p4_1600x640.py:151-152,236): 6x256x40x100 (L = 24 000),
    900 queries, bf16, forward AND backward against the float64 oracle (eval mode: no 24 000-wide mask tensors)."""
    oracle = O.seeded_head(0, None, num_query=900)
    head = make_pair(pa, oracle, num_query=900)
    metas = O.synthetic_img_metas(1, 6, (640, 1600), seed=6)
    g = torch.Generator().manual_seed(6)
    feats = torch.randn(1, 6, 256, 40, 100, generator=g)
    g_cls, g_box = torch.randn(6, 1, 900, 10, generator=g), torch.randn(6, 1, 900, 10, generator=g)
    _bf16_grad_case(pa, head, oracle, feats, metas, g_cls, g_box)


def test_headv2_800x320_bf16_training_step(pa):
    """BASELINE configs[4] as stated: PETRv2, 12 views x 20x50 (HW = 1 000: ragged K segments in the bf16 weight
    gradients), 900 queries, bf16 forward and backward."""
    kw = dict(num_query=900, v2=True, with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10)
    oracle = O.seeded_head(1, None, **kw)
    head = pa.build_head(pa.petrv2_head_cfg(num_query=900))
    head.load_state_dict(oracle.state_dict())
    head = head.cuda().eval()
    metas = O.synthetic_img_metas(1, 12, (320, 800), seed=4, with_time=True)
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(1, 12, 256, 20, 50, generator=g)
    g_cls, g_box = torch.randn(6, 1, 900, 10, generator=g), torch.randn(6, 1, 900, 10, generator=g)
    _bf16_grad_case(pa, head, oracle, feats, metas, g_cls, g_box)


def test_head_bf16_training_mode_dropout_900_queries_c5(pa):
    """c5 shape, 900 queries, TRAINING mode (all six dropout sites of every layer, masks exported to the oracle), bf16."""
    oracle = O.seeded_head(0, None, num_query=900)
    head = make_pair(pa, oracle, num_query=900).train()
    head._dropout_seed_override = 4242
    metas = O.synthetic_img_metas(1, 6, (512, 1408), seed=1)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(1, 6, 256, 16, 44, generator=g)
    g_cls, g_box = torch.randn(6, 1, 900, 10, generator=g), torch.randn(6, 1, 900, 10, generator=g)
    masks = _dropout_masks(4242, 0.1, 1, 900, 6 * 16 * 44)
    _bf16_grad_case(pa, head, oracle, feats, metas, g_cls, g_box, masks)


def test_head_bf16_attention_masked_batch2_and_v2(pa, golden_dir):
    """ragged / masked key axis through the bf16 kernel inside the head (toy fixture of the reference, batch 2 oracle)."""
    fx = np.load(os.path.join(golden_dir, 'head_toy_masked.npz'))
    oracle = O.seeded_head(2, 1234, num_query=16)
    head = make_pair(pa, oracle, num_query=16)
    head.attn_dtype = 'bf16'
    feats = torch.from_numpy(fx['feats'])
    with torch.no_grad():
        want = oracle([feats], metas_from(fx))
        got = head([feats.cuda()], metas_from(fx))
    e_cls, e_box = rel(got['all_cls_scores'], want['all_cls_scores']), rel(got['all_bbox_preds'], want['all_bbox_preds'])
    print(f'bf16 attention vs fp32 oracle, toy masked (48 keys, no averaging): cls {e_cls:.2e} bbox {e_box:.2e}')
    assert e_cls < 1e-2 and e_box < 1e-2       # SURVEY 8(d) config 3's expected bf16 tolerance


def test_headv2_800x320_forward_backward_runs(pa):
    """BASELINE configs[4] shape: PETRv2, 12 views x 20x50 (H*W = 1000 is NOT a multiple of 32: exercises the ragged
    K-segment tiles of the gradient contractions), 900 queries: forward parity + a finite, reproducible backward."""
    kw = dict(num_query=900, v2=True, with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10)
    oracle = O.seeded_head(1, None, **kw)
    head = pa.build_head(pa.petrv2_head_cfg(num_query=900))
    head.load_state_dict(oracle.state_dict())
    head = head.cuda().eval()
    metas = O.synthetic_img_metas(1, 12, (320, 800), seed=4, with_time=True)
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(1, 12, 256, 20, 50, generator=g)
    g_cls, g_box = torch.randn(6, 1, 900, 10, generator=g), torch.randn(6, 1, 900, 10, generator=g)
    fo = feats.clone().requires_grad_(True)
    want = oracle([fo], metas)
    (want['all_cls_scores'] * g_cls).sum().add((want['all_bbox_preds'] * g_box).sum()).backward()
    fg = feats.cuda().requires_grad_(True)
    got = head([fg], metas)
    torch.autograd.backward([got['all_cls_scores'], got['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    assert rel(got['all_cls_scores'], want['all_cls_scores']) < REL
    assert rel(got['all_bbox_preds'], want['all_bbox_preds']) < REL
    d = (fg.grad.double().cpu() - fo.grad.double())
    assert d.norm().item() / fo.grad.double().norm().item() < 5e-3
    wg = dict(oracle.named_parameters())
    for name in ['input_proj.weight', 'position_encoder.0.weight', 'adapt_pos3d.0.weight', 'fpe.conv_reduce.weight',
                 'reg_branches.3.task_heads.2.2.weight', 'transformer.decoder.layers.0.attentions.1.attn.in_proj_weight']:
        p = dict(head.named_parameters())[name]
        e = (p.grad.double().cpu() - wg[name].grad.double()).norm().item() / wg[name].grad.double().norm().item()
        assert e < 5e-3, (name, e)


def test_headv2_800x320_forward_bf16_mode(pa):
    """BASELINE configs[4] shape (PETRv2, 12 views x 20x50, 900 queries) in the bf16 inference mode: bf16 K/V projections,
    bf16 feature-guided-PE and position-embedding contractions, bf16 K/V cross-attention; against the fp32 CPU oracle."""
    kw = dict(num_query=900, v2=True, with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10)
    oracle = O.seeded_head(1, None, **kw)
    head = pa.build_head(pa.petrv2_head_cfg(num_query=900))
    head.load_state_dict(oracle.state_dict())
    head = head.cuda().eval()
    head.attn_dtype = 'bf16'
    metas = O.synthetic_img_metas(1, 12, (320, 800), seed=4, with_time=True)
    feats = torch.randn(1, 12, 256, 20, 50, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = oracle([feats], metas)
        got = head([feats.cuda()], metas)
    e_cls, e_box = rel(got['all_cls_scores'], want['all_cls_scores']), rel(got['all_bbox_preds'], want['all_bbox_preds'])
    print(f'bf16 mode vs fp32 oracle at v2 800x320: cls {e_cls:.2e} bbox {e_box:.2e}')
    assert e_cls < REL_BF16 and e_box < REL_BF16


@pytest.mark.parametrize('extra', [[], ['--force-reducer']])
def test_bench_contract(extra):
    """bench.py prints ONE JSON line with the driver's contract keys (+ roofline / cpu_baseline objects); the
    --force-reducer variant drives the data-parallel path (staged backward + RCCL) on one rank, including the
    'no collectives after the timed region' rule that N > 1 launches depend on."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-cpu-baseline'] + extra,
                       capture_output=True, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, MASTER_PORT='29577'))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['unit'] == 'samples/s' and d['dtype'] == 'f32' and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    assert 0.2 < rf['frac'] < 1.0
    assert (rf['traffic'] is None and rf['traffic_source'] is None) or (rf['traffic'] > 9.5e6 and 'profiles/' in rf['traffic_source'])
    assert abs(d['value'] - 1000.0 / d['ms_per_step']) < 0.02 * d['value']
    assert d['sustained']['steps'] >= 3 and d['sustained']['ms_per_step'] > 0
    assert 0.0 < d['step_frac_of_peak'] < 1.0
    # the bf16 leg of the headline workload and every other BASELINE single-GPU workload, fp32 and bf16, in the same line
    assert d['bf16']['ms_per_step'] > 0 and d['bf16']['mha_bwd_cross']['mean_launch_us'] > 0
    assert set(d['workloads']) == {'p4_1408', 'p4_1600', 'v2_800'}
    for w in d['workloads'].values():
        for mode in ('fp32', 'bf16'):
            assert w[mode]['ms_per_step'] > 0 and 0.0 < w[mode]['roofline']['frac'] < 1.0
        # (no ordering assertion between the two modes: with three timed steps one host hiccup - seen once: 22.9 ms - decides it)


@pytest.mark.parametrize('launcher', ['torch.distributed.run', 'self'])
def test_bench_two_rank_control_flow_on_one_gpu(launcher):
    """The N > 1 launch of bench.py (torch.distributed.run, one rank per process) rehearsed on this one-GPU box: both
    ranks on card 0, gloo instead of RCCL (PETR_BENCH_DEVICE / PETR_BENCH_BACKEND).  Guards the rule that nothing
    behind the timed region issues a collective on rank 0 only (cpu baseline, loss and bf16 legs are single-process
    legs) - such a leg deadlocks the real 2/4/8-GPU runs.  Timing is meaningless here; completion is the test."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
    env = dict(os.environ, PETR_BENCH_DEVICE='0', PETR_BENCH_BACKEND='gloo')
    if launcher == 'self':     # `python bench.py --gpus 2` with no launcher: bench.py starts its own two ranks
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 2 if 'global_batch' in d['config'] else True
    assert d['cpu_baseline'] is None and d['with_loss'] is None        # single-process legs are skipped


# ------------------------------------------------------------------ data parallel: the REAL head on two ranks
def _dp_worker(rank, world, port, out_dir):
    """one data-parallel rank on card 0 (gloo stands in for RCCL on a one-GPU box): real PETRHead, its own sample,
    staged backward + BucketedGradAllReduce; rank 1 starts from perturbed weights to prove the constructor's broadcast."""
    import torch.distributed as dist
    import petr_amd
    from petr_amd.dist import BucketedGradAllReduce
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=64))
    head.init_weights()
    head = head.cuda().eval()
    head._context()                              # side streams before the process group
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    if rank == 1:
        with torch.no_grad():
            head.flat_parameters().mul_(1.5)
    red = BucketedGradAllReduce(head, merge=2)
    metas = O.synthetic_img_metas(2, 2, (128, 192), (100, 150), seed=5)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2, 2, 256, 4, 6, generator=g)
    g_cls, g_box = torch.randn(6, 2, 64, 10, generator=g), torch.randn(6, 2, 64, 10, generator=g)
    x = feats[rank:rank + 1].cuda().requires_grad_(True)
    head.zero_grad_flat()
    out = head([x], metas[rank:rank + 1])
    torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']],
                            [g_cls[:, rank:rank + 1].cuda().contiguous(), g_box[:, rank:rank + 1].cuda().contiguous()])
    red.finish()
    torch.cuda.synchronize()
    torch.save({'grad': head.flat_gradients().cpu(), 'params': head.flat_parameters().cpu(), 'd_feats': x.grad.cpu()},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_real_head_two_ranks_equal_batch2(pa, tmp_path):
    """two ranks, one sample each, real head + staged backward + bucketed all-reduce (AVG) == 1/2 x the
    gradient of ONE process running both samples as a batch of 2 (the upstream gradient is given per sample, so the
    batch-2 gradient is the sum over the samples).  Both ranks on card 0, gloo instead of RCCL."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f'rank{r}.pt')) for r in range(2))
    assert torch.equal(r0['params'], r1['params'])                    # broadcast: rank 1's perturbation is gone
    torch.manual_seed(0)
    head = pa.build_head(pa.petr_head_cfg(num_query=64))
    head.init_weights()
    head = head.cuda().eval()
    assert torch.equal(head.flat_parameters().cpu(), r0['params'])
    metas = O.synthetic_img_metas(2, 2, (128, 192), (100, 150), seed=5)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2, 2, 256, 4, 6, generator=g)
    g_cls, g_box = torch.randn(6, 2, 64, 10, generator=g), torch.randn(6, 2, 64, 10, generator=g)
    x = feats.cuda().requires_grad_(True)
    head.zero_grad_flat()
    out = head([x], metas)
    torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls.cuda(), g_box.cuda()])
    want = 0.5 * head.flat_gradients().cpu().double()
    n = head.gradient_buckets()[-1][1]                                # code_weights beyond it carry no gradient
    for r in (r0, r1):
        got = r['grad'].double()
        err = (got[:n] - want[:n]).abs().max().item() / want[:n].abs().max().item()
        assert err < 2e-5, err                                        # float-atomic ordering + batch-2 vs batch-1 tiling
    assert torch.equal(r0['grad'], r1['grad'])                        # both ranks hold the same averaged gradient
    assert (torch.cat([r0['d_feats'], r1['d_feats']]).double() - x.grad.cpu().double()).abs().max().item() < 1e-4 * \
        x.grad.abs().max().item()


def test_head_autograd_guards_and_deepcopy(pa):
    """(i) an in-place edit of all_bbox_preds between forward and backward trips autograd's version check (the
    box backward reads that buffer); (ii) a second backward of one forward is refused (its workspace is back in the pool);
    (iii) a frozen first parameter does not re-zero the flat gradient on every backward (gradient accumulation);
    (iv) copy.deepcopy of a head that has already run gives an independent, working head."""
    import copy
    oracle = O.seeded_head(2, 1234, num_query=16)
    head = make_pair(pa, oracle, num_query=16)
    metas = O.synthetic_img_metas(1, 2, (128, 192), (100, 150), seed=3)
    feats = torch.randn(1, 2, 256, 4, 6, generator=torch.Generator().manual_seed(0)).cuda()
    out = head([feats.clone().requires_grad_(True)], metas)
    out['all_bbox_preds'][..., 0] += 1.0
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        (out['all_cls_scores'].sum() + out['all_bbox_preds'].sum()).backward()
    out = head([feats.clone().requires_grad_(True)], metas)
    loss = out['all_cls_scores'].sum() + out['all_bbox_preds'].sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='already been backpropagated'):
        loss.backward()
    # (iii) accumulation with the first flat parameter frozen
    first = head._grad_views[0][0]
    first.requires_grad_(False)
    head.zero_grad_flat()
    for _ in range(2):
        o2 = head([feats], metas)
        (o2['all_cls_scores'].sum() + o2['all_bbox_preds'].sum()).backward()
    twice = head.flat_gradients().clone()
    head.zero_grad_flat()
    o2 = head([feats], metas)
    (o2['all_cls_scores'].sum() + o2['all_bbox_preds'].sum()).backward()
    once = head.flat_gradients().clone()
    assert (twice - 2 * once).abs().max().item() < 1e-4 * once.abs().max().item()
    first.requires_grad_(True)
    # (iv) deepcopy after use
    twin = copy.deepcopy(head)
    with torch.no_grad():
        a, b = head([feats], metas), twin([feats], metas)
    assert torch.equal(a['all_cls_scores'], b['all_cls_scores'])
    with torch.no_grad():
        twin.input_proj.weight.mul_(2.0)                      # independent storage
        c = head([feats], metas)
    assert torch.equal(a['all_cls_scores'], c['all_cls_scores'])


def test_head_release_gives_back_streams_and_workspaces(pa):
    """release(): the stream context and the pooled workspaces go away now (a process that builds several heads one after
    the other must not leave live side streams behind - they would share hardware queues with the next head's), and the
    head keeps working: the next forward re-creates both and returns the same bits (eval forward is reproducible)."""
    oracle = O.seeded_head(2, 77, num_query=16)
    head = make_pair(pa, oracle, num_query=16).eval()
    metas = O.synthetic_img_metas(1, 2, (128, 192), (100, 150), seed=5)
    feats = torch.randn(1, 2, 256, 4, 6, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        a = head([feats], metas)
        assert head._ctx is not None
        head.release()
        assert head._ctx is None and not head._runs and not head._free_ws
        b = head([feats], metas)
    assert head._ctx is not None
    assert torch.equal(a['all_cls_scores'], b['all_cls_scores']) and torch.equal(a['all_bbox_preds'], b['all_bbox_preds'])
