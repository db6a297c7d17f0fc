"""Per-operator Python entry points over the C ABI (torch tensors in, torch tensors out).

Torch is plumbing here: device memory, the current HIP stream, nothing else.  Every function
launches hand-written gfx950 kernels through libpetr_hip.so on ``torch.cuda.current_stream()``;
there is no fallback path.
"""
import ctypes as C

import torch

from . import _C


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32, 'expected a float32 CUDA tensor'
    return t


def coords3d(img2lidar, depth, B, N, H, W, pad_h, pad_w, position_range, eps=1e-5, want_volume=True, want_mask=False):
    """K1: [B*N,3D,H,W] logit volume (+ optional coords_mask [B,N,H,W] bool)."""
    L = _C.lib()
    D = depth.numel()
    dev = depth.device
    out = torch.empty((B * N, 3 * D, H, W), dtype=torch.float32, device=dev) if want_volume else None
    cmask = torch.empty((B, N, H, W), dtype=torch.uint8, device=dev) if want_mask else None
    a = _C.Coords3dArgs(_ptr(_f32(img2lidar).contiguous()), _ptr(_f32(depth)), _ptr(out), _ptr(cmask), B, N, H, W, D,
                        float(pad_h), float(pad_w), (C.c_float * 6)(*[float(v) for v in position_range]), float(eps))
    _C.check(L.petr_coords3d_fwd(C.byref(a), _stream()), 'petr_coords3d_fwd')
    return out, (cmask.bool() if cmask is not None else None)


def sine3d(mask, dim_t, B, N, H, W, normalize=True, scale=6.283185307179586, eps=1e-6, offset=0.0):
    """K3a: SinePositionalEncoding3D features [B,N,3F,H,W]; mask uint8/bool [B,N,H,W] or None (all valid)."""
    L = _C.lib()
    F = dim_t.numel()
    if mask is not None:
        mask = mask.to(torch.uint8).contiguous()
    out = torch.empty((B, N, 3 * F, H, W), dtype=torch.float32, device=dim_t.device)
    a = _C.Sine3dArgs(_ptr(mask), _ptr(_f32(dim_t)), _ptr(out), B, N, H, W, F, int(normalize), float(scale), float(eps),
                      float(offset))
    _C.check(L.petr_sine3d_fwd(C.byref(a), _stream()), 'petr_sine3d_fwd')
    return out


def posemb3d(pos, dim_t):
    L = _C.lib()
    n, F = pos.shape[0], dim_t.numel()
    out = torch.empty((n, 3 * F), dtype=torch.float32, device=pos.device)
    _C.check(L.petr_posemb3d_fwd(_ptr(_f32(pos).contiguous()), _ptr(_f32(dim_t)), _ptr(out), n, F, _stream()),
             'petr_posemb3d_fwd')
    return out


def posemb3d_bwd(pos, dim_t, dout):
    L = _C.lib()
    n, F = pos.shape[0], dim_t.numel()
    dpos = torch.zeros_like(pos)
    _C.check(L.petr_posemb3d_bwd(_ptr(_f32(pos).contiguous()), _ptr(dim_t), _ptr(_f32(dout).contiguous()), _ptr(dpos), n,
                                 F, _stream()), 'petr_posemb3d_bwd')
    return dpos


def gemm_raw(**kw):
    """Direct access to petr_gemm; keyword names are the struct fields (tensors for pointers)."""
    L = _C.lib()
    g = _C.GemmArgs()
    keep = []
    for k, v in kw.items():
        if k == 'drop':
            v = _C.dropout(v)
        elif isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(g, k, v)
    _C.check(L.petr_gemm(C.byref(g), _stream()), 'petr_gemm')


def linear(x, w, bias=None, relu=False, residual=None, a2=None, a2_rows=0, a2_ncols=0, out=None, accumulate=False,
           split_k=1, drop=None):
    """y = drop(act((x [+ a2]) @ w.T + bias [+ residual]));  x [M,K] row-major, w [N,K] (nn.Linear layout);
    drop = (seed, site, p) or None."""
    M, K = x.shape
    N = w.shape[0]
    if split_k > 1:
        out = torch.empty((split_k, M, N), dtype=torch.float32, device=x.device)
    elif out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    flags = (_C.GEMM_RELU if relu else 0) | (_C.GEMM_ACCUMULATE if accumulate else 0)
    kw = dict(a=_f32(x), lda=x.stride(0), a_kcontig=1, b=_f32(w), ldb=w.stride(0), b_kcontig=1, c=out, ldc=N, M=M, N=N, K=K,
              flags=flags, alpha=1.0, split_k=split_k, c_split_stride=M * N)
    if bias is not None:
        kw['bias'] = bias
    if residual is not None:
        kw.update(r=residual, ldr=residual.stride(0))
    if a2 is not None:
        kw.update(a2=a2, a2_rows=a2_rows, a2_ncols=a2_ncols)
    if drop is not None:
        kw['drop'] = drop
    gemm_raw(**kw)
    return out


def wgrad_grouped(items):
    """``petr_wgrad_grouped``: items = [(dy [K, M], x [K, N], dw [M, N] (+=), db [M] or None, ksplit)], one launch."""
    arr = (_C.WgradItem * len(items))()
    for i, (dy, x, dw, db, ks) in enumerate(items):
        arr[i] = _C.WgradItem(_ptr(_f32(dy)), dy.stride(0), _ptr(_f32(x)), x.stride(0), _ptr(dw), dw.stride(0),
                              _ptr(db) if db is not None else None, dw.shape[0], dw.shape[1], dy.shape[0], ks)
    _C.check(_C.lib().petr_wgrad_grouped(arr, len(items), _stream()), 'petr_wgrad_grouped')


def branch_fwd(x, w1, b1, w2, b2, w3=None, b3=None, ln1=None, ln2=None, eps=1e-5, save=True):
    """``petr_branch_fwd``: out = Linear3(act2(Linear2(act1(Linear1(x))))) over rows [G, rows, 256]; weights in nn.Linear layout
    with a leading group dimension ([G, 256, 256], [G, 256], w3 [G, n_out, 256]); ln = (gamma [G, 256], beta [G, 256]) makes the
    activation ReLU(LayerNorm(.)).  Returns a dict with out and the rows the backward reads (h1, y1, h2, y2, mean / rstd)."""
    G, rows, Cc = x.shape
    dev = x.device
    w1t, w2t = w1.transpose(1, 2).contiguous(), w2.transpose(1, 2).contiguous()
    # one flat parameter buffer per kind so that every pointer shares the group stride
    def cat(*ts):
        flat = torch.cat([t.reshape(G, -1) for t in ts], dim=1).contiguous()
        offs, o = [], 0
        for t in ts:
            offs.append(o)
            o += t[0].numel()
        return flat, offs
    ts = [b1, b2] + ([w3, b3] if w3 is not None else []) + (list(ln1) if ln1 else []) + (list(ln2) if ln2 else [])
    pad = (-sum(t[0].numel() for t in ts)) % 4
    if pad:
        ts.append(torch.zeros(G, pad, device=dev))
    flat, offs = cat(*[_f32(t) for t in ts])
    ptr = lambda i: flat.data_ptr() + 4 * offs[i]
    a = _C.BranchFwdArgs()
    a.x = _ptr(_f32(x))
    a.w1t, a.w2t, a.wt_gs = _ptr(w1t), _ptr(w2t), Cc * Cc
    a.b1, a.b2, a.param_gs = ptr(0), ptr(1), flat.shape[1]
    i = 2
    n_out = 0
    if w3 is not None:
        a.w3, a.b3 = ptr(i), ptr(i + 1)
        n_out = w3.shape[1]
        i += 2
    if ln1:
        a.g1, a.be1 = ptr(i), ptr(i + 1)
        i += 2
    if ln2:
        a.g2, a.be2 = ptr(i), ptr(i + 1)
        i += 2
    res = {}
    if save:
        for k in ('y1', 'y2') + (('h1',) if ln1 else ()) + (('h2',) if ln2 else ()):
            res[k] = torch.empty((G, rows, Cc), dtype=torch.float32, device=dev)
            setattr(a, k, _ptr(res[k]))
        for k, on in (('mean1', ln1), ('rstd1', ln1), ('mean2', ln2), ('rstd2', ln2)):
            if on:
                res[k] = torch.empty((G, rows), dtype=torch.float32, device=dev)
                setattr(a, k, _ptr(res[k]))
    if w3 is not None:
        res['out'] = torch.empty((G, rows, n_out), dtype=torch.float32, device=dev)
        a.out, a.n_out = _ptr(res['out']), n_out
    a.rows, a.groups, a.eps = rows, G, eps
    _C.check(_C.lib().petr_branch_fwd(C.byref(a), _stream()), 'petr_branch_fwd')
    res['_keep'] = (flat, w1t, w2t)
    res['_launch'] = lambda: _C.check(_C.lib().petr_branch_fwd(C.byref(a), _stream()), 'petr_branch_fwd')    # timing scripts
    return res


def branch_bwd(fwd, w1, w2, w3=None, d_out=None, d_y2=None, ln1=None, ln2=None):
    """``petr_branch_bwd`` on the tensors ``branch_fwd`` saved (``fwd``): returns dict(d_x, d_h1, d_h2[, dg1, dbe1, dg2, dbe2]).
    Weights in nn.Linear layout with the group dimension in front; ln = (gamma [G, 256], beta) as in the forward."""
    G, rows, Cc = fwd['y1'].shape
    dev = fwd['y1'].device
    # one flat buffer so that w1 / w2 / w3 / gamma share the group stride
    ts = [w1, w2] + ([w3] if w3 is not None else []) + ([ln2[0]] if ln2 else []) + ([ln1[0]] if ln1 else [])
    flat = torch.cat([_f32(t).reshape(G, -1) for t in ts], dim=1).contiguous()
    offs, o = [], 0
    for t in ts:
        offs.append(o)
        o += t[0].numel()
    assert flat.shape[1] % 4 == 0
    ptr = lambda i: flat.data_ptr() + 4 * offs[i]
    a = _C.BranchBwdArgs()
    a.w1, a.w2, a.param_gs = ptr(0), ptr(1), flat.shape[1]
    i = 2
    keep = [flat]
    if w3 is not None:
        d_out = _f32(d_out).contiguous()
        a.w3, a.d_out, a.n_out = ptr(i), _ptr(d_out), w3.shape[1]
        keep.append(d_out)
        dw3 = torch.zeros((G, flat.shape[1]), dtype=torch.float32, device=dev)      # dW3 / db3 at the group stride of w3
        db3 = torch.zeros((G, flat.shape[1]), dtype=torch.float32, device=dev)
        a.dw3, a.db3 = dw3.data_ptr() + 4 * offs[i], db3.data_ptr() + 4 * offs[i]
        keep += [dw3, db3]
        i += 1
    else:
        d_y2 = _f32(d_y2).contiguous()
        a.d_y2 = _ptr(d_y2)
        keep.append(d_y2)
    res = {k: torch.empty((G, rows, Cc), dtype=torch.float32, device=dev) for k in ('d_x', 'd_h1', 'd_h2')}
    grads = torch.zeros((G, flat.shape[1]), dtype=torch.float32, device=dev)     # dgamma in the parameters' own layout
    betas = torch.zeros((G, flat.shape[1]), dtype=torch.float32, device=dev)
    if ln2:
        a.g2, a.h2, a.mean2, a.rstd2 = ptr(i), _ptr(fwd['h2']), _ptr(fwd['mean2']), _ptr(fwd['rstd2'])
        a.dg2, a.dbe2 = grads.data_ptr() + 4 * offs[i], betas.data_ptr() + 4 * offs[i]
        res['dg2'], res['dbe2'] = grads[:, offs[i]:offs[i] + Cc], betas[:, offs[i]:offs[i] + Cc]
        i += 1
    if ln1:
        a.g1, a.h1, a.mean1, a.rstd1 = ptr(i), _ptr(fwd['h1']), _ptr(fwd['mean1']), _ptr(fwd['rstd1'])
        a.dg1, a.dbe1 = grads.data_ptr() + 4 * offs[i], betas.data_ptr() + 4 * offs[i]
        res['dg1'], res['dbe1'] = grads[:, offs[i]:offs[i] + Cc], betas[:, offs[i]:offs[i] + Cc]
        i += 1
    a.y1, a.y2 = _ptr(fwd['y1']), _ptr(fwd['y2'])
    a.d_h2, a.d_h1, a.d_x = _ptr(res['d_h2']), _ptr(res['d_h1']), _ptr(res['d_x'])
    a.rows, a.groups = rows, G
    _C.check(_C.lib().petr_branch_bwd(C.byref(a), _stream()), 'petr_branch_bwd')
    if w3 is not None:
        n3 = w3.shape[1]
        res['dw3'] = dw3[:, offs[2]:offs[2] + n3 * Cc].view(G, n3, Cc)
        res['db3'] = db3[:, offs[2]:offs[2] + n3]
    res['_keep'] = keep
    res['_launch'] = lambda: _C.check(_C.lib().petr_branch_bwd(C.byref(a), _stream()), 'petr_branch_bwd')
    return res


def dropout_mask(drop, rows, cols, device='cuda'):
    """The keep mask (bool [rows, cols]) a kernel applies for drop = (seed, site, p): parity tests hand it to the oracle."""
    L = _C.lib()
    keep = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    d = _C.dropout(drop)
    _C.check(L.petr_dropout_mask(C.byref(d), C.c_long(rows), C.c_long(cols), _ptr(keep), _stream()), 'petr_dropout_mask')
    return keep.bool()


def attn_out_ln(a, w, bias, residual, gamma, beta, partials=None, n_split=1, BHQ=None, attn_scale=1.0, drop=None, add2=None,
                add2_rows=0, eps=1e-5, w2=None, bias2=None):
    """petr_attn_out_ln: [merge of the attention partials ->] a W^T + bias -> dropout -> + residual -> LayerNorm (-> + add2).
    ``a``: [M, 256] attention output (n_split <= 1) or an uninitialised [M, 256] buffer that receives the merged output.
    w, w2: nn.Linear layout [out, in] (transposed here: the kernel reads them k-major).  Returns (y, y2 or None, z, mean, rstd, lse or None)."""
    L = _C.lib()
    M = a.shape[0]
    y, z = torch.empty_like(a), torch.empty_like(a)
    y2 = torch.empty_like(a) if add2 is not None else None
    mean, rstd = torch.empty(M, device=a.device), torch.empty(M, device=a.device)
    B, H, Q = BHQ if BHQ is not None else (1, 8, M)
    lse = torch.empty(B * H * Q, device=a.device) if n_split > 1 else None
    o_part = partials if n_split > 1 else None
    ml_part = partials[n_split * B * H * Q * 32:] if n_split > 1 else None
    wT = w.t().contiguous()                                   # the kernel streams the weights k-major
    w2T = w2.t().contiguous() if w2 is not None else None
    args = _C.AttnOutLnArgs(_ptr(a), _ptr(o_part), _ptr(ml_part), n_split, B, H, Q, float(attn_scale), _ptr(lse), _ptr(wT), _ptr(bias),
                            _ptr(residual), _C.dropout(drop), _ptr(gamma), _ptr(beta), float(eps), _ptr(z), _ptr(mean), _ptr(rstd),
                            _ptr(y), _ptr(y2), _ptr(add2), add2_rows, M, _ptr(w2T), _ptr(bias2), None)
    out2 = torch.empty_like(a) if w2 is not None else None
    args.out2 = _ptr(out2)
    _C.check(L.petr_attn_out_ln(C.byref(args), _stream()), 'petr_attn_out_ln')
    if w2 is not None:
        return y, y2, z, mean, rstd, lse, out2
    return y, y2, z, mean, rstd, lse


def ln_proj(x, gamma, beta, w2, bias2=None, bias=None, residual=None, drop=None, add2=None, add2_rows=0, n2_pos=0, eps=1e-5):
    """petr_ln_proj: LN(drop(sum_p x[p] + bias) + residual) and the projections of its result in one launch.
    x [M,256] or [P,M,256]; w2 [256 n2, 256].  Returns (y, y2 or None, z, mean, rstd, out2 [M, 256 n2])."""
    L = _C.lib()
    if x.dim() == 3:
        P, M, Cc = x.shape
    else:
        P, (M, Cc) = 1, x.shape
    n2 = w2.shape[0] // 256
    y, z = torch.empty((M, Cc), device=x.device), torch.empty((M, Cc), device=x.device)
    y2 = torch.empty_like(y) if add2 is not None else None
    mean, rstd = torch.empty(M, device=x.device), torch.empty(M, device=x.device)
    out2 = torch.empty((M, 256 * n2), device=x.device)
    w2T = w2.t().contiguous()                                 # [256, 256 n2]: the kernel streams the weight k-major
    a = _C.LnProjArgs(_ptr(_f32(x)), P, M * Cc, _ptr(bias), _ptr(residual), _C.dropout(drop), _ptr(gamma), _ptr(beta), float(eps),
                      _ptr(z), _ptr(mean), _ptr(rstd), _ptr(y), _ptr(y2), _ptr(add2), add2_rows, M, _ptr(w2T), _ptr(bias2), _ptr(out2),
                      n2, n2_pos)
    _C.check(L.petr_ln_proj(C.byref(a), _stream()), 'petr_ln_proj')
    return y, y2, z, mean, rstd, out2


def ln_bwd_proj(z, mean, rstd, gamma, dy, w, dy_residual=None, drop=None, alpha=1.0, relu_mask=None, pre_a=None, pre_w=None):
    """petr_ln_bwd_proj: LayerNorm backward + dx = (dz_drop or dz) @ w for the nn.Linear weight w [256, K_in].
    dy [M,256] or [P,M,256].  Returns (dz, dz_drop or None, dgamma, dbeta, out [M, K_in])."""
    L = _C.lib()
    if dy is None:                       # the upstream gradient comes from the leading product (+ dy_residual) alone
        P, (M, Cc) = 0, z.shape
    elif dy.dim() == 3:
        P, M, Cc = dy.shape
    else:
        P, (M, Cc) = 1, dy.shape
    w = w.contiguous() if w is not None else None       # dx = dz W: the nn.Linear weight is already k-major for this product
    pre_w = pre_w.contiguous() if pre_w is not None else None
    n2 = w.shape[1] // 256 if w is not None else 0      # w None: LayerNorm backward behind the leading product only
    pre_n = pre_a.shape[1] // 256 if pre_a is not None else 0
    dz = torch.empty((M, Cc), device=z.device)
    dzd = torch.empty_like(dz) if drop is not None else None
    dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
    out = torch.empty((M, 256 * n2), device=z.device) if n2 else None
    dyc = _f32(dy).contiguous() if dy is not None else None
    a = _C.LnBwdProjArgs(_ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dyc), P, M * Cc, _ptr(dy_residual),
                         _ptr(dz), _ptr(dzd), _C.dropout(drop), _ptr(dg), _ptr(db), M, _ptr(w), n2, float(alpha), _ptr(relu_mask),
                         _ptr(out), _ptr(pre_a), _ptr(pre_w), pre_n)
    _C.check(L.petr_ln_bwd_proj(C.byref(a), _stream()), 'petr_ln_bwd_proj')
    return dz, dzd, dg, db, out


def ffn_fwd(x, w1, b1, w2, n_split=4, drop=None, store_hidden=True, transposed=None):
    """petr_ffn_fwd: hidden = drop(relu(x w1^T + b1)) and the per-slice partial sums of hidden w2^T in one launch.
    x [M,256], w1 [F,256], w2 [256,F] (nn.Linear layout; the kernel reads the transposes, made here unless ``transposed`` =
    (w1t, w2t) is given).  Returns (hidden [M,F] or None, part [n_split, M, 256])."""
    L = _C.lib()
    M, F = x.shape[0], w1.shape[0]
    w1t, w2t = transposed if transposed is not None else (w1.t().contiguous(), w2.t().contiguous())
    hidden = torch.empty((M, F), device=x.device) if store_hidden else None
    part = torch.empty((n_split, M, 256), device=x.device)
    a = _C.FfnFwdArgs(_ptr(_f32(x)), _ptr(w1t), _ptr(b1), _ptr(w2t), _ptr(hidden), _ptr(part), M * 256, _C.dropout(drop), M, F, n_split)
    _C.check(L.petr_ffn_fwd(C.byref(a), _stream()), 'petr_ffn_fwd')
    return hidden, part


def ffn_bwd(dy, w1, w2, hidden, alpha=1.0, n_split=4):
    """petr_ffn_bwd: d_hidden = alpha * (dy @ w2) masked by hidden > 0, and the per-slice partial sums of d_hidden @ w1.
    dy [M,256], w1 [F,256], w2 [256,F], hidden [M,F].  Returns (d_hidden [M,F], part [n_split, M, 256])."""
    L = _C.lib()
    M, F = dy.shape[0], w1.shape[0]
    dh = torch.empty((M, F), device=dy.device)
    part = torch.empty((n_split, M, 256), device=dy.device)
    a = _C.FfnBwdArgs(_ptr(_f32(dy)), _ptr(w2), _ptr(hidden), float(alpha), _ptr(w1), _ptr(dh), _ptr(part), M * 256, M, F, n_split)
    _C.check(L.petr_ffn_bwd(C.byref(a), _stream()), 'petr_ffn_bwd')
    return dh, part


def dropout_bits(drop, BH, Q, L, device='cuda'):
    """The attention-dropout mask of ``dropout_mask(drop, BH * Q, L)`` packed for the attention kernels:
    (query-major words for ``mha_fwd*``, key-major words for ``mha_bwd*``), both int32 [petr_dropout_bits_words]."""
    Lb = _C.lib()
    n = Lb.petr_dropout_bits_words(BH, Q, L)
    bq = torch.empty(n, dtype=torch.int32, device=device)
    bk = torch.empty(n, dtype=torch.int32, device=device)
    d = _C.dropout(drop)
    _C.check(Lb.petr_dropout_bits(C.byref(d), BH, Q, L, _ptr(bq), _ptr(bk), _stream()), 'petr_dropout_bits')
    return bq, bk


def conv1x1(x, w, bias=None, relu=False, out=None, accumulate=False):
    """x [V, C_in, HW] (NCHW views) -> token-major [V*HW, C_out]; w [C_out, C_in]."""
    V, Cin, HW = x.shape
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((V * HW, Cout), dtype=torch.float32, device=x.device)
    flags = (_C.GEMM_RELU if relu else 0) | (_C.GEMM_ACCUMULATE if accumulate else 0)
    kw = dict(a=_f32(x), lda=HW, a_kcontig=0, a_bs0=Cin * HW, b=_f32(w), ldb=w.stride(0), b_kcontig=1, c=out, ldc=Cout,
              c_bs0=HW * Cout, M=HW, N=Cout, K=Cin, nb0=V, flags=flags, alpha=1.0)
    if bias is not None:
        kw['bias'] = bias
    gemm_raw(**kw)
    return out


def colsum(x, out=None, accumulate=False):
    L = _C.lib()
    M, N = x.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.petr_colsum_workspace_bytes(N) // 4, dtype=torch.float32, device=x.device)
    _C.check(L.petr_colsum(_ptr(_f32(x)), x.stride(0), M, N, _ptr(out), int(accumulate), _ptr(ws), _stream()), 'petr_colsum')
    return out


def layernorm(x, gamma, beta, bias=None, residual=None, relu=False, nan_to_num=False, eps=1e-5, save_stats=False,
              drop=None):
    """y = LN(drop(sum_p x[p] + bias) + residual); x [M,C] or [P,M,C] (split-K partial slabs); drop = (seed, site, p)."""
    L = _C.lib()
    if x.dim() == 3:
        P, M, Cc = x.shape
    else:
        P, (M, Cc) = 1, x.shape
    y = torch.empty((M, Cc), dtype=torch.float32, device=x.device)
    z = mean = rstd = None
    if save_stats:
        z = torch.empty_like(y)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    flags = (_C.LN_RELU if relu else 0) | (_C.LN_NAN_TO_NUM if nan_to_num else 0)
    a = _C.LayerNormArgs(_ptr(_f32(x)), P, M * Cc, _ptr(bias), _ptr(residual), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(z),
                         _ptr(mean), _ptr(rstd), M, Cc, float(eps), flags, None, None, 0, _C.dropout(drop))
    _C.check(L.petr_layernorm_fwd(C.byref(a), _stream()), 'petr_layernorm_fwd')
    return (y, z, mean, rstd) if save_stats else y


def layernorm_bwd(z, mean, rstd, gamma, dy, y=None, relu=False, drop=None):
    """Returns (dz, dgamma, dbeta) and, with drop = (seed, site, p), also dz * keep / (1 - p)."""
    L = _C.lib()
    M, Cc = z.shape
    dz = torch.empty_like(z)
    dz_drop = torch.empty_like(z) if drop is not None else None
    dgamma = torch.zeros_like(gamma)
    dbeta = torch.zeros_like(gamma)
    ws = torch.empty(L.petr_layernorm_bwd_workspace_bytes(M, Cc) // 4, dtype=torch.float32, device=z.device)
    a = _C.LayerNormBwdArgs(_ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(_f32(dy).contiguous()), _ptr(y), _ptr(dz),
                            _ptr(dgamma), _ptr(dbeta), _ptr(ws), M, Cc, _C.LN_RELU if relu else 0, 0, 1, 0, None,
                            _ptr(dz_drop), _C.dropout(drop))
    _C.check(L.petr_layernorm_bwd(C.byref(a), _stream()), 'petr_layernorm_bwd')
    return (dz, dgamma, dbeta) if drop is None else (dz, dgamma, dbeta, dz_drop)


def _bhsd(t):
    assert t.dim() == 4 and t.stride(3) == 1 and t.shape[3] == 32, 'expected a [B,H,S,32] view with contiguous head dim'
    return t.stride(0), t.stride(1), t.stride(2)


def mha_fwd(q, k, v, key_padding_mask=None, scale=None, n_split=0, need_lse=True, dynamic=False, drop=None, drop_bits=None,
            defer_merge=False):
    """softmax(scale q k^T + mask) v for [B,H,S,32] (strided) views.  Returns (o [B,H,Q,32], lse [B,H,Q]).
    ``dynamic``: the L-split workers draw K/V tiles from per-query-block ticket counters (zeroed here)."""
    L = _C.lib()
    B, H, Q, _ = q.shape
    Lk = k.shape[2]
    scale = float(scale if scale is not None else 32 ** -0.5)
    o = torch.empty((B, H, Q, 32), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, H, Q), dtype=torch.float32, device=q.device) if need_lse else None
    ns = n_split if n_split > 0 else L.petr_mha_choose_split(B, H, Q, Lk)
    nbytes = L.petr_mha_fwd_workspace_bytes(B, H, Q, Lk, ns)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=q.device)
    kpm = key_padding_mask.to(torch.uint8).contiguous() if key_padding_mask is not None else None
    sched = torch.zeros(B * H * ((Q + 127) // 128), dtype=torch.int32, device=q.device) if dynamic else None
    a = _C.MhaFwdArgs(_ptr(_f32(q)), *_bhsd(q), _ptr(_f32(k)), *_bhsd(k), _ptr(_f32(v)), *_bhsd(v), _ptr(o), *_bhsd(o),
                      _ptr(lse), _ptr(kpm), B, H, Q, Lk, scale, ns, _ptr(ws), nbytes, _C.dropout(drop), _ptr(sched), _ptr(drop_bits),
                      int(defer_merge))
    _C.check(L.petr_mha_fwd(C.byref(a), _stream()), 'petr_mha_fwd')
    if defer_merge:
        return ws, ns          # the L-split partials (o_part, then ml_part) and their count; o / lse are not written
    return o, lse


def cast_bf16(x):
    """fp32 -> bfloat16 copy (round to nearest even) on the current stream: the K/V operands of ``mha_fwd_bf16``."""
    L = _C.lib()
    x = _f32(x).contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _C.check(L.petr_cast_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), 'petr_cast_bf16')
    return y


def mha_fwd_bf16(q, k, v, key_padding_mask=None, scale=None, n_split=0, need_lse=True, drop=None, drop_bits=None):
    """``mha_fwd`` with bfloat16 K/V ([B,H,L,32] strided views), fp32 Q / output / softmax (BASELINE configs 3-5)."""
    L = _C.lib()
    assert k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16, 'mha_fwd_bf16: K and V must be bfloat16'
    B, H, Q, _ = q.shape
    Lk = k.shape[2]
    scale = float(scale if scale is not None else 32 ** -0.5)
    o = torch.empty((B, H, Q, 32), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, H, Q), dtype=torch.float32, device=q.device) if need_lse else None
    nbytes = L.petr_mha_fwd_bf16_workspace_bytes(B, H, Q, Lk, n_split)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=q.device)
    kpm = key_padding_mask.to(torch.uint8).contiguous() if key_padding_mask is not None else None
    a = _C.MhaFwdArgs(_ptr(_f32(q)), *_bhsd(q), _ptr(k), *_bhsd(k), _ptr(v), *_bhsd(v), _ptr(o), *_bhsd(o),
                      _ptr(lse), _ptr(kpm), B, H, Q, Lk, scale, n_split, _ptr(ws), nbytes, _C.dropout(drop), None, _ptr(drop_bits))
    _C.check(L.petr_mha_fwd_bf16(C.byref(a), _stream()), 'petr_mha_fwd_bf16')
    return o, lse


def mha_bwd(q, k, v, o, do, lse, key_padding_mask=None, scale=None, drop=None, drop_bits=None):
    L = _C.lib()
    B, H, Q, _ = q.shape
    Lk = k.shape[2]
    scale = float(scale if scale is not None else 32 ** -0.5)
    dq = torch.zeros((B, H, Q, 32), dtype=torch.float32, device=q.device)
    dk = torch.zeros((B, H, Lk, 32), dtype=torch.float32, device=q.device)
    dv = torch.zeros_like(dk)
    nbytes = L.petr_mha_bwd_workspace_bytes(B, H, Q, Lk)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=q.device)
    kpm = key_padding_mask.to(torch.uint8).contiguous() if key_padding_mask is not None else None
    a = _C.MhaBwdArgs(_ptr(_f32(q)), *_bhsd(q), _ptr(_f32(k)), *_bhsd(k), _ptr(_f32(v)), *_bhsd(v), _ptr(_f32(o)), *_bhsd(o),
                      _ptr(_f32(do)), *_bhsd(do), _ptr(lse), _ptr(kpm), _ptr(dq), *_bhsd(dq), _ptr(dk), *_bhsd(dk),
                      _ptr(dv), *_bhsd(dv), B, H, Q, Lk, scale, _ptr(ws), nbytes, _C.dropout(drop), _ptr(drop_bits))
    _C.check(L.petr_mha_bwd(C.byref(a), _stream()), 'petr_mha_bwd')
    return dq, dk, dv


def mha_bwd_bf16(q, k, v, o, do, lse, key_padding_mask=None, scale=None, drop=None, overwrite=False, dkv_bf16=False,
                 drop_bits=None):
    """``mha_bwd`` with bfloat16 K/V ([B,H,L,32] strided views): the gradient of ``mha_fwd_bf16``; fp32 dq, dk, dv.
    ``overwrite``: dk / dv are stored into UNINITIALISED buffers (``dkv_overwrite``) instead of accumulated into zeros."""
    L = _C.lib()
    assert k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16, 'mha_bwd_bf16: K and V must be bfloat16'
    B, H, Q, _ = q.shape
    Lk = k.shape[2]
    scale = float(scale if scale is not None else 32 ** -0.5)
    dq = torch.zeros((B, H, Q, 32), dtype=torch.float32, device=q.device)
    if overwrite:
        dt = torch.bfloat16 if dkv_bf16 else torch.float32          # dkv_bf16: the kernel stores bf16 (needs overwrite)
        dk = torch.full((B, H, Lk, 32), float('nan'), dtype=dt, device=q.device)      # poison: must be overwritten
        dv = torch.full_like(dk, float('nan'))
    else:
        dk = torch.zeros((B, H, Lk, 32), dtype=torch.float32, device=q.device)
        dv = torch.zeros_like(dk)
    kpm = key_padding_mask.to(torch.uint8).contiguous() if key_padding_mask is not None else None
    a = _C.MhaBwdBf16Args(_ptr(_f32(q)), *_bhsd(q), _ptr(k), *_bhsd(k), _ptr(v), *_bhsd(v), _ptr(_f32(o)), *_bhsd(o),
                          _ptr(_f32(do)), *_bhsd(do), _ptr(lse), _ptr(kpm), _ptr(dq), *_bhsd(dq), _ptr(dk), *_bhsd(dk),
                          _ptr(dv), *_bhsd(dv), B, H, Q, Lk, scale, None, 0, _C.dropout(drop), _ptr(drop_bits), int(overwrite), int(dkv_bf16))
    _C.check(L.petr_mha_bwd_bf16(C.byref(a), _stream()), 'petr_mha_bwd_bf16')
    return dq, dk, dv


def bbox_epilogue(reg, ref, Q, pc_range, time_div=0.0, eps=1e-5):
    L = _C.lib()
    rows, code = reg.shape
    out = torch.empty_like(reg)
    a = _C.BboxArgs(_ptr(_f32(reg).contiguous()), _ptr(_f32(ref).contiguous()), _ptr(out), rows, Q, code,
                    (C.c_float * 6)(*[float(v) for v in pc_range]), float(time_div), float(eps))
    _C.check(L.petr_bbox_epilogue_fwd(C.byref(a), _stream()), 'petr_bbox_epilogue_fwd')
    return out


def bbox_epilogue_bwd(out, ref, dout, Q, pc_range, time_div=0.0, eps=1e-5):
    L = _C.lib()
    rows, code = out.shape
    dreg = torch.empty_like(out)
    dref = torch.zeros_like(ref)
    a = _C.BboxArgs(None, _ptr(_f32(ref).contiguous()), _ptr(_f32(out).contiguous()), rows, Q, code,
                    (C.c_float * 6)(*[float(v) for v in pc_range]), float(time_div), float(eps))
    _C.check(L.petr_bbox_epilogue_bwd(C.byref(a), _ptr(_f32(dout).contiguous()), _ptr(dreg), _ptr(dref), _stream()),
             'petr_bbox_epilogue_bwd')
    return dreg, dref
