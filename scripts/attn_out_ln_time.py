"""petr_attn_out_ln alone against the launches it replaces (900 rows, training mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
B, Q, H, C = 1, 900, 8, 256
for L in (900, 4224):
    q, k, v = (torch.randn(B, n, C, generator=g).cuda().view(B, n, H, 32).permute(0, 2, 1, 3) for n in (Q, L, L))
    w, bias, res = torch.randn(C, C, generator=g).cuda() * 0.06, torch.randn(C, generator=g).cuda(), torch.randn(Q, C, generator=g).cuda()
    gamma, beta, pos = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda(), torch.randn(Q, C, generator=g).cuda()
    parts, ns = ops.mha_fwd(q, k, v, drop=(1, 2, 0.1), defer_merge=True)
    a = torch.empty(Q, C, device='cuda')
    o, _ = ops.mha_fwd(q, k, v, drop=(1, 2, 0.1))
    ao = o.permute(0, 2, 1, 3).reshape(Q, C).contiguous()
    fused = t(lambda: ops.attn_out_ln(a, w, bias, res, gamma, beta, partials=parts, n_split=ns, BHQ=(B, H, Q), attn_scale=1.1, drop=(1, 3, 0.1), add2=pos, add2_rows=Q))
    lin = t(lambda: ops.linear(ao, w, bias))
    y = ops.linear(ao, w, bias)
    ln = t(lambda: ops.layernorm(y, gamma, beta, residual=res, save_stats=True, drop=(1, 3, 0.1)))
    print(f'L={L} (n_split {ns}): fused {fused:.1f} us (wrapper allocates 6 tensors) | linear {lin:.1f} us + layernorm {ln:.1f} us (+ the merge kernel ~6 us)')
