"""Where a query tile of petr_mha_bwd (fp32) spends its cycles: s_memtime stamps of the diagnostic build
(make -C petr_amd/csrc EXTRA=-DPETR_DIAG_BWD_STAMPS OUT=../lib/libpetr_hip_stamps.so OBJDIR=../lib/obj_stamps), wave 0 of
every workgroup, summed per phase in the front of the (otherwise unused) workspace.
   PETR_HIP_LIB=petr_amd/lib/libpetr_hip_stamps.so python scripts/bwd32_stamps.py [drop_p]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petr_amd import ops, _C
from petr_amd.ops import _ptr, _bhsd, _stream
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
Lb = _C.lib()
g = torch.Generator().manual_seed(0)
SK = os.environ.get('PETR_MHA_BWD_SK', '1') != '0'
NAMES = (('visit prologue', 'S/dP mfma', 'exp+dV/dK+ds->lds', 'barrier', 'dQ operands+stage+gload', 'dQ mfma+atomics', '-', 'dK/dV flush') if SK else
         ('prologue', 'S/dP mfma', 'exp+dV/dK+ds->lds', 'barrier1', 'stage+dQ mfma+red', 'barrier2', 'sum+atomics', 'epilogue'))
for name, Q, L in (('self', 900, 900), ('c5', 900, 4224), ('p4_1600', 900, 24000)):
    mk = lambda n: torch.randn(1, n, 256, generator=g).cuda().view(1, n, 8, 32).permute(0, 2, 1, 3)
    q, do, k, v = mk(Q), mk(Q), mk(L), mk(L)
    drop = (1234, 3, p) if p > 0 else None
    o, lse = ops.mha_fwd(q, k, v, drop=drop)
    o = o.permute(0, 2, 1, 3).contiguous().view(1, Q, 8, 32).permute(0, 2, 1, 3)
    dq = torch.zeros(1, Q, 256, device='cuda').view(1, Q, 8, 32).permute(0, 2, 1, 3)
    dk = torch.zeros(1, L, 256, device='cuda').view(1, L, 8, 32).permute(0, 2, 1, 3)
    dv = torch.zeros(1, L, 256, device='cuda').view(1, L, 8, 32).permute(0, 2, 1, 3)
    nbytes = Lb.petr_mha_bwd_workspace_bytes(1, 8, Q, L)
    ws = torch.zeros(max(nbytes // 8, 16), dtype=torch.int64, device='cuda')
    a = _C.MhaBwdArgs(_ptr(q), *_bhsd(q), _ptr(k), *_bhsd(k), _ptr(v), *_bhsd(v), _ptr(o), *_bhsd(o), _ptr(do), *_bhsd(do),
                      _ptr(lse), None, _ptr(dq), *_bhsd(dq), _ptr(dk), *_bhsd(dk), _ptr(dv), *_bhsd(dv), 1, 8, Q, L,
                      32 ** -0.5, _ptr(ws), nbytes, _C.dropout(drop))
    for _ in range(20):
        _C.check(Lb.petr_mha_bwd(C.byref(a), _stream()), 'petr_mha_bwd')
    torch.cuda.synchronize()
    ws.zero_()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _C.check(Lb.petr_mha_bwd(C.byref(a), _stream()), 'petr_mha_bwd')
    e1.record()
    torch.cuda.synchronize()
    acc = ws[:10].cpu().tolist()
    tiles, wgs = acc[8] / n, acc[9] / n
    print(f'{name} Q={Q} L={L} drop={p}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/launch, {wgs:.0f} workgroups / pair visits, {tiles:.0f} tile visits')
    tot = sum(acc[:8]) / n
    for i, nm in enumerate(NAMES):
        per = acc[i] / n / (wgs if i in (0, 7) else tiles)
        print(f'   {nm:22s} {per:9.0f} cycles per {"workgroup" if i in (0, 7) else "tile"}   ({acc[i] / n / tot * 100:5.1f} % of wave-0 lifetime)')
    print(f'   mean wave-0 lifetime {tot / wgs:.0f} cycles; MFMA floor per tile (80 x 64) = 5120 cycles per wave', flush=True)
