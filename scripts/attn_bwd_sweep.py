"""diagnostic: attention backward time vs q_splits (env override), c5 cross / self shapes."""
import sys, os, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.getcwd())
    import torch
    from petr_amd import ops
    L = int(sys.argv[1])
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, 8, 900, 32, generator=g).cuda(); k = torch.randn(1, 8, L, 32, generator=g).cuda(); v = torch.randn(1, 8, L, 32, generator=g).cuda()
    do = torch.randn(1, 8, 900, 32, generator=g).cuda()
    o, lse = ops.mha_fwd(q, k, v)
    for _ in range(3): ops.mha_bwd(q, k, v, o, do, lse)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.mha_bwd(q, k, v, o, do, lse)
    e1.record(); torch.cuda.synchronize()
    print(f'L={L} q_splits={os.environ.get("PETR_MHA_BWD_QSPLITS","auto")}: {e0.elapsed_time(e1)/20*1e3:.1f} us (incl. 3 zero-fills + delta kernel + python)', flush=True)
else:
    for L in (4224, 900):
        for qs in ('auto', '1', '2', '3', '4', '5', '6', '8', '10', '15'):
            env = dict(os.environ)
            if qs != 'auto': env['PETR_MHA_BWD_QSPLITS'] = qs
            subprocess.run([sys.executable, __file__, str(L)], env=env)
