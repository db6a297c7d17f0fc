"""diagnostic: does a HIP graph shorten the inference forward?  Capture petr_head_fwd (eval mode: no per-step kernel
arguments change) with torch.cuda.CUDAGraph and compare replay time with the plain enqueue."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import petr_amd
import bench
n, h, w, ph, pw, _ = bench.WORKLOADS['c5']
head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=900)); head.init_weights(); head = head.cuda().eval()
metas = bench.synthetic_metas(1, n, (ph, pw), seed=0)
feats = torch.randn(1, n, 256, h, w, generator=torch.Generator().manual_seed(0)).cuda()
def fwd():
    with torch.no_grad():
        return head([feats], metas)
for _ in range(10): fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): fwd()
torch.cuda.synchronize()
print(f'plain   : {(time.perf_counter()-t0)*10:.3f} ms per forward')
# capture the raw launch (static run object: same workspace, same outputs)
run = head._prepare(feats, metas); run.time_div = 0.0
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): head._launch_forward(run, feats)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        head._launch_forward(run, feats)
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
    print(f'graph   : {(time.perf_counter()-t0)*10:.3f} ms per forward')
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:400])
