"""diagnostic: is the training step host-bound?  time to ENQUEUE n steps vs time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
if len(sys.argv) > 2 and sys.argv[2] == 'early':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import petr_amd
import bench
n, h, w, ph, pw, _ = bench.WORKLOADS['c5']
head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=900)); head.init_weights(); head = head.cuda().train()
metas = bench.synthetic_metas(1, n, (ph, pw), seed=0)
g = torch.Generator().manual_seed(0)
feats = torch.randn(1, n, 256, h, w, generator=g).cuda().requires_grad_(True)
gc, gb = torch.randn(6, 1, 900, 10, generator=g).cuda(), torch.randn(6, 1, 900, 10, generator=g).cuda()
reducer = None
if len(sys.argv) > 1 and sys.argv[1] == 'reducer':
    import torch.distributed as dist
    from petr_amd.dist import BucketedGradAllReduce
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    reducer = BucketedGradAllReduce(head, merge=int(os.environ.get('MERGE', '2')), force=True)
def step():
    head.zero_grad_flat(); feats.grad = None
    out = head([feats], metas)
    torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [gc, gb])
    if reducer is not None: reducer.finish()
for _ in range(10): step()
torch.cuda.synchronize()
for N in (1, 5, 50):
    t0 = time.perf_counter()
    for _ in range(N): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{N} steps: enqueue {(t1-t0)/N*1e3:.3f} ms/step, until done {(t2-t0)/N*1e3:.3f} ms/step')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
