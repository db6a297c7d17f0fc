"""CPU, world_size 2, gloo: the bucketed gradient all-reduce (petr_amd.dist) sums/averages exactly
the flat ranges the backward stages finalise, and equals a single-process reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeHead:
    """stands in for PETRHead on the CPU: same three members the reducer uses."""

    def __init__(self, n, stages, seed):
        g = torch.Generator().manual_seed(seed)
        self._flat = torch.zeros(n)
        self._flat_grad = torch.randn(n, generator=g)
        self._stages = stages
        self._stage_hook = None

    def flat_parameters(self):
        return self._flat

    def gradient_buckets(self):
        return self._stages


def _worker(rank, world, port, n, stages, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from petr_amd.dist import BucketedGradAllReduce
    head = _FakeHead(n, stages, seed=100 + rank)
    head._flat += float(rank + 1)          # ranks start from DIFFERENT parameters ...
    red = BucketedGradAllReduce(head, merge=2, average=True)
    assert torch.equal(head._flat, torch.full((n,), 1.0))      # ... and leave the constructor with rank 0's (broadcast)
    assert [b[0] for b in red.buckets] == [1, 3, 4]
    before = head._flat_grad.clone()
    with red.no_sync():                    # gradient accumulation: stage hooks inside no_sync() exchange nothing
        for s in range(len(stages)):
            head._stage_hook(s)
        red.finish()
    assert torch.equal(head._flat_grad, before)
    tail = head._flat_grad[stages[-1][1]:].clone()
    for s in range(len(stages)):          # what PETRHead._launch_backward does after each stage
        head._stage_hook(s)
    red.finish()
    assert torch.equal(head._flat_grad[stages[-1][1]:], tail)      # beyond the last bucket: untouched
    q.put((rank, head._flat_grad.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    n = 1000
    stages = [(0, 100), (100, 400), (400, 650), (650, 900), (900, 990)]
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, stages, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g0 = _FakeHead(n, stages, 100)._flat_grad
    g1 = _FakeHead(n, stages, 101)._flat_grad
    want = (g0 + g1) / 2
    assert torch.allclose(got[0][:990], want[:990], rtol=0, atol=1e-7)
    assert torch.equal(got[0][:990], got[1][:990])
    assert torch.equal(got[0][990:], g0[990:]) and torch.equal(got[1][990:], g1[990:])


def _avg_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from petr_amd.losses import LossConfig, synced_avg_factors
    num_pos = [40, 7][rank]                 # this rank's matched boxes; 900 queries x 1 sample each
    out = []
    for sync in (False, True):
        cfg = LossConfig(bg_cls_weight=0.1, sync_cls_avg_factor=sync)
        out.append(synced_avg_factors(num_pos, 900, cfg, torch.device('cpu')).tolist())
    q.put((rank, out))
    dist.destroy_process_group()


def test_loss_normalisers_reduce_mean_world2():
    """mmdet reduce_mean semantics of the loss normalisers (petr_head.py:620-622, 628-631) over two gloo ranks:
    num_total_pos is always the mean over ranks, cls_avg_factor only with sync_cls_avg_factor; one process: None."""
    from petr_amd.losses import LossConfig, synced_avg_factors
    assert synced_avg_factors(5, 900, LossConfig(), torch.device('cpu')) is None        # no process group
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_avg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    local = [40 + (900 - 40) * 0.1, 7 + (900 - 7) * 0.1]
    for rank in range(2):
        (cls_nosync, pos_nosync), (cls_sync, pos_sync) = res[rank]
        assert abs(pos_nosync - 23.5) < 1e-5 and abs(pos_sync - 23.5) < 1e-5
        assert abs(cls_nosync - local[rank]) < 1e-4
        assert abs(cls_sync - 0.5 * (local[0] + local[1])) < 1e-4
