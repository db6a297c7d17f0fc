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
    red = BucketedGradAllReduce(head, merge=2, average=True)
    assert [b[0] for b in red.buckets] == [1, 3, 4]
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
