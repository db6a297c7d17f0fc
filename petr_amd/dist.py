"""Data-parallel gradient exchange for the PETRHead path.

The path shards by sample (one process per GPU, B samples each, no cross-sample op in the head), so
the only exchange is the SUM/AVG all-reduce of the gradient vector.  Because the head keeps ONE flat
gradient buffer whose layout is ordered by backward completion time, the exchange is a few large
contiguous RCCL all-reduces (``backend='nccl'`` is RCCL on ROCm; xGMI inside a node) issued from the
backward's stage hook on a side stream while later stages are still computing.

Initialisation order (measured on MI355X / ROCm 7.2): create the head and run ``head._context()``
(its side streams) BEFORE ``torch.distributed.init_process_group('nccl')``; the other order costs ~15 % of the step.

Replaces: mmcv ``MMDistributedDataParallel`` built by mmdet3d ``train_model`` (reference entry
tools/train.py:246), i.e. torch DDP's per-parameter-bucket NCCL all-reduce.
"""
import torch
import torch.distributed as dist


class BucketedGradAllReduce:
    """Overlaps the all-reduce of finished gradient ranges with the rest of the backward.

    ``merge`` consecutive backward stages form one bucket (xGMI is point-to-point: fewer, larger
    collectives amortise the per-collective latency better than many small ones)."""

    def __init__(self, head, group=None, merge=2, average=True, force=False, broadcast_parameters=True):
        self.head = head
        self.group = group
        self.average = average
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force          # issue the collectives even for a single rank (tests the stream/event path)
        self._sync = True           # False inside no_sync(): gradient accumulation, no exchange
        if self.world > 1 and broadcast_parameters:
            # every rank starts from rank 0's parameters, as DDP (the thing this replaces) does at construction: ONE
            # broadcast of the flat buffer.  Without it identical weights would rest on every rank seeding the same
            # RNG, and a checkpoint loaded on rank 0 only would silently diverge.
            dist.broadcast(head.flat_parameters(), src=dist.get_global_rank(group, 0) if group is not None else 0,
                           group=group)
        stages = head.gradient_buckets()
        self.n_stages = len(stages)
        self.buckets = []           # (last_stage, begin, end)
        i = 0
        while i < len(stages):
            j = min(i + merge, len(stages))
            self.buckets.append((j - 1, stages[i][0], stages[j - 1][1]))
            i = j
        self._by_stage = {b[0]: b for b in self.buckets}
        self._works = []
        self._cuda = head.flat_parameters().is_cuda
        # the exchange rides on the head's second side stream (the one that carries the least weight-gradient work): a stream of
        # its own became a FIFTH stream, was multiplexed onto the compute stream's hardware queue and stalled the backward at
        # every bucket boundary until the stage's side work and the collective were done (kernel trace, c5: 180 us per boundary)
        self._comm = None
        if self._cuda:
            self._comm = head.side_stream(1) if hasattr(head, 'side_stream') else None
            if self._comm is None:
                self._comm = torch.cuda.Stream()
        head._stage_hook = self._on_stage
        head._stage_hook_stages = sorted(self._by_stage)     # the backward is cut only where a bucket ends

    def no_sync(self):
        """Context manager for gradient accumulation (DDP's ``no_sync``): backward passes inside it add into the flat
        gradient buffer without any exchange; the first backward OUTSIDE it all-reduces the accumulated sum.  (Calling
        the exchange on every micro-step instead would average gradients that were already averaged.)"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = prev
        return ctx()

    def _on_stage(self, stage):
        b = self._by_stage.get(stage)
        if b is None or not self._sync or (self.world == 1 and not self.force):
            return
        flat = self.head._flat_grad[b[1]:b[2]]
        if self._cuda:
            # the exchange stream (not the compute stream) waits for the stage: its kernels on the compute stream
            # and its weight-gradient contractions on the head's side streams
            self.head.join_streams_into(self._comm)
            with torch.cuda.stream(self._comm):
                self._issue(flat)
        else:
            self._issue(flat)

    def _issue(self, flat):
        if self.average and self._cuda and dist.get_backend(self.group) == 'nccl':
            self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:   # gloo has no AVG
            self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                                flat if self.average else None))

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (call after backward)."""
        for work, flat in self._works:
            work.wait()
            if flat is not None:
                flat.div_(self.world)
        self._works.clear()
        if self._cuda:
            torch.cuda.current_stream().wait_stream(self._comm)

    def detach(self):
        self.head._stage_hook = None
        self.head._stage_hook_stages = None


def all_reduce_flat(flat_grad, world, group=None, average=True):
    """Single-shot (non-overlapped) exchange of a whole flat gradient buffer; used by the gloo CPU tests."""
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat_grad.div_(world)
    return flat_grad
