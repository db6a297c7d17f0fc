#!/usr/bin/env python
"""Benchmark of the PETRHead hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch of synthetic input on every rank:
zero the flat gradient buffer, PETRHead forward, backward from a fixed seeded upstream gradient, and
(N>1) the bucketed RCCL all-reduce of the gradients overlapped with the backward.  Inputs (features,
upstream gradients) are resident in HBM before the timed region; ``img_metas`` stay host-side as in the
reference (float64 ``lidar2img`` inverted on the host every forward, petr_head.py:308-315).

One JSON line on rank 0: metric/value/unit as BASELINE.json (samples/s fwd+bwd, whole job), plus
  roofline      time-dominant kernel = cross-attention BACKWARD (mha_bwd_sk_kernel); roofline['forward'] = the cross-attention
                forward (mha_fwd_kernel); each: algorithmic FLOPs per launch
                (4*Q*L*C) / its mean launch duration measured with HIP events on the launch
                stream inside real steps, against the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch path) timed on this box's host cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N views, H, W, pad_h, pad_w, description)
    'c5': (6, 16, 44, 512, 1408, 'petr_r50dcn_gridmask_c5 head-only: 6x(256x16x44) features'),
    'p4_1408': (6, 32, 88, 512, 1408, 'petr_r50dcn_gridmask_p4 1408x512: 6x(256x32x88) features'),
    'p4_1600': (6, 40, 100, 640, 1600, 'petr_vovnet_p4 1600x640: 6x(256x40x100) features'),
    # PETRv2Head (fpe + RegLayer + with_time), two frames = 12 views
    'v2_800': (12, 20, 50, 320, 800, 'petrv2_vovnet_p4 800x320, two frames: 12x(256x20x50) features, PETRv2Head'),
}
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
HBM_PEAK_GBS = 8000.0


def log(msg):
    print(f'[bench] {msg}', file=sys.stderr, flush=True)


def synthetic_metas(batch, n_views, pad_hw, seed):
    """nuScenes-like lidar2img; kept in bench.py so the product path never imports oracle/."""
    import numpy as np
    metas = []
    for b in range(batch):
        rng = np.random.RandomState(seed + b)
        r = pad_hw[1] / 1600.0
        K = np.eye(4)
        K[0, 0] = K[1, 1] = 1266.0 * r
        K[0, 2], K[1, 2] = 816.0 * r, 491.0 * r
        yaws = np.deg2rad([0.0, -55.0, 55.0, 180.0, 110.0, -110.0])
        swap = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)
        mats = []
        for i in range(n_views):
            c, s = np.cos(yaws[i % 6]), np.sin(yaws[i % 6])
            Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)
            R = swap @ Rz.T
            t = rng.uniform(-1.5, 1.5, size=3)
            rt = np.eye(4)
            rt[:3, :3] = R
            rt[:3, 3] = -R @ t
            mats.append(K @ rt)
        half = n_views // 2
        metas.append({'pad_shape': [(pad_hw[0], pad_hw[1], 3)] * n_views, 'img_shape': [(pad_hw[0], pad_hw[1], 3)] * n_views,
                      'lidar2img': mats,
                      # PETRv2 (with_time): current frame at t = 0, the previous sweep ~0.5 s earlier
                      'timestamp': [0.0] * half + [0.5 + 0.01 * i for i in range(n_views - half)]})
    return metas


def synthetic_gt(batch, n_gt, seed=0, num_classes=10, device='cpu'):
    """Ground truth of the shape nuScenes gives: per sample [G, 9] gravity-centre boxes (centre, dims, yaw, vx, vy)
    inside the point-cloud range, and [G] labels."""
    g = torch.Generator().manual_seed(seed)
    boxes, labels = [], []
    for _ in range(batch):
        c = (torch.rand(n_gt, 3, generator=g) - 0.5) * torch.tensor([90.0, 90.0, 6.0])
        dims = torch.rand(n_gt, 3, generator=g) * torch.tensor([3.0, 8.0, 2.5]) + 0.4
        yaw = (torch.rand(n_gt, 1, generator=g) - 0.5) * 6.283
        vel = torch.randn(n_gt, 2, generator=g)
        boxes.append(torch.cat([c, dims, yaw, vel], 1).to(device))
        labels.append(torch.randint(0, num_classes, (n_gt,), generator=g).to(device))
    return boxes, labels


def cpu_baseline(workload, batch, num_query, budget_s=25.0, train=True):
    """The oracle (CPU restatement of the reference's PyTorch path) on the host cores: fwd+bwd (training mode:
    torch's own dropouts active, like the GPU leg) and fwd-only (eval mode)."""
    from oracle import petr_oracle as O
    n, h, w, ph, pw, _ = WORKLOADS[workload]
    # the box's CPU share, not the host's core count (oversubscribed OpenMP threads crawl)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(cores, 16)))
    v2 = workload.startswith('v2')
    kw = dict(v2=True, with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10) if v2 else {}
    head = O.seeded_head(0, None, num_query=num_query, **kw)
    metas = O.synthetic_img_metas(batch, n, (ph, pw), seed=0, with_time=v2)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(batch, n, 256, h, w, generator=g)
    g_cls, g_box = torch.randn(6, batch, num_query, 10, generator=g), torch.randn(6, batch, num_query, 10, generator=g)

    def step():
        head.zero_grad(set_to_none=True)
        out = head([feats], metas)
        torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls, g_box])

    head.train(train)
    step()   # warm-up
    log(f'cpu baseline: warm-up done, {torch.get_num_threads()} threads')
    t0 = time.perf_counter()
    n_steps = 0
    while n_steps < 3 or (time.perf_counter() - t0 < budget_s * 0.6 and n_steps < 20):
        step()
        n_steps += 1
    dt = (time.perf_counter() - t0) / n_steps
    head.eval()
    with torch.no_grad():
        head([feats], metas)
        t1 = time.perf_counter()
        n_f = 0
        while n_f < 3 or (time.perf_counter() - t1 < budget_s * 0.3 and n_f < 20):
            head([feats], metas)
            n_f += 1
        dtf = (time.perf_counter() - t1) / n_f
    return {'value': round(batch / dt, 4), 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n_steps} fwd+bwd steps in {"training" if train else "eval"} mode (and {n_f} eval-mode forwards) of '
                      f'the same workload, batch {batch}, fp32',
            'fwd_value': round(batch / dtf, 4), 'ms_per_step': round(dt * 1e3, 2), 'fwd_ms': round(dtf * 1e3, 2)}


# algorithmic work per sample, GFLOP of one forward
HEAD_FWD_GFLOP = {'c5': 63.7, 'p4_1408': 183.6, 'p4_1600': 250.9, 'v2_800': 137.3}
ATTN_FWD_GFLOP = {'c5': 23.36, 'p4_1408': 93.43, 'p4_1600': 132.71, 'v2_800': 66.36}


def step_gflop(workload, batch):
    """forward + backward: 3x the forward contractions, and the attention backward recomputes P (5 products for the
    forward's 2): + 0.5x the attention core."""
    return batch * (3.0 * HEAD_FWD_GFLOP[workload] + 0.5 * ATTN_FWD_GFLOP[workload])


PMC_TRAFFIC_FILES = ('r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')     # newest first
PROF_NAMES = {1: 'mha_fwd_self', 17: 'mha_fwd_cross', 2: 'mha_bwd_self', 18: 'mha_bwd_cross', 4: 'coords3d'}


def profile_kernels(step_fn, n_steps):
    """mean duration of the tagged dispatches (HIP events attached to the dispatch on its launch stream) inside real steps"""
    from petr_amd import _C
    L = _C.lib()
    cap = n_steps * 64
    _C.check(L.petr_prof_begin(cap), 'petr_prof_begin')
    for _ in range(n_steps):
        step_fn()
    torch.cuda.synchronize()
    ms = (C.c_float * cap)()
    tags = (C.c_int * cap)()
    cnt = C.c_int()
    _C.check(L.petr_prof_end(ms, tags, cap, C.byref(cnt)), 'petr_prof_end')
    acc = {}
    for i in range(cnt.value):
        acc.setdefault(tags[i], []).append(ms[i])
    return {PROF_NAMES.get(t, str(t)): {'launches': len(v), 'mean_us': round(sum(v) / len(v) * 1e3, 2)} for t, v in acc.items()}


def build_workload(workload, batch, num_query, dev, rank=0, need_grad=True):
    import petr_amd
    n, h, w, ph, pw, _ = WORKLOADS[workload]
    torch.manual_seed(0)                     # identical weights on every rank (reference init rules)
    v2 = workload.startswith('v2')
    head = petr_amd.build_head((petr_amd.petrv2_head_cfg if v2 else petr_amd.petr_head_cfg)(num_query=num_query))
    head.init_weights()
    head = head.to(dev)
    metas = synthetic_metas(batch, n, (ph, pw), seed=rank * 1000)
    g = torch.Generator().manual_seed(1234 + rank)           # rank-offset seed: every rank has its own samples
    feats = torch.randn(batch, n, 256, h, w, generator=g).to(dev).requires_grad_(need_grad)
    g_cls = torch.randn(6, batch, num_query, 10, generator=g).to(dev)
    g_box = torch.randn(6, batch, num_query, 10, generator=g).to(dev)
    return head, metas, feats, g_cls, g_box


def extra_workload(workload, dev, num_query, steps, warmup):
    """One BASELINE single-GPU workload (B = 1 per GPU, the reference's samples_per_gpu) in fp32 and in bf16: training
    step time, cross-attention kernel times from dispatch events, attention roofline fraction against the dtype's peak."""
    B = 1
    n, h, w, _, _, desc = WORKLOADS[workload]
    Ltok = n * h * w
    head, metas, feats, g_cls, g_box = build_workload(workload, B, num_query, dev)
    head.train()
    torch.manual_seed(2000)

    def step():
        head.zero_grad_flat()
        feats.grad = None
        out = head([feats], metas)
        torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls, g_box])

    res = {'desc': desc, 'tokens': Ltok}
    for mode in ('fp32', 'bf16'):
        head.attn_dtype = mode
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        head.eval()
        with torch.no_grad():
            for _ in range(2):
                head([feats], metas)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                head([feats], metas)
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t1) / steps * 1e3
        head.train()
        k = profile_kernels(step, min(steps, 10))
        peak = BF16_MFMA_PEAK_TFLOPS if mode == 'bf16' else FP32_MFMA_PEAK_TFLOPS
        leg = {'ms_per_step': round(ms, 4), 'samples_per_s': round(B / (ms * 1e-3), 2), 'fwd_ms': round(fwd_ms, 4),
               'step_tflops': round(step_gflop(workload, B) / ms, 1),
               'step_frac_of_peak': round(step_gflop(workload, B) / ms / peak, 4)}
        if 'mha_fwd_cross' in k:
            us = k['mha_fwd_cross']['mean_us']
            ach = 4.0 * B * num_query * Ltok * 256 / (us * 1e-6) / 1e12
            leg['roofline'] = {'kernel': 'cross-attention forward (training variant: dropout inside)', 'bound': 'mfma',
                               'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                               'mean_launch_us': us}
        if 'mha_bwd_cross' in k:
            us = k['mha_bwd_cross']['mean_us']
            ach = 10.0 * B * num_query * Ltok * 256 / (us * 1e-6) / 1e12
            leg['mha_bwd_cross'] = {'mean_launch_us': us, 'achieved': round(ach, 2), 'frac': round(ach / peak, 4)}
        res[mode] = leg
    head.attn_dtype = 'fp32'
    head.release()             # its side streams must not outlive it (the next workload's streams would share their queues)
    del head, feats, g_cls, g_box
    torch.cuda.empty_cache()
    return res


def neck_fold_leg(dev, num_query, steps):
    """SURVEY 8(f) rank 4, inference: backbone maps -> CPFPN -> PETRHead.forward at the maps of BASELINE configs[3]
    (petr_vovnet_gridmask_p4_1600x640.py:38-42: 768 / 1024 channels at 40x100 / 20x50, six views), once as the reference
    composes it (3x3 output conv -> NCHW map -> input_proj) and once with input_proj folded into the 3x3 conv
    (CPFPN.forward_folded -> PETRHead.forward_projected)."""
    import petr_amd
    torch.manual_seed(0)
    neck = petr_amd.build_neck(dict(type='CPFPN', in_channels=[768, 1024], out_channels=256, num_outs=2))
    neck.init_weights()
    neck = neck.to(dev).eval()
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=num_query)).to(dev).eval()
    metas = synthetic_metas(1, 6, (640, 1600), seed=0)
    g = torch.Generator().manual_seed(77)
    xs = [torch.randn(6, 768, 40, 100, generator=g).to(dev), torch.randn(6, 1024, 20, 50, generator=g).to(dev)]

    def plain():
        head(petr_amd.glue.reshape_backbone_feats(list(neck(xs)), 1), metas)

    def folded():
        head.forward_projected(neck.forward_folded(xs, head).view(1, 6, 40, 100, 256), metas)

    res = {'what': 'inference forward, backbone maps (6 x 768 x 40 x 100, 6 x 1024 x 20 x 50) -> CPFPN -> PETRHead, ms per sample'}
    with torch.no_grad():
        for dt in ('fp32', 'bf16'):
            head.attn_dtype = dt
            for name, fn in (('neck_then_input_proj', plain), ('input_proj_folded', folded)):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
                res[f'{name}_{dt}_ms'] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    head.release()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='c5', choices=sorted(WORKLOADS))
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help='arithmetic of the timed step: fp32 (BASELINE configs[1], the headline) or bf16 (configs 2-4: '
                         'token-sized contractions + cross-attention on bf16 MFMA, fp32 accumulate)')
    ap.add_argument('--timed-only', action='store_true',
                    help='profiling aid: stop after the timed region (no forward-only / kernel-event / loss / bf16 / CPU legs)')
    ap.add_argument('--no-extra-workloads', action='store_true',
                    help='skip the other BASELINE single-GPU workloads (p4_1408, p4_1600, v2_800; fp32 and bf16) that '
                         'the default N=1 run times after the headline')
    ap.add_argument('--batch', type=int, default=1, help='samples per GPU (the reference configs use 1)')
    ap.add_argument('--queries', type=int, default=900)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--fwd-only', action='store_true', help='time the forward only (diagnostic; not the metric)')
    ap.add_argument('--force-reducer', action='store_true',
                    help='diagnostic: run the bucketed RCCL gradient exchange even on one rank (cost of the DP path)')
    ap.add_argument('--eval-mode', action='store_true',
                    help='time fwd+bwd with the dropouts off (diagnostic; the metric is the training step, dropout 0.1)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks ourselves (one process per GPU, the
        # reference's tools/dist_train.sh:7-9 does the same with torch.distributed.launch).  This parent has not touched
        # the GPU; it only waits for the child and passes its exit code on (never exec after a GPU init on this pool).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    # rehearsal knobs (not used by the driver): PETR_BENCH_DEVICE pins every rank to one GPU and PETR_BENCH_BACKEND=gloo
    # replaces RCCL, so that the N > 1 control flow (collectives, barriers, teardown) can be run on a one-GPU box
    dev_index = int(os.environ.get('PETR_BENCH_DEVICE', local_rank))
    backend = os.environ.get('PETR_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    import torch.distributed as dist
    import petr_amd
    from petr_amd import _C
    from petr_amd.dist import BucketedGradAllReduce

    n, h, w, ph, pw, desc = WORKLOADS[args.workload]
    B, Q = args.batch, args.queries
    head, metas, feats, g_cls, g_box = build_workload(args.workload, B, Q, dev, rank, need_grad=not args.fwd_only)
    # the metric is the TRAINING step: train() = the reference's dropouts (p = 0.1, six per decoder layer) are active
    head.train(not (args.eval_mode or args.fwd_only))
    head.attn_dtype = args.dtype
    torch.manual_seed(1000 + rank)           # the per-forward dropout seeds are drawn from this stream: one per rank
    reducer = None
    if world > 1 or args.force_reducer:
        # ORDER MATTERS (measured, MI355X / ROCm 7.2): the head's side streams must exist before the RCCL communicator
        # is created.  With the process group initialised first the same step ran 6.2 ms instead of 5.4 ms on one rank
        # (cross-stream event waits got slower), so: build the head, create its stream context, then init RCCL.
        head._context()
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        reducer = BucketedGradAllReduce(head, merge=2, force=args.force_reducer)

    def step():
        if args.fwd_only:
            with torch.no_grad():
                head([feats], metas)
            return
        head.zero_grad_flat()
        feats.grad = None
        out = head([feats], metas)
        torch.autograd.backward([out['all_cls_scores'], out['all_bbox_preds']], [g_cls, g_box])
        if reducer is not None:
            reducer.finish()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if rank == 0:
        log('warm-up done')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if reducer is not None:
        # everything below (forward-only, kernel-timing and loss legs) runs per rank without talking to the others,
        # some of it on rank 0 only: no collective may be issued from here on
        reducer.detach()
        reducer = None
    if rank == 0:
        log(f'timed region done: {elapsed / args.steps * 1e3:.3f} ms/step')
    if args.timed_only:
        if rank == 0:
            print(json.dumps({'metric': 'samples/sec PETRHead fwd+bwd', 'value': round(world * B * args.steps / elapsed, 3),
                              'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'dtype': args.dtype,
                              'config': {'workload': args.workload, 'timed_only': True}}), flush=True)
        if world > 1 or args.force_reducer:
            dist.destroy_process_group()
        return
    # ---- a sustained region of >= 1 s (the driver's K steps at ~5 ms each are a 0.1 s region): same step, more of them ----
    sustained = None
    if world == 1 and not args.fwd_only:
        n_sus = max(args.steps, int(1000.0 / max(elapsed / args.steps * 1e3, 1e-3)) + 1)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        sus = time.perf_counter() - ts
        sustained = {'steps': n_sus, 'ms_per_step': round(sus / n_sus * 1e3, 4), 'samples_per_s': round(B * n_sus / sus, 2)}
    # ---- forward-only rate (for the ">= 10x the host-CPU forward" target), same steady state ----
    fwd_ms = None
    was_training = head.training
    if not args.fwd_only:
        head.eval()                      # inference forward
        with torch.no_grad():
            for _ in range(3):
                head([feats], metas)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                head([feats], metas)
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t1) / args.steps * 1e3
        head.train(was_training)

    # ---- roofline leg: HIP events attached to the tagged dispatches inside real steps (rank 0) ----
    roofline = None
    bf16_leg = None
    kernels = {}
    peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == 'bf16' else FP32_MFMA_PEAK_TFLOPS
    Ltok = n * h * w
    if rank == 0:
        prof_steps = min(args.steps, 20)
        kernels = profile_kernels(step, prof_steps)
        if 'mha_fwd_cross' in kernels:
            us = kernels['mha_fwd_cross']['mean_us']
            flops = 4.0 * B * Q * Ltok * 256            # QK^T + PV of one layer
            ach = flops / (us * 1e-6) / 1e12
            # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (scripts/pmc_traffic.sh,
            # method inside the file); only the workload / dtype / batch that was profiled gets a number, everything else null
            traffic, traffic_source = None, None
            for fn in PMC_TRAFFIC_FILES:
                try:
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                    ent = (pmc.get(f'{args.workload}_{args.dtype}') or pmc[args.workload])['mha_fwd_cross']
                    if B == 1 and ent.get('dtype', 'fp32') == args.dtype:
                        traffic, traffic_source = ent['traffic_bytes'], f'profiles/{fn} (separate rocprofv3 --pmc passes)'
                        break
                except (OSError, KeyError, ValueError, TypeError):
                    continue
            roofline = {'kernel': ('mha_fwd_bf16_kernel' if args.dtype == 'bf16' else 'mha_fwd_kernel') +
                                  ' (cross-attention, one decoder layer)', 'bound': 'mfma',
                        'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                        'flops_per_launch': flops, 'mean_launch_us': us}
        if roofline is not None and was_training:
            # the same kernel without the dropout of the probabilities (eval-mode / inference forwards)
            head.eval()

            def fwd_only():
                with torch.no_grad():
                    head([feats], metas)
            ke = profile_kernels(fwd_only, prof_steps)
            head.train(was_training)
            if 'mha_fwd_cross' in ke:
                us_e = ke['mha_fwd_cross']['mean_us']
                ach_e = roofline['flops_per_launch'] / (us_e * 1e-6) / 1e12
                roofline['timed_variant'] = 'training mode: dropout (p = 0.1) of the attention probabilities inside the kernel'
                roofline['inference_variant'] = {'mean_launch_us': round(us_e, 2), 'achieved': round(ach_e, 2),
                                                 'frac': round(ach_e / peak, 4)}
        # ---- the same workload in bf16 (BASELINE configs 2-4 are bf16; configs[1], the headline, is fp32): training step
        #      AND inference forward, not part of `value` ----
        if roofline is not None and world == 1 and not args.fwd_only and args.dtype == 'fp32':
            head.attn_dtype = 'bf16'
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            bt_ms = (time.perf_counter() - t3) / args.steps * 1e3
            kb = profile_kernels(step, prof_steps)
            head.eval()
            with torch.no_grad():
                for _ in range(5):
                    head([feats], metas)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(args.steps):
                    head([feats], metas)
                torch.cuda.synchronize()
                bf_ms = (time.perf_counter() - t3) / args.steps * 1e3
            head.attn_dtype = 'fp32'
            head.train(was_training)
            bf16_leg = {'what': 'same workload with attn_dtype=bf16: token-sized contractions (forward, input- and weight-gradients) '
                                'on bf16 MFMA with fp32 accumulation, K/V stored as bf16, bf16 cross-attention forward and backward, '
                                'fp32 softmax / LayerNorm / parameters / gradients; training step (dropout 0.1) and eval forward',
                        'ms_per_step': round(bt_ms, 4), 'samples_per_s': round(B / (bt_ms * 1e-3), 2),
                        'fwd_ms': round(bf_ms, 4), 'fwd_samples_per_s': round(B / (bf_ms * 1e-3), 2),
                        'peak_tflops': BF16_MFMA_PEAK_TFLOPS}
            for nm, mult in (('mha_fwd_cross', 4.0), ('mha_bwd_cross', 10.0)):
                if nm in kb:
                    us_b = kb[nm]['mean_us']
                    ach_b = mult * B * Q * Ltok * 256 / (us_b * 1e-6) / 1e12
                    bf16_leg[nm] = {'mean_launch_us': us_b, 'achieved_tflops': round(ach_b, 1),
                                    'frac': round(ach_b / BF16_MFMA_PEAK_TFLOPS, 4)}
            bf16_leg['bound'] = ('softmax instruction stream and LDS/barrier latency, not MFMA (head_dim 32: 4 MFMA per 32x32 '
                                 'score block in the forward, 10 in the backward; DESIGN.md section 4)')
        if 'mha_bwd_cross' in kernels:
            us = kernels['mha_bwd_cross']['mean_us']
            kernels['mha_bwd_cross']['tflops'] = round(10.0 * B * Q * Ltok * 256 / (us * 1e-6) / 1e12, 2)
            kernels['mha_bwd_cross']['frac'] = round(kernels['mha_bwd_cross']['tflops'] / peak, 4)
            # `roofline` describes the TIME-DOMINANT kernel of the step: the cross-attention backward (one launch per decoder
            # layer; 10 Q L C algorithmic FLOP: five products).  The cross-attention forward keeps its object
            # under roofline['forward'] (it was `roofline` itself in rounds 1-2).
            fwd_obj = roofline
            flops_b = 10.0 * B * Q * Ltok * 256
            ach_b = flops_b / (us * 1e-6) / 1e12
            traffic_b, src_b, alg_b = None, None, None
            for fn in PMC_TRAFFIC_FILES:
                try:
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                    ent = (pmc.get(f'{args.workload}_{args.dtype}') or pmc[args.workload])['mha_bwd_cross']
                    if B == 1 and ent.get('dtype', 'fp32') == args.dtype:
                        traffic_b, alg_b = ent['traffic_bytes'], ent.get('algorithmic_min_bytes')
                        src_b = f'profiles/{fn} (separate rocprofv3 --pmc passes)'
                        break
                except (OSError, KeyError, ValueError, TypeError):
                    continue
            n_layers = 6
            roofline = {'kernel': ('mha_bwd_bf16_kernel' if args.dtype == 'bf16' else 'mha_bwd_sk_kernel') +
                                  ' (cross-attention backward, one decoder layer): the time-dominant kernel of the step',
                        'bound': 'mfma', 'achieved': round(ach_b, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach_b / peak, 4),
                        'traffic': traffic_b, 'traffic_source': src_b, 'algorithmic_min_bytes': alg_b,
                        'flops_per_launch': flops_b, 'mean_launch_us': us,
                        'share_of_step': round(n_layers * us * 1e-3 / (elapsed / args.steps * 1e3), 4),
                        'forward': fwd_obj}
        if 'coords3d' in kernels:
            us = kernels['coords3d']['mean_us']
            kernels['coords3d']['gbs'] = round(B * Ltok * 192 * 4 / (us * 1e-6) / 1e9, 1)   # volume bytes
            kernels['coords3d']['hbm_frac'] = round(kernels['coords3d']['gbs'] / HBM_PEAK_GBS, 4)

    # ---- the full training step of the reference: forward -> PETRHead.loss -> backward ----
    loss_leg = None
    # single-process runs only: PETRHead.loss averages its normalisers over the process group (mmdet reduce_mean), a
    # collective that a rank-0-only leg must not issue
    if rank == 0 and world == 1 and not args.fwd_only:
        gt_b, gt_l = synthetic_gt(B, 40, seed=7, device=dev)     # 40 ground-truth boxes per sample

        def loss_step():
            head.zero_grad_flat()
            feats.grad = None
            out = head([feats], metas)
            ld = head.loss(gt_b, gt_l, out)
            sum(ld.values()).backward()

        for _ in range(5):
            loss_step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            loss_step()
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t2) / args.steps * 1e3
        with torch.no_grad():
            out = head([feats], metas)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            head.loss(gt_b, gt_l, out)
        torch.cuda.synchronize()
        loss_ms = (time.perf_counter() - t3) / args.steps * 1e3
        loss_leg = {'fwd_loss_bwd_ms_per_step': round(full_ms, 4), 'samples_per_s': round(B / (full_ms * 1e-3), 2),
                    'loss_call_ms': round(loss_ms, 4),
                    'what': 'forward -> PETRHead.loss (device cost matrix + Hungarian assignment + focal/L1 + gradients, '
                            '6 levels, 40 ground-truth boxes per sample) -> backward'}
    if rank == 0:
        log('kernel timing done')

    # ---- the other BASELINE single-GPU workloads (configs[2] p4_1408, configs[3] per-GPU p4_1600, configs[4] per-GPU
    #      v2_800), fp32 and bf16, same step / same timing method, fewer steps; no CPU baselines for these ----
    workloads = None
    if rank == 0 and world == 1 and not args.fwd_only and not args.no_extra_workloads and args.workload == 'c5' \
            and args.batch == 1 and not args.eval_mode:
        head.release()
        del head, feats, g_cls, g_box
        torch.cuda.empty_cache()
        workloads = {}
        for name in ('p4_1408', 'p4_1600', 'v2_800'):
            workloads[name] = extra_workload(name, dev, Q, steps=min(args.steps, 20), warmup=min(max(args.warmup, 3), 5))
            log(f'workload {name}: fp32 {workloads[name]["fp32"]["ms_per_step"]} ms/step, '
                f'bf16 {workloads[name]["bf16"]["ms_per_step"]} ms/step')
    neck_leg = None
    if workloads is not None:
        neck_leg = neck_fold_leg(dev, Q, steps=min(args.steps, 20))
        log(f'neck fold leg: {neck_leg}')
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, B, Q, train=was_training)

    if rank == 0:
        samples = world * B * args.steps
        ms_step = elapsed / args.steps * 1e3
        prec = 'bf16 operands / fp32 accumulate' if args.dtype == 'bf16' else 'fp32'
        out = {
            'metric': 'samples/sec PETRHead fwd+bwd' if not args.fwd_only else 'samples/sec PETRHead fwd',
            'value': round(samples / elapsed, 3), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'{desc}, {Q} queries, 6 decoder layers, {prec}'
                                   + (' (BASELINE configs[1])' if args.workload == 'c5' and args.dtype == 'fp32' else ''),
                       'global_batch': world * B, 'per_gpu_batch': B, 'parallelism': f'dp{world}',
                       'dropout': ('0.1 at all six sites of every decoder layer (training mode; the CPU baseline too)'
                                   if was_training else 'off (eval mode, as the CPU baseline)'),
                       'fwd_only_leg': 'eval mode (inference forward)'},
            'fwd_ms': round(fwd_ms, 4) if fwd_ms is not None else None,
            'fwd_samples_per_s': round(B / (fwd_ms * 1e-3), 2) if fwd_ms else None,
            # whole-step fraction of the matrix peak next to the kernel fraction in `roofline` (algorithmic FLOP of fwd + bwd)
            'step_tflops': None if args.fwd_only else round(step_gflop(args.workload, B) * world / ms_step, 1),
            'step_frac_of_peak': None if args.fwd_only else round(step_gflop(args.workload, B) / ms_step / peak, 4),
            'sustained': sustained,
            'roofline': roofline, 'kernels': kernels, 'cpu_baseline': cpu, 'with_loss': loss_leg,
            'bf16': bf16_leg, 'workloads': workloads, 'neck_fold': neck_leg,
        }
        if cpu and fwd_ms:
            out['fwd_speedup_vs_cpu'] = round((B / (fwd_ms * 1e-3)) / cpu['fwd_value'], 1)
            out['fwdbwd_speedup_vs_cpu'] = round(out['value'] / cpu['value'], 1)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_reducer:
        if world > 1:
            dist.barrier()           # rank 0 ran a few extra legs: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
