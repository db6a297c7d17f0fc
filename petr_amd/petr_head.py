"""PETRHead — host mirror of reference models/dense_heads/petr_head.py:47-468 (forward path).

Same constructor arguments, sub-module / parameter names (state_dict keys load reference ``.pth``
heads, including the legacy-key remap of ``_load_from_state_dict`` :336-364) and the same
``forward(mlvl_feats, img_metas) -> dict`` contract.  The arithmetic is ONE call into libpetr_hip.so
(``petr_head_fwd``; ``petr_head_bwd`` for the gradient) — there is no torch-operator fallback.

Host-side work kept in Python because the reference does it on the host too:
  * ``np.linalg.inv(lidar2img)`` in float64 per view (petr_head.py:308-315), uploaded as fp32;
  * the padding mask, in closed form instead of ``ones -> zero-fill -> F.interpolate`` (:383-394):
    nearest-neighbour source index = floor(dst * in/out), so
    ``mask[i,j] = (floor(i*pad_h/H) >= img_h) or (floor(j*pad_w/W) >= img_w)``.
"""
import copy
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _C, losses, ops
from .positional_encoding import sine_dim_t
from .registry import build_positional_encoding, build_transformer, register


def pos2posemb3d(pos, num_pos_feats=128, temperature=10000):
    """reference petr_head.py:31-43 through the ``petr_posemb3d_fwd`` kernel ([n,3] -> [n,384])."""
    dim_t = sine_dim_t(num_pos_feats, temperature).to(pos.device)
    shp = pos.shape[:-1]
    return ops.posemb3d(pos.reshape(-1, 3).contiguous(), dim_t).view(*shp, 3 * num_pos_feats)


def padding_mask_closed_form(img_metas, num_cams, feat_hw):
    """uint8 [B,N,H,W] equal to the reference's interpolated mask (petr_head.py:383-394)."""
    H, W = feat_hw
    pad_h, pad_w, _ = img_metas[0]['pad_shape'][0]
    # torch nearest: src = min(int(floorf(dst * scale)), in - 1), scale = float(in) / out  (float32)
    sh, sw = np.float32(pad_h) / np.float32(H), np.float32(pad_w) / np.float32(W)
    src_h = np.minimum(np.floor(np.arange(H, dtype=np.float32) * sh).astype(np.int64), pad_h - 1)
    src_w = np.minimum(np.floor(np.arange(W, dtype=np.float32) * sw).astype(np.int64), pad_w - 1)
    B = len(img_metas)
    mask = np.zeros((B, num_cams, H, W), dtype=np.uint8)
    for b in range(B):
        for n in range(num_cams):
            img_h, img_w, _ = img_metas[b]['img_shape'][n]
            mask[b, n] = (src_h[:, None] >= img_h) | (src_w[None, :] >= img_w)
    return mask


def depth_bins(depth_num, depth_start, position_range, LID):
    """reference petr_head.py:293-301 with the same torch ops (host, once)."""
    index = torch.arange(start=0, end=depth_num, step=1).float()
    if LID:
        index_1 = index + 1
        bin_size = (position_range[3] - depth_start) / (depth_num * (1 + depth_num))
        return depth_start + bin_size * index * index_1
    bin_size = (position_range[3] - depth_start) / depth_num
    return depth_start + bin_size * index


class SELayer(nn.Module):
    """parameter container of reference petrv2_head.py:48-60 (the gate runs in petr_gate_fwd/_bwd)."""

    def __init__(self, channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = nn.ReLU()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = nn.Sigmoid()


class RegLayer(nn.Module):
    """parameter container of reference petrv2_head.py:63-95."""

    def __init__(self, embed_dims=256, shared_reg_fcs=2, group_reg_dims=(2, 1, 3, 2, 2)):
        super().__init__()
        reg_branch = []
        for _ in range(shared_reg_fcs):
            reg_branch += [nn.Linear(embed_dims, embed_dims), nn.ReLU(), nn.Dropout(0.0)]
        self.reg_branch = nn.Sequential(*reg_branch)
        self.task_heads = nn.ModuleList([
            nn.Sequential(nn.Linear(embed_dims, embed_dims), nn.ReLU(), nn.Linear(embed_dims, d)) for d in group_reg_dims])


class _HeadFn(torch.autograd.Function):
    """autograd node around petr_head_fwd / petr_head_bwd.  Parameter gradients are written by the
    kernels straight into the head's flat gradient buffer (each ``p.grad`` is a view of it), the
    return value only carries d(feats)."""

    @staticmethod
    def forward(ctx, head, feats, anchor, run):
        ctx.head, ctx.run = head, run
        ctx.feats_requires_grad = feats.requires_grad
        cls, bbox = head._launch_forward(run, feats)
        # the box backward recomputes the sigmoid derivative from the OUTPUT buffer (petr_bbox_epilogue_bwd), so an
        # in-place edit of all_bbox_preds between forward and backward (the reference edits it in place) must trip
        # autograd's version check instead of silently corrupting the gradients
        ctx.save_for_backward(bbox)
        return cls, bbox

    @staticmethod
    def backward(ctx, d_cls, d_bbox):
        head, run = ctx.head, ctx.run
        _ = ctx.saved_tensors            # raises if all_bbox_preds was modified in place
        if run.consumed:
            raise _C.PetrHipError('this forward has already been backpropagated: its workspace went back to the pool '
                                  '(retain_graph / a second backward is not supported; run the forward again)')
        run.consumed = True
        d_feats = head._launch_backward(run, d_cls.contiguous(), d_bbox.contiguous(), ctx.feats_requires_grad)
        return None, d_feats, None, None


class _PinnedRing:
    """Asynchronous host -> device upload of a small per-call array (img2lidar, the padding mask).

    A copy from pageable memory blocks the host until everything queued on the stream before it has run - the whole
    previous training step - so the device idles at every step boundary while the host catches up (~130 us per step on
    MI355X, measured in the kernel trace).  The array goes through a ring of page-locked buffers instead; a slot is reused
    only after the copy that read it has completed (its event), which with four slots never waits in practice."""

    def __init__(self, shape, dtype, device, slots=4):
        self.device = device
        self.bufs = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def upload(self, arr):
        i, self.i = self.i, (self.i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        self.bufs[i].numpy()[...] = arr
        out = self.bufs[i].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.events[i] = ev
        return out


class _Run:
    """Everything one forward needs to keep alive until its backward: workspace, outputs, host inputs."""
    __slots__ = ('cfg', 'io', 'ws', 'feats', 'img2lidar', 'mask', 'cls', 'bbox', 'key', 'time_div', 'consumed')


@register('HEADS')
class PETRHead(nn.Module):
    _version = 2

    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None,
                 sync_cls_avg_factor=False,
                 positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True),
                 code_weights=None, bbox_coder=None, loss_cls=None, loss_bbox=None, loss_iou=None, train_cfg=None,
                 test_cfg=dict(max_per_img=100), with_position=True, with_multiview=False, depth_step=0.8, depth_num=64,
                 LID=False, depth_start=1, position_range=[-65, -65, -8.0, 65, 65, 8.0], init_cfg=None,
                 normedlinear=False, **kwargs):
        super().__init__()
        self._v2 = bool(kwargs.pop('_v2', False))
        self.with_fpe = bool(kwargs.get('with_fpe', False)) and self._v2
        self.with_time = bool(kwargs.get('with_time', False)) and self._v2
        self.with_multi = bool(kwargs.get('with_multi', False)) and self._v2
        self.group_reg_dims = tuple(kwargs.get('group_reg_dims', (2, 1, 3, 2, 2)))
        assert not self.with_multi or self.group_reg_dims == (2, 1, 3, 2, 2), 'RegLayer groups are (2,1,3,2,2)'
        self.code_size = kwargs.get('code_size', 10)
        cw = code_weights if code_weights is not None else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]
        cw = list(cw)[:self.code_size]
        self.sync_cls_avg_factor = sync_cls_avg_factor
        # not a reference option: 'bf16' = BASELINE configs 3-5: token-sized contractions and the cross-attention on the
        # bf16 matrix cores, forward and backward (see _launch_forward)
        self.attn_dtype = kwargs.get('attn_dtype', 'fp32')
        self.num_query, self.num_classes, self.in_channels = num_query, num_classes, in_channels
        self.num_reg_fcs = num_reg_fcs
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.loss_cfg = dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_iou=loss_iou)
        self.fp16_enabled = False
        self.embed_dims = 256                       # petr_head.py:175
        self.depth_step, self.depth_num = depth_step, depth_num
        self.position_dim = 3 * depth_num
        self.position_range = list(position_range)
        self.LID, self.depth_start = LID, depth_start
        self.position_level = kwargs.get('position_level', 0) if self._v2 else 0   # petrv2_head.py:165,239
        self.with_position, self.with_multiview = with_position, with_multiview
        assert 'num_feats' in positional_encoding
        num_feats = positional_encoding['num_feats']
        assert num_feats * 2 == self.embed_dims, \
            f'embed_dims should be exactly 2 times of num_feats. Found {self.embed_dims} and {num_feats}.'
        assert num_reg_fcs == 2, 'the fused branches kernel chain is built for num_reg_fcs=2 (every reference config)'
        assert not normedlinear, 'normedlinear=True is not used by any reference config'
        assert with_position and with_multiview, \
            'the fused executor implements the with_position=with_multiview=True path of the BASELINE configs'
        self.num_pred = 6                           # petr_head.py:192
        self.normedlinear = normedlinear
        use_sigmoid = True if loss_cls is None else loss_cls.get('use_sigmoid', False)
        self.cls_out_channels = num_classes if use_sigmoid else num_classes + 1
        self._use_sigmoid = use_sigmoid
        # construction order follows the reference (:208-215) so that a seeded init draws the same numbers
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.code_weights = nn.Parameter(torch.tensor(cw, requires_grad=False), requires_grad=False)
        self.pc_range = list(bbox_coder['pc_range']) if bbox_coder is not None else [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
        self.bbox_coder_cfg = bbox_coder
        self._init_layers()
        dec = self.transformer.decoder
        self._nl = dec.num_layers
        self._ffn = dec.layers[0].ffns[0].feedforward_channels
        self._heads = dec.layers[0].attentions[0].num_heads
        self.register_buffer('_depth', depth_bins(depth_num, depth_start, self.position_range, LID), persistent=False)
        self.register_buffer('_dim_t', sine_dim_t(num_feats, positional_encoding.get('temperature', 10000)),
                             persistent=False)
        self._flat = None
        self._flat_grad = None
        self._layout = None
        self._runs = {}
        self._free_ws = {}
        self._rings = {}
        self._anchor = None
        self._ctx = None

    # ------------------------------------------------------------------ construction
    def _init_layers(self):
        """reference petr_head.py:217-274."""
        E = self.embed_dims
        self.input_proj = nn.Conv2d(self.in_channels, E, kernel_size=1)
        cls_branch = []
        for _ in range(self.num_reg_fcs):
            cls_branch += [nn.Linear(E, E), nn.LayerNorm(E), nn.ReLU(inplace=True)]
        cls_branch.append(nn.Linear(E, self.cls_out_channels))
        fc_cls = nn.Sequential(*cls_branch)
        if self.with_multi:
            reg_branch = RegLayer(E, self.num_reg_fcs, self.group_reg_dims)
        else:
            reg_branch = []
            for _ in range(self.num_reg_fcs):
                reg_branch += [nn.Linear(E, E), nn.ReLU()]
            reg_branch.append(nn.Linear(E, self.code_size))
            reg_branch = nn.Sequential(*reg_branch)
        if self._v2:   # PETRv2Head deep-copies one template per level (petrv2_head.py:304-307)
            self.cls_branches = nn.ModuleList([copy.deepcopy(fc_cls) for _ in range(self.num_pred)])
            self.reg_branches = nn.ModuleList([copy.deepcopy(reg_branch) for _ in range(self.num_pred)])
        else:          # the SAME module in all num_pred slots (petr_head.py:244-247): shared weights, aliased keys
            self.cls_branches = nn.ModuleList([fc_cls for _ in range(self.num_pred)])
            self.reg_branches = nn.ModuleList([reg_branch for _ in range(self.num_pred)])
        self.adapt_pos3d = nn.Sequential(nn.Conv2d(E * 3 // 2, E * 4, 1), nn.ReLU(), nn.Conv2d(E * 4, E, 1))
        self.position_encoder = nn.Sequential(nn.Conv2d(self.position_dim, E * 4, 1), nn.ReLU(), nn.Conv2d(E * 4, E, 1))
        self.reference_points = nn.Embedding(self.num_query, 3)
        self.query_embedding = nn.Sequential(nn.Linear(E * 3 // 2, E), nn.ReLU(), nn.Linear(E, E))
        if self.with_fpe:
            self.fpe = SELayer(E)

    def init_weights(self):
        """reference petr_head.py:276-284."""
        self.transformer.init_weights()
        nn.init.uniform_(self.reference_points.weight.data, 0, 1)
        if self._use_sigmoid:
            bias_init = float(-np.log((1 - 0.01) / 0.01))   # mmcv bias_init_with_prob(0.01)
            for m in self.cls_branches:
                nn.init.constant_(m[-1].bias, bias_init)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """legacy DETR key names -> current ones (reference petr_head.py:345-359)."""
        version = local_metadata.get('version', None)
        if (version is None or version < 2) and self.__class__ in (PETRHead, PETRv2Head):
            convert_dict = {'.self_attn.': '.attentions.0.', '.multihead_attn.': '.attentions.1.',
                            '.decoder.norm.': '.decoder.post_norm.'}
            for k in list(state_dict.keys()):
                for ori_key, convert_key in convert_dict.items():
                    if ori_key in k:
                        state_dict[k.replace(ori_key, convert_key)] = state_dict[k]
                        del state_dict[k]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    # ------------------------------------------------------------------ flat parameter storage
    def _base_config(self):
        cfg = _C.HeadConfig()
        cfg.B, cfg.N, cfg.C_in, cfg.H, cfg.W = 1, 1, self.in_channels, 2, 2
        cfg.num_query, cfg.num_layers, cfg.num_heads = self.num_query, self._nl, self._heads
        cfg.embed_dims, cfg.ffn_dims, cfg.depth_num = self.embed_dims, self._ffn, self.depth_num
        cfg.num_classes, cfg.code_size = self.cls_out_channels, self.code_size
        cfg.v2, cfg.with_fpe, cfg.with_time, cfg.with_multi = int(self._v2), int(self.with_fpe), int(self.with_time), int(self.with_multi)
        cfg.shared_branches, cfg.LID, cfg.depth_start = int(not self._v2), int(self.LID), float(self.depth_start)
        cfg.position_range = (C.c_float * 6)(*[float(v) for v in self.position_range])
        cfg.pc_range = (C.c_float * 6)(*[float(v) for v in self.pc_range])
        cfg.pad_h = cfg.pad_w = 0.0
        cfg.has_mask, cfg.training = 0, 1
        return cfg

    def _flatten(self):
        """(Re)build the flat fp32 buffer on the parameters' device and alias every Parameter onto it."""
        L = _C.lib()
        lay = _C.HeadLayout()
        _C.check(L.petr_head_layout(C.byref(self._base_config()), C.byref(lay)), 'petr_head_layout')
        sd = dict(self.named_parameters(remove_duplicate=False))
        dev = self.input_proj.weight.device
        flat = torch.zeros(lay.total, dtype=torch.float32, device=dev)
        grad = torch.zeros(lay.total, dtype=torch.float32, device=dev)
        self._grad_views = []
        seen = {}
        for i in range(lay.count):
            name = lay.name[i].value.decode()
            shape = tuple(lay.shape[i][j] for j in range(lay.ndim[i]))
            p = sd.get(name)
            if p is None:
                raise _C.PetrHipError(f'layout names a tensor the module does not have: {name}')
            if tuple(p.shape) != shape:
                raise _C.PetrHipError(f'{name}: module shape {tuple(p.shape)} != layout shape {shape}')
            if id(p) in seen:
                assert seen[id(p)] == lay.offset[i], f'aliased tensor {name} must share one slot'
                continue
            off, n = lay.offset[i], p.numel()
            seen[id(p)] = off
            view = flat[off:off + n].view(shape)
            view.copy_(p.data)
            p.data = view
            self._grad_views.append((p, grad[off:off + n].view(shape)))
        missing = [k for k, p in sd.items() if id(p) not in seen]
        if missing:
            raise _C.PetrHipError(f'parameters without a slot in the flat layout: {missing}')
        self._flat, self._flat_grad, self._layout = flat, grad, lay
        self._runs.clear()
        self._free_ws.clear()
        self._anchor = torch.zeros((), device=dev, requires_grad=True)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._flat = None          # storage moved or changed dtype: re-flatten lazily
        return out

    def _ensure_flat(self):
        w = self.input_proj.weight
        if (self._flat is None or w.device != self._flat.device
                or w.data_ptr() < self._flat.data_ptr()
                or w.data_ptr() >= self._flat.data_ptr() + self._flat.numel() * 4):
            if w.dtype != torch.float32:
                raise _C.PetrHipError('PETRHead runs fp32 parameters (the reference head is fp32: fp16_enabled=False)')
            self._flatten()

    def flat_parameters(self):
        self._ensure_flat()
        return self._flat

    def flat_gradients(self):
        """The single contiguous gradient buffer (every ``p.grad`` is a view of it)."""
        self._ensure_flat()
        self._attach_grads(zero_if_detached=False)
        return self._flat_grad

    def zero_grad_flat(self):
        """Clear all gradients with ONE memset of the flat buffer (instead of 222 per-tensor kernels)."""
        self._ensure_flat()
        self._attach_grads(zero_if_detached=False)
        self._flat_grad.zero_()

    def gradient_buckets(self):
        """[(begin, end)] flat ranges that become final after each backward stage, in completion order."""
        self._ensure_flat()
        L = _C.lib()
        cfg = self._base_config()
        out = []
        for s in range(L.petr_head_bwd_num_stages(C.byref(cfg))):
            b, e = C.c_long(), C.c_long()
            _C.check(L.petr_head_bwd_stage_range(C.byref(cfg), s, C.byref(b), C.byref(e)), 'petr_head_bwd_stage_range')
            out.append((b.value, e.value))
        return out

    def _attach_grads(self, zero_if_detached=True):
        # sentinel = the first parameter that takes a gradient (a frozen first parameter never gets .grad, which would
        # re-zero the flat buffer on every backward and break gradient accumulation)
        sentinel = next(((p, gv) for p, gv in self._grad_views if p.requires_grad), None)
        if sentinel is None:
            return
        if sentinel[0].grad is None or sentinel[0].grad.data_ptr() != sentinel[1].data_ptr():
            if zero_if_detached:
                self._flat_grad.zero_()
            for p, gv in self._grad_views:
                if p.requires_grad:
                    p.grad = gv

    # ------------------------------------------------------------------ host-side input preparation
    def _prepare(self, feats, img_metas, projected=False):
        if projected:       # feats = the projected memory [B, N, H, W, C] (forward_projected)
            B, N, H, W, Cm = feats.shape
            assert Cm == self.embed_dims, f'expected {self.embed_dims} memory channels, got {Cm}'
        else:
            B, N, Cin, H, W = feats.shape
            assert Cin == self.in_channels, f'expected {self.in_channels} input channels, got {Cin}'
        assert len(img_metas) == B
        pad_h, pad_w, _ = img_metas[0]['pad_shape'][0]
        mask_np = padding_mask_closed_form(img_metas, N, (H, W))
        has_mask = bool(mask_np.any())
        key = (B, N, H, W, int(pad_h), int(pad_w), has_mask, feats.device)
        L = _C.lib()
        if key not in self._runs:
            cfg = self._base_config()
            cfg.B, cfg.N, cfg.H, cfg.W = B, N, H, W
            cfg.pad_h, cfg.pad_w, cfg.has_mask = float(pad_h), float(pad_w), int(has_mask)
            ws_bytes = L.petr_head_workspace_bytes(C.byref(cfg))
            if ws_bytes == 0:
                raise _C.PetrHipError(f'petr_head_workspace_bytes: {L.petr_last_error().decode()}')
            self._runs[key] = (cfg, ws_bytes)
            self._free_ws[key] = []
        cfg, ws_bytes = self._runs[key]
        run = _Run()
        run.key, run.cfg, run.time_div, run.consumed = key, cfg, 0.0, False
        pool = self._free_ws[key]
        run.ws = pool.pop() if pool else torch.empty(ws_bytes // 4, dtype=torch.float32, device=feats.device)
        # img2lidar: float64 inverse per view on the host, as the reference (petr_head.py:308-315)
        l2i = np.asarray([[np.asarray(m) for m in meta['lidar2img']] for meta in img_metas], dtype=np.float64)
        i2l = np.linalg.inv(l2i).astype(np.float32).reshape(B * N, 16)
        run.img2lidar = self._upload('img2lidar', i2l, feats.device)
        run.mask = self._upload('mask', mask_np, feats.device) if has_mask else None
        return run

    def _upload(self, name, arr, device):
        arr = np.ascontiguousarray(arr)
        key = (name, arr.shape, arr.dtype.str, device)
        ring = self._rings.get(key)
        if ring is None:
            ring = self._rings[key] = _PinnedRing(arr.shape, torch.from_numpy(arr).dtype, device)
        with torch.cuda.device(device):
            return ring.upload(arr)

    def _context(self):
        """side streams for the work off the critical path (petr_ctx); PETR_AMD_SIDE_STREAMS=0 serialises."""
        n = int(os.environ.get('PETR_AMD_SIDE_STREAMS', '2'))
        if n <= 0:
            return None
        if self._ctx is None:
            self._ensure_flat()
            with torch.cuda.device(self._flat.device):      # the side streams belong to the head's device
                h = C.c_void_p()
                _C.check(_C.lib().petr_ctx_create(C.byref(h), n), 'petr_ctx_create')
            self._ctx = h
        return self._ctx

    def __getstate__(self):
        """pickling / copy.deepcopy: drop the process-local handles (stream context, pooled workspaces, live runs)."""
        st = self.__dict__.copy()
        for k in ('_ctx', '_last_run', '_stage_hook', '_stage_hook_stages'):
            st.pop(k, None)
        st['_ctx'] = None
        st['_runs'], st['_free_ws'], st['_rings'] = {}, {}, {}
        st['_flat'] = st['_flat_grad'] = st['_layout'] = st['_anchor'] = None      # re-flattened lazily on first use
        st.pop('_grad_views', None)
        return st

    def release(self):
        """Give back the process-local GPU resources now: the side streams / events of the stream context and the pooled
        workspaces.  The head stays usable (both are re-created on the next forward).  Streams that outlive their head
        keep their hardware queues: a process that builds several heads one after the other (bench.py's workloads) should
        not leave it to the garbage collector."""
        ctx, self._ctx = getattr(self, '_ctx', None), None
        if ctx is not None:
            torch.cuda.synchronize()
            _C.lib().petr_ctx_destroy(ctx)
        self._runs, self._free_ws, self._rings = {}, {}, {}
        self._last_run = None

    def __del__(self):
        ctx = getattr(self, '_ctx', None)
        if ctx is not None:
            try:
                _C.lib().petr_ctx_destroy(ctx)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

    def _time_div(self, img_metas, batch_size):
        """mean_time_stamp of reference petrv2_head.py:499-505 (B = 1: SURVEY §7.3)."""
        if not self.with_time:
            return 0.0
        if batch_size != 1:
            raise NotImplementedError('with_time: the reference broadcast [B,Q,2] / [B] only works for B = 1')
        ts = np.asarray([np.asarray(m['timestamp']) for m in img_metas], dtype=np.float32).reshape(batch_size, -1, 6)
        return float((ts[:, 1, :] - ts[:, 0, :]).mean(-1)[0])

    def _launch_forward(self, run, feats, projected=False):
        with torch.cuda.device(self._flat.device):       # streams / side streams of the head's own device
            return self._launch_forward_impl(run, feats, projected)

    def _launch_forward_impl(self, run, feats, projected=False):
        L = _C.lib()
        B, N = run.cfg.B, run.cfg.N
        run.feats = feats.contiguous()
        run.cls = torch.empty((self._nl, B, self.num_query, self.cls_out_channels), dtype=torch.float32, device=feats.device)
        run.bbox = torch.empty((self._nl, B, self.num_query, self.code_size), dtype=torch.float32, device=feats.device)
        io = _C.HeadIO()
        io.params = self._flat.data_ptr()
        io.feats = None if projected else run.feats.data_ptr()
        io.memory_in = run.feats.data_ptr() if projected else None
        io.img2lidar = run.img2lidar.data_ptr()
        io.depth = self._depth.data_ptr()
        io.dim_t = self._dim_t.data_ptr()
        io.mask = run.mask.data_ptr() if run.mask is not None else None
        io.time_div = float(run.time_div)
        io.all_cls_scores = run.cls.data_ptr()
        io.all_bbox_preds = run.bbox.data_ptr()
        io.ws = run.ws.data_ptr()
        io.ws_bytes = run.ws.numel() * 4
        io.ctx = self._context()
        # training mode: the decoder's six dropout layers per decoder layer run inside the kernels from a
        # counter-based generator; a fresh seed per forward is drawn from torch's default CPU generator
        # (torch.manual_seed makes runs repeatable).  eval(): p = 0, the dropout-free kernels.
        drop_p = float(self._dropout_p()) if self.training else 0.0
        io.dropout_p, io.dropout_seed = drop_p, 0
        if drop_p > 0.0:
            seed = getattr(self, '_dropout_seed_override', None)
            io.dropout_seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        self._last_dropout = (int(io.dropout_seed), drop_p)
        # attn_dtype = 'bf16' (attribute, default 'fp32'; BASELINE configs 3-5): what torch.autocast(bfloat16) would do
        # to the token-sized part of the head - every L-row contraction (input_proj, both position-embedding MLPs, the
        # K/V projections, PETRv2's fpe) and the cross-attention round their operands to bf16 and accumulate in fp32, in
        # the forward and in the backward (petr_mha_bwd_bf16, gemm_bf16.hip); parameters, gradients, softmax statistics,
        # LayerNorms and the 900-row query side stay fp32.
        attn_dtype = getattr(self, 'attn_dtype', 'fp32')
        if attn_dtype not in ('fp32', 'bf16'):
            raise ValueError(f"attn_dtype must be 'fp32' or 'bf16', got {attn_dtype!r}")
        io.attn_bf16 = 1 if attn_dtype == 'bf16' else 0
        run.io = io
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _C.check(L.petr_head_fwd(C.byref(run.cfg), C.byref(io), stream), 'petr_head_fwd')
        return run.cls, run.bbox

    def side_stream(self, i=1):
        """Side stream ``i`` of this head's stream context as a ``torch.cuda.ExternalStream`` (None without a context): the
        gradient exchange rides on it instead of opening one more stream (HIP multiplexes streams onto 4 hardware queues)."""
        ctx = self._context()
        if ctx is None:
            return None
        h = C.c_void_p()
        if _C.lib().petr_ctx_side_stream(ctx, int(i), C.byref(h)) != 0:
            return None
        return torch.cuda.ExternalStream(h.value, device=self._flat.device)

    def join_streams_into(self, target_stream):
        """Make ``target_stream`` (a torch.cuda.Stream) wait for everything this head has enqueued so far on the
        current stream and on its side streams (petr_ctx_join_into): used by the gradient exchange between backward
        stages, so that the compute stream never stops at a stage boundary."""
        cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _C.check(_C.lib().petr_ctx_join_into(self._context(), cur, C.c_void_p(target_stream.cuda_stream)),
                 'petr_ctx_join_into')

    def _dropout_p(self):
        """The one dropout rate of the decoder (reference configs: attn_drop = dropout_layer = ffn_drop = 0.1).
        The executor applies a single rate to all six sites of a layer and refuses anything else loudly."""
        rates = set()
        for layer in self.transformer.decoder.layers:
            for att in layer.attentions:
                rates.add(float(att.attn_drop_p))
                rates.add(float(att.dropout_layer.p) if isinstance(att.dropout_layer, nn.Dropout) else 0.0)
                if float(att.proj_drop.p) != 0.0:
                    raise _C.PetrHipError('proj_drop != 0 is not implemented (no reference config uses it)')
            for ffn in layer.ffns:
                rates.add(float(ffn.layers[0][2].p))
                rates.add(float(ffn.layers[2].p))
        if len(rates) != 1:
            raise _C.PetrHipError(f'the fused executor needs one dropout rate for the whole decoder, got {sorted(rates)}')
        return rates.pop()

    def _launch_backward(self, run, d_cls, d_bbox, want_dfeats, stage_hook=None):
        with torch.cuda.device(self._flat.device):
            return self._launch_backward_impl(run, d_cls, d_bbox, want_dfeats, stage_hook)

    def _launch_backward_impl(self, run, d_cls, d_bbox, want_dfeats, stage_hook=None):
        L = _C.lib()
        self._attach_grads()
        g = _C.HeadGrads()
        g.d_cls, g.d_bbox = d_cls.data_ptr(), d_bbox.data_ptr()
        g.d_params = self._flat_grad.data_ptr()
        d_feats = torch.empty_like(run.feats) if want_dfeats else None
        g.d_feats = d_feats.data_ptr() if want_dfeats else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        n_stages = L.petr_head_bwd_num_stages(C.byref(run.cfg))
        hook = stage_hook or getattr(self, '_stage_hook', None)
        if hook is None:
            _C.check(L.petr_head_bwd(C.byref(run.cfg), C.byref(run.io), C.byref(g), 0, n_stages, stream), 'petr_head_bwd')
        else:   # data-parallel: hand each finished gradient bucket to the all-reducer while backward continues
            # one call per bucket (stage ranges end where the hook wants to be called), not per stage
            ends = getattr(self, '_stage_hook_stages', None) or list(range(n_stages))
            ends = sorted(set(int(e) for e in ends if 0 <= int(e) < n_stages) | {n_stages - 1})
            begin = 0
            for e in ends:
                _C.check(L.petr_head_bwd(C.byref(run.cfg), C.byref(run.io), C.byref(g), begin, e + 1, stream), 'petr_head_bwd')
                for st in range(begin, e + 1):
                    hook(st)
                begin = e + 1
        self._free_ws[run.key].append(run.ws)
        return d_feats

    # ------------------------------------------------------------------ the hot path
    def forward(self, mlvl_feats, img_metas):
        """reference petr_head.py:366-468.  mlvl_feats[0]: [B,N,C_in,H,W] fp32 CUDA; returns the same dict."""
        x = mlvl_feats[self.position_level]
        if not x.is_cuda:
            raise _C.PetrHipError('PETRHead (petr_amd) runs on the GPU only; there is no CPU fallback')
        if x.dtype != torch.float32:
            raise _C.PetrHipError('PETRHead expects fp32 features (the reference forces fp32: petr3d.py:101)')
        self._ensure_flat()
        run = self._prepare(x, img_metas)
        run.time_div = self._time_div(img_metas, x.shape[0])
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p, _ in self._grad_views)):
            cls, bbox = _HeadFn.apply(self, x, self._anchor, run)
        else:
            cls, bbox = self._launch_forward(run, x)
            self._free_ws[run.key].append(run.ws)     # stream-ordered reuse: the next forward runs after this one
        self._last_run = run
        return {'all_cls_scores': cls, 'all_bbox_preds': bbox, 'enc_cls_scores': None, 'enc_bbox_preds': None}

    def forward_projected(self, memory, img_metas):
        """Inference entry for a producer that already applied ``input_proj`` ("fusing [the neck's] last
        conv with input_proj removes one full feature-map round trip"; ``CPFPN.forward_folded``): ``memory`` is the projected
        map, channels-last ``[B, N, H, W, embed_dims]`` fp32 = the token order of reference petr_transformer.py:90.  Everything
        behind petr_head.py:390 is the ordinary forward; there is no gradient through this entry (``eval()`` only)."""
        if not memory.is_cuda or memory.dtype != torch.float32:
            raise _C.PetrHipError('PETRHead.forward_projected expects an fp32 CUDA tensor [B, N, H, W, C]')
        if self.training:
            raise _C.PetrHipError('PETRHead.forward_projected is an inference path (the folded conv has no input_proj gradient): call eval()')
        self._ensure_flat()
        run = self._prepare(memory, img_metas, projected=True)
        run.time_div = self._time_div(img_metas, memory.shape[0])
        with torch.no_grad():
            cls, bbox = self._launch_forward(run, memory, projected=True)
        self._free_ws[run.key].append(run.ws)
        self._last_run = run
        return {'all_cls_scores': cls, 'all_bbox_preds': bbox, 'enc_cls_scores': None, 'enc_bbox_preds': None}

    def workspace_view(self, name, run=None):
        """Named view into the forward workspace of the most recent call (tests / per-module API)."""
        run = run or self._last_run
        L = _C.lib()
        off, n = C.c_long(), C.c_long()
        _C.check(L.petr_head_ws_view(C.byref(run.cfg), name.encode(), C.byref(off), C.byref(n)), 'petr_head_ws_view')
        return run.ws[off.value:off.value + n.value]

    def position_embeding(self, img_feats, img_metas, masks=None):
        """reference petr_head.py:286-334: returns (coords_position_embeding [B,N,256,H,W], coords_mask)."""
        B, N, _, H, W = img_feats[self.position_level].shape
        dev = img_feats[self.position_level].device
        pad_h, pad_w, _ = img_metas[0]['pad_shape'][0]
        l2i = np.asarray([[np.asarray(m) for m in meta['lidar2img']] for meta in img_metas], dtype=np.float64)
        i2l = torch.from_numpy(np.linalg.inv(l2i).astype(np.float32).reshape(B * N, 16)).to(dev)
        vol, cmask = ops.coords3d(i2l, self._depth.to(dev), B, N, H, W, pad_h, pad_w, self.position_range,
                                  want_mask=True)
        if masks is not None:
            cmask = masks | cmask
        with torch.no_grad():
            pe0, pe2 = self.position_encoder[0], self.position_encoder[2]
            hid = ops.conv1x1(vol.view(B * N, -1, H * W), pe0.weight.view(pe0.out_channels, -1), pe0.bias, relu=True)
            out = ops.linear(hid, pe2.weight.view(pe2.out_channels, -1), pe2.bias)
        pe = out.view(B, N, H, W, self.embed_dims).permute(0, 1, 4, 2, 3).contiguous()
        return pe, cmask

    # ------------------------------------------------------------------ §8(f) "next" rows
    def _loss_config(self):
        lc, lb = self.loss_cfg.get('loss_cls') or {}, self.loss_cfg.get('loss_bbox') or {}
        assert lc.get('type', 'FocalLoss') == 'FocalLoss' and lc.get('use_sigmoid', True), \
            'the device loss implements the sigmoid FocalLoss of every reference config'
        assert lb.get('type', 'L1Loss') == 'L1Loss'
        cls_w, box_w = float(lc.get('loss_weight', 2.0)), float(lb.get('loss_weight', 0.25))
        assigner = (self.train_cfg or {}).get('assigner') if isinstance(self.train_cfg, dict) else None
        if assigner is not None:     # petr_head.py:149-156
            assert cls_w == assigner['cls_cost']['weight'], \
                'The classification weight for loss and matcher should be exactly the same.'
            assert box_w == assigner['reg_cost']['weight'], \
                'The regression L1 weight for loss and matcher should be exactly the same.'
        return losses.LossConfig(self.num_classes, self.code_weights.detach().cpu().tolist(), cls_w, box_w,
                                 float(lc.get('alpha', 0.25)), float(lc.get('gamma', 2.0)), 0.0,
                                 sync_cls_avg_factor=self.sync_cls_avg_factor)

    def loss(self, gt_bboxes_list, gt_labels_list, preds_dicts, gt_bboxes_ignore=None):
        """reference petr_head.py:646-728.  Cost matrix, Hungarian assignment, focal + L1 loss and their gradients for
        all decoder levels run in one call on the device (petr_loss_fwd_bwd); no host round trip."""
        assert gt_bboxes_ignore is None, f'{self.__class__.__name__} only supports for gt_bboxes_ignore setting to None.'
        # multi-process: num_total_pos (and cls_avg_factor with sync_cls_avg_factor) are averaged over the default
        # process group inside head_loss, as mmdet's reduce_mean does (petr_head.py:620-622, 628-631)
        return losses.head_loss(self._loss_config(), gt_bboxes_list, gt_labels_list, preds_dicts)

    def get_bboxes(self, preds_dicts, img_metas, rescale=False):
        """reference petr_head.py:730-751 (NMSFreeCoder.decode + gravity centre -> bottom centre)."""
        if getattr(self, 'bbox_coder', None) is None:
            cfg = dict(self.bbox_coder_cfg or dict(pc_range=self.pc_range,
                                                   post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_num=300))
            cfg.pop('type', None)
            cfg.setdefault('num_classes', self.num_classes)
            self.bbox_coder = losses.NMSFreeCoder(**cfg)
        return losses.get_bboxes(self.bbox_coder, preds_dicts, img_metas, rescale)


@register('HEADS')
class PETRv2Head(PETRHead):
    """reference models/dense_heads/petrv2_head.py:99-540: PETRHead + feature-guided PE (``with_fpe``, SELayer),
    grouped regression (``with_multi``, RegLayer), velocity / dt (``with_time``), ``position_level``, one deep copy
    of the branches per decoder level.  Same executor, ``v2`` switches on."""

    def __init__(self, *args, with_fpe=False, with_time=False, with_multi=False, group_reg_dims=(2, 1, 3, 2, 2),
                 position_level=0, **kwargs):
        super().__init__(*args, _v2=True, with_fpe=with_fpe, with_time=with_time, with_multi=with_multi,
                         group_reg_dims=group_reg_dims, position_level=position_level, **kwargs)
