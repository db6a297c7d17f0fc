"""Host side of the two steps right behind the hot path: the training loss with Hungarian
matching and the NMS-free decode, both as ONE C-ABI call into libpetr_hip.so (petr_amd/csrc/loss.hip).

Mirrors (reference projects/mmdet3d_plugin/): ``PETRHead.loss`` / ``get_bboxes``
(models/dense_heads/petr_head.py:646-751), ``HungarianAssigner3D`` (core/bbox/assigners/hungarian_assigner_3d.py),
``NMSFreeCoder`` (core/bbox/coders/nms_free_coder.py), ``normalize_bbox`` / ``denormalize_bbox`` (core/bbox/util.py).
torch is plumbing here (tensors, autograd node, ``topk``); the arithmetic — cost matrix, assignment, focal / L1
loss, their gradients, box denormalisation — runs in the HIP kernels.
"""
import ctypes as C

import torch

from . import _C


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class LossConfig:
    """Hyper-parameters the reference spreads over loss_cls / loss_bbox / train_cfg.assigner / code_weights
    (configs/petr/petr_r50dcn_gridmask_c5.py:45-110); the assigner's cost weights must equal the loss weights
    (asserted by the reference, petr_head.py:149-156)."""

    def __init__(self, num_classes=10, code_weights=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2), cls_weight=2.0,
                 bbox_weight=0.25, alpha=0.25, gamma=2.0, bg_cls_weight=0.0, sync_cls_avg_factor=False):
        self.sync_cls_avg_factor = bool(sync_cls_avg_factor)
        self.num_classes = num_classes
        self.code_weights = [float(v) for v in code_weights] + [0.0] * (10 - len(code_weights))
        self.cls_weight, self.bbox_weight, self.alpha, self.gamma = cls_weight, bbox_weight, alpha, gamma
        self.bg_cls_weight = bg_cls_weight


def synced_avg_factors(num_pos, num_queries_total, cfg, device, group=None):
    """The two normalisers the reference averages over the ranks (mmdet ``reduce_mean``): ``num_total_pos`` always
    (petr_head.py:628-631) and ``cls_avg_factor`` when ``sync_cls_avg_factor`` (:620-622).  Returns a float32 tensor
    ``[cls_avg_factor, num_total_pos]`` on ``device`` holding the values BEFORE their ``max(., 1)`` clamps (the kernels
    clamp), or ``None`` in a single process (the kernels then derive both from ``num_pos``).  The all-reduce is
    enqueued like any other collective (RCCL: on the current stream; no ``.item()``, no host round trip)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    world = dist.get_world_size(group)
    cls_avg = float(num_pos) + (float(num_queries_total) - float(num_pos)) * float(cfg.bg_cls_weight)
    t = torch.tensor([cls_avg if cfg.sync_cls_avg_factor else 0.0, float(num_pos)], dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t /= world
    if not cfg.sync_cls_avg_factor:
        t[0] = cls_avg                      # stays this rank's own value
    return t


_last_flag = None
_last_cost = None


def loss_label_errors():
    """True if the most recent ``PETRHead.loss`` call saw a ground-truth label outside [0, num_classes) (the kernels
    guard such labels instead of indexing with them; reading the flag synchronises with the device, so it is a
    separate, optional call - e.g. once per epoch or in a debugging run)."""
    if _last_flag is None:
        return False
    return bool(_last_flag.view(torch.int32)[0].item() != 0)


class _LossFn(torch.autograd.Function):
    """[NL, 2] losses; the kernels already produced the gradients of their sum, backward only weights the level
    slices by the incoming gradient."""

    @staticmethod
    def forward(ctx, cls, box, gt_boxes, gt_labels, gt_offsets, counts, cfg, avg=None):
        L = _C.lib()
        NL, B, Q, NC = cls.shape
        CS = box.shape[-1]
        cls, box = cls.contiguous().float(), box.contiguous().float()
        gtot, gmax = int(sum(counts)), int(max(counts) if counts else 0)
        losses = torch.empty((NL, 2), dtype=torch.float32, device=cls.device)
        d_cls, d_box = torch.empty_like(cls), torch.empty_like(box)
        assigned = torch.empty((NL, B, Q), dtype=torch.int32, device=cls.device)
        nbytes = L.petr_loss_workspace_bytes(NL, B, Q, gtot)
        ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=cls.device)
        a = _C.LossArgs()
        a.cls, a.box = cls.data_ptr(), box.data_ptr()
        a.gt_boxes = gt_boxes.data_ptr() if gtot else None
        a.gt_labels = gt_labels.data_ptr() if gtot else None
        offs = (C.c_int * (B + 1))(*gt_offsets)
        a.gt_offsets = C.cast(offs, C.c_void_p)
        a.NL, a.B, a.Q, a.NC, a.CS, a.Gtot, a.Gmax = NL, B, Q, NC, CS, gtot, gmax
        a.num_pos = int(sum(min(c, Q) for c in counts))
        a.cls_weight, a.bbox_weight, a.alpha, a.gamma = cfg.cls_weight, cfg.bbox_weight, cfg.alpha, cfg.gamma
        a.bg_cls_weight = cfg.bg_cls_weight
        a.code_weights = (C.c_float * 10)(*cfg.code_weights[:10])
        a.losses, a.d_cls, a.d_box, a.assigned = losses.data_ptr(), d_cls.data_ptr(), d_box.data_ptr(), assigned.data_ptr()
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 8
        a.avg_factors = avg.data_ptr() if avg is not None else None
        _C.check(L.petr_loss_fwd_bwd(C.byref(a), _stream()), 'petr_loss_fwd_bwd')
        # the label-error flag word sits behind the cost matrix and the level sums (petr_hip.h); kept for loss_label_errors()
        global _last_flag, _last_cost
        _last_cost = ws[:NL * max(gtot, 1) * Q].view(NL, max(gtot, 1), Q)      # the kernels' own cost matrix (tests)
        _last_flag = ws[NL * max(gtot, 1) * Q + 2 * NL:NL * max(gtot, 1) * Q + 2 * NL + 1]
        ctx.save_for_backward(d_cls, d_box)
        ctx.mark_non_differentiable(assigned)
        ctx.n_levels = NL
        # 2*NL separate scalar outputs (what the reference's dict holds): the caller's sum / weighting then costs one
        # tiny autograd node per entry it touches instead of a select + select_backward (a zero-filled [NL, 2]
        # tensor each) per entry
        return (assigned,) + losses.flatten().unbind(0)

    @staticmethod
    def backward(ctx, _g_assigned, *g_losses):
        d_cls, d_box = ctx.saved_tensors
        if all(g is None for g in g_losses):
            return (None,) * 8
        zero = d_cls.new_zeros(())
        g = torch.stack([zero if x is None else x.to(d_cls.dtype) for x in g_losses]).view(ctx.n_levels, 2)
        return d_cls * g[:, 0].view(-1, 1, 1, 1), d_box * g[:, 1].view(-1, 1, 1, 1), None, None, None, None, None, None


def _gt_tensor(boxes, device):
    """petr_head.py:697-699: ``cat(gravity_center, tensor[:, 3:])`` for box objects; plain [G, 9] tensors pass."""
    if hasattr(boxes, 'gravity_center') and hasattr(boxes, 'tensor'):
        boxes = torch.cat((boxes.gravity_center, boxes.tensor[:, 3:]), dim=1)
    return boxes.to(device=device, dtype=torch.float32)


def head_loss(cfg, gt_bboxes_list, gt_labels_list, preds_dicts, return_assignment=False, group=None, avg_factors=None):
    """``PETRHead.loss`` (petr_head.py:646-728): dict with 'loss_cls', 'loss_bbox' (last level) and
    'd{i}.loss_cls', 'd{i}.loss_bbox' (earlier levels).  With an initialised process group of more than one rank the
    normalisers are averaged over ``group`` as the reference does (``synced_avg_factors``); ``avg_factors`` overrides
    them with a caller-supplied device tensor ``[cls_avg_factor, num_total_pos]``."""
    all_cls, all_box = preds_dicts['all_cls_scores'], preds_dicts['all_bbox_preds']
    assert preds_dicts.get('enc_cls_scores') is None, 'two-stage (enc_*) outputs are not produced by PETRHead'
    if not all_cls.is_cuda:
        raise _C.PetrHipError('PETRHead.loss (petr_amd) runs on the GPU only; there is no CPU fallback')
    dev = all_cls.device
    B = all_cls.shape[1]
    assert len(gt_bboxes_list) == B and len(gt_labels_list) == B
    boxes = [_gt_tensor(b, dev) for b in gt_bboxes_list]
    counts = [int(b.shape[0]) for b in boxes]
    for b in boxes:
        assert b.dim() == 2 and b.shape[1] == 9, 'ground-truth boxes are [G, 9] (centre, dims, yaw, vx, vy)'
    gt_boxes = torch.cat(boxes, 0).contiguous() if sum(counts) else torch.zeros((0, 9), device=dev)
    gt_labels = torch.cat([t.to(dev).long() for t in gt_labels_list], 0).contiguous() if sum(counts) else \
        torch.zeros((0,), dtype=torch.long, device=dev)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    gt_offsets = offs            # host metadata: goes into the kernel arguments, no H2D copy
    Q = all_cls.shape[2]
    if avg_factors is None:
        avg_factors = synced_avg_factors(sum(min(c, Q) for c in counts), B * Q, cfg, dev, group)
    res = _LossFn.apply(all_cls, all_box, gt_boxes, gt_labels, gt_offsets, counts, cfg, avg_factors)
    assigned, flat = res[0], res[1:]          # flat[2*l] = loss_cls of level l, flat[2*l+1] = loss_bbox
    out = {}
    n = all_cls.shape[0]
    out['loss_cls'], out['loss_bbox'] = flat[2 * (n - 1)], flat[2 * (n - 1) + 1]
    for i in range(n - 1):
        out[f'd{i}.loss_cls'], out[f'd{i}.loss_bbox'] = flat[2 * i], flat[2 * i + 1]
    return (out, assigned) if return_assignment else out


class NMSFreeCoder:
    """core/bbox/coders/nms_free_coder.py:16-120."""

    def __init__(self, pc_range, voxel_size=None, post_center_range=None, max_num=100, score_threshold=None,
                 num_classes=10, **kwargs):
        self.pc_range, self.voxel_size, self.post_center_range = pc_range, voxel_size, post_center_range
        self.max_num, self.score_threshold, self.num_classes = max_num, score_threshold, num_classes

    def encode(self):
        pass

    def _decode_batch(self, cls, box, bottom_center):
        """[B, Q, NC] logits + [B, Q, CS] boxes -> per-sample dicts, ONE launch (petr_decode_topk): sigmoid, top-``max_num``
        selection, gather, box denormalisation and the range filter run on the device; torch only allocates the outputs and
        applies the keep mask (a boolean index: the number of kept boxes is data dependent, as in the reference).
        Order among equal scores: by logit descending, then flattened (query, class) index ascending - the reference's
        ``sigmoid().topk()`` leaves the order of equal fp32 scores to the implementation."""
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only support post_center_range is not None for now!')
        L = _C.lib()
        B, Q, NC = cls.shape
        k = int(self.max_num)
        cls, box = cls.contiguous().float(), box.contiguous().float()
        dev = cls.device
        boxes = torch.empty((B, k, 9), dtype=torch.float32, device=dev)
        scores = torch.empty((B, k), dtype=torch.float32, device=dev)
        labels = torch.empty((B, k), dtype=torch.long, device=dev)
        index = torch.empty((B, k), dtype=torch.long, device=dev)
        keep = torch.empty((B, k), dtype=torch.uint8, device=dev)
        thr = float(self.score_threshold) if self.score_threshold is not None else -1.0      # negative: no score filter (None)
        a = _C.DecodeTopkArgs(_ptr(cls), _ptr(box), _ptr(boxes), _ptr(scores), _ptr(labels), _ptr(keep), _ptr(index), B, Q, NC,
                              box.shape[-1], k, (C.c_float * 6)(*[float(v) for v in self.post_center_range]), thr, int(bottom_center))
        _C.check(L.petr_decode_topk(C.byref(a), _stream()), 'petr_decode_topk')
        self._last_index = index
        out = []
        for i in range(B):
            m = keep[i].bool()
            out.append({'bboxes': boxes[i][m], 'scores': scores[i][m], 'labels': labels[i][m]})
        return out

    def decode_single(self, cls_scores, bbox_preds, bottom_center=False):
        """nms_free_coder.py:48-97 for one sample: the same device path as ``decode`` with a batch of one."""
        return self._decode_batch(cls_scores[None], bbox_preds[None], bottom_center)[0]

    def decode(self, preds_dicts, bottom_center=False):
        """nms_free_coder.py:99-120: the last decoder level of every sample."""
        return self._decode_batch(preds_dicts['all_cls_scores'][-1], preds_dicts['all_bbox_preds'][-1], bottom_center)


def get_bboxes(coder, preds_dicts, img_metas, rescale=False):
    """``PETRHead.get_bboxes`` (petr_head.py:730-751): [boxes, scores, labels] per sample, boxes wrapped by
    ``img_metas[i]['box_type_3d']`` when the caller provides it (mmdet3d's LiDARInstance3DBoxes)."""
    ret = []
    for i, p in enumerate(coder.decode(preds_dicts, bottom_center=True)):
        boxes = p['bboxes']
        wrap = img_metas[i].get('box_type_3d') if img_metas is not None and i < len(img_metas) else None
        if wrap is not None:
            boxes = wrap(boxes, boxes.size(-1))
        ret.append([boxes, p['scores'], p['labels']])
    return ret
