"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the two steps right behind the hot path:
the training loss with Hungarian matching and the NMS-free decode.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this file.

Plain PyTorch restatement of

* reference-owned code (pinned by ``oracle/make_golden_loss.py``, which runs the reference's own source on the same
  inputs): ``PETRHead.loss / loss_single / get_targets / _get_target_single / get_bboxes``
  (projects/mmdet3d_plugin/models/dense_heads/petr_head.py:470-751), ``HungarianAssigner3D.assign``
  (core/bbox/assigners/hungarian_assigner_3d.py:61-143), ``BBox3DL1Cost`` (core/bbox/match_costs/match_cost.py:6-27),
  ``normalize_bbox / denormalize_bbox`` (core/bbox/util.py:38-87), ``NMSFreeCoder.decode`` (core/bbox/coders/
  nms_free_coder.py:48-120);
* third-party code that is NOT vendored in the reference and is restated from the published mmdet 2.24.1
  (requirements.txt:14) — **parity unpinned**, no reference artefact covers it: ``FocalLoss`` / ``py_sigmoid_focal_loss``,
  ``L1Loss``, ``weight_reduce_loss`` (sum / (avg_factor + eps)), ``FocalLossCost``, ``PseudoSampler``, ``multi_apply``,
  ``reduce_mean`` (identity on one process).
"""
import torch
import torch.nn.functional as F

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:  # pragma: no cover
    linear_sum_assignment = None


# --------------------------------------------------------------------------- reference: core/bbox/util.py
def normalize_bbox(bboxes, pc_range=None):
    """util.py:38-58: (cx, cy, cz, w, l, h, rot, vx, vy) -> (cx, cy, log w, log l, cz, log h, sin, cos, vx, vy)."""
    cx, cy, cz = bboxes[..., 0:1], bboxes[..., 1:2], bboxes[..., 2:3]
    w, l, h = bboxes[..., 3:4].log(), bboxes[..., 4:5].log(), bboxes[..., 5:6].log()
    rot = bboxes[..., 6:7]
    if bboxes.size(-1) > 7:
        return torch.cat((cx, cy, w, l, cz, h, rot.sin(), rot.cos(), bboxes[..., 7:8], bboxes[..., 8:9]), dim=-1)
    return torch.cat((cx, cy, w, l, cz, h, rot.sin(), rot.cos()), dim=-1)


def denormalize_bbox(nb, pc_range=None):
    """util.py:60-87."""
    rot = torch.atan2(nb[..., 6:7], nb[..., 7:8])
    cx, cy, cz = nb[..., 0:1], nb[..., 1:2], nb[..., 4:5]
    w, l, h = nb[..., 2:3].exp(), nb[..., 3:4].exp(), nb[..., 5:6].exp()
    if nb.size(-1) > 8:
        return torch.cat([cx, cy, cz, w, l, h, rot, nb[:, 8:9], nb[:, 9:10]], dim=-1)
    return torch.cat([cx, cy, cz, w, l, h, rot], dim=-1)


# --------------------------------------------------------------------------- mmdet 2.24.1 (restated, unpinned)
def weight_reduce_loss(loss, weight=None, avg_factor=None):
    """mmdet/models/losses/utils.py (reduction='mean' with avg_factor): sum / (avg_factor + eps)."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean()
    return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)


def sigmoid_focal_loss(pred, labels, weight=None, gamma=2.0, alpha=0.25, avg_factor=None, loss_weight=1.0):
    """mmdet FocalLoss(use_sigmoid=True).forward -> py_sigmoid_focal_loss; labels == num_classes is background."""
    num_classes = pred.size(1)
    target = F.one_hot(labels, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * target + p * (1 - target)
    focal_weight = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction='none') * focal_weight
    if weight is not None and weight.shape != loss.shape:
        weight = weight.view(-1, 1)
    return loss_weight * weight_reduce_loss(loss, weight, avg_factor)


def l1_loss(pred, target, weight=None, avg_factor=None, loss_weight=1.0):
    """mmdet L1Loss.forward (reduction='mean')."""
    if target.numel() == 0:
        return pred.sum() * 0
    return loss_weight * weight_reduce_loss((pred - target).abs(), weight, avg_factor)


def focal_loss_cost(cls_pred, gt_labels, weight=1.0, alpha=0.25, gamma=2.0, eps=1e-12):
    """mmdet FocalLossCost.__call__ (logits in)."""
    p = cls_pred.sigmoid()
    neg_cost = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos_cost = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    return (pos_cost[:, gt_labels] - neg_cost[:, gt_labels]) * weight


# --------------------------------------------------------------------------- reference: match cost + assigner
def bbox3d_l1_cost(bbox_pred, gt_bboxes, weight=1.0):
    """match_cost.py:15-27."""
    return torch.cdist(bbox_pred, gt_bboxes, p=1) * weight


def hungarian_assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, cls_weight=2.0, reg_weight=0.25):
    """hungarian_assigner_3d.py:61-143.  Returns assigned_gt_inds [num_query] (0 = background, k = gt k-1)."""
    num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
    assigned = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
    if num_gts == 0 or num_bboxes == 0:
        if num_gts == 0:
            assigned[:] = 0
        return assigned
    cost = focal_loss_cost(cls_pred, gt_labels, cls_weight) + \
        bbox3d_l1_cost(bbox_pred[:, :8], normalize_bbox(gt_bboxes)[:, :8], reg_weight)
    cost = torch.nan_to_num(cost.detach().cpu(), nan=100.0, posinf=100.0, neginf=-100.0)
    rows, cols = linear_sum_assignment(cost)
    assigned[:] = 0
    assigned[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
    return assigned


# --------------------------------------------------------------------------- reference: PETRHead loss / decode
class LossCfg:
    """The hyper-parameters of configs/petr/petr_r50dcn_gridmask_c5.py:45-110."""

    def __init__(self, num_classes=10, code_weights=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2),
                 cls_weight=2.0, bbox_weight=0.25, alpha=0.25, gamma=2.0, bg_cls_weight=0.0,
                 pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
                 post_center_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0), max_num=300, score_threshold=None):
        self.num_classes, self.code_weights = num_classes, torch.tensor(code_weights)
        self.cls_weight, self.bbox_weight, self.alpha, self.gamma = cls_weight, bbox_weight, alpha, gamma
        self.bg_cls_weight, self.pc_range = bg_cls_weight, list(pc_range)
        self.post_center_range, self.max_num, self.score_threshold = list(post_center_range), max_num, score_threshold


def get_target_single(cfg, cls_score, bbox_pred, gt_labels, gt_bboxes):
    """petr_head.py:470-524 (PseudoSampler: positives = assigned queries)."""
    num_bboxes = bbox_pred.size(0)
    assigned = hungarian_assign(bbox_pred, cls_score, gt_bboxes, gt_labels, cfg.cls_weight, cfg.bbox_weight)
    pos_inds = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
    neg_inds = torch.nonzero(assigned == 0, as_tuple=False).squeeze(-1).unique()
    pos_gt = assigned[pos_inds] - 1
    labels = gt_bboxes.new_full((num_bboxes,), cfg.num_classes, dtype=torch.long)
    labels[pos_inds] = gt_labels[pos_gt]
    label_weights = gt_bboxes.new_ones(num_bboxes)
    code_size = gt_bboxes.size(1)
    bbox_targets = torch.zeros_like(bbox_pred)[..., :code_size]
    bbox_weights = torch.zeros_like(bbox_pred)
    bbox_weights[pos_inds] = 1.0
    bbox_targets[pos_inds] = gt_bboxes[pos_gt]
    return labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds, assigned


def loss_single(cfg, cls_scores, bbox_preds, gt_bboxes_list, gt_labels_list, reduce_mean=None):
    """petr_head.py:573-644 for one decoder level; also returns the assignments [B, num_query].
    ``reduce_mean``: stand-in for mmdet's cross-rank mean (a callable float -> float); None = one process (identity).
    ``cfg.sync_cls_avg_factor`` (default False) as in the reference (:620-622)."""
    num_imgs = cls_scores.size(0)
    tg = [get_target_single(cfg, cls_scores[i], bbox_preds[i], gt_labels_list[i], gt_bboxes_list[i]) for i in range(num_imgs)]
    labels = torch.cat([t[0] for t in tg], 0)
    label_weights = torch.cat([t[1] for t in tg], 0)
    bbox_targets = torch.cat([t[2] for t in tg], 0)
    bbox_weights = torch.cat([t[3] for t in tg], 0)
    num_total_pos = sum(t[4].numel() for t in tg)
    num_total_neg = sum(t[5].numel() for t in tg)
    cls_scores = cls_scores.reshape(-1, cls_scores.size(-1))
    reduce_mean = reduce_mean or (lambda v: v)
    cls_avg_factor = num_total_pos * 1.0 + num_total_neg * cfg.bg_cls_weight
    if getattr(cfg, 'sync_cls_avg_factor', False):
        cls_avg_factor = reduce_mean(cls_avg_factor)                          # :620-622
    cls_avg_factor = max(cls_avg_factor, 1)
    loss_cls = sigmoid_focal_loss(cls_scores, labels, label_weights, cfg.gamma, cfg.alpha, cls_avg_factor, cfg.cls_weight)
    num_total_pos = max(float(reduce_mean(float(num_total_pos))), 1.0)        # clamp(reduce_mean(.), min=1).item(), :630-631
    bbox_preds = bbox_preds.reshape(-1, bbox_preds.size(-1))
    normalized = normalize_bbox(bbox_targets)
    isnotnan = torch.isfinite(normalized).all(dim=-1)
    bbox_weights = bbox_weights * cfg.code_weights.to(bbox_weights)
    loss_bbox = l1_loss(bbox_preds[isnotnan, :10], normalized[isnotnan, :10], bbox_weights[isnotnan, :10], num_total_pos,
                        cfg.bbox_weight)
    return torch.nan_to_num(loss_cls), torch.nan_to_num(loss_bbox), torch.stack([t[6] for t in tg])


def head_loss(cfg, gt_bboxes_list, gt_labels_list, preds, reduce_mean=None):
    """petr_head.py:646-728.  ``gt_bboxes_list``: per image [G, 9] (gravity centre, dims, yaw, vx, vy), i.e. what
    ``torch.cat((boxes.gravity_center, boxes.tensor[:, 3:]), 1)`` yields (:697-699)."""
    all_cls, all_box = preds['all_cls_scores'], preds['all_bbox_preds']
    out, assigns = {}, []
    n = len(all_cls)
    for lvl in range(n):
        lc, lb, a = loss_single(cfg, all_cls[lvl], all_box[lvl], gt_bboxes_list, gt_labels_list, reduce_mean)
        assigns.append(a)
        key = '' if lvl == n - 1 else f'd{lvl}.'
        out[key + 'loss_cls'], out[key + 'loss_bbox'] = lc, lb
    return out, torch.stack(assigns)


def decode_single(cfg, cls_scores, bbox_preds):
    """nms_free_coder.py:48-97."""
    scores, idx = cls_scores.sigmoid().view(-1).topk(cfg.max_num)
    labels = idx % cfg.num_classes
    boxes = denormalize_bbox(bbox_preds[idx // cfg.num_classes])
    rng = torch.tensor(cfg.post_center_range, device=scores.device)
    mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
    if cfg.score_threshold:
        mask &= scores > cfg.score_threshold
    return {'bboxes': boxes[mask], 'scores': scores[mask], 'labels': labels[mask]}


def get_bboxes(cfg, preds, img_metas=None):
    """petr_head.py:730-751: last level, decode, z = gravity centre -> bottom centre; returns [boxes, scores, labels]
    per sample (the box_type_3d wrapper of the reference is the caller's)."""
    out = []
    cls, box = preds['all_cls_scores'][-1], preds['all_bbox_preds'][-1]
    for i in range(cls.size(0)):
        p = decode_single(cfg, cls[i], box[i])
        b = p['bboxes'].clone()
        b[:, 2] = b[:, 2] - b[:, 5] * 0.5
        out.append([b, p['scores'], p['labels']])
    return out


def synthetic_gt(batch, n_gt, seed=0, num_classes=10):
    """Ground truth of the shape nuScenes gives: [G, 9] gravity-centre boxes inside the point-cloud range."""
    g = torch.Generator().manual_seed(seed)
    boxes, labels = [], []
    for b in range(batch):
        n = n_gt[b] if isinstance(n_gt, (list, tuple)) else n_gt
        c = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([90.0, 90.0, 6.0])
        dims = torch.rand(n, 3, generator=g) * torch.tensor([3.0, 8.0, 2.5]) + 0.4
        yaw = (torch.rand(n, 1, generator=g) - 0.5) * 6.283
        vel = torch.randn(n, 2, generator=g)
        boxes.append(torch.cat([c, dims, yaw, vel], 1))
        labels.append(torch.randint(0, num_classes, (n,), generator=g))
    return boxes, labels
