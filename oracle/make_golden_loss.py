"""Generate tests/golden/loss_*.npz / decode_*.npz from the reference source.  BUILD-CONTAINER ONLY
(same rules as oracle/make_golden.py: /root/reference is read in place, nothing of it is copied).

What executes as the reference wrote it (loaded by path, third-party imports replaced by inert stand-ins):
``PETRHead.loss / loss_single / get_targets / _get_target_single / get_bboxes`` (petr_head.py:470-751),
``HungarianAssigner3D.assign``, ``BBox3DL1Cost``, ``normalize_bbox`` / ``denormalize_bbox`` (source text of
core/bbox/util.py:38-87), ``NMSFreeCoder.decode``.  What is a restatement of un-vendored mmdet 2.24.1 (parity
unpinned): FocalLoss, L1Loss, FocalLossCost, PseudoSampler/SamplingResult, AssignResult, multi_apply, reduce_mean —
they are taken from oracle/loss_oracle.py so that the reference code and the oracle see the same third-party math.

For each case: run the reference, run oracle/loss_oracle.py, assert they agree, store inputs + reference outputs.
Usage:  python oracle/make_golden_loss.py
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import loss_oracle as LO  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

REF, OUT = MG.REF, MG.OUT


# ---- restated mmdet containers (no arithmetic) ----
class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class BaseAssigner:
    pass


class BaseBBoxCoder:
    def __init__(self, **kw):
        pass


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds.long(), :] if gt_bboxes.numel() else gt_bboxes.view(-1, 9)


class PseudoSampler:
    def sample(self, assign_result, bboxes, gt_bboxes, **kw):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result)


class FocalLossCost:
    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        return LO.focal_loss_cost(cls_pred, gt_labels, self.weight, self.alpha, self.gamma, self.eps)


class FocalLoss:
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0, **kw):
        self.use_sigmoid, self.gamma, self.alpha, self.loss_weight = use_sigmoid, gamma, alpha, loss_weight

    def __call__(self, pred, target, weight=None, avg_factor=None):
        return LO.sigmoid_focal_loss(pred, target, weight, self.gamma, self.alpha, avg_factor, self.loss_weight)


class L1Loss:
    def __init__(self, loss_weight=1.0, **kw):
        self.loss_weight = loss_weight

    def __call__(self, pred, target, weight=None, avg_factor=None):
        return LO.l1_loss(pred, target, weight, avg_factor, self.loss_weight)


def multi_apply(func, *args, **kwargs):
    results = map(lambda *a: func(*a, **kwargs), *args)
    return tuple(map(list, zip(*results)))


def reference_util_functions():
    """normalize_bbox / denormalize_bbox from the source text of core/bbox/util.py (the top of the file needs a
    package-relative import that does not resolve outside the plugin package)."""
    src = open(os.path.join(REF, 'core/bbox/util.py')).read()
    ns = {'torch': torch}
    exec(compile(src[src.index('def normalize_bbox'):], 'util.py', 'exec'), ns)
    return ns['normalize_bbox'], ns['denormalize_bbox']


def main():
    warnings.simplefilter('ignore')
    torch.set_num_threads(8)
    MG.install_stubs()
    sys.modules['mmdet.models.utils.transformer'].inverse_sigmoid = MG.inverse_sigmoid_from_reference()
    ref_norm, ref_denorm = reference_util_functions()
    match_costs = MG.Registry('MATCH_COST')
    assigners, coders = MG.Registry('BBOX_ASSIGNERS'), MG.Registry('BBOX_CODERS')
    match_costs.table['FocalLossCost'] = FocalLossCost
    match_costs.table['IoUCost'] = lambda weight=0.0, **k: None

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    mod('mmdet.core.bbox', BaseBBoxCoder=BaseBBoxCoder)
    mod('mmdet.core.bbox.builder', BBOX_ASSIGNERS=assigners, BBOX_CODERS=coders)
    mod('mmdet.core.bbox.assigners', AssignResult=AssignResult, BaseAssigner=BaseAssigner)
    mod('mmdet.core.bbox.match_costs', build_match_cost=match_costs.build)
    mod('mmdet.core.bbox.match_costs.builder', MATCH_COST=match_costs)
    mod('mmdet.core.bbox.iou_calculators', bbox_overlaps=None)
    sys.modules['projects.mmdet3d_plugin.core.bbox.util'].normalize_bbox = ref_norm
    sys.modules['projects.mmdet3d_plugin.core.bbox.util'].denormalize_bbox = ref_denorm
    sys.modules['mmdet.core'].multi_apply = multi_apply
    sys.modules['mmdet.core'].reduce_mean = lambda t: t
    MG.load_by_path('ref_match_cost', 'core/bbox/match_costs/match_cost.py')
    asg_mod = MG.load_by_path('ref_assigner', 'core/bbox/assigners/hungarian_assigner_3d.py')
    coder_mod = MG.load_by_path('ref_coder', 'core/bbox/coders/nms_free_coder.py')
    MG.load_by_path('ref_positional_encoding', 'models/utils/positional_encoding.py')
    tr_mod = MG.load_by_path('ref_petr_transformer', 'models/utils/petr_transformer.py')
    MG.REG['ATTENTION'].table['MultiheadAttention'] = tr_mod.PETRMultiheadAttention
    head_mod = MG.load_by_path('ref_petr_head', 'models/dense_heads/petr_head.py')
    RefHead = head_mod.PETRHead
    pcr = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]

    class LossHead:      # the reference's methods on a bare attribute holder (no network needed for these steps)
        loss, loss_single = RefHead.loss, RefHead.loss_single
        get_targets, _get_target_single, get_bboxes = RefHead.get_targets, RefHead._get_target_single, RefHead.get_bboxes

    ref = LossHead()
    ref.num_classes = ref.cls_out_channels = 10
    ref.bg_cls_weight, ref.sync_cls_avg_factor, ref.pc_range = 0, False, pcr
    ref.code_weights = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2])
    ref.assigner = asg_mod.HungarianAssigner3D(cls_cost=dict(type='FocalLossCost', weight=2.0),
                                               reg_cost=dict(type='BBox3DL1Cost', weight=0.25),
                                               iou_cost=dict(type='IoUCost', weight=0.0), pc_range=pcr)
    ref.sampler = PseudoSampler()
    ref.loss_cls = FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0)
    ref.loss_bbox = L1Loss(loss_weight=0.25)
    ref.bbox_coder = coder_mod.NMSFreeCoder(pc_range=pcr, post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                            max_num=300, voxel_size=[0.2, 0.2, 8], num_classes=10)
    cfg = LO.LossCfg()

    for name, B, Q, n_gt, seed in [('loss_toy', 2, 40, [7, 0], 3), ('loss_q900', 1, 900, [37], 4)]:
        g = torch.Generator().manual_seed(seed)
        cls = (torch.randn(6, B, Q, 10, generator=g) * 2 - 2).requires_grad_(True)
        box = torch.randn(6, B, Q, 10, generator=g).requires_grad_(True)
        gt_boxes, gt_labels = LO.synthetic_gt(B, n_gt, seed=seed)
        # the reference wants objects with .gravity_center / .tensor (LiDARInstance3DBoxes): bottom-centre tensor
        objs = []
        for t in gt_boxes:
            bottom = t.clone()
            bottom[:, 2] -= t[:, 5] * 0.5
            objs.append(types.SimpleNamespace(gravity_center=t[:, :3].clone(), tensor=bottom))
        preds = {'all_cls_scores': cls, 'all_bbox_preds': box, 'enc_cls_scores': None, 'enc_bbox_preds': None}
        want = ref.loss(objs, gt_labels, preds)
        total = sum(v for v in want.values())
        d_cls, d_box = torch.autograd.grad(total, [cls, box])
        got, assigns = LO.head_loss(cfg, gt_boxes, gt_labels, {'all_cls_scores': cls.detach(), 'all_bbox_preds': box.detach()})
        assert set(got) == set(want), (sorted(got), sorted(want))
        for k in want:
            MG.close(got[k], want[k].detach(), 1e-6 * max(1.0, want[k].abs().item()), f'{name}:{k}')
        keys = sorted(want)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), cls=cls.detach().numpy(), box=box.detach().numpy(),
                            gt_boxes=np.concatenate([t.numpy() for t in gt_boxes]).astype(np.float32),
                            gt_labels=np.concatenate([t.numpy() for t in gt_labels]).astype(np.int64),
                            gt_counts=np.array([t.shape[0] for t in gt_boxes], dtype=np.int64),
                            loss_keys=np.array(keys), loss_values=np.array([want[k].item() for k in keys], dtype=np.float64),
                            d_cls=d_cls.numpy(), d_box=d_box.numpy(), assigned=assigns.numpy())
        print(name, {k: round(want[k].item(), 5) for k in keys[:4]}, 'positives', int((assigns > 0).sum()))

    # ---- decode / get_bboxes ----
    g = torch.Generator().manual_seed(9)
    cls = torch.randn(6, 2, 900, 10, generator=g) * 2 - 3
    box = torch.randn(6, 2, 900, 10, generator=g)
    box[..., 0:2] *= 40
    box[..., 4] *= 6                                   # some centres fall outside post_center_range
    metas = [{'box_type_3d': lambda t, dim: t} for _ in range(2)]
    want = ref.get_bboxes({'all_cls_scores': cls.clone(), 'all_bbox_preds': box.clone()}, metas)
    got = LO.get_bboxes(cfg, {'all_cls_scores': cls, 'all_bbox_preds': box})
    store = dict(cls=cls.numpy(), box=box.numpy())
    for i, (w, o) in enumerate(zip(want, got)):
        for j, nm in enumerate(['bboxes', 'scores', 'labels']):
            MG.close(o[j], w[j], 1e-6 if j < 2 else 0, f'decode[{i}].{nm}')
            store[f'{nm}{i}'] = w[j].numpy()
    np.savez_compressed(os.path.join(OUT, 'decode_q900.npz'), **store)
    print('decode_q900 kept', [w[0].shape[0] for w in want])
    # normalize/denormalize known answers
    t = LO.synthetic_gt(1, 5, seed=1)[0][0]
    MG.close(LO.normalize_bbox(t), ref_norm(t, pcr), 0, 'normalize_bbox')
    MG.close(LO.denormalize_bbox(LO.normalize_bbox(t)), ref_denorm(ref_norm(t, pcr), pcr), 0, 'denormalize_bbox')
    print('loss/decode fixtures written to', OUT)


if __name__ == '__main__':
    main()
