"""CPU oracle for the PETRHead hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the arithmetic of the
reference's PETRHead forward path.  It exists so that the HIP path can be
checked against it; it is never imported by the product package
(``petr_amd``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

``torch.float32`` / ``.float()`` of the reference are written ``torch.get_default_dtype()`` here: with
the default dtype (float32) the arithmetic is the reference's bit for bit (tests/test_oracle_golden.py);
under ``torch.set_default_dtype(torch.float64)`` the same code is a float64 yardstick used by the
gradient tests to tell fp32 rounding noise from real disagreement.

Every function cites the reference lines it follows (paths relative to
``/root/reference/projects/mmdet3d_plugin``).  Pinning status:

* ``inverse_sigmoid``, ``pos2posemb3d``, ``sine_positional_encoding_3d``,
  ``coords3d_volume``/``position_embeding`` and ``PETRMultiheadAttention`` are
  checked function-by-function against the reference source itself (loaded by
  path with inert mmcv/mmdet stubs) by ``oracle/make_golden.py``; the vectors
  it emits live in ``tests/golden``.
* The mmcv-1.4.0 pieces that are not vendored in the reference
  (``BaseTransformerLayer`` dispatch loop, ``FFN``, ``TransformerLayerSequence``)
  are restated from their published behaviour; the dispatch loop is pinned by
  the in-tree copy ``models/utils/multi_atten_decoder_layer.py:204-293``, the
  inside of ``FFN`` is "parity unpinned" (no reference test or fixture covers
  it; the reference has no tests at all).
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# elementary functions
# --------------------------------------------------------------------------- #
def inverse_sigmoid(x, eps=1e-5):
    """models/utils/detr.py:15-30 (in-tree copy of mmdet's inverse_sigmoid)."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def pos2posemb3d(pos, num_pos_feats=128, temperature=10000):
    """models/dense_heads/petr_head.py:31-43.  Output order (y, x, z)."""
    scale = 2 * math.pi
    pos = pos * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.get_default_dtype(), device=pos.device)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = pos[..., 0, None] / dim_t
    pos_y = pos[..., 1, None] / dim_t
    pos_z = pos[..., 2, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=-1).flatten(-2)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=-1).flatten(-2)
    pos_z = torch.stack((pos_z[..., 0::2].sin(), pos_z[..., 1::2].cos()), dim=-1).flatten(-2)
    return torch.cat((pos_y, pos_x, pos_z), dim=-1)


def sine_dim_t(num_feats=128, temperature=10000):
    """models/utils/positional_encoding.py:82-84 (and petr_head.py:34-35)."""
    dim_t = torch.arange(num_feats, dtype=torch.get_default_dtype())
    return temperature ** (2 * (dim_t // 2) / num_feats)


def sine_positional_encoding_3d(mask, num_feats=128, temperature=10000,
                                normalize=False, scale=2 * math.pi, eps=1e-6,
                                offset=0.):
    """models/utils/positional_encoding.py:58-100.  mask [B,N,H,W] -> [B,N,3F,H,W]."""
    mask = mask.to(torch.int)
    not_mask = 1 - mask
    n_embed = not_mask.cumsum(1, dtype=torch.get_default_dtype())
    y_embed = not_mask.cumsum(2, dtype=torch.get_default_dtype())
    x_embed = not_mask.cumsum(3, dtype=torch.get_default_dtype())
    if normalize:
        n_embed = (n_embed + offset) / (n_embed[:, -1:, :, :] + eps) * scale
        y_embed = (y_embed + offset) / (y_embed[:, :, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.get_default_dtype(), device=mask.device)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_n = n_embed[:, :, :, :, None] / dim_t
    pos_x = x_embed[:, :, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, :, None] / dim_t
    B, N, H, W = mask.size()
    pos_n = torch.stack((pos_n[..., 0::2].sin(), pos_n[..., 1::2].cos()), dim=4).view(B, N, H, W, -1)
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(B, N, H, W, -1)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(B, N, H, W, -1)
    return torch.cat((pos_n, pos_y, pos_x), dim=4).permute(0, 1, 4, 2, 3)


def padding_masks(batch_size, num_cams, img_metas, feat_hw, dtype=torch.get_default_dtype()):
    """models/dense_heads/petr_head.py:383-394: ones, zero the valid region, nearest-resize, to bool."""
    input_img_h, input_img_w, _ = img_metas[0]['pad_shape'][0]
    masks = torch.ones((batch_size, num_cams, input_img_h, input_img_w), dtype=dtype)
    for img_id in range(batch_size):
        for cam_id in range(num_cams):
            img_h, img_w, _ = img_metas[img_id]['img_shape'][cam_id]
            masks[img_id, cam_id, :img_h, :img_w] = 0
    return F.interpolate(masks, size=feat_hw).to(torch.bool)


def depth_bins(depth_num, depth_start, position_range, LID):
    """models/dense_heads/petr_head.py:293-301."""
    index = torch.arange(start=0, end=depth_num, step=1).to(torch.get_default_dtype())
    if LID:
        index_1 = index + 1
        bin_size = (position_range[3] - depth_start) / (depth_num * (1 + depth_num))
        return depth_start + bin_size * index * index_1
    bin_size = (position_range[3] - depth_start) / depth_num
    return depth_start + bin_size * index


def img2lidar_matrices(img_metas):
    """models/dense_heads/petr_head.py:308-315: fp64 numpy inverse per view, then fp32."""
    out = []
    for img_meta in img_metas:
        out.append(np.asarray([np.linalg.inv(m) for m in img_meta['lidar2img']]))
    return torch.tensor(np.asarray(out), dtype=torch.float32).to(torch.get_default_dtype())


def coords3d_volume(B, N, H, W, img_metas, depth_num=64, depth_start=1,
                    position_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0), LID=True,
                    masks=None):
    """models/dense_heads/petr_head.py:286-331 up to (and including) inverse_sigmoid.

    Returns (logit volume [B*N, 3*D, H, W], coords_mask [B,N,H,W] bool,
    normalised coords [B,N,W,H,D,3] before the logit)."""
    eps = 1e-5
    pad_h, pad_w, _ = img_metas[0]['pad_shape'][0]
    coords_h = torch.arange(H).to(torch.get_default_dtype()) * pad_h / H
    coords_w = torch.arange(W).to(torch.get_default_dtype()) * pad_w / W
    coords_d = depth_bins(depth_num, depth_start, position_range, LID)
    D = coords_d.shape[0]
    coords = torch.stack(torch.meshgrid([coords_w, coords_h, coords_d], indexing='ij')).permute(1, 2, 3, 0)
    coords = torch.cat((coords, torch.ones_like(coords[..., :1])), -1)
    coords[..., :2] = coords[..., :2] * torch.maximum(coords[..., 2:3], torch.ones_like(coords[..., 2:3]) * eps)
    img2lidars = img2lidar_matrices(img_metas)
    coords = coords.view(1, 1, W, H, D, 4, 1).repeat(B, N, 1, 1, 1, 1, 1)
    img2lidars = img2lidars.view(B, N, 1, 1, 1, 4, 4).repeat(1, 1, W, H, D, 1, 1)
    coords3d = torch.matmul(img2lidars, coords).squeeze(-1)[..., :3]
    coords3d[..., 0:1] = (coords3d[..., 0:1] - position_range[0]) / (position_range[3] - position_range[0])
    coords3d[..., 1:2] = (coords3d[..., 1:2] - position_range[1]) / (position_range[4] - position_range[1])
    coords3d[..., 2:3] = (coords3d[..., 2:3] - position_range[2]) / (position_range[5] - position_range[2])
    normalised = coords3d.clone()
    coords_mask = (coords3d > 1.0) | (coords3d < 0.0)
    coords_mask = coords_mask.flatten(-2).sum(-1) > (D * 0.5)
    if masks is not None:
        coords_mask = masks | coords_mask.permute(0, 1, 3, 2)
    else:
        coords_mask = coords_mask.permute(0, 1, 3, 2)
    coords3d = coords3d.permute(0, 1, 4, 5, 3, 2).contiguous().view(B * N, -1, H, W)
    return inverse_sigmoid(coords3d), coords_mask, normalised


# --------------------------------------------------------------------------- #
# restated mmcv pieces (third-party, not vendored in the reference)
# --------------------------------------------------------------------------- #
class MultiheadAttentionWrapper(nn.Module):
    """mmcv ``MultiheadAttention`` == models/utils/petr_transformer.py:228-367
    (``PETRMultiheadAttention`` is an in-tree copy of it).  ``dropout`` kwarg maps to
    attn_drop and to dropout_layer.drop_prob (:258-265)."""

    def __init__(self, embed_dims, num_heads, dropout=0.0, batch_first=False):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, dropout)
        self.proj_drop = nn.Dropout(0.0)
        self.dropout_layer = nn.Dropout(dropout)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_pos=None, attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        masks = kwargs.get('masks')
        if masks is not None:
            # training-mode parity: the two dropouts with SUPPLIED keep masks (exported from the HIP kernels'
            # counter-based generator) instead of torch's RNG stream
            keep_p, keep_o, scale = masks
            out = self._attention_with_mask(query, key, value, key_padding_mask, keep_p, scale)
            lq, bs, c = out.shape
            keep_o = keep_o.view(bs, lq, c).permute(1, 0, 2).to(out.dtype)      # kernel rows are b*Q + q
            return identity + out * keep_o * scale
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        return identity + self.dropout_layer(self.proj_drop(out))

    def _attention_with_mask(self, query, key, value, key_padding_mask, keep_p, scale):
        """torch.nn.MultiheadAttention.forward written out (in_proj, 1/sqrt(d) scaling, softmax, dropout on the
        probabilities, out_proj) so that the probability dropout can take a given mask
        ``keep_p [(b*H+h)*Lq + q, key]``."""
        a = self.attn
        lq, bs, e = query.shape
        lk = key.shape[0]
        h, hd = self.num_heads, e // self.num_heads
        w, b = a.in_proj_weight, a.in_proj_bias
        q = F.linear(query, w[:e], b[:e]).reshape(lq, bs * h, hd).transpose(0, 1) * hd ** -0.5
        k = F.linear(key, w[e:2 * e], b[e:2 * e]).reshape(lk, bs * h, hd).transpose(0, 1)
        v = F.linear(value, w[2 * e:], b[2 * e:]).reshape(lk, bs * h, hd).transpose(0, 1)
        att = q @ k.transpose(1, 2)
        if key_padding_mask is not None:
            att = att.view(bs, h, lq, lk).masked_fill(key_padding_mask[:, None, None, :], float('-inf')).view(bs * h, lq, lk)
        att = torch.softmax(att, dim=-1)
        att = att * keep_p.view(bs * h, lq, lk).to(att.dtype) * scale
        out = (att @ v).transpose(0, 1).reshape(lq, bs, e)
        return F.linear(out, a.out_proj.weight, a.out_proj.bias)


class FFN(nn.Module):
    """mmcv 1.4.0 ``FFN`` (parity unpinned: not vendored, no fixture).
    layers = Sequential(Sequential(Linear, ReLU, Dropout), Linear, Dropout); x + layers(x)."""

    def __init__(self, embed_dims=256, feedforward_channels=2048, ffn_drop=0.0):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims),
            nn.Dropout(ffn_drop))

    def forward(self, x, identity=None, masks=None):
        if identity is None:
            identity = x
        if masks is not None:            # supplied keep masks, rows b*Q + q (see MultiheadAttentionWrapper)
            keep_h, keep_o, scale = masks
            lq, bs, _ = x.shape
            hid = torch.relu(self.layers[0][0](x))
            hid = hid * keep_h.view(bs, lq, -1).permute(1, 0, 2).to(hid.dtype) * scale
            out = self.layers[1](hid)
            return identity + out * keep_o.view(bs, lq, -1).permute(1, 0, 2).to(out.dtype) * scale
        out = self.layers(x)
        return identity + out


class DecoderLayer(nn.Module):
    """mmcv ``BaseTransformerLayer`` with operation_order
    ('self_attn','norm','cross_attn','norm','ffn','norm') — dispatch loop pinned by
    models/utils/multi_atten_decoder_layer.py:204-293; ctor layout :72-158;
    PETRTransformerDecoderLayer models/utils/petr_transformer.py:113-224 (checkpointing is
    a memory device only and is not restated)."""

    operation_order = ('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')

    def __init__(self, embed_dims=256, num_heads=8, feedforward_channels=2048, dropout=0.1, ffn_dropout=0.1):
        super().__init__()
        self.attentions = nn.ModuleList([
            MultiheadAttentionWrapper(embed_dims, num_heads, dropout),
            MultiheadAttentionWrapper(embed_dims, num_heads, dropout)])
        self.ffns = nn.ModuleList([FFN(embed_dims, feedforward_channels, ffn_dropout)])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims) for _ in range(3)])
        self.pre_norm = False

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None,
                key_padding_mask=None, query_key_padding_mask=None, masks=None):
        """``masks`` (tests only): dict with the six keep masks of this layer, keys 'sp','so','cp','co','fh','fo',
        and 'scale' = 1/(1-p); None = the modules' own (RNG / eval) dropout."""
        norm_index = attn_index = ffn_index = 0
        identity = query
        m = masks
        for layer in self.operation_order:
            if layer == 'self_attn':
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos, attn_mask=None,
                    key_padding_mask=query_key_padding_mask,
                    masks=None if m is None else (m['sp'], m['so'], m['scale']))
                attn_index += 1
                identity = query
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=key_pos, attn_mask=None,
                    key_padding_mask=key_padding_mask,
                    masks=None if m is None else (m['cp'], m['co'], m['scale']))
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None,
                                             masks=None if m is None else (m['fh'], m['fo'], m['scale']))
                ffn_index += 1
        return query


class Decoder(nn.Module):
    """models/utils/petr_transformer.py:401-447 (return_intermediate=True, shared post_norm)."""

    def __init__(self, num_layers=6, **layer_kw):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayer(**layer_kw) for _ in range(num_layers)])
        self.post_norm = nn.LayerNorm(layer_kw.get('embed_dims', 256))

    def forward(self, query, dropout_masks=None, **kw):
        intermediate = []
        for i, layer in enumerate(self.layers):
            query = layer(query, masks=None if dropout_masks is None else dropout_masks[i], **kw)
            intermediate.append(self.post_norm(query))
        return torch.stack(intermediate)


class Transformer(nn.Module):
    """models/utils/petr_transformer.py:34-109."""

    def __init__(self, num_layers=6, **layer_kw):
        super().__init__()
        self.decoder = Decoder(num_layers, **layer_kw)

    def init_weights(self):
        # petr_transformer.py:62-67 + mmcv xavier_init (weight xavier-uniform gain 1, bias 0)
        for m in self.modules():
            if hasattr(m, 'weight') and m.weight is not None and m.weight.dim() > 1:
                nn.init.xavier_uniform_(m.weight, gain=1)
                if hasattr(m, 'bias') and m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x, mask, query_embed, pos_embed, reg_branch=None, dropout_masks=None):
        bs, n, c, h, w = x.shape
        memory = x.permute(1, 3, 4, 0, 2).reshape(-1, bs, c)
        pos_embed = pos_embed.permute(1, 3, 4, 0, 2).reshape(-1, bs, c)
        query_embed = query_embed.unsqueeze(1).repeat(1, bs, 1)
        mask = mask.view(bs, -1)
        target = torch.zeros_like(query_embed)
        out_dec = self.decoder(query=target, key=memory, value=memory, key_pos=pos_embed,
                               query_pos=query_embed, key_padding_mask=mask, dropout_masks=dropout_masks)
        out_dec = out_dec.transpose(1, 2)
        memory = memory.reshape(n, h, w, bs, c).permute(3, 0, 4, 1, 2)
        return out_dec, memory


# --------------------------------------------------------------------------- #
# PETRv2 extras
# --------------------------------------------------------------------------- #
class SELayer(nn.Module):
    """models/dense_heads/petrv2_head.py:48-60."""

    def __init__(self, channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = nn.ReLU()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = nn.Sigmoid()

    def forward(self, x, x_se):
        x_se = self.conv_expand(self.act1(self.conv_reduce(x_se)))
        return x * self.gate(x_se)


class RegLayer(nn.Module):
    """models/dense_heads/petrv2_head.py:63-95."""

    def __init__(self, embed_dims=256, shared_reg_fcs=2, group_reg_dims=(2, 1, 3, 2, 2)):
        super().__init__()
        reg_branch = []
        for _ in range(shared_reg_fcs):
            reg_branch += [nn.Linear(embed_dims, embed_dims), nn.ReLU(), nn.Dropout(0.0)]
        self.reg_branch = nn.Sequential(*reg_branch)
        self.task_heads = nn.ModuleList([
            nn.Sequential(nn.Linear(embed_dims, embed_dims), nn.ReLU(), nn.Linear(embed_dims, d))
            for d in group_reg_dims])

    def forward(self, x):
        reg_feat = self.reg_branch(x)
        return torch.cat([th(reg_feat.clone()) for th in self.task_heads], -1)


# --------------------------------------------------------------------------- #
# the head
# --------------------------------------------------------------------------- #
class PETRHeadOracle(nn.Module):
    """models/dense_heads/petr_head.py:47-468 (forward path only) and, with
    ``v2=True``, models/dense_heads/petrv2_head.py:99-540 (forward deltas:
    fpe :464-466, with_time :499-505,520-521, RegLayer, deep-copied branches :304-307).

    State-dict keys follow the reference."""

    def __init__(self, num_classes=10, in_channels=256, num_query=900, num_reg_fcs=2,
                 num_layers=6, embed_dims=256, num_heads=8, feedforward_channels=2048,
                 dropout=0.1, with_position=True, with_multiview=True, depth_num=64,
                 LID=True, depth_start=1, position_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0),
                 pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), code_size=10,
                 v2=False, with_fpe=False, with_time=False, with_multi=False,
                 group_reg_dims=(2, 1, 3, 2, 2), code_weights=None):
        super().__init__()
        assert with_position and with_multiview, 'oracle covers the BASELINE configs only'
        self.num_query, self.embed_dims, self.code_size = num_query, embed_dims, code_size
        self.depth_num, self.LID, self.depth_start = depth_num, LID, depth_start
        self.position_range, self.pc_range = list(position_range), list(pc_range)
        self.position_dim = 3 * depth_num
        self.v2, self.with_fpe, self.with_time = v2, with_fpe, with_time
        self.num_pred = num_layers
        # construction order follows the reference (transformer in __init__ :210 before
        # _init_layers :215) so that a seeded init reproduces the reference's draws
        self.transformer = Transformer(num_layers, embed_dims=embed_dims, num_heads=num_heads,
                                       feedforward_channels=feedforward_channels,
                                       dropout=dropout, ffn_dropout=dropout)
        self.input_proj = nn.Conv2d(in_channels, embed_dims, kernel_size=1)
        cls_branch = []
        for _ in range(num_reg_fcs):
            cls_branch += [nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True)]
        cls_branch.append(nn.Linear(embed_dims, num_classes))
        fc_cls = nn.Sequential(*cls_branch)
        if v2 and with_multi:
            reg_branch = RegLayer(embed_dims, num_reg_fcs, group_reg_dims)
        else:
            reg_branch = []
            for _ in range(num_reg_fcs):
                reg_branch += [nn.Linear(embed_dims, embed_dims), nn.ReLU()]
            reg_branch.append(nn.Linear(embed_dims, code_size))
            reg_branch = nn.Sequential(*reg_branch)
        if v2:   # petrv2_head.py:304-307 deep copies
            self.cls_branches = nn.ModuleList([copy.deepcopy(fc_cls) for _ in range(self.num_pred)])
            self.reg_branches = nn.ModuleList([copy.deepcopy(reg_branch) for _ in range(self.num_pred)])
        else:    # petr_head.py:244-247 the SAME module in all six slots
            self.cls_branches = nn.ModuleList([fc_cls for _ in range(self.num_pred)])
            self.reg_branches = nn.ModuleList([reg_branch for _ in range(self.num_pred)])
        self.adapt_pos3d = nn.Sequential(
            nn.Conv2d(embed_dims * 3 // 2, embed_dims * 4, 1), nn.ReLU(), nn.Conv2d(embed_dims * 4, embed_dims, 1))
        self.position_encoder = nn.Sequential(
            nn.Conv2d(self.position_dim, embed_dims * 4, 1), nn.ReLU(), nn.Conv2d(embed_dims * 4, embed_dims, 1))
        self.reference_points = nn.Embedding(num_query, 3)
        self.query_embedding = nn.Sequential(
            nn.Linear(embed_dims * 3 // 2, embed_dims), nn.ReLU(), nn.Linear(embed_dims, embed_dims))
        if v2 and with_fpe:
            self.fpe = SELayer(embed_dims)
        cw = code_weights if code_weights is not None else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]
        self.code_weights = nn.Parameter(torch.tensor(list(cw)[:code_size]), requires_grad=False)

    def init_weights(self):
        """petr_head.py:276-284 (bias_init_with_prob(0.01) = -log((1-p)/p))."""
        self.transformer.init_weights()
        nn.init.uniform_(self.reference_points.weight.data, 0, 1)
        bias_init = float(-np.log((1 - 0.01) / 0.01))
        for m in self.cls_branches:
            nn.init.constant_(m[-1].bias, bias_init)

    def position_embeding(self, B, N, H, W, img_metas, masks):
        vol, coords_mask, _ = coords3d_volume(B, N, H, W, img_metas, self.depth_num, self.depth_start,
                                              self.position_range, self.LID, masks)
        pe = self.position_encoder(vol)
        return pe.view(B, N, self.embed_dims, H, W), coords_mask

    def forward(self, mlvl_feats, img_metas, return_intermediates=False, dropout_masks=None):
        """``dropout_masks`` (tests only): per decoder layer the six keep masks + scale, see DecoderLayer.forward."""
        x = mlvl_feats[0]
        batch_size, num_cams = x.size(0), x.size(1)
        x = self.input_proj(x.flatten(0, 1))
        x = x.view(batch_size, num_cams, *x.shape[-3:])
        masks = padding_masks(batch_size, num_cams, img_metas, x.shape[-2:])
        H, W = x.shape[-2:]
        coords_pe, _ = self.position_embeding(batch_size, num_cams, H, W, img_metas, masks)
        if self.v2 and self.with_fpe:
            coords_pe = self.fpe(coords_pe.flatten(0, 1), x.flatten(0, 1)).view(x.size())
        pos_embed = coords_pe
        sin_embed = sine_positional_encoding_3d(masks, num_feats=self.embed_dims // 2, normalize=True)
        sin_embed = self.adapt_pos3d(sin_embed.flatten(0, 1)).view(x.size())
        pos_embed = pos_embed + sin_embed
        reference_points = self.reference_points.weight
        query_embeds = self.query_embedding(pos2posemb3d(reference_points))
        reference_points = reference_points.unsqueeze(0).repeat(batch_size, 1, 1)
        outs_dec, _ = self.transformer(x, masks, query_embeds, pos_embed, self.reg_branches, dropout_masks=dropout_masks)
        outs_dec = torch.nan_to_num(outs_dec)
        if self.v2 and self.with_time:
            # petrv2_head.py:499-505 — parity is defined for B=1
            time_stamp = torch.tensor(np.asarray([np.asarray(m['timestamp']) for m in img_metas]),
                                      dtype=x.dtype).view(batch_size, -1, 6)
            mean_time_stamp = (time_stamp[:, 1, :] - time_stamp[:, 0, :]).mean(-1)
        outputs_classes, outputs_coords = [], []
        for lvl in range(outs_dec.shape[0]):
            reference = inverse_sigmoid(reference_points.clone())
            outputs_class = self.cls_branches[lvl](outs_dec[lvl])
            tmp = self.reg_branches[lvl](outs_dec[lvl])
            tmp[..., 0:2] += reference[..., 0:2]
            tmp[..., 0:2] = tmp[..., 0:2].sigmoid()
            tmp[..., 4:5] += reference[..., 2:3]
            tmp[..., 4:5] = tmp[..., 4:5].sigmoid()
            if self.v2 and self.with_time:
                tmp[..., 8:] = tmp[..., 8:] / mean_time_stamp
            outputs_classes.append(outputs_class)
            outputs_coords.append(tmp)
        all_cls_scores = torch.stack(outputs_classes)
        all_bbox_preds = torch.stack(outputs_coords)
        pc = self.pc_range
        all_bbox_preds[..., 0:1] = all_bbox_preds[..., 0:1] * (pc[3] - pc[0]) + pc[0]
        all_bbox_preds[..., 1:2] = all_bbox_preds[..., 1:2] * (pc[4] - pc[1]) + pc[1]
        all_bbox_preds[..., 4:5] = all_bbox_preds[..., 4:5] * (pc[5] - pc[2]) + pc[2]
        outs = {'all_cls_scores': all_cls_scores, 'all_bbox_preds': all_bbox_preds,
                'enc_cls_scores': None, 'enc_bbox_preds': None}
        if return_intermediates:
            outs['_outs_dec'] = outs_dec
            outs['_pos_embed'] = pos_embed
            outs['_memory'] = x
            outs['_query_embeds'] = query_embeds
            outs['_masks'] = masks
        return outs


def perturb_parameters(module, seed, scale=0.02):
    """Add seeded N(0, scale^2) noise to every trainable parameter (each distinct tensor once)."""
    if seed is None:
        return
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.requires_grad:
                p.add_(scale * torch.randn(p.shape, generator=g))


def seeded_head(seed, perturb_seed=None, **kw):
    """Oracle head whose weights are a pure function of (seed, perturb_seed): reference init rules
    (petr_head.py:276-284, petr_transformer.py:62-67) drawn in the reference's construction order,
    then perturb_parameters.  oracle/make_golden.py asserts tensor-for-tensor equality with the
    reference head built the same way, so fixtures need not store weights."""
    torch.manual_seed(seed)
    h = PETRHeadOracle(**kw)
    h.init_weights()
    perturb_parameters(h, perturb_seed)
    return h.eval()


# --------------------------------------------------------------------------- #
# synthetic inputs; convention of datasets/nuscenes_dataset.py:53-66
# --------------------------------------------------------------------------- #
def synthetic_lidar2img(num_views, img_w, seed=0):
    """lidar2img = viewpad(K) @ lidar2cam_rt.T with nuScenes-like intrinsics and six yaw angles."""
    rng = np.random.RandomState(seed)
    r = img_w / 1600.0
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 1266.0 * r
    K[0, 2], K[1, 2] = 816.0 * r, 491.0 * r
    yaws = np.deg2rad([0.0, -55.0, 55.0, 180.0, 110.0, -110.0])
    # camera axes (x right, y down, z forward) from lidar axes (x right, y forward, z up)
    swap = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)
    mats = []
    for i in range(num_views):
        yaw = yaws[i % 6]
        c, s = np.cos(yaw), np.sin(yaw)
        Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)
        R = swap @ Rz.T
        t = rng.uniform(-1.5, 1.5, size=3)
        rt = np.eye(4)
        rt[:3, :3] = R
        rt[:3, 3] = -R @ t
        mats.append(K @ rt)
    return mats


def synthetic_img_metas(batch, num_views, pad_hw, img_hw=None, seed=0, with_time=False):
    img_hw = img_hw or pad_hw
    metas = []
    for b in range(batch):
        m = {'pad_shape': [(pad_hw[0], pad_hw[1], 3)] * num_views,
             'img_shape': [(img_hw[0], img_hw[1], 3)] * num_views,
             'lidar2img': synthetic_lidar2img(num_views, pad_hw[1], seed=seed + b)}
        if with_time:
            half = num_views // 2
            m['timestamp'] = [0.0] * half + [0.5 + 0.01 * i for i in range(num_views - half)]
        metas.append(m)
    return metas
