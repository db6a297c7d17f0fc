"""PETRTransformer family — host mirror of reference models/utils/petr_transformer.py.

Same class names, constructor arguments, parameter names (state_dict keys) and ``forward`` contracts
as the reference; the arithmetic runs in libpetr_hip.so (``petr_amd.ops``), never in torch operators.
These per-module forwards are the inference-time drop-in surface (no autograd graph is built); the
training path is the fused executor behind ``PETRHead.forward``, which owns the same parameters.

The three mmcv classes the reference inherits from (``BaseTransformerLayer``, ``TransformerLayerSequence``,
``FFN``) are not vendored by the reference; their layout is restated here only as PARAMETER CONTAINERS
(names per SURVEY §8(b), dispatch order per the reference's in-tree copy
models/utils/multi_atten_decoder_layer.py:204-293).
"""
import copy
import warnings

import torch
import torch.nn as nn

from . import ops
from .registry import ATTENTION, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE, register


def _rows(t):
    """[S, B, C] (seq-first, as torch.nn.MultiheadAttention) -> contiguous batch-major rows [B*S, C]."""
    return t.transpose(0, 1).contiguous().view(-1, t.shape[-1])


def _unrows(t, S, B):
    return t.view(B, S, -1).transpose(0, 1).contiguous()


def _attention(attn, q_rows, k_rows, v_rows, B, Lq, Lk, key_padding_mask, q_pos=None, k_pos=None):
    """in-proj -> flash attention core -> out-proj through the C ABI; rows are batch-major [B*L, C]."""
    C = attn.embed_dim
    H = attn.num_heads
    W, b = attn.in_proj_weight, attn.in_proj_bias
    q = ops.linear(q_rows, W[:C], b[:C], a2=q_pos, a2_rows=(q_pos.shape[0] if q_pos is not None else 0))
    k = ops.linear(k_rows, W[C:2 * C], b[C:2 * C], a2=k_pos, a2_rows=(k_pos.shape[0] if k_pos is not None else 0))
    v = ops.linear(v_rows, W[2 * C:], b[2 * C:])
    qv = q.view(B, Lq, H, C // H).permute(0, 2, 1, 3)
    kv = k.view(B, Lk, H, C // H).permute(0, 2, 1, 3)
    vv = v.view(B, Lk, H, C // H).permute(0, 2, 1, 3)
    o, _ = ops.mha_fwd(qv, kv, vv, key_padding_mask, scale=(C // H) ** -0.5, need_lse=False)
    return o.permute(0, 2, 1, 3).reshape(B * Lq, C)


@register('ATTENTION')
class PETRMultiheadAttention(nn.Module):
    """reference petr_transformer.py:228-367 (an in-tree copy of mmcv's MultiheadAttention wrapper)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=dict(type='Dropout', drop_prob=0.),
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        dropout_layer = copy.deepcopy(dropout_layer) if dropout_layer else None
        if 'dropout' in kwargs:   # petr_transformer.py:258-265
            warnings.warn('The arguments `dropout` in MultiheadAttention has been deprecated, now you can separately '
                          'set `attn_drop`(float), proj_drop(float), and `dropout_layer`(dict) ', DeprecationWarning)
            attn_drop = kwargs['dropout']
            if dropout_layer is not None:
                dropout_layer['drop_prob'] = kwargs.pop('dropout')
            else:
                kwargs.pop('dropout')
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn_drop_p = attn_drop
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)   # parameter container
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(dropout_layer['drop_prob']) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        assert attn_mask is None, 'attn_mask is not used on the PETR path'
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None:
            if query_pos.shape == key.shape:
                key_pos = query_pos
            else:
                warnings.warn(f'position encoding of key is missing in {self.__class__.__name__}.')
        if self.batch_first:
            query, key, value, identity = (t.transpose(0, 1) for t in (query, key, value, identity))
            query_pos = query_pos.transpose(0, 1) if query_pos is not None else None
            key_pos = key_pos.transpose(0, 1) if key_pos is not None else None
        Lq, B, C = query.shape
        Lk = key.shape[0]
        with torch.no_grad():
            q_pos = _rows(query_pos) if query_pos is not None else None
            k_pos = _rows(key_pos) if key_pos is not None else None
            o = _attention(self.attn, _rows(query), _rows(key), _rows(value), B, Lq, Lk, key_padding_mask, q_pos, k_pos)
            out = ops.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=_rows(identity))
        out = _unrows(out, Lq, B)
        return out.transpose(0, 1) if self.batch_first else out


# mmcv's own `MultiheadAttention` (cfg type of the self-attention, c5:64-68) has the same body
ATTENTION.register_module(name='MultiheadAttention', force=True)(PETRMultiheadAttention)


class FFN(nn.Module):
    """mmcv FFN as a parameter container: layers.0.0 = Linear(C,F), layers.1 = Linear(F,C)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0., **kwargs):
        super().__init__()
        assert num_fcs == 2, 'the reference configs use ffn_num_fcs=2'
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

    def forward(self, x, identity=None):
        S, B, C = x.shape
        with torch.no_grad():
            rows = _rows(x)
            h = ops.linear(rows, self.layers[0][0].weight, self.layers[0][0].bias, relu=True)
            out = ops.linear(h, self.layers[1].weight, self.layers[1].bias,
                             residual=rows if identity is None else _rows(identity))
        return _unrows(out, S, B)


class _LayerNorm(nn.LayerNorm):
    def forward(self, x):
        shp = x.shape
        with torch.no_grad():
            y = ops.layernorm(x.contiguous().view(-1, shp[-1]), self.weight, self.bias, eps=self.eps)
        return y.view(shp)


@register('TRANSFORMER_LAYER')
class PETRTransformerDecoderLayer(nn.Module):
    """reference petr_transformer.py:113-224 over mmcv BaseTransformerLayer."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2, with_cp=True,
                 batch_first=False, **kwargs):
        super().__init__()
        assert operation_order is not None and len(operation_order) == 6
        assert set(operation_order) == set(['self_attn', 'norm', 'cross_attn', 'ffn'])
        assert norm_cfg.get('type', 'LN') == 'LN'
        self.operation_order = tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        self.use_checkpoint = with_cp   # memory device of the reference; the flash kernels never store scores
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.attentions = nn.ModuleList()
        for cfg in attn_cfgs:
            cfg = copy.deepcopy(cfg)
            cfg.setdefault('batch_first', batch_first)
            self.attentions.append(ATTENTION.build(cfg))
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList([FFN(self.embed_dims, feedforward_channels, ffn_num_fcs, ffn_dropout)
                                   for _ in range(operation_order.count('ffn'))])
        self.norms = nn.ModuleList([_LayerNorm(self.embed_dims) for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        for layer in self.operation_order:
            if layer == 'self_attn':
                query = self.attentions[attn_index](query, query, query, identity if self.pre_norm else None,
                                                    query_pos=query_pos, key_pos=query_pos,
                                                    key_padding_mask=query_key_padding_mask)
                attn_index += 1
                identity = query
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](query, key, value, identity if self.pre_norm else None,
                                                    query_pos=query_pos, key_pos=key_pos,
                                                    key_padding_mask=key_padding_mask)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


@register('TRANSFORMER_LAYER_SEQUENCE')
class PETRTransformerDecoder(nn.Module):
    """reference petr_transformer.py:401-447 over mmcv TransformerLayerSequence."""

    def __init__(self, transformerlayers=None, num_layers=None, post_norm_cfg=dict(type='LN'), return_intermediate=False,
                 init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList([TRANSFORMER_LAYER.build(c) for c in transformerlayers])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm
        self.return_intermediate = return_intermediate
        self.post_norm = _LayerNorm(self.embed_dims) if post_norm_cfg is not None else None

    def forward(self, query, *args, **kwargs):
        kwargs.pop('reg_branch', None)
        if not self.return_intermediate:
            for layer in self.layers:
                query = layer(query, *args, **kwargs)
            if self.post_norm:
                query = self.post_norm(query)[None]
            return query
        intermediate = []
        for layer in self.layers:
            query = layer(query, *args, **kwargs)
            intermediate.append(self.post_norm(query) if self.post_norm is not None else query)
        return torch.stack(intermediate)


@register('TRANSFORMER_LAYER_SEQUENCE')
class PETRTransformerEncoder(nn.Module):
    """reference petr_transformer.py:371-397; no reference config builds it (encoder=None)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError('PETRTransformerEncoder is unused by every reference config (SURVEY §2 #2)')


@register('TRANSFORMER')
class PETRTransformer(nn.Module):
    """reference petr_transformer.py:34-109."""

    def __init__(self, encoder=None, decoder=None, init_cfg=None, cross=False):
        super().__init__()
        assert encoder is None, 'the PETR configs are decoder-only'
        self.encoder = None
        self.decoder = TRANSFORMER_LAYER_SEQUENCE.build(decoder)
        self.embed_dims = self.decoder.embed_dims
        self.cross = cross

    def init_weights(self):
        # petr_transformer.py:62-67 + mmcv xavier_init(distribution='uniform'): weight xavier-uniform, bias 0
        for m in self.modules():
            if hasattr(m, 'weight') and isinstance(m.weight, torch.Tensor) and m.weight.dim() > 1:
                nn.init.xavier_uniform_(m.weight, gain=1)
                if getattr(m, 'bias', None) is not None:
                    nn.init.constant_(m.bias, 0)
        self._is_init = True

    def forward(self, x, mask, query_embed, pos_embed, reg_branch=None):
        bs, n, c, h, w = x.shape
        memory = x.permute(1, 3, 4, 0, 2).reshape(-1, bs, c)
        pos_embed = pos_embed.permute(1, 3, 4, 0, 2).reshape(-1, bs, c)
        query_embed = query_embed.unsqueeze(1).repeat(1, bs, 1)
        mask = mask.view(bs, -1)
        target = torch.zeros_like(query_embed)
        out_dec = self.decoder(query=target, key=memory, value=memory, key_pos=pos_embed, query_pos=query_embed,
                               key_padding_mask=mask if bool(mask.any()) else None)
        out_dec = out_dec.transpose(1, 2)
        memory = memory.reshape(n, h, w, bs, c).permute(3, 0, 4, 1, 2)
        return out_dec, memory
