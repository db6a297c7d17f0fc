"""SinePositionalEncoding3D — host mirror of reference models/utils/positional_encoding.py:15-110.

Same constructor, same ``forward(mask[B,N,H,W]) -> [B,N,3*num_feats,H,W]``; the arithmetic runs in
the ``petr_sine3d_fwd`` HIP kernel (no torch fallback).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .registry import register


def sine_dim_t(num_feats, temperature):
    """dim_t exactly as the reference builds it (positional_encoding.py:82-84), on the host once."""
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    return temperature ** (2 * (dim_t // 2) / num_feats)


@register('POSITIONAL_ENCODING')
class SinePositionalEncoding3D(nn.Module):
    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6, offset=0.,
                 init_cfg=None):
        super().__init__()
        if normalize:
            assert isinstance(scale, (float, int)), 'when normalize is set, scale should be provided and in ' \
                f'float or int type, found {type(scale)}'
        self.num_feats = num_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = scale
        self.eps = eps
        self.offset = offset
        self.register_buffer('_dim_t', sine_dim_t(num_feats, temperature), persistent=False)

    def forward(self, mask):
        assert mask.dim() == 4, 'mask must be [B, N, H, W]'
        B, N, H, W = mask.shape
        if not self.normalize:
            raise NotImplementedError('normalize=False is not used by any reference config')
        dim_t = self._dim_t.to(mask.device)
        return ops.sine3d(mask, dim_t, B, N, H, W, normalize=True, scale=float(self.scale), eps=float(self.eps),
                          offset=float(self.offset))

    def __repr__(self):
        return (f'{self.__class__.__name__}(num_feats={self.num_feats}, temperature={self.temperature}, '
                f'normalize={self.normalize}, scale={self.scale}, eps={self.eps})')
