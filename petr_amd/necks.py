"""CPFPN — host mirror of reference models/necks/cp_fpn.py:18-210 (the producer of the 256-channel
map the head's ``input_proj`` consumes at ``position_level`` 0).

Same constructor arguments and state_dict keys (``lateral_convs.{i}.conv.{weight,bias}``, ``fpn_convs.0.conv.{weight,bias}``:
the reference builds ONE 3x3 output conv, for level 0 only, :128-138) and the same ``forward(inputs) -> tuple`` of NCHW maps.
The arithmetic is libpetr_hip.so (no torch operators, no fallback):

  * lateral 1x1 convs = ``petr_gemm`` reading the NCHW backbone maps in place (K-major operand).  Level 0's lateral is
    written straight into the interior of a zero-bordered channels-last buffer ``[V, H+2, W+2, C]``; the others are produced
    channel-major (weights as the A operand, ``PETR_GEMM_BIAS_M``), i.e. already in the NCHW layout the caller gets back;
  * top-down path (:175-186) = ``petr_fpn_upsample_add`` (nearest source pixel exactly as ``F.interpolate``);
  * the 3x3 output conv = ONE contraction with three K segments (one per kernel row): in the padded channels-last map the
    three taps of a kernel row are 3*C CONTIGUOUS floats starting at pixel (h+dy, w), so a (token, row) operand is just a
    row of length 3*C at a row stride of C (overlapping rows), and the kernel-row offset is a K-segment stride of (W+2)*C;
    the weights are repacked once to ``[dy][out][dx*C + c]``.  Written channel-major, so level 0 also comes out NCHW.

Inference path (the neck's own backward feeds the backbone, which is outside the hot path): tensors that require grad are
refused loudly.  ``add_extra_convs`` / extra max-pool levels / norm / activation are not used by any PETR config
(petr_r50dcn_gridmask_p4.py:45-49, petr_vovnet_gridmask_p4_1600x640.py:38-42) and are refused as well.
"""
import torch
import torch.nn as nn

from . import _C, ops
from .registry import register


class ConvModule(nn.Module):
    """parameter container with mmcv ConvModule's key names (``conv.weight`` / ``conv.bias``); conv only (no norm / act)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding)


@register('NECKS')
class CPFPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.fp16_enabled = False
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        if add_extra_convs or num_outs != self.backbone_end_level - start_level:
            raise _C.PetrHipError('CPFPN (petr_amd): extra levels (add_extra_convs / max-pool) are not used by any PETR config')
        if conv_cfg is not None or norm_cfg is not None or act_cfg is not None:
            raise _C.PetrHipError('CPFPN (petr_amd): conv_cfg / norm_cfg / act_cfg are None in every PETR config')
        if self.upsample_cfg.get('mode', 'nearest') != 'nearest' or 'scale_factor' in self.upsample_cfg:
            raise _C.PetrHipError("CPFPN (petr_amd): upsample_cfg must be dict(mode='nearest') (cp_fpn.py:80)")
        assert out_channels % 4 == 0
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1))
            if i == 0:      # cp_fpn.py:128: ONE output conv, for backbone level 0 only
                self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1))
        self._packed = None

    def init_weights(self):
        """init_cfg = Xavier / uniform on every Conv2d (cp_fpn.py:81-82; mmcv: bias 0)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=1)
                nn.init.constant_(m.bias, 0)

    def _packed_w3(self):
        """3x3 weights [out, c, dy, dx] -> [dy][out][dx*C + c] (one K segment per kernel row), cached per weight version."""
        w = self.fpn_convs[0].conv.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, w.detach().permute(2, 0, 3, 1).contiguous().view(3, self.out_channels, 3 * self.out_channels))
        return self._packed[1]

    @torch.no_grad()
    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        xs = [inputs[i + self.start_level] for i in range(len(self.lateral_convs))]
        for x in xs:
            if not x.is_cuda:
                raise _C.PetrHipError('CPFPN (petr_amd) runs on the GPU only; there is no CPU fallback')
            if x.dtype != torch.float32:
                raise _C.PetrHipError('CPFPN expects fp32 maps (fp16_enabled = False, cp_fpn.py:92)')
        C_ = self.out_channels
        dev = xs[0].device
        V = xs[0].shape[0]
        lat = [None] * len(xs)
        # laterals of the upper levels: channel-major (weights as the A operand, bias per output row) = NCHW
        for i in range(1, len(xs)):
            x = xs[i].contiguous()
            _, Ci, Hi, Wi = x.shape
            conv = self.lateral_convs[i].conv
            out = torch.empty((V, C_, Hi, Wi), dtype=torch.float32, device=dev)
            ops.gemm_raw(a=conv.weight.view(C_, Ci), lda=Ci, a_kcontig=1, b=x, ldb=Hi * Wi, b_kcontig=0, b_bs0=Ci * Hi * Wi,
                         c=out, ldc=Hi * Wi, c_bs0=C_ * Hi * Wi, bias=conv.bias, M=C_, N=Hi * Wi, K=Ci, nb0=V, nb1=1,
                         flags=_C.GEMM_BIAS_M, alpha=1.0)
            lat[i] = out
        # level 0's lateral: token-major rows into the interior of the zero-bordered channels-last map
        x0 = xs[0].contiguous()
        _, C0, H, W = x0.shape
        pad = torch.zeros((V, H + 2, W + 2, C_), dtype=torch.float32, device=dev)
        conv0 = self.lateral_convs[0].conv
        interior = pad[:, 1:, 1:]
        ops.gemm_raw(a=x0, lda=H * W, a_kcontig=0, a_bs0=C0 * H * W, a_bs1=W, b=conv0.weight.view(C_, C0), ldb=C0, b_kcontig=1,
                     c=interior, ldc=C_, c_bs0=(H + 2) * (W + 2) * C_, c_bs1=(W + 2) * C_, bias=conv0.bias, M=W, N=C_, K=C0,
                     nb0=V, nb1=H, flags=0, alpha=1.0)
        # top-down path (cp_fpn.py:175-186): lat[i-1] += nearest-upsampled lat[i]
        L = _C.lib()
        for i in range(len(xs) - 1, 0, -1):
            _, _, Hs, Ws = lat[i].shape
            if i - 1 == 0:
                _C.check(L.petr_fpn_upsample_add(interior.data_ptr(), (H + 2) * (W + 2) * C_, 1, (W + 2) * C_, C_,
                                                 lat[i].data_ptr(), V, C_, H, W, Hs, Ws, ops._stream()), 'petr_fpn_upsample_add')
            else:
                Hd, Wd = lat[i - 1].shape[2:]
                _C.check(L.petr_fpn_upsample_add(lat[i - 1].data_ptr(), C_ * Hd * Wd, Hd * Wd, Wd, 1, lat[i].data_ptr(), V, C_,
                                                 Hd, Wd, Hs, Ws, ops._stream()), 'petr_fpn_upsample_add')
        # 3x3 output conv of level 0 (cp_fpn.py:190-192): one contraction, K = 9*C in three kernel-row segments
        w3 = self._packed_w3()
        conv3 = self.fpn_convs[0].conv
        out0 = torch.empty((V, C_, H, W), dtype=torch.float32, device=dev)
        ops.gemm_raw(a=w3, lda=3 * C_, a_kcontig=1, b=pad, ldb=C_, b_kcontig=1, b_bs0=(H + 2) * (W + 2) * C_, b_bs1=(W + 2) * C_,
                     c=out0, ldc=H * W, c_bs0=C_ * H * W, c_bs1=W, bias=conv3.bias, M=C_, N=W, K=9 * C_, nb0=V, nb1=H,
                     k_seg=3 * C_, a_seg_stride=C_ * 3 * C_, b_seg_stride=(W + 2) * C_, flags=_C.GEMM_BIAS_M, alpha=1.0)
        return tuple([out0] + lat[1:])
