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

Backward (training: the reference neck is a trainable module inside the step, cp_fpn.py:159-210, checkpointed at :165-193):
``forward`` is a ``torch.autograd.Function`` whose backward is the same library -
  * 3x3 conv: the output gradient goes NCHW -> zero-bordered channels-last (``petr_nchw_to_padded_nhwc``); the input gradient is
    the forward's K-segmented contraction with the flipped / transposed weights; the weight gradient is ONE contraction per
    call (three kernel-row batches x views, K segments over the image rows, float-atomic accumulation) and the bias gradient
    rides on it as the column sums of its A operand;
  * top-down adds: ``petr_fpn_upsample_add_bwd`` (gather form of the nearest-upsample adjoint, fixed summation order);
  * lateral 1x1 convs: weight gradient = K-major x K-major contraction over (view, pixel) segments, input gradient = the
    transposed-weight contraction written channel-major (NCHW), i.e. what the backbone receives.
``add_extra_convs`` / extra max-pool levels / norm / activation are not used by any PETR config
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

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        xs = [inputs[i + self.start_level] for i in range(len(self.lateral_convs))]
        for x in xs:
            if not x.is_cuda:
                raise _C.PetrHipError('CPFPN (petr_amd) runs on the GPU only; there is no CPU fallback')
            if x.dtype != torch.float32:
                raise _C.PetrHipError('CPFPN expects fp32 maps (fp16_enabled = False, cp_fpn.py:92)')
        params = []
        for m in self.lateral_convs:
            params += [m.conv.weight, m.conv.bias]
        params += [self.fpn_convs[0].conv.weight, self.fpn_convs[0].conv.bias]
        return _CPFPNFunction.apply(self, len(xs), *xs, *params)

    # ------------------------------------------------------------------ inference: the head's input_proj folded into the 3x3 conv
    def _folded_w3(self, head):
        """input_proj(conv3x3(x)) = one 3x3 conv: W' = W_proj W_3 ([out', out] x [out, c, dy, dx]), b' = W_proj b_3 + b_proj
        (cp_fpn.py:190-192 followed by petr_head.py:390; valid because the reference puts no norm / activation between them).
        Formed once per weight version in float64 on the host, packed [dy][out'][dx*C + c] like the unfolded weights."""
        w3, b3 = self.fpn_convs[0].conv.weight, self.fpn_convs[0].conv.bias
        wp, bp = head.input_proj.weight, head.input_proj.bias
        key = tuple((t.data_ptr(), t._version) for t in (w3, b3, wp, bp)) + (w3.device,)
        cached = getattr(self, '_folded', None)
        if cached is None or cached[0] != key:
            C_ = self.out_channels
            Ch = wp.shape[0]
            assert wp.shape[1] == C_, f'input_proj expects {wp.shape[1]} channels, the neck produces {C_}'
            wp64 = wp.detach().double().cpu().view(Ch, C_)
            wf = (wp64 @ w3.detach().double().cpu().view(C_, -1)).view(Ch, C_, 3, 3)
            bf = wp64 @ b3.detach().double().cpu() + bp.detach().double().cpu()
            packed = wf.permute(2, 0, 3, 1).contiguous().view(3, Ch, 3 * C_).float().to(w3.device)
            self._folded = (key, packed, bf.float().to(w3.device))
        return self._folded[1], self._folded[2]

    def forward_folded(self, inputs, head):
        """Inference path of SURVEY §8(f) rank 4: level 0 of the neck with ``head.input_proj`` folded into the 3x3 output conv,
        written token-major.  Returns the head's projected memory ``[V, H, W, embed_dims]`` (V = B * N views; reshape to
        ``[B, N, H, W, C]`` for ``PETRHead.forward_projected``): the NCHW map of level 0 and the 1x1 contraction over it - one
        full feature-map round trip - do not exist.  No gradient (the fold has no separate parameters)."""
        assert len(inputs) == len(self.in_channels)
        if self.start_level != 0 or head.position_level != 0:
            raise _C.PetrHipError('forward_folded: the 3x3 output conv belongs to backbone level 0 (cp_fpn.py:128) = position_level 0')
        xs = [inputs[i + self.start_level].detach() for i in range(len(self.lateral_convs))]
        for x in xs:
            if not x.is_cuda or x.dtype != torch.float32:
                raise _C.PetrHipError('CPFPN.forward_folded expects fp32 CUDA maps; there is no CPU fallback')
        C_ = self.out_channels
        wf, bf = self._folded_w3(head)
        Ch = wf.shape[1]
        with torch.no_grad():
            _, pad = self._laterals_topdown(xs)
            V, Hp, Wp_, _ = pad.shape
            H, W = Hp - 2, Wp_ - 2
            mem = torch.empty((V, H, W, Ch), dtype=torch.float32, device=pad.device)
            ops.gemm_raw(a=pad, lda=C_, a_kcontig=1, a_bs0=(H + 2) * (W + 2) * C_, a_bs1=(W + 2) * C_, b=wf, ldb=3 * C_, b_kcontig=1,
                         c=mem, ldc=Ch, c_bs0=H * W * Ch, c_bs1=W * Ch, bias=bf, M=W, N=Ch, K=9 * C_, nb0=V, nb1=H, k_seg=3 * C_,
                         a_seg_stride=(W + 2) * C_, b_seg_stride=Ch * 3 * C_, flags=0, alpha=1.0)
        return mem

    # ------------------------------------------------------------------ forward arithmetic (libpetr_hip.so)
    def _forward_impl(self, xs):
        C_ = self.out_channels
        V = xs[0].shape[0]
        dev = xs[0].device
        lat, pad = self._laterals_topdown(xs)
        H, W = pad.shape[1] - 2, pad.shape[2] - 2
        # 3x3 output conv of level 0 (cp_fpn.py:190-192): one contraction, K = 9*C in three kernel-row segments
        w3 = self._packed_w3()
        conv3 = self.fpn_convs[0].conv
        out0 = torch.empty((V, C_, H, W), dtype=torch.float32, device=dev)
        ops.gemm_raw(a=w3, lda=3 * C_, a_kcontig=1, b=pad, ldb=C_, b_kcontig=1, b_bs0=(H + 2) * (W + 2) * C_, b_bs1=(W + 2) * C_,
                     c=out0, ldc=H * W, c_bs0=C_ * H * W, c_bs1=W, bias=conv3.bias.detach(), M=C_, N=W, K=9 * C_, nb0=V, nb1=H,
                     k_seg=3 * C_, a_seg_stride=C_ * 3 * C_, b_seg_stride=(W + 2) * C_, flags=_C.GEMM_BIAS_M, alpha=1.0)
        return [out0] + lat[1:], pad

    def _laterals_topdown(self, xs):
        """lateral 1x1 convs + top-down adds: (lat list with lat[0] = None, zero-bordered channels-last level-0 map)."""
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
            ops.gemm_raw(a=conv.weight.detach().view(C_, Ci), lda=Ci, a_kcontig=1, b=x, ldb=Hi * Wi, b_kcontig=0, b_bs0=Ci * Hi * Wi,
                         c=out, ldc=Hi * Wi, c_bs0=C_ * Hi * Wi, bias=conv.bias.detach(), M=C_, N=Hi * Wi, K=Ci, nb0=V, nb1=1,
                         flags=_C.GEMM_BIAS_M, alpha=1.0)
            lat[i] = out
        # level 0's lateral: token-major rows into the interior of the zero-bordered channels-last map
        x0 = xs[0].contiguous()
        _, C0, H, W = x0.shape
        pad = torch.zeros((V, H + 2, W + 2, C_), dtype=torch.float32, device=dev)
        conv0 = self.lateral_convs[0].conv
        interior = pad[:, 1:, 1:]
        ops.gemm_raw(a=x0, lda=H * W, a_kcontig=0, a_bs0=C0 * H * W, a_bs1=W, b=conv0.weight.detach().view(C_, C0), ldb=C0, b_kcontig=1,
                     c=interior, ldc=C_, c_bs0=(H + 2) * (W + 2) * C_, c_bs1=(W + 2) * C_, bias=conv0.bias.detach(), M=W, N=C_, K=C0,
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
        return lat, pad

    # ------------------------------------------------------------------ backward arithmetic (libpetr_hip.so)
    def _backward_impl(self, xs, pad, grads, need_dx):
        """grads[i]: gradient of output i (NCHW) or None.  Returns (dx list, [(dW, db) per lateral], dW3, db3)."""
        C_ = self.out_channels
        n = len(xs)
        dev = xs[0].device
        V = xs[0].shape[0]
        L = _C.lib()
        _, C0, H, W = xs[0].shape
        dlat = [None] * n          # gradient of the final (post top-down) lateral maps; level 0 channels-last [V, H, W, C]
        dw3 = db3 = None
        if grads[0] is not None:
            g0 = grads[0].contiguous()
            gpad = torch.zeros((V, H + 2, W + 2, C_), dtype=torch.float32, device=dev)
            _C.check(L.petr_nchw_to_padded_nhwc(g0.data_ptr(), gpad.data_ptr(), V, C_, H, W, ops._stream()), 'petr_nchw_to_padded_nhwc')
            # dW3 in the packed layout [dy][out][dx*C + c] (+ bias gradient = row sums of g0, one copy per kernel row)
            dw3p = torch.zeros((3, C_, 3 * C_), dtype=torch.float32, device=dev)
            db3x = torch.zeros((3, C_), dtype=torch.float32, device=dev)
            ops.gemm_raw(a=g0, lda=H * W, a_kcontig=1, a_bs0=C_ * H * W, a_bs1=0, b=pad, ldb=C_, b_kcontig=0,
                         b_bs0=(H + 2) * (W + 2) * C_, b_bs1=(W + 2) * C_, c=dw3p, ldc=3 * C_, c_bs0=0, c_bs1=C_ * 3 * C_,
                         M=C_, N=3 * C_, K=H * W, k_seg=W, a_seg_stride=W, b_seg_stride=(W + 2) * C_, nb0=V, nb1=3,
                         a_colsum=db3x, cs_bs0=0, cs_bs1=C_, flags=_C.GEMM_ATOMIC, alpha=1.0)
            dw3 = dw3p.view(3, C_, 3, C_).permute(1, 3, 0, 2).contiguous()          # [out, c, dy, dx]
            db3 = db3x[0].clone()
            # d(lat0) = the forward's contraction over the bordered gradient map with the flipped, transposed weights
            w3 = self.fpn_convs[0].conv.weight.detach()
            w3t = w3.flip(2, 3).permute(2, 1, 3, 0).contiguous().view(3, C_, 3 * C_)     # [dy'][c][dx'*C + out]
            d0 = torch.empty((V, H, W, C_), dtype=torch.float32, device=dev)
            ops.gemm_raw(a=gpad, lda=C_, a_kcontig=1, a_bs0=(H + 2) * (W + 2) * C_, a_bs1=(W + 2) * C_, b=w3t, ldb=3 * C_, b_kcontig=1,
                         c=d0, ldc=C_, c_bs0=H * W * C_, c_bs1=W * C_, M=W, N=C_, K=9 * C_, nb0=V, nb1=H, k_seg=3 * C_,
                         a_seg_stride=(W + 2) * C_, b_seg_stride=C_ * 3 * C_, flags=0, alpha=1.0)
            dlat[0] = d0
        # top-down adjoint, ascending: d(lat_i) = g_i + adjoint-upsample(d(lat_{i-1}))
        for i in range(1, n):
            _, _, Hs, Ws = xs[i].shape
            have = grads[i] is not None
            if dlat[i - 1] is None and not have:
                continue
            d = grads[i].contiguous().clone() if have else torch.zeros((V, C_, Hs, Ws), dtype=torch.float32, device=dev)
            if dlat[i - 1] is not None:
                Hd, Wd = xs[i - 1].shape[2:]
                if i - 1 == 0:
                    strides = (Hd * Wd * C_, 1, Wd * C_, C_)
                else:
                    strides = (C_ * Hd * Wd, Hd * Wd, Wd, 1)
                _C.check(L.petr_fpn_upsample_add_bwd(d.data_ptr(), dlat[i - 1].data_ptr(), *strides, V, C_, Hd, Wd, Hs, Ws, 1,
                                                     ops._stream()), 'petr_fpn_upsample_add_bwd')
            dlat[i] = d
        dxs, dwb = [None] * n, [(None, None)] * n
        for i in range(n):
            if dlat[i] is None:
                continue
            x = xs[i].contiguous()
            _, Ci, Hi, Wi = x.shape
            HW = Hi * Wi
            wl = self.lateral_convs[i].conv.weight.detach().view(C_, Ci)
            dw = torch.zeros((C_, Ci), dtype=torch.float32, device=dev)
            db = torch.zeros((C_,), dtype=torch.float32, device=dev)
            tiles = ((C_ + 63) // 64) * ((Ci + 63) // 64)
            sk = max(1, min(8, 512 // tiles, HW // 64))
            if i == 0:      # channels-last gradient [V, HW, C]
                ops.gemm_raw(a=dlat[0], lda=C_, a_kcontig=0, a_seg_stride=HW * C_, b=x, ldb=HW, b_kcontig=1, b_seg_stride=Ci * HW,
                             c=dw, ldc=Ci, M=C_, N=Ci, K=V * HW, k_seg=HW, a_colsum=db, split_k=sk, flags=_C.GEMM_ATOMIC, alpha=1.0)
                if need_dx[0]:
                    dx = torch.empty_like(x)
                    ops.gemm_raw(a=wl, lda=Ci, a_kcontig=0, b=dlat[0], ldb=C_, b_kcontig=1, b_bs0=HW * C_, c=dx, ldc=HW, c_bs0=Ci * HW,
                                 M=Ci, N=HW, K=C_, nb0=V, flags=0, alpha=1.0)
                    dxs[0] = dx
            else:           # NCHW gradient [V, C, HW]
                ops.gemm_raw(a=dlat[i], lda=HW, a_kcontig=1, a_seg_stride=C_ * HW, b=x, ldb=HW, b_kcontig=1, b_seg_stride=Ci * HW,
                             c=dw, ldc=Ci, M=C_, N=Ci, K=V * HW, k_seg=HW, a_colsum=db, split_k=sk, flags=_C.GEMM_ATOMIC, alpha=1.0)
                if need_dx[i]:
                    dx = torch.empty_like(x)
                    ops.gemm_raw(a=wl, lda=Ci, a_kcontig=0, b=dlat[i], ldb=HW, b_kcontig=0, b_bs0=C_ * HW, c=dx, ldc=HW, c_bs0=Ci * HW,
                                 M=Ci, N=HW, K=C_, nb0=V, flags=0, alpha=1.0)
                    dxs[i] = dx
            dwb[i] = (dw.view(C_, Ci, 1, 1), db)
        return dxs, dwb, dw3, db3


class _CPFPNFunction(torch.autograd.Function):
    """CPFPN.forward as one autograd node: inputs = the backbone maps and the neck's parameters, outputs = the NCHW maps."""

    @staticmethod
    def forward(ctx, neck, n, *args):
        xs = [a.detach() for a in args[:n]]
        outs, pad = neck._forward_impl(xs)
        ctx.neck, ctx.n = neck, n
        ctx.save_for_backward(*xs, pad)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        neck, n = ctx.neck, ctx.n
        saved = ctx.saved_tensors
        xs, pad = list(saved[:n]), saved[n]
        need_dx = [ctx.needs_input_grad[2 + i] for i in range(n)]
        with torch.no_grad():
            dxs, dwb, dw3, db3 = neck._backward_impl(xs, pad, list(grads), need_dx)
        out = [None, None] + dxs
        for dw, db in dwb:
            out += [dw, db]
        out += [dw3, db3]
        return tuple(out)
