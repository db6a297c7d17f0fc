"""Generate tests/golden/cpfpn_*.npz from the reference's CPFPN.  BUILD-CONTAINER ONLY (needs /root/reference).

cp_fpn.py imports mmcv.cnn.ConvModule, mmcv.runner.{BaseModule, auto_fp16} and mmdet.models.NECKS, which are absent:
the file is loaded BY PATH after inert stand-ins are placed in sys.modules.  ``ConvModule`` is restated for the only
form CPFPN uses in the PETR configs (conv_cfg = norm_cfg = act_cfg = None: a bare nn.Conv2d held as ``.conv``); everything
that executes in ``CPFPN.__init__`` / ``forward`` is the reference's own code.  The script checks oracle/neck_oracle.py
against it and stores inputs, parameters and the reference's outputs.   Usage:  python oracle/make_golden_neck.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference/projects/mmdet3d_plugin/models/necks/cp_fpn.py'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
from oracle import neck_oracle as NO  # noqa: E402


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 inplace=True):
        super().__init__()
        assert conv_cfg is None and norm_cfg is None and act_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)

    def forward(self, x):
        return self.conv(x)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()


class _Reg:
    def register_module(self, *a, **k):
        return lambda cls: cls


def load_reference():
    for name in ('mmcv', 'mmcv.cnn', 'mmcv.runner', 'mmdet', 'mmdet.models'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['mmcv.cnn'].ConvModule = ConvModule
    sys.modules['mmcv.runner'].BaseModule = BaseModule
    sys.modules['mmcv.runner'].auto_fp16 = lambda *a, **k: (lambda fn: fn)
    sys.modules['mmdet.models'].NECKS = _Reg()
    spec = importlib.util.spec_from_file_location('ref_cp_fpn', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CPFPN


def main():
    CPFPN = load_reference()
    cases = {
        # name: (in_channels, V, sizes per level)
        'cpfpn_toy': ([48, 80], 3, [(6, 10), (3, 5)]),
        'cpfpn_odd3': ([24, 40, 56], 2, [(7, 9), (4, 5), (2, 3)]),      # three levels, sizes that do not halve exactly
    }
    for name, (chans, V, sizes) in cases.items():
        torch.manual_seed(11)
        ref = CPFPN(in_channels=chans, out_channels=64, num_outs=len(chans)).eval()      # 64: keeps the fixture small
        for m in ref.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.uniform_(m.bias, -0.5, 0.5)
        g = torch.Generator().manual_seed(5)
        inputs = [torch.randn(V, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
        with torch.no_grad():
            want = ref(inputs)
            got = NO.cpfpn_forward(ref.state_dict(), inputs)
        for a, b in zip(want, got):
            assert torch.equal(a, b), 'oracle/neck_oracle.py disagrees with the reference CPFPN'
        fx = {f'in{i}': x.numpy() for i, x in enumerate(inputs)}
        fx.update({f'out{i}': y.numpy() for i, y in enumerate(want)})
        fx.update({'param:' + k: v.numpy() for k, v in ref.state_dict().items()})
        fx['keys'] = np.array(sorted(ref.state_dict().keys()))
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **fx)
        print(name, 'levels', len(chans), 'outputs', [tuple(y.shape) for y in want], 'state_dict keys', len(ref.state_dict()))


if __name__ == '__main__':
    main()
