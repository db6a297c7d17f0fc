"""TEST INFRASTRUCTURE (checker only; never imported by petr_amd, bench's timed legs or the product path).

CPU restatement of the reference's CPFPN forward (projects/mmdet3d_plugin/models/necks/cp_fpn.py:159-210) for the
configuration every PETR config uses (two or more backbone levels, num_outs = levels, no extra convs, no norm / act):
lateral 1x1 convs (:164-167), top-down nearest-upsample adds (:175-186), ONE 3x3 output conv on level 0 (:190-192).
Pinned by tests/golden/cpfpn_*.npz, which oracle/make_golden_neck.py writes from the reference class itself.
"""
import torch.nn.functional as F


def cpfpn_forward(state, inputs):
    """state: dict with lateral_convs.{i}.conv.{weight,bias}, fpn_convs.0.conv.{weight,bias}; inputs: list of NCHW maps."""
    n = len(inputs)
    lat = [F.conv2d(inputs[i], state[f'lateral_convs.{i}.conv.weight'], state[f'lateral_convs.{i}.conv.bias']) for i in range(n)]
    for i in range(n - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    out0 = F.conv2d(lat[0], state['fpn_convs.0.conv.weight'], state['fpn_convs.0.conv.bias'], padding=1)
    return tuple([out0] + lat[1:])
