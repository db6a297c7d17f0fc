"""Head hyper-parameters of the reference configs, as the ``pts_bbox_head=dict(...)`` they appear in.

c5   : projects/configs/petr/petr_r50dcn_gridmask_c5.py:45-98 (in_channels 2048 there; the BASELINE
       benchmark feeds 256-channel synthetic features)
p4   : petr_r50dcn_gridmask_p4.py:45-53 / petr_vovnet_gridmask_p4_*.py (same head, in_channels=256)
"""

POINT_CLOUD_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]

# feature-map shapes (N, H, W, pad_h, pad_w) of the BASELINE.json configs
SHAPES = {
    'c5': (6, 16, 44, 512, 1408),            # petr_r50dcn_gridmask_c5, stride 32
    'p4_1408': (6, 32, 88, 512, 1408),       # petr_r50dcn_gridmask_p4, stride 16
    'p4_1600': (6, 40, 100, 640, 1600),      # petr_vovnet_gridmask_p4_1600x640
    'v2_800': (12, 20, 50, 320, 800),        # petrv2_vovnet_gridmask_p4_800x320 (two frames)
    'toy': (2, 4, 6, 128, 192),
}


def petr_head_cfg(in_channels=256, num_query=900, num_layers=6, feedforward_channels=2048, **overrides):
    cfg = dict(
        type='PETRHead', num_classes=10, in_channels=in_channels, num_query=num_query, LID=True, with_position=True,
        with_multiview=True, position_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], normedlinear=False,
        transformer=dict(
            type='PETRTransformer',
            decoder=dict(
                type='PETRTransformerDecoder', return_intermediate=True, num_layers=num_layers,
                transformerlayers=dict(
                    type='PETRTransformerDecoderLayer',
                    attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1),
                               dict(type='PETRMultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1)],
                    feedforward_channels=feedforward_channels, ffn_dropout=0.1, with_cp=True,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))),
        bbox_coder=dict(type='NMSFreeCoder', post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                        pc_range=POINT_CLOUD_RANGE, max_num=300, voxel_size=[0.2, 0.2, 8], num_classes=10),
        positional_encoding=dict(type='SinePositionalEncoding3D', num_feats=128, normalize=True),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_bbox=dict(type='L1Loss', loss_weight=0.25), loss_iou=dict(type='GIoULoss', loss_weight=0.0),
        train_cfg=None)
    cfg.update(overrides)
    return cfg


def petrv2_head_cfg(in_channels=256, num_query=900, **overrides):
    """projects/configs/petrv2/petrv2_vovnet_gridmask_p4_800x320.py:41-96"""
    cfg = petr_head_cfg(in_channels=in_channels, num_query=num_query)
    cfg.update(type='PETRv2Head', with_fpe=True, with_time=True, with_multi=True, code_weights=[1.0] * 10)
    cfg.update(overrides)
    return cfg
