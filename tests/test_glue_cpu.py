"""petr_amd.glue (the caller-side contract of the hot path) against the fixture produced by
executing the reference's own source text (oracle/make_golden_glue.py): calibration -> lidar2img, augmentation
updates, sweep assembly, feature reshape.  Pure host code: runs without a GPU."""
import os

import numpy as np
import torch

from petr_amd import glue


def _fx(golden_dir):
    return np.load(os.path.join(golden_dir, 'glue_calib.npz'))


def test_calibration_to_lidar2img(golden_dir):
    fx = _fx(golden_dir)
    for i in range(fx['K'].shape[0]):
        r = glue.lidar2img_from_calib(fx['K'][i], fx['rot'][i], fx['trans'][i])
        assert np.array_equal(r['lidar2img'], fx['lidar2img'][i])
        assert np.array_equal(r['intrinsics'], fx['intrinsics'][i])
        assert np.array_equal(r['extrinsics'], fx['extrinsics'][i])
    l2i = glue.lidar2img_from_parts(list(fx['intrinsics']), list(fx['extrinsics']))
    assert np.array_equal(np.stack(l2i), fx['lidar2img'])


def test_augmentation_updates(golden_dir):
    fx = _fx(golden_dir)
    k, l2i = glue.resize_intrinsics(list(fx['intrinsics']), list(fx['extrinsics']), float(fx['w_scale']), float(fx['h_scale']))
    assert np.array_equal(np.stack(k), fx['resized_k']) and np.array_equal(np.stack(l2i), fx['resized_l2i'])
    k, l2i = glue.apply_ida(list(fx['intrinsics']), list(fx['extrinsics']), list(fx['idas']))
    assert np.array_equal(np.stack(k), fx['ida_k']) and np.array_equal(np.stack(l2i), fx['ida_l2i'])
    assert np.array_equal(np.stack(fx['intrinsics']), fx['intrinsics'])            # inputs untouched
    rot = glue.rotate_bev_along_z(list(fx['lidar2img']), 0.3)
    assert np.array_equal(np.stack(rot), fx['rotated']) and rot[0].dtype == np.float32
    sc = glue.scale_xyz(rot, 1.07)
    assert np.array_equal(np.stack(sc), fx['scaled'])


def test_sweep_assembly_and_metas():
    sensors = ['CAM_A', 'CAM_B']
    cur = {'timestamp': 100.0, 'img_timestamp': [99.95, 99.96], 'lidar2img': [np.eye(4), 2 * np.eye(4)],
           'intrinsics': [np.eye(4)] * 2, 'extrinsics': [np.eye(4)] * 2}
    sweep = {s: {'timestamp': (99.5 + 0.01 * i) * 1e6, 'lidar2img': (3 + i) * np.eye(4), 'intrinsics': np.eye(4),
                 'extrinsics': np.eye(4)} for i, s in enumerate(sensors)}
    out = glue.append_sweep(cur, sweep, sensors)
    assert len(out['lidar2img']) == 4 and out['lidar2img'][3][0, 0] == 4
    assert np.allclose(out['timestamp'], [0.05, 0.04, 0.5, 0.49])
    metas = glue.make_img_metas([out['lidar2img']], (320, 800), timestamps=[out['timestamp']])
    assert metas[0]['pad_shape'] == [(320, 800, 3)] * 4 and metas[0]['img_shape'] == metas[0]['pad_shape']
    assert len(metas[0]['lidar2img']) == 4 and metas[0]['timestamp'] == out['timestamp']


def test_reshape_backbone_feats():
    f = [torch.arange(2 * 6 * 3 * 4 * 5, dtype=torch.float32).view(12, 3, 4, 5)]
    out = glue.reshape_backbone_feats(f, 2)
    assert out[0].shape == (2, 6, 3, 4, 5) and out[0].data_ptr() == f[0].data_ptr()
    assert torch.equal(out[0][1, 0], f[0][6])
