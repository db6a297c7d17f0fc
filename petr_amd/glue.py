"""Caller-side contract of the hot path: how the only non-tensor input of ``PETRHead.forward``
— ``img_metas[b]['lidar2img']`` (+ ``pad_shape`` / ``img_shape`` / ``timestamp``) — is produced from calibration,
and how backbone features are shaped for the head.  Host-side numpy / torch plumbing, mirroring (reference
projects/mmdet3d_plugin/):

* ``datasets/nuscenes_dataset.py:53-78``       calibration -> ``lidar2img`` / ``intrinsics`` / ``extrinsics``
* ``datasets/pipelines/transform_3d.py:314-324,398-401``   image resize / crop-flip-rotate (``ida_mat``) updates
* ``datasets/pipelines/transform_3d.py:517-548``           BEV rotation / scaling of the lidar frame
* ``datasets/pipelines/loading.py:60-118``                 multi-sweep (PETRv2) view + timestamp assembly
* ``models/detectors/petr3d.py:95-99``                     ``[B*N, C, H, W] -> [B, N, C, H, W]``

Conventions are the reference's, including the transposed storage of ``extrinsics`` (``lidar2cam_rt`` holds the
rotation transposed and the translation in its last ROW; ``lidar2img = intrinsics @ extrinsics.T``).
"""
import numpy as np
import torch


def lidar2img_from_calib(cam_intrinsic, sensor2lidar_rotation, sensor2lidar_translation):
    """nuscenes_dataset.py:56-69 for one camera.  Returns dict(lidar2img, intrinsics, extrinsics), 4x4 float64."""
    lidar2cam_r = np.linalg.inv(np.asarray(sensor2lidar_rotation, dtype=np.float64))
    lidar2cam_t = np.asarray(sensor2lidar_translation, dtype=np.float64) @ lidar2cam_r.T
    lidar2cam_rt = np.eye(4)
    lidar2cam_rt[:3, :3] = lidar2cam_r.T
    lidar2cam_rt[3, :3] = -lidar2cam_t
    intrinsic = np.asarray(cam_intrinsic, dtype=np.float64)
    viewpad = np.eye(4)
    viewpad[:intrinsic.shape[0], :intrinsic.shape[1]] = intrinsic
    return dict(lidar2img=viewpad @ lidar2cam_rt.T, intrinsics=viewpad, extrinsics=lidar2cam_rt)


def lidar2img_from_parts(intrinsics, extrinsics):
    """transform_3d.py:324,401: ``intrinsics[i] @ extrinsics[i].T`` for every view."""
    return [intrinsics[i] @ extrinsics[i].T for i in range(len(extrinsics))]


def resize_intrinsics(intrinsics, extrinsics, w_scale, h_scale):
    """transform_3d.py:314-324 (ResizeMultiview3D): focal lengths and principal point follow the image resize.
    Returns (new intrinsics, new lidar2img); inputs are not modified."""
    out = []
    for k in intrinsics:
        k = np.array(k, dtype=np.float64, copy=True)
        k[0, 0] *= w_scale
        k[0, 2] *= w_scale
        k[1, 1] *= h_scale
        k[1, 2] *= h_scale
        out.append(k)
    return out, lidar2img_from_parts(out, extrinsics)


def apply_ida(intrinsics, extrinsics, ida_mats):
    """transform_3d.py:398-401 (ResizeCropFlipImage): ``intrinsics[:3,:3] = ida_mat @ intrinsics[:3,:3]`` per view
    (``ida_mat``: the 3x3 image-space resize/crop/flip/rotate matrix of ``_img_transform`` :414-440)."""
    out = []
    for k, ida in zip(intrinsics, ida_mats):
        k = np.array(k, dtype=np.float64, copy=True)
        k[:3, :3] = np.asarray(ida, dtype=np.float64) @ k[:3, :3]
        out.append(k)
    return out, lidar2img_from_parts(out, extrinsics)


def _right_multiply_f32(lidar2img, mat_inv):
    # the reference does this in float32 torch and stores float32 numpy (transform_3d.py:526,545)
    return [(torch.tensor(m).float() @ mat_inv).numpy() for m in lidar2img]


def rotate_bev_along_z(lidar2img, angle):
    """transform_3d.py:517-529 (GlobalRotScaleTransImage): the lidar frame is rotated by ``angle`` about z."""
    rot_cos, rot_sin = torch.cos(torch.tensor(angle)), torch.sin(torch.tensor(angle))
    rot_mat = torch.tensor([[rot_cos, -rot_sin, 0, 0], [rot_sin, rot_cos, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return _right_multiply_f32(lidar2img, torch.inverse(rot_mat))


def scale_xyz(lidar2img, scale_ratio):
    """transform_3d.py:531-548: isotropic scaling of the lidar frame."""
    rot_mat = torch.tensor([[scale_ratio, 0, 0, 0], [0, scale_ratio, 0, 0], [0, 0, scale_ratio, 0], [0, 0, 0, 1]])
    return _right_multiply_f32(lidar2img, torch.inverse(rot_mat))


def append_sweep(results, sweep, sensors):
    """loading.py:104-118 (LoadMultiViewImageFromMultiSweepsFiles) without the image I/O: the previous sweep's views
    are appended to ``results['lidar2img' / 'intrinsics' / 'extrinsics']`` and ``results['timestamp']`` becomes the
    list of per-view time offsets ``lidar_timestamp - view_timestamp`` (seconds; current views first, :63-68)."""
    lidar_timestamp = results['timestamp']
    ts = [lidar_timestamp - t for t in results['img_timestamp']]
    ts.extend(lidar_timestamp - sweep[s]['timestamp'] / 1e6 for s in sensors)
    for s in sensors:
        results['lidar2img'].append(sweep[s]['lidar2img'])
        results['intrinsics'].append(sweep[s]['intrinsics'])
        results['extrinsics'].append(sweep[s]['extrinsics'])
    results['timestamp'] = ts
    return results


def make_img_metas(lidar2img_per_sample, pad_hw, img_hw=None, timestamps=None):
    """The dict ``PETRHead.forward`` reads (petr_head.py:288,311-312,383-388; petrv2_head.py:502)."""
    metas = []
    for b, mats in enumerate(lidar2img_per_sample):
        n = len(mats)
        ih, iw = img_hw if img_hw is not None else pad_hw
        m = {'pad_shape': [(pad_hw[0], pad_hw[1], 3)] * n, 'img_shape': [(ih, iw, 3)] * n,
             'lidar2img': [np.asarray(x) for x in mats]}
        if timestamps is not None:
            m['timestamp'] = list(timestamps[b])
        metas.append(m)
    return metas


def reshape_backbone_feats(img_feats, batch_size):
    """petr3d.py:95-99: every level ``[B*N, C, H, W] -> [B, N, C, H, W]`` (a view)."""
    out = []
    for f in img_feats:
        bn, c, h, w = f.size()
        out.append(f.view(batch_size, int(bn / batch_size), c, h, w))
    return out
