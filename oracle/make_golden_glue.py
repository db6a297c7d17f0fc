"""Generate tests/golden/glue_calib.npz from the reference source.  BUILD-CONTAINER ONLY (same rules as make_golden.py).

The calibration / augmentation arithmetic of the reference's data pipeline lives inside dataset and pipeline classes
that need mmdet3d; the few statements that matter are executed from the reference's own SOURCE TEXT (read in place,
nothing copied): ``nuscenes_dataset.py:56-69`` (calibration -> lidar2img), ``transform_3d.py:517-548``
(``rotate_bev_along_z`` / ``scale_xyz`` as written), ``transform_3d.py:314-324`` and ``:398-401`` (intrinsics
updates).  Inputs are seeded; outputs become the fixture petr_amd.glue is tested against.
"""
import os
import sys
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference/projects/mmdet3d_plugin'
OUT = os.path.join(ROOT, 'tests', 'golden')


def lines(rel, a, b):
    src = open(os.path.join(REF, rel)).read().split('\n')
    return textwrap.dedent('\n'.join(src[a - 1:b]))


def main():
    rng = np.random.RandomState(0)
    n = 6
    cams = []
    for i in range(n):
        yaw = rng.uniform(-np.pi, np.pi)
        c, s = np.cos(yaw), np.sin(yaw)
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]) @ np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
        cams.append(dict(sensor2lidar_rotation=rot, sensor2lidar_translation=rng.uniform(-1.5, 1.5, 3),
                         cam_intrinsic=np.array([[1266.4, 0, 816.3], [0, 1266.4, 491.5], [0, 0, 1.0]]) + rng.uniform(-5, 5, (3, 3)) * np.eye(3),
                         timestamp=1.5e15 + i * 1e4))
    # ---- nuscenes_dataset.py:56-69, the body of the per-camera loop ----
    body = lines('datasets/nuscenes_dataset.py', 57, 69)
    lidar2img, intr, extr = [], [], []
    for cam_info in cams:
        ns = dict(np=np, cam_info=cam_info, intrinsics=intr, extrinsics=extr, lidar2img_rts=lidar2img)
        exec(compile(body, 'nuscenes_dataset.py:57-69', 'exec'), ns)
    # ---- transform_3d.py:515-548: the two methods as written (they only touch results['lidar2img']) ----
    meth = lines('datasets/pipelines/transform_3d.py', 517, 548)
    ns = dict(torch=torch, np=np)
    exec(compile(meth, 'transform_3d.py:517-548', 'exec'), ns)
    res = {'lidar2img': [m.copy() for m in lidar2img]}
    ns['rotate_bev_along_z'](None, res, 0.3)
    rotated = [m.copy() for m in res['lidar2img']]
    ns['scale_xyz'](None, res, 1.07)
    scaled = [m.copy() for m in res['lidar2img']]
    # ---- transform_3d.py:314-324 (resize) and :398-401 (ida) on copies ----
    w_scale, h_scale = 0.88, 0.8711111
    results = {'intrinsics': [k.copy() for k in intr], 'extrinsics': extr}
    for i in range(n):
        exec(compile(lines('datasets/pipelines/transform_3d.py', 315, 318), 'transform_3d.py:315-318', 'exec'),
             dict(results=results, i=i, w_scale=w_scale, h_scale=h_scale))
    exec(compile(lines('datasets/pipelines/transform_3d.py', 324, 324), 'transform_3d.py:324', 'exec'), dict(results=results, range=range, len=len))
    resized_k, resized_l2i = [k.copy() for k in results['intrinsics']], [m.copy() for m in results['lidar2img']]
    idas = [np.array([[0.5 + 0.01 * i, 0.02, -30.0 + i], [-0.02, 0.5 + 0.01 * i, -100.0], [0, 0, 1.0]]) for i in range(n)]
    results = {'intrinsics': [k.copy() for k in intr], 'extrinsics': extr}
    for i in range(n):
        exec(compile(lines('datasets/pipelines/transform_3d.py', 398, 398), 'transform_3d.py:398', 'exec'),
             dict(results=results, i=i, ida_mat=idas[i]))
    exec(compile(lines('datasets/pipelines/transform_3d.py', 401, 401), 'transform_3d.py:401', 'exec'), dict(results=results, range=range, len=len))
    ida_k, ida_l2i = results['intrinsics'], results['lidar2img']
    np.savez_compressed(os.path.join(OUT, 'glue_calib.npz'),
                        rot=np.stack([c['sensor2lidar_rotation'] for c in cams]), trans=np.stack([c['sensor2lidar_translation'] for c in cams]),
                        K=np.stack([c['cam_intrinsic'] for c in cams]),
                        lidar2img=np.stack(lidar2img), intrinsics=np.stack(intr), extrinsics=np.stack(extr),
                        rotated=np.stack(rotated), scaled=np.stack(scaled), w_scale=w_scale, h_scale=h_scale,
                        resized_k=np.stack(resized_k), resized_l2i=np.stack(resized_l2i), idas=np.stack(idas),
                        ida_k=np.stack(ida_k), ida_l2i=np.stack(ida_l2i))
    print('glue_calib.npz written;', 'lidar2img[0] =\n', lidar2img[0])


if __name__ == '__main__':
    main()
