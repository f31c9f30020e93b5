"""TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.

numpy restatement of the geometry of ``backproject_frame`` in the reference's ``qed-init-pc`` tool
(/root/reference/qed_splatter/create_init_pointcloud.py:148-196):

  * depth cleaning (:164-167): non-finite and non-positive depths become 0; a frame with no positive depth
    yields no cloud (:169-171);
  * ``_opengl_c2w_to_opencv_w2c`` (:61-70): negate the camera's Y and Z columns, invert;
  * ``o3d.t.geometry.PointCloud.create_from_depth_image(depth, K, w2c, depth_scale=1.0, depth_max, stride,
    with_normals=False)`` -- third-party Open3D (un-vendored): every stride-th pixel (u, v) whose depth d
    satisfies 0 < d < depth_max is unprojected to the camera point ((u - cx) d / fx, (v - cy) d / fy, d) and
    carried to the world by the inverse extrinsic.

PARITY UNPINNED for the Open3D part (open3d is not installed here and the reference holds no fixture for
it); the pose conversion is plain linear algebra.  Point order is not part of the contract (Open3D compacts
in parallel); tests compare sorted point sets.
"""
from __future__ import annotations

import numpy as np


def opengl_c2w_to_opencv_w2c(c2w_opengl: np.ndarray) -> np.ndarray:
    c2w = np.eye(4, dtype=np.float64)
    c2w[:3, :4] = np.asarray(c2w_opengl, dtype=np.float64)[:3, :4]
    c2w[:3, 1:3] *= -1
    return np.linalg.inv(c2w)


def backproject_frame(depth: np.ndarray, K: np.ndarray, c2w_opengl: np.ndarray, depth_max: float = 100.0,
                      stride: int = 1) -> np.ndarray:
    """-> [n,3] float64 world points in row-major pixel order (empty when no depth is valid)."""
    depth = np.array(depth, dtype=np.float64)
    depth[~np.isfinite(depth)] = 0.0
    depth[depth <= 0.0] = 0.0
    if not np.any(depth > 0.0):
        return np.zeros((0, 3))
    w2c = opengl_c2w_to_opencv_w2c(c2w_opengl)
    c2w = np.linalg.inv(w2c)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    vs, us = np.meshgrid(np.arange(0, depth.shape[0], stride), np.arange(0, depth.shape[1], stride), indexing="ij")
    d = depth[vs, us]
    ok = (d > 0.0) & (d < depth_max)
    u, v, d = us[ok].astype(np.float64), vs[ok].astype(np.float64), d[ok]
    pts_cam = np.stack([(u - cx) * d / fx, (v - cy) * d / fy, d, np.ones_like(d)], axis=-1)
    return (pts_cam @ c2w.T)[:, :3]


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    """Open3D voxel_down_sample on positions: one point per occupied voxel = the mean of its members
    (voxel index = floor(p / voxel_size)); output order unspecified."""
    if len(points) == 0:
        return points
    keys = np.floor(points / voxel_size).astype(np.int64)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    n = inv.max() + 1
    out = np.zeros((n, 3))
    np.add.at(out, inv, points)
    return out / np.bincount(inv, minlength=n)[:, None]
