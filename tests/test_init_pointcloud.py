"""SURVEY 8(f) rank 4, second half: depth back-projection for the initial point cloud (qed-init-pc,
create_init_pointcloud.py:148-196).  CPU: the numpy oracle on closed-form cases.  GPU: the HIP path vs it."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import backproject_oracle as B


def _pose(seed):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    c2w = np.eye(4)
    c2w[:3, :3] = R
    c2w[:3, 3] = rng.normal(size=3) * 3
    return c2w


def test_oracle_closed_form():
    K = np.array([[100.0, 0, 2.0], [0, 50.0, 1.0], [0, 0, 1]])
    depth = np.zeros((3, 5)); depth[1, 2] = 4.0; depth[2, 4] = 2.0; depth[0, 0] = np.nan; depth[0, 1] = -1.0; depth[0, 2] = 200.0
    # identity OpenGL pose: camera looks down -Z, Y up -> OpenCV point (x, y, d) is world (x, -y, -d)
    pts = B.backproject_frame(depth, K, np.eye(4), depth_max=100.0)
    assert pts.shape == (2, 3)
    np.testing.assert_allclose(pts[0], [0.0, 0.0, -4.0], atol=1e-12)                 # principal point
    np.testing.assert_allclose(pts[1], [(4 - 2) * 2 / 100, -(2 - 1) * 2 / 50, -2.0], atol=1e-12)
    assert B.backproject_frame(np.zeros((4, 4)), K, np.eye(4)).shape == (0, 3)      # :169-171
    assert B.backproject_frame(depth, K, np.eye(4), depth_max=100.0, stride=2).shape == (1, 3)   # (1,2) is off the stride-2 grid
    w2c = B.opengl_c2w_to_opencv_w2c(_pose(1))
    assert np.allclose(w2c[:3, :3] @ w2c[:3, :3].T, np.eye(3), atol=1e-12)
    v = B.voxel_down_sample(np.array([[0.1, 0.1, 0.1], [0.3, 0.3, 0.3], [1.2, 0.0, 0.0]]), 1.0)
    assert sorted(map(tuple, np.round(v, 6))) == [(0.2, 0.2, 0.2), (1.2, 0.0, 0.0)]


def _sorted_rows(a):
    a = np.asarray(a, dtype=np.float64)
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,stride,seed", [(48, 64, 1, 0), (270, 480, 2, 1), (1080, 1920, 3, 2), (5, 7, 4, 3)])
def test_backproject_matches_oracle(cuda, h, w, stride, seed):
    from qed_splatter_amd.init_pointcloud import backproject_depth, voxel_down_sample
    rng = np.random.default_rng(seed)
    depth = rng.uniform(0.2, 12.0, size=(h, w)).astype(np.float32)
    depth[rng.uniform(size=(h, w)) < 0.1] = 0.0
    depth[rng.uniform(size=(h, w)) < 0.02] = np.nan
    depth[rng.uniform(size=(h, w)) < 0.02] = -3.0
    depth[rng.uniform(size=(h, w)) < 0.02] = 150.0                      # beyond depth_max
    K = np.array([[0.8 * w, 0, w / 2 - 0.3], [0, 0.82 * w, h / 2 + 0.7], [0, 0, 1.0]])
    c2w = _pose(seed)
    want = B.backproject_frame(depth, K, c2w, depth_max=100.0, stride=stride)
    got = backproject_depth(torch.from_numpy(depth).to(cuda), K[0, 0], K[1, 1], K[0, 2], K[1, 2], torch.from_numpy(c2w),
                            depth_max=100.0, stride=stride)
    assert got.shape == (want.shape[0], 3)
    # row-major pixel order on both sides
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-5)
    vd = voxel_down_sample(got, 0.25).cpu().numpy()
    vw = B.voxel_down_sample(got.cpu().numpy().astype(np.float64), 0.25)
    assert vd.shape == vw.shape

    def by_voxel(a):                                     # a voxel's mean lies inside it: the voxel index is a key
        k = np.floor(a.astype(np.float64) / 0.25).astype(np.int64)
        return a[np.lexsort(k.T[::-1])]
    np.testing.assert_allclose(by_voxel(vd), by_voxel(vw), atol=1e-4)


@pytest.mark.gpu
def test_backproject_no_valid_depth(cuda):
    from qed_splatter_amd.init_pointcloud import backproject_depth
    d = torch.full((32, 32), float("nan"), device=cuda)
    d[0, :4] = 0.0
    assert backproject_depth(d, 30.0, 30.0, 16.0, 16.0, torch.eye(4)).shape == (0, 3)


# ---- pinned by the reference's own helpers (tests/golden/make_reference_kats.py: create_init_pointcloud.py:49-70) ----
def _kats():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.npz"))


def test_pose_conversion_and_intrinsics_match_the_reference_vectors():
    """`_opengl_c2w_to_opencv_w2c` and `_frame_intrinsics` executed from the reference: the oracle's restatement of the
    first and the product's host helper for the second give the same numbers."""
    import json
    from qed_splatter_amd.init_pointcloud import frame_intrinsics
    k = _kats()
    for c2w, w2c in zip(k["ip_c2w_opengl"], k["ip_w2c_opencv"]):
        np.testing.assert_allclose(B.opengl_c2w_to_opencv_w2c(c2w), w2c.astype(np.float64), rtol=0, atol=2e-6)
    for (contents, frame), want in zip(json.loads(str(k["ip_intrinsics_cases"])), k["ip_intrinsics_out"]):
        fx, fy, cx, cy = frame_intrinsics(contents, frame)
        np.testing.assert_array_equal(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32), want)


@pytest.mark.gpu
def test_backprojected_points_return_to_their_pixels_through_the_reference_w2c(cuda):
    """qed_backproject_depth takes the OpenGL camera-to-world pose; the reference hands Open3D the OpenCV world-to-camera
    matrix it derives from it.  With the reference's OWN matrices (known-answer vectors): every point the kernel emits,
    carried back by that matrix, is the pinhole un-projection of its pixel ((u - cx) d / fx, (v - cy) d / fy, d)."""
    from qed_splatter_amd.init_pointcloud import backproject_depth
    k = _kats()
    h, w = 37, 53
    rng = np.random.default_rng(11)
    depth = rng.uniform(0.5, 9.0, size=(h, w)).astype(np.float32)
    depth[rng.uniform(size=(h, w)) < 0.15] = 0.0
    fx, fy, cx, cy = 61.5, 63.25, 26.0, 18.5
    vs, us = np.nonzero(depth > 0)
    d = depth[vs, us].astype(np.float64)
    cam = np.stack([(us - cx) * d / fx, (vs - cy) * d / fy, d], -1)
    for c2w, w2c in zip(k["ip_c2w_opengl"], k["ip_w2c_opencv"]):
        pts = backproject_depth(torch.from_numpy(depth).to(cuda), fx, fy, cx, cy, torch.from_numpy(c2w)).cpu().numpy()
        assert pts.shape == cam.shape
        back = pts.astype(np.float64) @ w2c[:3, :3].astype(np.float64).T + w2c[:3, 3].astype(np.float64)
        np.testing.assert_allclose(back, cam, rtol=0, atol=2e-4 * float(np.abs(c2w[:3, 3]).max() + 9.0))
