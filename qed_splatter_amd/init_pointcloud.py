"""Depth back-projection for the initial point cloud: the geometry of the reference's
``qed-init-pc`` tool (create_init_pointcloud.py:148-196) on the GPU.  File handling (transforms.json, PLY
caches, the pairwise on-disk merge of :100-145) stays with the reference tool; this module replaces the
per-frame Open3D calls: ``backproject_depth`` for ``create_from_depth_image`` and ``voxel_down_sample`` for
the method of the same name.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib as L


def frame_intrinsics(contents: dict, frame: dict):
    """(fx, fy, cx, cy) of one frame of a transforms.json: frame-level values win over the file's, and a missing ``fl_y``
    falls back to the FRAME's ``fl_x`` before the file's (create_init_pointcloud.py:49-56; pinned by
    tests/golden/reference_kats.npz ``ip_intrinsics_*``)."""
    fl_x = float(frame.get("fl_x", contents["fl_x"]))
    fl_y = float(frame.get("fl_y", contents.get("fl_y", fl_x)))
    return fl_x, fl_y, float(frame.get("cx", contents["cx"])), float(frame.get("cy", contents["cy"]))


@torch.no_grad()
def backproject_depth(depth: Tensor, fx: float, fy: float, cx: float, cy: float, c2w_opengl: Tensor,
                      depth_max: float = 100.0, stride: int = 1) -> Tensor:
    """depth [H,W] (metres; non-finite / non-positive = invalid) on the GPU, OpenGL camera-to-world [3,4] or
    [4,4] -> world points [n,3] in row-major pixel order.  One host read (the point count)."""
    lib = L.load()
    assert depth.is_cuda and depth.dim() == 2
    depth = depth.to(torch.float32).contiguous()
    H, W = depth.shape
    gw, gh = (W + stride - 1) // stride, (H + stride - 1) // stride
    cap = gw * gh
    pts = torch.empty(cap, 3, dtype=torch.float32, device=depth.device)
    n_pts = torch.zeros(1, dtype=torch.int32, device=depth.device)
    work = torch.empty(int(lib.qed_backproject_workspace_ints(H, W, stride)), dtype=torch.int32, device=depth.device)
    status = torch.zeros(4, dtype=torch.int32, device=depth.device)
    pose = [float(v) for v in torch.as_tensor(c2w_opengl, dtype=torch.float32).cpu()[:3, :4].reshape(-1)]
    h_pose = (C.c_float * 12)(*pose)
    L.check(lib.qed_backproject_depth(H, W, L.ptr(depth), float(fx), float(fy), float(cx), float(cy),
                                      C.cast(h_pose, C.c_void_p), float(depth_max), int(stride), cap, L.ptr(pts),
                                      L.ptr(n_pts), L.ptr(work), L.ptr(status), torch.cuda.current_stream().cuda_stream),
            "qed_backproject_depth")
    return pts[: int(n_pts)]


@torch.no_grad()
def voxel_down_sample(points: Tensor, voxel_size: float) -> Tensor:
    """One point per occupied voxel: the mean of its members (Open3D semantics).  Device-side torch ops
    (unique over voxel keys + index_add); offline tool, not a training hot path."""
    if points.shape[0] == 0:
        return points
    keys = torch.floor(points / voxel_size).to(torch.int64)
    keys = keys - keys.min(dim=0).values
    span = keys.max(dim=0).values + 1
    flat = (keys[:, 0] * span[1] + keys[:, 1]) * span[2] + keys[:, 2]
    _, inv, counts = torch.unique(flat, return_inverse=True, return_counts=True)
    out = torch.zeros(counts.shape[0], 3, dtype=points.dtype, device=points.device)
    out.index_add_(0, inv, points)
    return out / counts[:, None].to(points.dtype)
