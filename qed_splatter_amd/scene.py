"""Synthetic random-Gaussian scenes of SURVEY.md section 8(d) (BASELINE.json configs): one CPU
``torch.Generator``, draws in a fixed order, so every consumer (bench, tests, oracle) sees the same
scene for the same seed.  Pure data generation: no rendering arithmetic lives here."""
from __future__ import annotations

import math
from typing import Dict

import torch
from torch import Tensor

SH_C0 = 0.28209479177387814


def synthetic_scene(n: int, width: int, height: int, seed: int, sh_degree: int = 3,
                    n_cameras: int = 1, dtype=torch.float32) -> Dict[str, Tensor]:
    """One CPU generator, draws in this fixed order."""
    g = torch.Generator().manual_seed(seed)
    tan30 = math.tan(math.radians(30.0))
    fx = 0.5 * width / tan30
    K = torch.tensor([[fx, 0.0, width / 2.0], [0.0, fx, height / 2.0], [0.0, 0.0, 1.0]])
    z = -(2.0 + 10.0 * torch.rand(n, generator=g))                          # U(-12,-2)
    x = (2.0 * torch.rand(n, generator=g) - 1.0) * z.abs() * tan30 * 1.05
    y = (2.0 * torch.rand(n, generator=g) - 1.0) * z.abs() * tan30 * 1.05 * (height / width)
    means = torch.stack([x, y, z], dim=-1)
    lo, hi = math.log(0.003), math.log(0.03)
    scales = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    quats = torch.randn(n, 4, generator=g)
    opac = -2.0 + 6.0 * torch.rand(n, 1, generator=g)
    fdc = (torch.rand(n, 3, generator=g) - 0.5) / SH_C0
    frest = 0.05 * torch.randn(n, (sh_degree + 1) ** 2 - 1, 3, generator=g)
    gt_rgb = torch.rand(height, width, 3, generator=g)
    gt_depth = 2.0 + 10.0 * torch.rand(height, width, 1, generator=g)
    gt_depth = torch.where(torch.rand(height, width, 1, generator=g) < 0.1, torch.zeros_like(gt_depth), gt_depth)
    c2ws = []
    for k in range(n_cameras):
        a = math.radians(5.0 * k)
        c2ws.append(torch.tensor([[math.cos(a), 0.0, math.sin(a), 0.0], [0.0, 1.0, 0.0, 0.0],
                                  [-math.sin(a), 0.0, math.cos(a), 0.0]]))
    out = dict(means=means, scales=scales, quats=quats, opacities=opac, features_dc=fdc,
               features_rest=frest, gt_rgb=gt_rgb, gt_depth=gt_depth,
               camera_to_worlds=torch.stack(c2ws), Ks=K[None].repeat(n_cameras, 1, 1),
               background=torch.zeros(3))
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in out.items()}
