"""Shared helpers for the parity tests (the oracle is the checker, never the thing under test)."""
from __future__ import annotations

import math

import torch

from oracle import splat_oracle as O

PARAM_NAMES = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")

# north_star tolerance: rendered RGB/depth and gradients within 1e-4 relative (fp32), measured
# against the largest magnitude of the reference tensor
REL_TOL = 1e-4


def scene(n, w, h, seed=1234, n_cameras=1, sh_degree=3):
    return O.synthetic_scene(n, w, h, seed=seed, n_cameras=n_cameras, sh_degree=sh_degree)


def activated(sc, dtype=torch.float64):
    """Arguments of the rasterization() call as the reference builds them (model.py:241,268-274)."""
    q = sc["quats"].to(dtype)
    return dict(
        means=sc["means"].to(dtype),
        quats=q / q.norm(dim=-1, keepdim=True),
        scales=torch.exp(sc["scales"].to(dtype)),
        opacities=torch.sigmoid(sc["opacities"].to(dtype)).squeeze(-1),
        colors=torch.cat((sc["features_dc"][:, None, :], sc["features_rest"]), dim=1).to(dtype),
        viewmats=O.get_viewmat(sc["camera_to_worlds"].to(dtype)),
        Ks=sc["Ks"].to(dtype),
    )


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a - b| / max |b|  (b = reference)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_close(a, b, tol=REL_TOL, what=""):
    e = max_rel(a, b)
    assert e <= tol, f"{what}: max-rel-err {e:.3e} > {tol:.1e}"
    return e


def to_dev(d, dev, dtype=torch.float32):
    return {k: (v.to(dev, dtype) if torch.is_tensor(v) and v.is_floating_point() else
                (v.to(dev) if torch.is_tensor(v) else v)) for k, v in d.items()}


def tile_grid(w, h):
    return math.ceil(w / 16), math.ceil(h / 16)
