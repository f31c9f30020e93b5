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


def elem_stats(a: torch.Tensor, b: torch.Tensor, atol_frac: float = 1e-6):
    """Element-wise comparison against the reference tensor b: the error of every element in units of its OWN
    tolerance 1e-4 |b_i| + atol_frac max|b| (the north_star's "<= 1e-4 relative" with an absolute floor far below
    anything that matters), plus the 99.9th percentile of the plain relative error where |b_i| is above the floor."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    if b.numel() == 0:
        return dict(worst=0.0, p999_rel=0.0, n=0)
    scale = float(b.abs().max()) + 1e-300
    err = (a - b).abs()
    worst = float((err / (REL_TOL * b.abs() + atol_frac * scale)).max())
    big = b.abs() > 1e-3 * scale
    rel = (err[big] / b.abs()[big]) if bool(big.any()) else torch.zeros(1, dtype=torch.float64)
    k = max(int(math.ceil(0.999 * rel.numel())) - 1, 0)
    return dict(worst=worst, p999_rel=float(rel.sort().values[k]), n=int(b.numel()))


def assert_close_elem(a, b, what="", atol_frac: float = 1e-6):
    """|a_i - b_i| <= 1e-4 |b_i| + atol_frac max|b| for EVERY element (a max-norm test lets a small-magnitude
    element be 100 % wrong).  Prints the 99.9th-percentile relative error (visible with pytest -s / on failure)."""
    st = elem_stats(a, b, atol_frac)
    print(f"[parity] {what}: worst element at {st['worst']:.3f} of its tolerance, p99.9 relative error "
          f"{st['p999_rel']:.2e} over {st['n']} elements")
    assert st["worst"] <= 1.0, (f"{what}: an element is off by {st['worst']:.2f} x (1e-4 |b| + {atol_frac:.0e} max|b|); "
                                f"p99.9 relative error {st['p999_rel']:.2e}")
    return st


def to_dev(d, dev, dtype=torch.float32):
    return {k: (v.to(dev, dtype) if torch.is_tensor(v) and v.is_floating_point() else
                (v.to(dev) if torch.is_tensor(v) else v)) for k, v in d.items()}


def tile_grid(w, h):
    return math.ceil(w / 16), math.ceil(h / 16)


def threshold_pixel_mask(ref, gt_rgb, gt_depth, margin_tol, edge_tol=1e-6):
    """[H,W,1] float64 mask for ``batch["mask"]``: 0 at the (few) pixels where the fp32 kernels and the fp64 oracle
    may legitimately take DIFFERENT discrete decisions -- an alpha >= 1/255 or T <= 1e-4 test within ``margin_tol`` of
    its threshold (the oracle reports every pixel's margin), a colour within ``edge_tol`` of a clamp edge, a prediction
    within ``edge_tol`` of its target (the kink of |x - y|).  The mask multiplies both images and both depths
    (model.py:93-97 and the parent's loss), so such a pixel passes NO gradient on either side and every Gaussian --
    also those that share a tile with it -- can be compared element by element.  ``ref`` = oracle.splatfacto_outputs
    (..., return_margin=True)."""
    safe = ref["info"]["margin"][0] > margin_tol                                        # [H,W]
    pre = (ref["render"][0, ..., :3] + (1 - ref["accumulation"]) * ref["background"]).detach()
    near = torch.minimum(pre.abs(), (pre - 1).abs())
    edge = ((near < edge_tol) & (near > 0)).any(-1)        # (exactly 0: an empty pixel on a black background, same on both sides)
    kink = ((ref["rgb"].detach() - gt_rgb.to(pre.dtype)).abs() < edge_tol).any(-1)
    if ref.get("depth") is not None and gt_depth is not None:
        kink = kink | ((ref["depth"].detach() - gt_depth.to(pre.dtype)).abs() < edge_tol)[..., 0]
    return (safe & ~edge & ~kink)[..., None].to(torch.float64)


def sweep_case(case: int, scale_boost: float = 2.5, cam_k: int = 0):
    """Scene ``case`` of the randomised end-to-end sweep (scripts/parity_sweep.py draws exactly this; the fixed GPU
    tests re-create single cases of it).  Returns dict(sc, w, h, n, deg, mode, use_mask, gen, n_moved, pre): the two
    per-GAUSSIAN non-smooth points (the SH colour clamp max(0, c + 0.5), the Jacobian clamp at the frustum rim) are
    moved off their edge in the scene before either side runs; ``gen`` is the case's generator after the scene draws
    (the sweep draws its random mask from it)."""
    g = torch.Generator().manual_seed(2024 + 7919 * case)
    w = int(torch.randint(40, 260, (1,), generator=g)); h = int(torch.randint(33, 200, (1,), generator=g))
    n = int(torch.randint(200, 6000, (1,), generator=g))
    deg = int(torch.randint(0, 4, (1,), generator=g))
    mode = "antialiased" if case % 3 == 1 else "classic"
    use_mask = case % 4 == 2
    sc = O.synthetic_scene(n, w, h, seed=1000 + case, n_cameras=cam_k + 1)
    sc["camera_to_worlds"] = sc["camera_to_worlds"][cam_k:cam_k + 1]
    sc["Ks"] = sc["Ks"][:1]
    sc["scales"] = sc["scales"] + float(torch.rand(1, generator=g)) * scale_boost      # up to e^boost x larger splats
    sc["opacities"] = sc["opacities"] + (float(torch.rand(1, generator=g)) - 0.7) * 4
    K = sc["Ks"][0]
    pre = None
    with torch.no_grad():
        vm = O.get_viewmat(sc["camera_to_worlds"].double())
        campos = torch.linalg.inv(vm)[0, :3, 3]
        fx, fy, cx, cy = K[0, 0].item(), K[1, 1].item(), K[0, 2].item(), K[1, 2].item()
        lxp, lxn = (w - cx) / fx + 0.3 * 0.5 * w / fx, cx / fx + 0.3 * 0.5 * w / fx
        lyp, lyn = (h - cy) / fy + 0.3 * 0.5 * h / fy, cy / fy + 0.3 * 0.5 * h / fy
        n_moved = 0
        for _ in range(4):
            coeffs = torch.cat([sc["features_dc"].double()[:, None, :], sc["features_rest"].double()], dim=1)
            pre = O.eval_sh(deg, sc["means"].double() - campos, coeffs[:, : (deg + 1) ** 2]) + 0.5
            near_clamp = pre.abs().min(dim=-1).values < 1e-4
            pc = (vm[0, :3, :3] @ sc["means"].double().T).T + vm[0, :3, 3]
            rx, ry = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
            near_jac = ((rx - lxp).abs() < 1e-5) | ((rx + lxn).abs() < 1e-5) | ((ry - lyp).abs() < 1e-5) | ((ry + lyn).abs() < 1e-5)
            if not bool((near_clamp | near_jac).any()):
                break
            n_moved += int((near_clamp | near_jac).sum())
            sc["features_dc"][near_clamp] += 1e-2 / O.SH_C0
            sc["means"][near_jac] *= 1.0 + 1e-3
    return dict(sc=sc, w=w, h=h, n=n, deg=deg, mode=mode, use_mask=use_mask, gen=g, n_moved=n_moved, pre=pre)


def sweep_nonsmooth_pixels(out, sc, margin=1e-4, kink_scale=1.0):
    """[H,W] bool: pixels of an oracle forward (splatfacto_outputs(..., return_margin=True)) at which the fp32 kernels
    and the fp64 oracle may legitimately take DIFFERENT sides of a non-smooth point (what the sweeps put into
    batch["mask"]): an alpha / T decision within ``margin`` of its cut, a pre-clamp colour within 2e-6 of 0 or 1, a
    prediction within rounding of its target (the kinks of the two L1 terms)."""
    with torch.no_grad():
        bad = ~(out["info"]["margin"][0] > margin)
        pre_rgb = out["render"][0, ..., :3] + (1 - out["accumulation"]) * sc["background"].double()
        near = torch.minimum(pre_rgb.abs(), (pre_rgb - 1).abs())
        edge = ((near < 2e-6 * kink_scale) & (near > 0)).any(dim=-1)       # exactly 0 (empty pixel): same on both sides
        if out.get("depth") is not None:
            dd = (out["depth"] - sc["gt_depth"].double()).abs()[..., 0]
            edge |= (dd < 4e-6 * kink_scale * sc["gt_depth"].double()[..., 0].abs()) & (sc["gt_depth"][..., 0] > 0)
        edge |= ((out["rgb"] - sc["gt_rgb"].double()).abs() < 2e-6 * kink_scale).any(dim=-1)
    return bad | edge, int(edge.sum())
