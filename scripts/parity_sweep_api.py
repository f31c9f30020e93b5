#!/usr/bin/env python3
"""Randomised parity sweep of the gsplat-compatible operator (rasterization(...), the call of model.py:267-288)
against the fp64 oracle: several cameras, RGB / RGB+D, SH degrees or plain colours, classic / antialiased,
backgrounds, near-plane and radius clipping -- the paths the fused training step does not take.  Integer outputs
(radii given, tile counts, sorted ids, offsets) must be identical; render / alpha on the oracle's safe pixels and
the input gradients of EVERY Gaussian (random upstream weights, zero on the oracle's threshold pixels on both sides:
such a pixel passes no gradient) within 1e-4 (the north_star tolerance) -- ``kept=1.00``."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import splat_oracle as O  # noqa: E402
from qed_splatter_amd.rasterization import rasterization  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
bad, t_start = 0, time.time()
only = int(os.environ["QED_SWEEP_CASE"]) if "QED_SWEEP_CASE" in os.environ else None
for case in (range(n_cases) if only is None else [only]):
    if time.time() - t_start > budget_s:
        print(f"time budget reached after {case} cases")
        break
    g = torch.Generator().manual_seed(99 + 104729 * case)
    ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))                       # noqa: E731
    w, h, n, C = ri(40, 220), ri(33, 170), ri(100, 3000), ri(1, 4)
    deg = [None, 0, 1, 2, 3][ri(0, 5)]
    mode = "antialiased" if ri(0, 3) == 0 else "classic"
    rmode = "RGB" if ri(0, 3) == 0 else "RGB+D"
    use_bg = ri(0, 2) == 1
    near = [0.01, 4.0][ri(0, 2)]                                                          # 4.0 clips the front of the scene
    rclip = [0.0, 3.0][ri(0, 2)]
    sc = O.synthetic_scene(n, w, h, seed=500 + case, n_cameras=C)
    sc["scales"] = sc["scales"] + float(torch.rand(1, generator=g)) * 2.0
    means, quats = sc["means"], sc["quats"]
    scales, opac = torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]).squeeze(-1)
    if deg is None:
        colors = torch.rand(n, 3, generator=g)
    else:
        colors = torch.cat([sc["features_dc"][:, None, :], sc["features_rest"]], dim=1)
    vm = O.get_viewmat(sc["camera_to_worlds"])
    ch = 3 if rmode == "RGB" else 4
    bgs = torch.rand(C, ch, generator=g) if use_bg else None
    names = ("means", "quats", "scales", "opacities", "colors")
    vals = dict(means=means, quats=quats, scales=scales, opacities=opac, colors=colors)
    gp = {k: v.to(dev).requires_grad_(True) for k, v in vals.items()}

    def gpu_call():
        return rasterization(viewmats=vm.to(dev), Ks=sc["Ks"].to(dev), width=w, height=h, render_mode=rmode,
                             sh_degree=deg, rasterize_mode=mode, backgrounds=bgs.to(dev) if use_bg else None,
                             near_plane=near, radius_clip=rclip, absgrad=True, **gp)
    # pass 1 (no gradients): the radii of this GPU run; the oracle's forward gives every pixel's margin to the nearest
    # alpha / transmittance threshold
    with torch.no_grad():
        _, _, info0 = gpu_call()
    op = {k: v.double().requires_grad_(True) for k, v in vals.items()}
    r_ref, a_ref, i_ref = O.rasterization(viewmats=vm.double(), Ks=sc["Ks"].double(), width=w, height=h, render_mode=rmode,
                                          sh_degree=deg, rasterize_mode=mode, near_plane=near, radius_clip=rclip,
                                          return_margin=True, radii_override=info0["radii"].cpu(), **op)
    safe = i_ref["margin"] > 1e-4                                             # [C,H,W]
    # pass 2: random upstream weights that are ZERO on the threshold pixels, on both sides -- such a pixel passes no
    # gradient, so every Gaussian (also those that share its tile) is compared
    wr = torch.rand(r_ref.shape, generator=g) * safe[..., None]
    wa = torch.rand(a_ref.shape, generator=g) * safe[..., None]
    render, alpha, info = gpu_call()
    ((render * wr.to(dev)).sum() + (alpha * wa.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(info["radii"], info0["radii"])
    if use_bg:                                              # gsplat: render += (1 - alpha) * background
        r_ref = r_ref + (1 - a_ref) * bgs.double()[:, None, None, :]
    l_ref = (r_ref * wr.double()).sum() + (a_ref * wa.double()).sum()
    if l_ref.requires_grad:                                 # nothing visible: the oracle's outputs are constants
        l_ref.backward()
    for v in op.values():
        if v.grad is None:
            v.grad = torch.zeros_like(v)
    ints_ok = (torch.equal(info["tiles_per_gauss"].cpu(), i_ref["tiles_per_gauss"]) and
               torch.equal(info["flatten_ids"].cpu(), i_ref["flatten_ids"].to(torch.int32)) and
               torch.equal(info["isect_offsets"].cpu(), i_ref["isect_offsets"].to(torch.int32)))
    e_r = float(((render.detach().cpu().double() - r_ref.detach()).abs() * safe[..., None]).max() / (r_ref.detach().abs().max() + 1e-30))
    e_a = float(((alpha.detach().cpu().double() - a_ref.detach()).abs() * safe[..., None]).max())
    keep = torch.ones(n, dtype=torch.bool)                  # every Gaussian is compared
    worst, where = 0.0, ""
    for k in names:
        a, b = gp[k].grad.cpu().double(), op[k].grad
        err = (a - b).abs().reshape(n, -1).amax(dim=1)
        err[~keep] = 0
        if keep.any():
            e = float(err.max() / (b[keep].abs().max() + 1e-30))
            if e > worst:
                worst, where = e, f"{k}[{int(err.argmax())}]"
    ok = ints_ok and e_r <= 1e-4 and e_a <= 1e-4 and worst <= 1e-4 and bool(keep.all())
    bad += not ok
    print(f"case {case:3d} {w:3d}x{h:3d} n={n:4d} C={C} deg={deg} {mode:11s} {rmode:5s} bg={int(use_bg)} near={near} clip={rclip} "
          f"visible={int((info['radii'] > 0).sum()):5d} M={info['flatten_ids'].numel():6d} ints={'ok' if ints_ok else 'DIFF'} "
          f"safe={float(safe.float().mean()):.4f} kept={float(keep.float().mean()):.2f} render={e_r:.1e} alpha={e_a:.1e} "
          f"grad={worst:.1e} {where} {'ok' if ok else 'VIOLATION'}", flush=True)
print(f"{bad} violation(s)")
sys.exit(1 if bad else 0)
