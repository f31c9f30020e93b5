#!/usr/bin/env python3
"""Where does K7's error in a dense scene come from?  (sweep case 112, `means` gradient 2.5e-4 off.)

The compositing backward alone, on the GPU's own projected splats, against (i) the fp64 oracle and (ii) the SAME oracle
evaluated in fp32 (the band an fp32 evaluation of these sums can be expected to land in), with a random upstream gradient
masked at the threshold pixels.  Knobs: QED_KEEP_T_FINAL=0 (T_final = 1 - alpha, gsplat's form), QED_SPLAT_LIB=<variant>
(e.g. the -DQED_K7_RCP_REFINE build).  Usage: dense_scene_diag.py [case ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import splat_oracle as O  # noqa: E402
from qed_splatter_amd import rasterization as R  # noqa: E402
from tests.util import activated, sweep_case, to_dev  # noqa: E402

R.KEEP_T_FINAL = os.environ.get("QED_KEEP_T_FINAL", "1") == "1"
dev = torch.device("cuda:0")
for case in ([int(a) for a in sys.argv[1:]] or [112]):
    cs = sweep_case(case)
    sc, w, h, n, deg, mode = cs["sc"], cs["w"], cs["h"], cs["n"], cs["deg"], cs["mode"]
    a = to_dev(activated(sc, torch.float32), dev)
    for k in ("means", "quats", "scales", "opacities", "colors"):
        a[k].requires_grad_(True)
    render, alpha, info = R.rasterization(
        means=a["means"], quats=a["quats"], scales=a["scales"], opacities=a["opacities"],
        colors=a["colors"][:, : (deg + 1) ** 2], viewmats=a["viewmats"], Ks=a["Ks"], width=w, height=h,
        render_mode="RGB+D", sh_degree=deg, absgrad=True, rasterize_mode=mode)
    M = int(info["n_isects"])

    def oracle(dt):
        m2 = info["means2d"].detach().cpu().to(dt).requires_grad_(True)
        con = info["conics"].detach().cpu().to(dt).requires_grad_(True)
        op = info["opacities"].detach().cpu().to(dt).requires_grad_(True)
        col = torch.cat([info["colors"].detach().cpu(), info["depths"].detach().cpu()[..., None]], -1).to(dt).requires_grad_(True)
        r, al, _, margin = O.composite_tiles(m2, con, col, op, w, h, 16, info["isect_offsets"].cpu(), info["flatten_ids"].cpu(),
                                             return_margin=True)
        return (m2, con, col, op), r, al, margin

    ins64, r64, a64, margin = oracle(torch.float64)
    g = torch.Generator().manual_seed(1)
    v_r = torch.randn(r64.shape, generator=g, dtype=torch.float64)
    v_a = torch.randn(a64.shape, generator=g, dtype=torch.float64)
    safe = (margin > 1e-5)[..., None].double()
    v_r, v_a = v_r * safe, v_a * safe
    (r64 * v_r).sum().add((a64 * v_a).sum()).backward()
    ins32, r32, a32, _ = oracle(torch.float32)
    (r32 * v_r.float()).sum().add((a32 * v_a.float()).sum()).backward()
    gpu_ins = [info["means2d"], info["conics"], info["colors"], info["opacities"], info["depths"]]
    grads = torch.autograd.grad((render * v_r.to(dev, torch.float32)).sum() + (alpha * v_a.to(dev, torch.float32)).sum(), gpu_ins)
    ref = [ins64[0].grad, ins64[1].grad, ins64[2].grad[..., :3], ins64[3].grad, ins64[2].grad[..., 3]]
    f32 = [ins32[0].grad, ins32[1].grad, ins32[2].grad[..., :3], ins32[3].grad, ins32[2].grad[..., 3]]
    tf = float(1 - a64.mean())
    print(f"case {case}: {w}x{h} n={n} {mode} M={M} mean T_final={tf:.3e} safe={float(safe.mean()):.4f} keep_t_final={R.KEEP_T_FINAL} "
          f"lib={os.path.basename(os.environ.get('QED_SPLAT_LIB', 'default'))}")
    for name, gg, b, f in zip(("v_means2d", "v_conics", "v_colors", "v_opac", "v_depths"), grads, ref, f32):
        s = float(b.abs().max())
        e = (gg.detach().cpu().double() - b).abs()
        ef = (f.double() - b).abs()
        i = int(e.reshape(-1).argmax())
        print(f"   {name:10s} max|ref|={s:.3e}  HIP err/max {float(e.max()) / s:.2e}   fp32-oracle err/max {float(ef.max()) / s:.2e}   "
              f"(at HIP's worst element: fp32-oracle {float(ef.reshape(-1)[i]) / s:.2e})")
