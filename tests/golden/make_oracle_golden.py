#!/usr/bin/env python3
"""Golden input/output vectors for the rasterizer part of the hot path, produced by the fp64 CPU
oracle (oracle/splat_oracle.py).  The reference holds no fixtures for this path and gsplat cannot
run here, so these pin the ORACLE (regression) and give the HIP path fixed targets;
they are not independent evidence of gsplat parity.

    python tests/golden/make_oracle_golden.py      ->  tests/golden/oracle_small.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import splat_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz")
NAMES = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")
CASES = {
    # name: (N, W, H, seed, rasterize_mode, sh_degree_to_use, scale_boost)
    "classic_deg3": (48, 40, 24, 101, "classic", 3, 3.2),
    "antialiased_deg2": (64, 64, 48, 102, "antialiased", 2, 3.0),
    "rgb_only_colors": (32, 32, 32, 103, "classic", None, 3.4),
}


def run_case(name):
    n, w, h, seed, mode, deg, boost = CASES[name]
    sc = O.synthetic_scene(n, w, h, seed=seed)
    sc["scales"] = sc["scales"] + boost           # larger splats: several Gaussians per pixel at this size
    if deg is None:                               # config.sh_degree == 0: features_rest is [N,0,3] (model.py:261-265)
        sc["features_rest"] = sc["features_rest"][:, :0, :]
    ps = {k: sc[k].double().requires_grad_(True) for k in NAMES}
    out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                               ps["features_rest"], sc["camera_to_worlds"].double(), sc["Ks"].double(), w, h,
                               sc["background"].double(), sh_degree_to_use=deg, rasterize_mode=mode,
                               return_margin=True)
    l_rgb = O.main_loss(out["rgb"], sc["gt_rgb"].double(), 0.2)    # (1 - l) L1 + l (1 - SSIM)
    l_d = O.depth_l1_loss(out["depth"], sc["gt_depth"].double(), None, 0.2)
    (l_rgb + l_d).backward()
    info = out["info"]
    d = {f"in_{k}": sc[k].numpy() for k in NAMES}
    d.update(in_camera_to_worlds=sc["camera_to_worlds"].numpy(), in_Ks=sc["Ks"].numpy(),
             in_gt_rgb=sc["gt_rgb"].numpy(), in_gt_depth=sc["gt_depth"].numpy(),
             in_size=np.array([w, h]), in_mode=np.array(mode), in_deg=np.array(-1 if deg is None else deg))
    d.update(render=out["render"].detach().numpy(), rgb=out["rgb"].detach().numpy(),
             depth=out["depth"].detach().numpy(), accumulation=out["accumulation"].detach().numpy(),
             loss_rgb=np.float64(l_rgb.item()), loss_depth=np.float64(l_d.item()),
             radii=info["radii"].numpy(), means2d=info["means2d"].detach().numpy(),
             means2d_grad=info["means2d"].grad.numpy(), depths=info["depths"].detach().numpy(),
             conics=info["conics"].detach().numpy(), isect_ids=info["isect_ids"].numpy(),
             flatten_ids=info["flatten_ids"].numpy(), isect_offsets=info["isect_offsets"].numpy(),
             last_ids=info["last_ids"].numpy(), margin=info["margin"].numpy(),
             tiles_per_gauss=info["tiles_per_gauss"].numpy())
    d.update({f"grad_{k}": ps[k].grad.numpy() for k in NAMES})
    return {f"{name}/{k}": v for k, v in d.items()}


def main():
    allv = {}
    for name in CASES:
        allv.update(run_case(name))
        print(name, "M =", allv[f"{name}/flatten_ids"].shape[0],
              "visible =", int((allv[f"{name}/radii"] > 0).sum()),
              "mean alpha = %.3f" % allv[f"{name}/accumulation"].mean())
    np.savez_compressed(OUT, **allv)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
