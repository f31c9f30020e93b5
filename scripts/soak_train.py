#!/usr/bin/env python3
"""Soak run: thousands of fused training steps with the reference's refinement schedule (densify / cull every
100 steps after 500, opacity reset at 3100), random cameras around the scene, async intersection counts --
checks that nothing goes non-finite, that N evolves sanely and that the async overflow protocol is never hit
silently."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.densify import DensifyConfig, Densifier  # noqa: E402
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

NAMES = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
dev = torch.device("cuda:0")
L.load()
n, w, h, n_cams = 30000, 640, 360, 12
sc = synthetic_scene(n, w, h, seed=5, n_cameras=n_cams)
K = sc["Ks"][0]
cfg = QEDSplatterModelConfig.synthetic(sh_degree_interval=1000)
gt_model = QEDSplatterModel(cfg, **{k: sc[k].to(dev) for k in NAMES})
gt_model.step = 10_000
gt_model.eval()
cams, batches = [], []
with torch.no_grad():
    for c in range(n_cams):
        cam = PinholeCameras(sc["camera_to_worlds"][c:c + 1].to(dev), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
        out = gt_model.get_outputs(cam)
        cams.append(cam)
        batches.append({"image": out["rgb"].contiguous(), "depth_image": out["depth"].contiguous()})
g = torch.Generator().manual_seed(1)
init = {k: sc[k].clone() for k in NAMES}
init["means"] += 0.02 * torch.randn(init["means"].shape, generator=g)
init["features_dc"] += 0.3 * torch.randn(init["features_dc"].shape, generator=g)
init["features_rest"] *= 0.0
model = QEDSplatterModel(cfg, **{k: v.to(dev) for k, v in init.items()})
opt = FlatAdam(model, means_schedule=FlatAdam.MEANS_SCHEDULE)
dens = Densifier(model, opt, DensifyConfig(densify_grad_thresh=0.004), num_train_data=n_cams, seed=0)
t0 = time.time()
n_hist, worst = [], 0.0
for step in range(steps):
    model.step = step
    c = int(torch.randint(0, n_cams, (1,), generator=g))
    for p in model.parameters():
        p.grad = None
    try:
        losses = model.fused_loss(cams[c], batches[c], sync=(step % 100 == 0), compact_sh_grad=True,
                                  frame_key=c)     # async between refinements; the camera index keeps a launch order per camera
    except L.QedSplatError as e:                                                       # async overflow protocol
        print(f"step {step}: {e}")
        losses = model.fused_loss(cams[c], batches[c], sync=True, compact_sh_grad=True, frame_key=c)
    model.backward_fused(losses)
    opt.step(fused_sh=True)
    dens.after_train(step)
    if step % dens.config.refine_every == 0:
        info = dens.refinement_after(step)
        if info["did_densify"] or info["n_culled"] or info["opacity_reset"]:
            n_hist.append((step, info["n_before"], info["n_after"], info["opacity_reset"]))
    if step % 250 == 0 or step == steps - 1:
        lv = float(losses["loss"].detach())
        assert math.isfinite(lv), (step, lv)
        assert bool(torch.isfinite(model.flat_params).all()), step
        print(f"step {step:5d} loss {lv:.5f} N {model.num_points} sh_deg {min(step // 1000, 3)} {time.time() - t0:.1f}s", flush=True)
print("refinements (step, N before, N after, opacity reset):", n_hist[:6], "...", n_hist[-3:])
print("soak OK")
