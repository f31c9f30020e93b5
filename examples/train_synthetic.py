#!/usr/bin/env python3
"""End-to-end use of the package on one GPU: fused training step (forward, (1-l) L1 + l (1-SSIM) + depth-L1
loss, backward, fused Adam with the reference's learning rates and "means" schedule), densification /
culling on the reference's schedule, and evaluation metrics without per-step host synchronisation.

    python examples/train_synthetic.py --gaussians 20000 --width 640 --height 360 --steps 700

The scene is the synthetic generator of SURVEY 8(d); the ground truth is rendered from the scene itself
and training starts from perturbed parameters, so the loss has somewhere to go.
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--steps", type=int, default=700)
    ap.add_argument("--log-every", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L.load()
    sc = synthetic_scene(a.gaussians, a.width, a.height, seed=a.seed)
    K = sc["Ks"][0]
    cam = PinholeCameras(sc["camera_to_worlds"].to(dev), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                         a.width, a.height)
    cfg = QEDSplatterModelConfig.synthetic(sh_degree_interval=1)
    # ground truth = a render of the unperturbed scene
    gt_model = QEDSplatterModel(cfg, **{k: sc[k].to(dev) for k in NAMES})
    gt_model.step = 10_000
    gt_model.eval()
    with torch.no_grad():
        gt = gt_model.get_outputs(cam)
    batch = {"image": gt["rgb"].contiguous(), "depth_image": gt["depth"].contiguous()}
    g = torch.Generator().manual_seed(a.seed + 1)
    init = {k: sc[k].clone() for k in NAMES}
    init["means"] += 0.01 * torch.randn(init["means"].shape, generator=g)
    init["features_dc"] += 0.3 * torch.randn(init["features_dc"].shape, generator=g)
    init["opacities"] -= 1.0
    model = QEDSplatterModel(cfg, **{k: v.to(dev) for k, v in init.items()})
    model.step = 10_000
    opt = FlatAdam(model, means_schedule=FlatAdam.MEANS_SCHEDULE)
    dens = Densifier(model, opt, DensifyConfig(), num_train_data=1, seed=a.seed)
    t0 = time.time()
    for step in range(a.steps):
        model.step = 10_000 + step
        for p in model.parameters():
            p.grad = None
        # (frame_key: the camera's index -- the compositing forward reuses the launch order of this camera's last frame)
        losses = model.fused_loss(cam, batch, compact_sh_grad=True, frame_key=0)
        model.backward_fused(losses)
        opt.step(fused_sh=True)       # SH-coefficient gradients expanded inside the Adam pass
        dens.after_train(step)
        if step % dens.config.refine_every == 0:
            info = dens.refinement_after(step)
            if info["did_densify"] or info["n_culled"]:
                print(f"step {step}: refinement {info}")
        if step % a.log_every == 0 or step == a.steps - 1:
            model.eval()
            with torch.no_grad():
                md = model.get_metrics_dict(model.get_outputs(cam), batch)
            model.train()
            print(f"step {step:5d}  loss {float(losses['loss']):.5f} (main {float(losses['main_loss']):.5f} depth "
                  f"{float(losses['depth_loss']):.5f})  psnr {float(md['rgb_psnr']):.2f}  ssim {float(md['rgb_ssim']):.4f}  "
                  f"abs_rel {float(md['depth_abs_rel']):.4f}  N {md['gaussian_count']}  {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
