#!/usr/bin/env python3
"""Per-entry-point HIP-event timings of one training step at a given size (development aid).

    python scripts/stage_times.py --gaussians 500000 --width 1920 --height 1080 --iters 5
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1235)
    ap.add_argument("--async-m", action="store_true")
    ap.add_argument("--compact", action="store_true", help="QED_F_SH_GRAD_COMPACT in project_bwd")
    ap.add_argument("--fused-sh", action="store_true", help="compact + qed_adam_step_sh")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L.load()
    t0 = time.time()
    sc = synthetic_scene(a.gaussians, a.width, a.height, seed=a.seed)
    sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    print(f"scene built in {time.time() - t0:.1f}s", flush=True)
    model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1),
                             **{k: sc[k] for k in ("means", "scales", "quats", "opacities", "features_dc",
                                                   "features_rest")})
    model.step = 30000
    K = sc["Ks"][0].cpu()
    cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                         a.width, a.height)
    batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
    opt = FlatAdam(model)
    L.TIMER.active = True
    for it in range(a.iters):
        L.TIMER.reset()
        for p in model.parameters():
            p.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = model.fused_loss(cam, batch, sync=not (a.async_m and it > 0), compact_sh_grad=a.compact or a.fused_sh)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        losses["loss"].backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step(fused_sh=a.fused_sh)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ks = L.TIMER.summary()
        M = model.info["n_isects"]
        print(f"iter {it}: fwd {1e3 * (t1 - t0):.3f} ms  bwd {1e3 * (t2 - t1):.3f} ms  adam {1e3 * (t3 - t2):.3f} ms  "
              f"M={M} visible={int((model.info['radii'] > 0).sum())}", flush=True)
        print("   " + "  ".join(f"{k[4:]}={v[1] * 1e3:.1f}us" for k, v in sorted(ks.items())), flush=True)
    alpha = model.info["last_ids"]
    print("losses", float(losses["main_loss"]), float(losses["depth_loss"]), flush=True)


if __name__ == "__main__":
    main()
