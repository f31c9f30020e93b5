#!/usr/bin/env python3
"""Per-entry-point HIP-event times of the reference-shaped route (eager) beside the fused step's, same scene, same process.

    python scripts/api_vs_fused_kernels.py [steps]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from qed_splatter_amd import _lib as L  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
args = argparse.Namespace(gaussians=500_000, width=1920, height=1080, steps=steps, warmup=10)
sc = bench.make_scene(args.gaussians, args.width, args.height, 0, dev)

# the reference-shaped route, eager, timed launches
L.TIMER.reset()
args.warmup = 10
bench.api_path_ms(args, sc, dev, "qed", graph_segments=False)          # (warm: allocator, capacity)
L.TIMER.reset()
L.TIMER.active = True
bench.api_path_ms(args, sc, dev, "qed", graph_segments=False)
torch.cuda.synchronize()
L.TIMER.active = False
api = L.TIMER.summary()

# the fused step
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1)
model = QEDSplatterModel(cfg, **{k: sc[k].clone() for k in ("means", "scales", "quats", "opacities", "features_dc",
                                                            "features_rest")})
model.step = 30000
model.train()
K = sc["Ks"][0].cpu()
w, h = args.width, args.height
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
opt = FlatAdam(model)
for it in range(steps + args.warmup * 2):
    if it == args.warmup * 2:
        L.TIMER.reset()
        L.TIMER.active = True
    for p in model.parameters():
        p.grad = None
    losses = model.fused_loss(cam, batch, sync=(it == 0), compact_sh_grad=True)
    model.backward_fused(losses)
    opt.step(fused_sh=True)
torch.cuda.synchronize()
L.TIMER.active = False
fused = L.TIMER.summary()
names = sorted(set(api) | set(fused), key=lambda k: -(api.get(k, (0, 0))[1] + fused.get(k, (0, 0))[1]))
print(f"{'entry point':34s} {'api n':>6s} {'api us':>9s} {'fused n':>8s} {'fused us':>9s}   (n = launches per step)")
ta = tf = 0.0
for k in names:
    a, f = api.get(k, (0, 0.0)), fused.get(k, (0, 0.0))
    ta += a[0] * a[1] / steps
    tf += f[0] * f[1] / steps
    print(f"{k:34s} {a[0] / steps:6.2f} {a[1] * 1e3:9.1f} {f[0] / steps:8.2f} {f[1] * 1e3:9.1f}")
print(f"{'sum per step (ms)':34s} {'':6s} {ta:9.3f} {'':8s} {tf:9.3f}")
