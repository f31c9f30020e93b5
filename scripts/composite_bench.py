#!/usr/bin/env python3
"""Median / min HIP-event time of every entry point over repeated forward+backward passes of the SAME scene (no
optimiser step, so the workload does not drift): the A/B tool for kernel variants (QED_SPLAT_LIB=... selects one).

    python scripts/composite_bench.py [iters] [gaussians width height [step]]   (step < 3: SH degree = step)
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, w, h = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (500_000, 1920, 1080)
dev = torch.device("cuda:0")
L.load()
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=1235).items()}
model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k] for k in
                         ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
model.step = int(sys.argv[5]) if len(sys.argv) > 5 else 30000
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
for it in range(iters + 3):
    if it == 3:
        L.TIMER.reset()
        L.TIMER.active = True
    for p in model.parameters():
        p.grad = None
    losses = model.fused_loss(cam, batch, sync=(it == 0), compact_sh_grad=True)
    model.backward_fused(losses)
torch.cuda.synchronize()
L.TIMER.active = False
out = []
for name, pairs in sorted(L.TIMER.events.items()):
    ms = [a.elapsed_time(b) * 1e3 for a, b in pairs]
    out.append(f"{name.replace('qed_', '')}={statistics.median(ms):.1f}/{min(ms):.1f}")
print(" ".join(out), "(median/min us)")
