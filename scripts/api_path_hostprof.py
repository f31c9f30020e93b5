#!/usr/bin/env python3
"""Where the HOST time of the reference-shaped route goes: cProfile over steady-state steps on a scene so small that the
GPU never limits (scripts/api_path_hosttime.py)."""
import cProfile
import functools
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from qed_splatter_amd.model import (FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig, QedAdam,  # noqa: E402
                                    exponential_decay_lr)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
n, w, h = 4000, 256, 192
sc = bench.make_scene(n, w, h, 0, dev)
cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1)
model = QEDSplatterModel(cfg, **{k: sc[k].clone() for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
model.step = 30000
model.train()
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
lrs = FlatAdam.DEFAULT_LRS
opts = {k: QedAdam([model.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in
        ("means", "features_dc", "features_rest", "opacities", "scales", "quats")}
sched = torch.optim.lr_scheduler.LambdaLR(opts["means"], lambda s: exponential_decay_lr(s, lrs["means"], 1.6e-6, 30000) / lrs["means"])


def step():
    for o in opts.values():
        o.zero_grad(set_to_none=True)
    outputs = model.get_outputs(cam)
    metrics = model.get_metrics_dict(outputs, batch)
    loss_dict = model.get_loss_dict(outputs, batch, metrics)
    functools.reduce(torch.add, loss_dict.values()).backward()
    for o in opts.values():
        o.step()
    sched.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
print(f"per-step figures = the totals below / {steps}")
st.sort_stats("cumulative").print_stats(60)
st.sort_stats("tottime").print_stats(40)
