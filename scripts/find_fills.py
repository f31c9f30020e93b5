#!/usr/bin/env python3
"""Which Python lines launch the small fill / copy kernels of one training step (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
L.load()
n, w, h = 50_000, 640, 360
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=1).items()}
model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k] for k in
                         ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
model.step = 30000
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
bg = torch.zeros(3, device=dev)
opt = FlatAdam(model)


def step():
    for p in model.parameters():
        p.grad = None
    losses = model.fused_loss(cam, batch, background=bg, sync=False)
    losses["loss"].backward()
    opt.step()


model.fused_loss(cam, batch, background=bg, sync=True)["loss"].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::ones_like", "aten::full"):
        st = [s for s in (e.stack or []) if "qed_splatter_amd" in s or "find_fills" in s][:3]
        print(e.name, getattr(e, "input_shapes", ""), " <- ", " | ".join(st))
