#!/usr/bin/env python3
"""One training step through the REFERENCE-SHAPED interface -- get_outputs() -> get_loss_dict() -> sum of the
losses -> backward() -> optimiser step, eager dispatch, torch doing the composite / clamp / depth fix-up / L1 as
in model.py:295-308 and :87-116 -- beside the fused step bench.py times.  What a user gets from the two-line
change of INTEGRATION.md before adopting fused_loss / hipGraph replay.

    python scripts/api_path_bench.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, w, h = 500_000, 1920, 1080
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
L.load()
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=1235).items()}
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
init = {k: sc[k] for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")}


def run(kind):
    model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1, background_color="black"), **init)
    model.step = 30000
    model.train()
    opt = FlatAdam(model, means_schedule=FlatAdam.MEANS_SCHEDULE)

    def step():
        for p in model.parameters():
            p.grad = None
        if kind == "api":
            out = model.get_outputs(cam)
            ld = model.get_loss_dict(out, batch)
            sum(ld.values()).backward()
            opt.step()
        else:
            model.backward_fused(model.fused_loss(cam, batch, sync=False, compact_sh_grad=True))
            opt.step(fused_sh=True)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for kind in ("api", "fused-eager"):
    print(f"{kind:12s} {run(kind):.3f} ms/step (eager dispatch, steps 5..{5 + steps} of training)")
