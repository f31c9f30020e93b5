#!/usr/bin/env python3
"""Time one refinement (classify + scan + emit) and the per-step statistics pass at config-B size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.densify import DensifyConfig, Densifier  # noqa: E402
from qed_splatter_amd.model import FlatAdam, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

dev = torch.device("cuda:0")
L.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
sc = synthetic_scene(n, 1920, 1080, seed=1235)
for rep in range(3):
    model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(), **{k: sc[k].to(dev) for k in
                                                          ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
    opt = FlatAdam(model)
    model.last_size = (1080, 1920)
    dz = Densifier(model, opt, DensifyConfig(), num_train_data=10)
    g = torch.Generator(device=dev).manual_seed(rep)
    dz.vis_counts = torch.randint(1, 40, (n,), device=dev, generator=g).float()
    dz.xys_grad_norm = torch.rand(n, device=dev, generator=g) * 2e-6 * dz.vis_counts
    dz.max_2Dsize = torch.rand(n, device=dev, generator=g) * 0.1
    L.TIMER.active = True
    L.TIMER.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = dz.refinement_after(700)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ks = L.TIMER.summary()
    print(f"refinement: {1e3 * (t1 - t0):.3f} ms wall  {info}")
    print("   " + "  ".join(f"{k[4:]}={v[1] * 1e3:.1f}us" for k, v in sorted(ks.items())), flush=True)
