#!/usr/bin/env python3
"""Per-rank COMPUTE of the data-parallel optimiser step that grows with the number of ranks: the SH part
of the optimiser (qed_adam_step_sh) rebuilds every Gaussian's 48 coefficient gradients from ALL gathered views -- one
(3 colour gradients + view matrix) message per rank -- so its arithmetic and its reads grow with the rank count while the
link budget of DESIGN.md section 5 only prices the bytes on the wire.  Synthetic gathered messages (random colour gradients,
yawed cameras), one GPU, HIP events, median over `iters` launches.

    python scripts/adam_sh_views_bench.py [iters]
"""
import ctypes as C
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import FlatAdam, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
lib = L.load()
print("qed_adam_step_sh (SH groups: features_dc + features_rest, degree 3) with n gathered views; median / min us over "
      f"{iters} launches")
for n, sizes in ((500_000, (1, 2, 4, 8)), (2_000_000, (1, 4))):
    sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, 64, 64, seed=1235).items()}
    model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(), **{k: sc[k] for k in
                             ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
    opt = FlatAdam(model)
    b = model.group_begin
    g = torch.zeros_like(model.flat_params)
    nv, row = 3 * n, 3 * n + 20
    for n_views in sizes:
        recv = torch.randn(n_views, row, device=dev) * 1e-3
        for r in range(n_views):                                    # the cameras of bench.py's ranks: yawed by 5 degrees each
            a = math.radians(5.0 * r)
            vm = torch.tensor([[math.cos(a), 0, -math.sin(a), 0], [0, -1, 0, 0], [math.sin(a), 0, -math.cos(a), 0], [0, 0, 0, 1.0]])
            recv[r, nv:nv + 16] = vm.reshape(-1).to(dev)
        model.last_compact = True
        model.last_sh_degree = 3
        model.sh_views = (n_views, recv[:, nv:], row, recv, row, 1.0 / n_views)
        for p_, off in zip(model.parameters(), b):
            p_.grad = g[off:off + p_.numel()].view(p_.shape)
        times = []
        for it in range(iters + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            opt.step(device_state=True, fused_sh=True, part=1)
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                times.append(e0.elapsed_time(e1) * 1e3)
        read_mb = (24 * 48 * n + 12 * n * n_views + 12 * n) / 1e6
        print(f"  N = {n:>9,d}  views = {n_views}:  {statistics.median(times):7.1f} / {min(times):7.1f} us   "
              f"(algorithmic bytes {read_mb:6.0f} MB -> {read_mb / statistics.median(times):.2f} TB/s)")
