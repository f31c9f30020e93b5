#!/usr/bin/env python3
"""How much work each stage of the compositing kernels does at a given size (diagnostic library built with
`python -m qed_splatter_amd.build --variant stats -DQED_COMPOSITE_STATS`).

    QED_SPLAT_LIB=qed_splatter_amd/lib/libqed_splat_stats.so python scripts/composite_stats.py [gaussians width height]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

n, w, h = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500_000, 1920, 1080)
dev = torch.device("cuda:0")
lib = L.load()
fn = lib._cdll.qed_debug_composite_stats
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=1235).items()}
model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k] for k in
                         ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
model.step = 30000
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
buf = (C.c_ulonglong * 32)()
for it in range(2):
    for p in model.parameters():
        p.grad = None
    fn(C.cast(buf, C.c_void_p), 1)
    losses = model.fused_loss(cam, batch, sync=True, compact_sh_grad=True)
    model.backward_fused(losses)
    fn(C.cast(buf, C.c_void_p), 1)
s = list(buf)
M = int(model.info["n_isects"])
print(f"N={n} {w}x{h} M={M} tiles={((w + 15) // 16) * ((h + 15) // 16)}")
names = {0: "fwd batches", 1: "fwd entries staged", 2: "fwd entries surviving the cull (any quadrant)", 3: "fwd quadrant visits",
         4: "fwd Gaussians visited", 5: "fwd accepted (pixel, Gaussian) pairs", 6: "fwd quadrant visits that accept no pixel", 8: "bwd batches", 9: "bwd entries staged", 10: "bwd entries surviving the cull",
         11: "bwd quadrant visits", 12: "bwd reductions (Gaussians with a valid pixel)", 13: "bwd flush groups (<= 4 rows each)",
         14: "bwd valid (pixel, Gaussian) pairs", 15: "bwd quadrant visits without a valid pixel", 16: "bwd whole-tile waves with work", 17: "bwd quadrant waves with work"}
for i, nm in names.items():
    print(f"  {nm:52s} {s[i]:12d}")
if s[12]:
    print(f"  bwd quadrant visits per reduction: {s[11] / s[12]:.2f}; valid pixels per quadrant visit: {s[14] / max(s[11], 1):.1f} of 64")
