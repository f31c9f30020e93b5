#!/usr/bin/env python3
"""Development aid: capture get_outputs' segment step by step (QED_SEG_MODE=thread_local|global|relaxed, QED_SEG_RETAIN=0|1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import splat_oracle as O  # noqa: E402
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("QED_SEG_LEGACY", "0") != "1":
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
w, h, n = 200, 136, 6000
sc = O.synthetic_scene(n, w, h, seed=23)
cfg = QEDSplatterModelConfig.synthetic(sh_degree_interval=1, graph_segments=False)
m = QEDSplatterModel(cfg, **{k: sc[k].to(dev) for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
m.step = 100
m.train()
K = sc["Ks"][0]
cam = PinholeCameras(sc["camera_to_worlds"][:1].to(dev), K[0, 0], K[1, 1], K[0, 2], K[1, 2], w, h)
for _ in range(3):
    out = m.get_outputs(cam)
    (out["rgb"].sum() + out["depth"].sum()).backward()
torch.cuda.synchronize()
print("eager steps done", flush=True)
mode = os.environ.get("QED_SEG_MODE", "thread_local")
from qed_splatter_amd.rasterization import rasterization  # noqa: E402
from qed_splatter_amd import _lib as L  # noqa: E402
c2w = cam.camera_to_worlds.clone()
intr = cam.intrinsics_fxfycxcy().clone()
bg = torch.zeros(3, device=dev)
holder = []
host = torch.zeros(4, dtype=torch.int32).pin_memory()
slot = (host.numpy(), host.data_ptr())
g1 = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g1, capture_error_mode=mode):
    with torch.enable_grad():
        vm = torch.empty(1, 4, 4, device=dev)
        Ks = torch.empty(1, 3, 3, device=dev)
        render, alpha, info = rasterization(
            means=m.means, quats=m.quats, scales=m.scales, opacities=m.opacities, colors=m.features_dc, viewmats=vm, Ks=Ks,
            width=w, height=h, render_mode="RGB+D", sh_degree=3, absgrad=True,
            _flags=L.F_LOG_SCALES | L.F_LOGIT_OPAC | L.F_TIGHT_TILES, _sh_rest=m.features_rest, _sync=False, _c2w=(c2w, intr),
            _post_background=bg, _vsplat_holder=holder, _means2d_leaf=True, _capture_slot=slot)
print("forward captured", flush=True)
rgb, depth = info["post_rgb"], info["post_depth"]
v_rgb, v_depth = torch.ones_like(rgb), torch.ones_like(depth)
vsplat = torch.zeros(info["radii"].numel(), L.VSPLAT_FLOATS, device=dev)
params = [m.gauss_params[k] for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")]
g2 = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
print("capturing backward", flush=True)
with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode=mode):
    holder[:] = [vsplat]
    grads = torch.autograd.grad([rgb, depth], params, [v_rgb, v_depth], retain_graph=os.environ.get("QED_SEG_RETAIN", "1") == "1",
                                allow_unused=True)
print("backward captured", flush=True)
host[0] = -1
g1.replay()
g2.replay()
torch.cuda.synchronize()
print("replayed: M", int(host[0]), "grad means", float(grads[0].abs().sum()), "eager", float(m.means.grad.abs().sum()) / 3, flush=True)
