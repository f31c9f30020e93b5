#!/usr/bin/env python3
"""Find which part of the step breaks hipGraph capture: runs each variant in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from qed_splatter_amd import _lib as L
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
from qed_splatter_amd.scene import synthetic_scene
from qed_splatter_amd.graph import GraphedTrainStep
variant = sys.argv[1]
dev = torch.device("cuda:0"); L.load()
import os; n, w, h = (int(x) for x in os.environ.get("PROBE_SIZE", "20000,320,240").split(","))
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=3).items()}
model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k] for k in ("means","scales","quats","opacities","features_dc","features_rest")})
model.step = 30000
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0,0]), float(K[1,1]), float(K[0,2]), float(K[1,2]), w, h)
batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
opt = FlatAdam(model)
bg = torch.zeros(3, device=dev)
from qed_splatter_amd.rasterization import rasterization
from qed_splatter_amd.model import get_viewmat
def step():
    for p in model.parameters(): p.grad = None
    if variant == "viewmat":
        return {"x": get_viewmat(cam.camera_to_worlds).sum()}
    if variant == "raster":
        vm = get_viewmat(cam.camera_to_worlds); Kd = cam.get_intrinsics_matrices()
        r, a, info = rasterization(model.means, model.quats, model.scales, model.opacities, model.features_dc, vm, Kd, w, h,
                                   render_mode="RGB+D", sh_degree=3, absgrad=True, _flags=L.F_LOG_SCALES | L.F_LOGIT_OPAC,
                                   _sh_rest=model.features_rest, _sync=False)
        return {"x": r.sum()}
    losses = model.fused_loss(cam, batch, background=bg, sync=False)
    if variant == "fwd": return losses
    losses["loss"].backward()
    if variant == "bwd": return losses
    opt.step(device_state=True)
    return losses
g = GraphedTrainStep(step, dev, warmup=2, check_every=0)
for _ in range(3): g.replay()
torch.cuda.synchronize()
print("OK", variant, {k: float(v) for k, v in g.outputs.items()})
''' % ROOT
for v in os.environ.get("PROBE_VARIANTS", "viewmat,raster,fwd,bwd,full").split(","):
    r = subprocess.run([sys.executable, "-c", CHILD, v], capture_output=True, text=True, timeout=120)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("OK") or "Error" in l or "Fatal" in l][-3:]
    print(f"== {v}: rc={r.returncode}", " | ".join(t[:160] for t in tail), flush=True)
