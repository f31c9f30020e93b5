import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
dev = torch.device("cuda:0")
n, w, h = 500000, 1920, 1080
sc = bench.make_scene(n, w, h, 0, dev)
cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1)
m = QEDSplatterModel(cfg, **{k: sc[k].clone() for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
m.step = 30000; m.train()
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
for i in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = m.get_outputs(cam)
    (out["rgb"].sum() + out["depth"].sum()).backward()
    torch.cuda.synchronize()
    print(i, f"{(time.perf_counter() - t) * 1e3:.2f} ms", "segments:", len(m._segments.segments) if "_segments" in m.__dict__ else 0, flush=True)
