#!/usr/bin/env python3
"""How many (Gaussian, tile) list entries could an exact-conservative test at binning time remove?
Counts, on a sample of the config-B scene: (a) gsplat's 3-sigma square, (b) that square intersected with
the bounding box of the alpha >= 1/255 ellipse, (c) tiles of (a) whose pixel-centre rectangle reaches
sigma <= ln(255 o) (what the compositing kernels' quadrant test does per 8x8 block)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

dev = torch.device("cuda:0")
L.load()
n, w, h = 500_000, 1920, 1080
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=1235).items()}
model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k] for k in
                         ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
model.step = 30000
model.eval()
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
with torch.no_grad():
    model.get_outputs(cam)
info = model.info
m2 = info["means2d"][0].cpu().double().numpy()
con = info["conics"][0].cpu().double().numpy()
op = info["opacities"][0].cpu().double().numpy()
rad = info["radii"][0].cpu().numpy()
vis = np.nonzero(rad > 0)[0]
rng = np.random.default_rng(0)
sample = rng.choice(vis, size=20000, replace=False)
TW, TH = (w + 15) // 16, (h + 15) // 16
tot_a = tot_b = tot_c = 0
per = []                                   # (tiles of (b), of those surviving the exact test) per sampled Gaussian
for i in sample:
    x, y, r = m2[i, 0], m2[i, 1], float(rad[i])
    a, b, c = con[i]
    tau = np.log(255.0 * op[i])
    x0, x1 = int(min(max(0, np.floor((x - r) / 16)), TW)), int(min(max(0, np.ceil((x + r) / 16)), TW))
    y0, y1 = int(min(max(0, np.floor((y - r) / 16)), TH)), int(min(max(0, np.ceil((y + r) / 16)), TH))
    na = (x1 - x0) * (y1 - y0)
    tot_a += na
    if tau <= 0 or na == 0:
        continue
    det = a * c - b * b
    ex, ey = np.sqrt(2 * tau * c / det), np.sqrt(2 * tau * a / det)      # half extents of sigma <= tau
    bx0, bx1 = max(x0, int(np.floor((x - ex) / 16))), min(x1, int(np.ceil((x + ex) / 16)))
    by0, by1 = max(y0, int(np.floor((y - ey) / 16))), min(y1, int(np.ceil((y + ey) / 16)))
    nb_ = max(bx1 - bx0, 0) * max(by1 - by0, 0)
    tot_b += nb_
    nc_ = 0
    # exact per tile: min of sigma over the pixel-centre rectangle [16tx+.5, 16tx+15.5] x [...]
    for ty in range(y0, y1):
        for tx in range(x0, x1):
            lx, hx, ly, hy = 16 * tx + 0.5, 16 * tx + 15.5, 16 * ty + 0.5, 16 * ty + 15.5
            in_b = bx0 <= tx < bx1 and by0 <= ty < by1
            if lx <= x <= hx and ly <= y <= hy:
                tot_c += 1
                nc_ += in_b
                continue
            best = np.inf
            for (fx, lo, hi, isx) in ((lx, ly, hy, True), (hx, ly, hy, True), (ly, lx, hx, False), (hy, lx, hx, False)):
                if isx:      # edge x = fx, y in [lo, hi]: minimise over y
                    dx = fx - x
                    ystar = y - (b / c) * dx
                    yy = min(max(ystar, lo), hi)
                    dy = yy - y
                else:
                    dy = fx - y
                    xstar = x - (b / a) * dy
                    xx = min(max(xstar, lo), hi)
                    dx = xx - x
                best = min(best, 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy)
            if best <= tau:
                tot_c += 1
                nc_ += in_b
    per.append((nb_, nc_))
print(f"sample {len(sample)}: 3-sigma square {tot_a} ({tot_a / len(sample):.2f}/Gaussian), + alpha bbox {tot_b} "
      f"({100 * tot_b / tot_a:.1f} %), exact per-tile test {tot_c} ({100 * tot_c / tot_a:.1f} %)")
per = np.array(per)
edges = [0, 1, 2, 4, 8, 16, 32, 64, 10 ** 9]
print("tiles of the alpha bbox per Gaussian -> share of the list, survivors of the exact test")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (per[:, 0] > lo) & (per[:, 0] <= hi)
    if m.any():
        print(f"  {lo + 1:>3}..{hi if hi < 10 ** 9 else 'inf':>3}: {m.sum():6d} Gaussians, {per[m, 0].sum():7d} entries "
              f"({100 * per[m, 0].sum() / per[:, 0].sum():5.1f} %), exact {per[m, 1].sum():7d} ({100 * per[m, 1].sum() / max(per[m, 0].sum(), 1):5.1f} %)")
w = per[: len(per) // 64 * 64, 0].reshape(-1, 64)
print(f"per 64 Gaussians: mean candidates {w.sum(1).mean():.0f}, largest single rectangle mean {w.max(1).mean():.1f}, "
      f"largest capped at 64 mean {np.minimum(w, 64).max(1).mean():.1f}")
