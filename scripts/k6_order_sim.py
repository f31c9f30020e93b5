#!/usr/bin/env python3
"""Would K6 gain from a costliest-first launch order predicted from the list LENGTH (known before the kernel runs)?
Greedy dispatch model (waves start in index order as slots free up; a wave's duration = its recorded work count) over the
real per-tile work counts of config B: raster order with the quadrant tail (the product), whole tiles ordered by list length,
whole tiles ordered by their true cost (the bound).  GPU needed only to obtain the counts."""
import heapq
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

NAMES = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")
dev = torch.device("cuda:0")
n, w, h = 500_000, 1920, 1080
sc = synthetic_scene(n, w, h, seed=1235)
K = sc["Ks"][0]
cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1)
cam = PinholeCameras(sc["camera_to_worlds"][:1].to(dev), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].to(dev), "depth_image": sc["gt_depth"].to(dev)}
model = QEDSplatterModel(cfg, **{k: sc[k].to(dev) for k in NAMES})
model.step = 3


def costs(shape):
    if shape is None:
        os.environ.pop("QED_COMPOSITE_WAVES", None)
    else:
        os.environ["QED_COMPOSITE_WAVES"] = shape
    out = model.fused_loss(cam, batch)
    node = out["loss"].grad_fn.next_functions[0][0]
    offs = model.info["isect_offsets"].flatten()
    return node.tile_cost.cpu().long(), offs.cpu().long(), int(model.info["n_isects"])


whole, offs, M = costs("tile")            # [T,4]: lane 0 = n_vis + 2 nb
quad, _, _ = costs("quadrant")            # [T,4]: n_vis_q + nb per quadrant
T = whole.shape[0]
lens = torch.diff(torch.cat([offs, torch.tensor([M])]))[:T]
cw = whole.sum(1)
print(f"tiles {T}  M {M}  whole-tile cost: mean {float(cw.float().mean()):.1f} max {int(cw.max())}  "
      f"corr(length, cost) = {float(torch.corrcoef(torch.stack([lens.float(), cw.float()]))[0, 1]):.3f}")


def makespan(durations, slots):
    heap = [0] * slots
    heapq.heapify(heap)
    end = 0
    for d in durations:
        t = heapq.heappop(heap) + d
        end = max(end, t)
        heapq.heappush(heap, t)
    return end


slots = 256 * 4 * 5
n_small = int(2.5 * slots / 4)
raster = list(range(T))                                     # (the XCD remap is a permutation of whole rows: same statistics)
prod = [int(cw[t]) for t in raster[:T - n_small]] + [int(quad[t, q]) for t in raster[T - n_small:] for q in range(4)]
ideal = float(cw.sum()) / slots
print(f"work / slot (whole tiles): {ideal:.0f}")
print(f"product  (raster + {n_small} tiles as quadrant waves): makespan {makespan(prod, slots)}  "
      f"(its own work / slot {sum(prod) / slots:.0f})")
print(f"raster, whole tiles only:                         makespan {makespan([int(c) for c in cw], slots)}")
by_len = torch.argsort(lens, descending=True)
print(f"whole tiles, longest list first:                  makespan {makespan([int(cw[t]) for t in by_len], slots)}")
by_cost = torch.argsort(cw, descending=True)
print(f"whole tiles, costliest first (bound):             makespan {makespan([int(cw[t]) for t in by_cost], slots)}")
for frac in (0.5, 1.0, 1.5):
    ns = int(frac * slots / 4)
    order = [int(t) for t in by_len]
    d = [int(cw[t]) for t in order[:T - ns]] + [int(quad[t, q]) for t in order[T - ns:] for q in range(4)]
    print(f"longest first + the {ns} shortest as quadrant waves:  makespan {makespan(d, slots)}")

for frac in (0.5, 1.0, 1.5, 2.0, 2.5):
    ns = int(frac * slots / 4)
    order = [int(t) for t in by_cost]
    d = [int(cw[t]) for t in order[:T - ns]] + [int(quad[t, q]) for t in order[T - ns:] for q in range(4)]
    print(f"costliest first + the {ns} cheapest as quadrant waves: makespan {makespan(d, slots)}  (work / slot {sum(d) / slots:.0f})")
# heavy tiles split first (K7's policy), then whole tiles costliest first
for frac in (0.25, 0.5, 1.0):
    ns = int(frac * slots / 4)
    order = [int(t) for t in by_cost]
    d = [int(quad[t, q]) for t in order[:ns] for q in range(4)] + [int(cw[t]) for t in order[ns:]]
    print(f"the {ns} costliest as quadrant waves first, then whole tiles costliest first: makespan {makespan(d, slots)}")
