#!/usr/bin/env python3
"""The launch order the ordering job produces (XCD-aware since round 5: 8 regions of the image, each costliest-first,
interleaved in step with the block index) against a host-built order of the same kind -- and, with a library from before
that change, against the plain costliest-first order (K7 245 -> 238 us, K6 110 -> 108).  K6 through qed_composite_fwd's tile_order, K7 through the order the fused step hands it.
Static config-B scene; HIP events around the two entry points."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
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
lib = L.load()

out = model.fused_loss(cam, batch, frame_key=0)
node = out["loss"].grad_fn.next_functions[0][0]
slot = model._frame_orders[(0, h, w)]
plain = slot[0].clone()
T = plain.numel() - 1
n_split = int(plain[T])
cost = node.tile_cost.sum(1).cpu()
model.backward_fused(out)


def xcd_order(n_regions=8):
    per = (T + n_regions - 1) // n_regions
    lists = []
    for r in range(n_regions):
        ids = torch.arange(r * per, min((r + 1) * per, T))
        lists.append(ids[torch.argsort(cost[ids], descending=True)].tolist())
    # the heaviest n_split tiles keep their place at the front (quadrant waves), taken out of their regions' lists
    head = plain[:n_split].cpu().tolist()
    hs = set(head)
    lists = [[t for t in l if t not in hs] for l in lists]
    order, ptr = list(head), [0] * n_regions
    for i in range(n_split, T):
        r = (i + 3 * n_split) % n_regions
        for k in range(n_regions):                     # (a region that ran out: the next one)
            rr = (r + k) % n_regions
            if ptr[rr] < len(lists[rr]):
                order.append(lists[rr][ptr[rr]])
                ptr[rr] += 1
                break
    assert sorted(order) == list(range(T))
    return torch.tensor(order + [n_split], dtype=torch.int32, device=dev)


def timed(which, reps=30):
    ev = {"qed_composite_fwd": [], "qed_composite_bwd": []}
    saved = {k: getattr(lib, k) for k in ev}

    def wrap(name):
        fn = saved[name]

        def f(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a)
            e1.record()
            ev[name].append((e0, e1))
            return r
        return f
    for k in ev:
        setattr(lib, k, wrap(k))
    try:
        for _ in range(reps):
            for p in model.parameters():
                p.grad = None
            slot[0].copy_(which)                          # the forward pass reads it ...
            o = model.fused_loss(cam, batch, frame_key=0)
            slot[0].copy_(which)                          # ... and the backward pass (the loss launch rewrote it)
            model.backward_fused(o)
    finally:
        for k, fn in saved.items():
            setattr(lib, k, fn)
    torch.cuda.synchronize()
    res = {}
    for k, l in ev.items():
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in l)
        res[k] = (round(t[len(t) // 2], 1), round(t[0], 1))
    return res


xo = xcd_order()
for name, o in (("device order (XCD-aware)", plain), ("host-built interleaved order", xo), ("device order (XCD-aware)", plain), ("host-built interleaved order", xo)):
    print(f"{name:28s}", timed(o))
