#!/usr/bin/env python3
"""A/B of K6 with a costliest-first launch order taken from an earlier frame of the same camera (development build with
-DQED_DEV_FWD_ORDER; QED_SPLAT_LIB must point at it).  Static config-B scene: the best case for the predictor."""
import ctypes as C
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
raw = C.CDLL(os.environ["QED_SPLAT_LIB"])               # (the same handle the binding holds: one copy of the hook's state)
raw.qed_dev_set_fwd_order.restype = None
raw.qed_dev_set_fwd_order.argtypes = [C.c_void_p, C.c_int64, C.c_int64]

out = model.fused_loss(cam, batch)
node = out["loss"].grad_fn.next_functions[0][0]
order = node.vsplat_holder[0]["order_ws"].clone()        # K7's order of this frame = the predictor for the next frame's K6
ref = model.info  # keep alive
T = order.numel() - 1
print("n_split", int(order[T]))


def time_k6(reps=40):
    # K6 alone through the fused forward under no_grad would skip tile_cost; time the whole forward's K6 with events around
    # the entry point instead: wrap the library call
    times = []
    fn = lib.qed_composite_fwd
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    state = {"i": 0}

    def wrapped(*a):
        e0, e1 = ev[state["i"]]
        e0.record()
        r = fn(*a)
        e1.record()
        state["i"] += 1
        return r
    lib.qed_composite_fwd = wrapped
    try:
        for _ in range(reps):
            model.fused_loss(cam, batch)
    finally:
        lib.qed_composite_fwd = fn
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]


base = time_k6()
img0 = model.fused_loss(cam, batch)["loss"].item()
for tail, split in ((0, -1), (1280, -1), (0, 160), (0, 320), (0, 640), (1280, 320), (640, 160), (0, 1000)):
    raw.qed_dev_set_fwd_order(order.data_ptr(), tail, split)
    t = time_k6()
    img1 = model.fused_loss(cam, batch)["loss"].item()
    print(f"order, split {split:4d}, tail {tail:5d}: K6 {t[0]:.1f} / {t[1]:.1f} us   (raster + quadrant tail: {base[0]:.1f} / {base[1]:.1f})   loss equal: {img0 == img1}")
raw.qed_dev_set_fwd_order(None, 0, -1)
print("back to product order:", time_k6())
