#!/usr/bin/env python3
"""SSIM kernels alone at a given image size (development aid for rocprofv3 runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402

H, W = 1080, 1920
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
lib = L.load()
torch.manual_seed(0)
render = torch.rand(H, W, 4, device=dev)
alpha = torch.rand(H, W, 1, device=dev)
bg = torch.rand(3, device=dev)
gt = torch.rand(H, W, 3, device=dev)
maps = torch.empty(lib.qed_ssim_maps_floats(H, W), device=dev)
ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), device=dev)
v = torch.empty(H, W, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for it in range(iters):
    ev[0].record()
    L.check(lib.qed_ssim_fwd(H, W, 4, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(gt), None, L.ptr(maps), L.ptr(ssum), st), "f")
    ev[1].record()
    L.check(lib.qed_ssim_bwd(H, W, 4, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(gt), None, L.ptr(maps), -1e-7, None, L.ptr(v), st), "b")
    ev[2].record()
    torch.cuda.synchronize()
    print(f"fwd {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us  bwd {ev[1].elapsed_time(ev[2]) * 1e3:.1f} us", flush=True)
# the plain-image forms (get_loss_dict / the rgb_ssim metric): no compositing, [H,W,3] images
rgb = torch.rand(H, W, 3, device=dev)
for it in range(iters):
    ev[0].record()
    L.check(lib.qed_ssim_fwd(H, W, 3, L.ptr(rgb), None, None, L.ptr(gt), None, L.ptr(maps), L.ptr(ssum), st), "f")
    ev[1].record()
    L.check(lib.qed_ssim_bwd(H, W, 3, L.ptr(rgb), None, None, L.ptr(gt), None, L.ptr(maps), -1e-7, None, L.ptr(v), st), "b")
    ev[2].record()
    torch.cuda.synchronize()
    print(f"plain images: fwd {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us  bwd {ev[1].elapsed_time(ev[2]) * 1e3:.1f} us", flush=True)
