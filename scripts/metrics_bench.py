#!/usr/bin/env python3
"""qed_image_metrics at 1080p (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from qed_splatter_amd.metrics import image_metrics, ssim_value  # noqa: E402

dev = torch.device("cuda:0")
H, W = 1080, 1920
pr, gr = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
pd, gd = torch.rand(H, W, 1, device=dev) * 5 + 0.2, torch.rand(H, W, 1, device=dev) * 5
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for _ in range(4):
    ev[0].record(); m = image_metrics(pr, gr, pd, gd); ev[1].record(); s = ssim_value(pr, gr); ev[2].record()
    torch.cuda.synchronize()
    print(f"image_metrics {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us  ssim_value {ev[1].elapsed_time(ev[2]) * 1e3:.1f} us", flush=True)
