#!/usr/bin/env python3
"""What the HBM of the box at hand delivers to simple streaming kernels (SURVEY 8d asks for the vendor peak and a
measured device-to-device copy side by side): torch copy / fill / sum of a 1 GiB buffer, HIP-event timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda:0")
n = 1 << 28                                           # 2^28 floats = 1 GiB
a = torch.rand(n, device=dev)
b = torch.empty_like(a)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


gib = n * 4
t = timed(lambda: b.copy_(a))
print(f"copy  1 GiB -> 1 GiB : {t * 1e6:8.1f} us  {2 * gib / t / 1e12:.2f} TB/s (read + write)")
t = timed(lambda: b.fill_(1.0))
print(f"fill  1 GiB          : {t * 1e6:8.1f} us  {gib / t / 1e12:.2f} TB/s (write)")
t = timed(lambda: a.sum())
print(f"sum   1 GiB          : {t * 1e6:8.1f} us  {gib / t / 1e12:.2f} TB/s (read)")
