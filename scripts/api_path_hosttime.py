#!/usr/bin/env python3
"""Host cost of one step of the reference-shaped route: the same loop on a scene so small that the GPU is idle most of the
time (the step time is then what Python + PyTorch dispatch + the ctypes calls cost on this box)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
for kind, sep in (("qed", False), ("qed", True), ("torch", False)):
    args = argparse.Namespace(gaussians=4000, width=256, height=192, steps=300, warmup=20)
    sc = bench.make_scene(args.gaussians, args.width, args.height, 0, dev)
    host = {}
    ms = bench.api_path_ms(args, sc, dev, kind, separate_params=sep, host=host)
    print(f"{kind:5s} separate={int(sep)}: {ms:.3f} ms/step at 4 000 Gaussians @ 256x192 (host enqueue {host['enqueue_ms_per_step']:.3f})")
