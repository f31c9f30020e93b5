#!/usr/bin/env python3
"""The reference-shaped training loop of bench.py (api_path_ms) on its own: zero_grad -> get_outputs -> get_metrics_dict ->
get_loss_dict -> sum -> backward -> six per-group optimisers, eager dispatch, config B.  For rocprofv3 runs.

    python scripts/api_path_bench.py [steps] [qed|torch]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
kind = sys.argv[2] if len(sys.argv) > 2 else "qed"
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
args = argparse.Namespace(gaussians=500_000, width=1920, height=1080, steps=steps, warmup=10)
sc = bench.make_scene(args.gaussians, args.width, args.height, 0, dev)
host = {}
ms = bench.api_path_ms(args, sc, dev, kind, host=host)
print(f"{kind}: {ms:.3f} ms/step (eager, steps 10..{10 + steps} of training); host enqueue {host['enqueue_ms_per_step']:.3f} ms/step")
