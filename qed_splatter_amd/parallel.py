"""Camera-sharded data parallelism: one process per GPU, every rank holds the
full Gaussian set and renders its own camera; after backward ONE collective averages the flat
59 N-float gradient buffer (RCCL over xGMI; `nccl` backend == RCCL on ROCm), then every rank takes the
identical fused Adam step.  The reference itself is single-GPU, one camera per step
(/root/reference/qed_splatter/model.py:210-212); there is no collective to translate.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def allreduce_flat_grad(model, world_size: int, group=None) -> torch.Tensor:
    """Average the model's six parameter gradients across ranks with a single all-reduce.

    The gradients already alias one contiguous allocation (``model.flat_grad()``), so nothing is
    packed or copied.  Loss is normalised by the global camera count, i.e. gradients are averaged.
    """
    g = model.flat_grad()
    if g is None:
        raise RuntimeError("no gradients to reduce: call backward() first")
    if world_size <= 1:
        return g
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(g, op=dist.ReduceOp.AVG, group=group)      # ncclAvg: no extra scaling pass
    else:                                                            # gloo (CPU tests) has no AVG
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        g.mul_(1.0 / world_size)
    return g


def allreduce_densification_stats(xys_absgrad_norm: torch.Tensor, vis_counts: torch.Tensor,
                                  max_radii: torch.Tensor, group=None) -> None:
    """Reduce what Nerfstudio's densifier accumulates from model.py:289-292 side effects:
    sum of |means2d.absgrad| norms and visibility counts (SUM), largest screen radius (MAX)."""
    dist.all_reduce(xys_absgrad_norm, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vis_counts, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group)
