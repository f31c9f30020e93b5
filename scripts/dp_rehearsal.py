#!/usr/bin/env python3
"""Two-rank rehearsal of the data-parallel step on ONE GPU (gloo, both ranks on cuda:0):
allreduce_and_step (chunked, pipelined) and exchange_grads_compact (geometry all-reduce + gathered colour
gradients, rebuilt either before the optimiser or inside it, or with both collectives left in flight and the optimiser
run in two parts behind them) must leave the same parameters as
allreduce_flat_grad + step.

    QED_BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29533 scripts/dp_rehearsal.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from qed_splatter_amd import _lib as L  # noqa: E402
from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig  # noqa: E402
from qed_splatter_amd.parallel import (allreduce_and_step, allreduce_flat_grad, backward_with_early_gather,  # noqa: E402
                                       exchange_grads_compact, exchange_grads_compact_begin, sparse_message_capacity)
from qed_splatter_amd.scene import synthetic_scene  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
L.load()
n, w, h = 20000, 320, 208
sc = synthetic_scene(n, w, h, seed=4, n_cameras=world)
names = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")
K = sc["Ks"][0]
cam = PinholeCameras(sc["camera_to_worlds"][rank:rank + 1].to(dev), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
batch = {"image": sc["gt_rgb"].to(dev), "depth_image": sc["gt_depth"].to(dev)}
models = [QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k].to(dev) for k in names}) for _ in range(7)]
opts = [FlatAdam(m, means_schedule=FlatAdam.MEANS_SCHEDULE) for m in models]
for m in models:
    m.step = 30000
for step in range(4):
    for i, (m, o) in enumerate(zip(models, opts)):
        for p in m.parameters():
            p.grad = None
        losses_i = m.fused_loss(cam, batch, compact_sh_grad=(i >= 2))
        if i == 5:                              # the all-gather leaves between the compositing and the projection backward
            backward_with_early_gather(m, losses_i, world)
        else:
            m.backward_fused(losses_i)
        if i == 0:
            allreduce_flat_grad(m, world)
            o.step()
        elif i == 1:
            allreduce_and_step(m, o, world, n_chunks=4)
        elif i == 2:
            exchange_grads_compact(m, world)
            o.step()
        elif i == 3:                            # gathered views feed the optimiser directly (qed_adam_step_sh)
            exchange_grads_compact(m, world, rebuild=False)
            o.step(fused_sh=True)
        else:                                   # both collectives in flight, optimiser in two parts behind them
            # i == 6: the SPARSE colour-gradient message (only the rows this rank's camera saw; capacity agreed at set-up;
            # forced below the dense break-even here: the synthetic scene is 95 % visible)
            if i == 6 and step == 0:
                sparse_cap = sparse_message_capacity(m, world) or (m.num_points + 3) // 4 * 4
            ex = exchange_grads_compact_begin(m, world, sparse_cap=sparse_cap if i == 6 else None)
            ex.wait_views()
            o.step(fused_sh=True, part=1)
            ex.wait_geometry()
            o.step(fused_sh=True, part=2)
torch.cuda.synchronize()
# the two models see gradients that differ in the last bits (atomic summation order in the compositing
# backward), so "same" is up to that noise amplified by four Adam steps
same = bool(((models[0].flat_params - models[1].flat_params).abs() <= 1e-5 + 1e-4 * models[0].flat_params.abs()).all())
gathered = [torch.empty_like(models[1].flat_params) for _ in range(world)]
dist.all_gather(gathered, models[1].flat_params.detach())
replicas = all(torch.equal(gathered[0], t) for t in gathered)
same2 = bool(((models[0].flat_params - models[2].flat_params).abs() <= 1e-5 + 1e-4 * models[0].flat_params.abs()).all())
same3 = bool(((models[0].flat_params - models[3].flat_params).abs() <= 1e-5 + 1e-4 * models[0].flat_params.abs()).all())
gathered3 = [torch.empty_like(models[3].flat_params) for _ in range(world)]
dist.all_gather(gathered3, models[3].flat_params.detach())
replicas = replicas and all(torch.equal(gathered3[0], t) for t in gathered3)
print(f"rank {rank}: chunked == plain: {same}; compact exchange == plain: {same2}; views -> optimiser == plain: {same3}; "
      f"replicas identical: {replicas}", flush=True)
same4 = bool(((models[0].flat_params - models[4].flat_params).abs() <= 1e-5 + 1e-4 * models[0].flat_params.abs()).all())
gathered4 = [torch.empty_like(models[4].flat_params) for _ in range(world)]
dist.all_gather(gathered4, models[4].flat_params.detach())
replicas4 = all(torch.equal(gathered4[0], t) for t in gathered4)
print(f"rank {rank}: overlapped exchange (gather | SH part | all-reduce | leading part) == plain: {same4}; "
      f"replicas identical: {replicas4}", flush=True)
same5 = bool(((models[0].flat_params - models[5].flat_params).abs() <= 1e-5 + 1e-4 * models[0].flat_params.abs()).all())
gathered6 = [torch.empty_like(models[5].flat_params) for _ in range(world)]
dist.all_gather(gathered6, models[5].flat_params.detach())
replicas5 = all(torch.equal(gathered6[0], t) for t in gathered6)
print(f"rank {rank}: all-gather issued ahead of the projection backward == plain: {same5}; replicas identical: {replicas5}", flush=True)
same6 = bool(((models[0].flat_params - models[6].flat_params).abs() <= 1e-5 + 1e-4 * models[0].flat_params.abs()).all())
gathered7 = [torch.empty_like(models[6].flat_params) for _ in range(world)]
dist.all_gather(gathered7, models[6].flat_params.detach())
replicas6 = all(torch.equal(gathered7[0], t) for t in gathered7)
print(f"rank {rank}: sparse colour-gradient message (capacity {sparse_cap} rows of {n}) == plain: {same6}; replicas identical: "
      f"{replicas6}", flush=True)
same = same and same2 and same3 and same4 and same5 and same6
replicas = replicas and replicas4 and replicas5 and replicas6

# ---- one rank's frame overflows its intersection buffer: EVERY rank must skip that step ----------------
# Per-rank cameras make the list length rank dependent, so an overflow need not hit all ranks in the same step; the rank
# it hits renders an empty frame and skips its update on the device.  The overflow word travels in the gathered message
# (or a MAX all-reduce on the plain path) and every rank's optimiser launches take the maximum as their skip flag.
import warnings  # noqa: E402
from qed_splatter_amd.rasterization import _workspace  # noqa: E402
ws = _workspace(dev)
skip_ok = True
for i, (m, o) in ((0, (models[0], opts[0])), (4, (models[4], opts[4]))):
    def one_step(sync):
        for p in m.parameters():
            p.grad = None
        m.backward_fused(m.fused_loss(cam, batch, sync=sync, compact_sh_grad=(i == 4)))
        if i == 0:
            allreduce_flat_grad(m, world)
            o.step()
        else:
            ex = exchange_grads_compact_begin(m, world)
            ex.wait_views()
            o.step(fused_sh=True, part=1)
            ex.wait_geometry()
            o.step(fused_sh=True, part=2)
    one_step(True)                               # calibrates this shape: the next call may leave the count on the device
    torch.cuda.synchronize()
    before = m.flat_params.detach().clone()
    mom = (o.exp_avg.clone(), o.exp_avg_sq.clone())
    M = int(m.info["n_isects"])
    if rank == world - 1:
        ws.capacity = max(M // 3, 1024)          # (what a sudden change of this rank's view would amount to)
    one_step(False)                              # rank world-1 overflows: empty frame there, skipped step EVERYWHERE
    torch.cuda.synchronize()
    unchanged = torch.equal(m.flat_params, before) and torch.equal(o.exp_avg, mom[0]) and torch.equal(o.exp_avg_sq, mom[1])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        one_step(False)                          # the rank that overflowed regrows and reads M back; all ranks train
    torch.cuda.synchronize()
    warned = any("rendered empty" in str(c.message) for c in caught)
    trained = not torch.equal(m.flat_params, before)
    gathered5 = [torch.empty_like(m.flat_params) for _ in range(world)]
    dist.all_gather(gathered5, m.flat_params.detach())
    identical = all(torch.equal(gathered5[0], t) for t in gathered5)
    print(f"rank {rank}: forced overflow on rank {world - 1}, {'plain all-reduce' if i == 0 else 'compact exchange'}: step skipped on this "
          f"rank: {unchanged}; warned here: {warned}; next step trained: {trained}; replicas identical: {identical}", flush=True)
    skip_ok = skip_ok and unchanged and trained and identical and (warned == (rank == world - 1))
dist.destroy_process_group()
sys.exit(0 if same and replicas and skip_ok else 1)
