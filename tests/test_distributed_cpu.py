"""World-size-2 `gloo` tests of the camera-sharded data-parallel path on CPU: the
flat-gradient all-reduce, the compact exchange (blocking and with both collectives left in flight), replica
consistency after the optimiser step, and the per-rank scene sharding bench.py uses."""
from __future__ import annotations

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import PARAM_NAMES, scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qed_splatter_amd.model import QEDSplatterModel
        from qed_splatter_amd.parallel import allreduce_densification_stats, allreduce_flat_grad
        sc = scene(50, 32, 32, seed=5)
        m = QEDSplatterModel(None, **{k: sc[k] for k in PARAM_NAMES})
        # rank-dependent gradients laid out as _ProjectSH.backward lays them out (one allocation)
        total = m.flat_params.numel()
        flat = torch.arange(total, dtype=torch.float32) * (rank + 1)
        off = 0
        for name in m.group_names:
            p = m.gauss_params[name]
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        g = allreduce_flat_grad(m, world)
        ok0 = int(m._dp_skip) == 0                                    # no rank overflowed: nobody skips
        want = torch.arange(total, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = ok0 and torch.allclose(g, want) and g.data_ptr() == m.means.grad.data_ptr()
        # identical update on every rank keeps the replicas bit-identical
        with torch.no_grad():
            m.flat_params.add_(g, alpha=-1e-3)
        gathered = [torch.empty_like(m.flat_params) for _ in range(world)]
        dist.all_gather(gathered, m.flat_params.detach())
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
        # densification statistics: SUM / SUM / MAX
        a = torch.full((50,), float(rank + 1))
        c = torch.full((50,), 1.0)
        r = torch.full((50,), float(10 * (rank + 1)))
        allreduce_densification_stats(a, c, r)
        ok = ok and float(a[0]) == sum(range(1, world + 1)) and float(c[0]) == world and float(r[0]) == 10 * world
        # Densifier.all_reduce_stats: vis_counts starts at ONE on every rank; exactly one "one" survives
        from qed_splatter_amd.densify import Densifier
        dz = Densifier.__new__(Densifier)
        dz.xys_grad_norm = torch.full((50,), 0.5 * (rank + 1))
        dz.vis_counts = torch.full((50,), 1.0 + 3 * (rank + 1))        # 1 + visits on this rank
        dz.max_2Dsize = torch.full((50,), 0.01 * (rank + 1))
        dz.all_reduce_stats()
        visits = 3 * sum(range(1, world + 1))
        ok = ok and float(dz.vis_counts[0]) == 1.0 + visits and abs(float(dz.max_2Dsize[0]) - 0.01 * world) < 1e-9
        ok = ok and abs(float(dz.xys_grad_norm[0]) - 0.5 * sum(range(1, world + 1))) < 1e-6
        # compact exchange, both collectives left in flight (what bench.py runs for N > 1) == the blocking form:
        # averaged geometry gradients, one (colour gradients + view matrix) message per rank, scale 1 / world
        from qed_splatter_amd import parallel as P
        from qed_splatter_amd.parallel import exchange_grads_compact, exchange_grads_compact_begin
        # ... and the optimiser's skip decision is collective: ONE rank's frame overflowed its intersection buffer (its
        # overflow word holds the count it needed) -> every rank's step takes the same non-zero skip word
        P._CPU_OVERFLOW_WORD[0] = 77 if rank == world - 1 else 0
        allreduce_flat_grad(m, world)
        ok = ok and int(m._dp_skip) == 77
        results = []
        for overlapped in (False, True, "prepared"):
            m2 = QEDSplatterModel(None, **{k: sc[k] for k in PARAM_NAMES})
            flat2 = (torch.arange(total, dtype=torch.float32) + 1.0) * (rank + 1)
            off = 0
            for name in m2.group_names:
                p = m2.gauss_params[name]
                p.grad = flat2[off:off + p.numel()].view(p.shape)
                off += p.numel()
            m2.last_compact = True
            m2.last_viewmat = torch.eye(4).reshape(1, 4, 4) * (rank + 2)
            if overlapped == "prepared":                             # message assembled and skip words folded by the caller
                P.prepare_compact_message(m2, world)                 # (what bench.py captures inside its graphs)
                ex = exchange_grads_compact_begin(m2, world, prepared=True, fold=False)
                ex.wait_views()
                ok = ok and int(m2._dp_skip) == 0                    # not folded yet
                P.fold_skip_words(m2)
                ex.wait_geometry()
            elif overlapped:
                ex = exchange_grads_compact_begin(m2, world)
                ex.wait_views()
                ex.wait_geometry()
                ex.wait_geometry()                                   # idempotent
            else:
                exchange_grads_compact(m2, world, rebuild=False)
            ok = ok and int(m2._dp_skip) == 77
            n_views, vms, vm_stride, v_views, view_stride, scale = m2.sh_views
            results.append((m2.flat_grad().clone(), v_views.clone(), n_views, vm_stride, view_stride, scale))
        b = m.group_begin
        nv = b[5] - b[4]
        mean_w = sum(range(1, world + 1)) / world
        for g2, recv, n_views, vm_stride, view_stride, scale in results:
            ok = ok and n_views == world and vm_stride == view_stride == nv + 20 and scale == 1.0 / world
            ok = ok and torch.allclose(g2[:b[4]], (torch.arange(b[4], dtype=torch.float32) + 1.0) * mean_w)
            for r in range(world):
                ok = ok and torch.equal(recv[r, :nv], (torch.arange(b[4], b[5], dtype=torch.float32) + 1.0) * (r + 1))
                ok = ok and torch.equal(recv[r, nv:nv + 16], (torch.eye(4) * (r + 2)).reshape(-1))
                ok = ok and int(recv[r, nv + 16:nv + 17].view(torch.int32)) == (77 if r == world - 1 else 0)
        for other in results[1:]:
            ok = ok and torch.equal(results[0][0][:b[4]], other[0][:b[4]]) and torch.equal(results[0][1], other[1])
        # the SPARSE colour-gradient message (only the rows of the Gaussians a rank's camera saw: index + 3 floats) leaves the
        # optimiser exactly what the dense message leaves it -- also when a rank saw NOTHING (rank 0 here), and the capacity
        # all ranks agree on covers the rank that saw the most
        P._CPU_OVERFLOW_WORD[0] = 0
        n_pts = m.num_points
        gsel = torch.Generator().manual_seed(100 + rank)
        visible = torch.zeros(n_pts, dtype=torch.bool) if rank == 0 else torch.rand(n_pts, generator=gsel) < 0.4
        views = {}
        for form in ("dense", "sparse", "sparse-overflow"):
            m3 = QEDSplatterModel(None, **{k: sc[k] for k in PARAM_NAMES})
            flat3 = (torch.arange(total, dtype=torch.float32) + 1.0) * (rank + 1)
            flat3[b[4]:b[5]] = flat3[b[4]:b[5]] * visible.repeat_interleave(3)          # unseen Gaussians: zero colour gradient
            off = 0
            for name in m3.group_names:
                p = m3.gauss_params[name]
                p.grad = flat3[off:off + p.numel()].view(p.shape)
                off += p.numel()
            m3.last_compact = True
            m3.last_viewmat = torch.eye(4).reshape(1, 4, 4) * (rank + 2)
            if form == "dense":
                ex = exchange_grads_compact_begin(m3, world)
            else:
                cap = P.sparse_message_capacity(m3, world, visible=visible, headroom=1.25)
                ok = ok and cap == 28                                  # 20 rows on rank 1, none on rank 0: 1.25 x 20 -> 28
                if form == "sparse-overflow":
                    cap = 4                                            # fewer rows than the other rank has: it must say so
                ex = exchange_grads_compact_begin(m3, world, sparse_cap=cap, visible=visible)
            ex.wait_views()
            ex.wait_geometry()
            views[form] = (m3.sh_views[3].clone(), m3.flat_grad().clone(), int(m3._dp_skip))
        d, sp, so = views["dense"], views["sparse"], views["sparse-overflow"]
        ok = ok and torch.equal(d[0][:, :nv + 17], sp[0][:, :nv + 17]) and torch.equal(d[1][:b[4]], sp[1][:b[4]])
        ok = ok and d[2] == 0 and sp[2] == 0
        ok = ok and float(sp[0][0, :nv].abs().max()) == 0.0 and float(sp[0][1, :nv].abs().max()) > 0.0   # rank 0 saw nothing
        ok = ok and so[2] > 4                                           # the overflowing rank's count: every rank skips the step
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_scene_sharding_by_rank():
    """Every rank sees the same Gaussians and its own camera (yaw 5 degrees * rank)."""
    import bench
    s0 = bench.make_scene(200, 64, 48, 0, torch.device("cpu"))
    s1 = bench.make_scene(200, 64, 48, 1, torch.device("cpu"))
    for k in PARAM_NAMES:
        assert torch.equal(s0[k], s1[k])
    assert s0["camera_to_worlds"].shape == s1["camera_to_worlds"].shape == (1, 3, 4)
    assert not torch.equal(s0["camera_to_worlds"], s1["camera_to_worlds"])
    yaw = torch.atan2(s1["camera_to_worlds"][0, 0, 2], s1["camera_to_worlds"][0, 0, 0])
    assert abs(float(torch.rad2deg(yaw)) - 5.0) < 1e-4
