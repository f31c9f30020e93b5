"""Camera-sharded data parallelism: one process per GPU, every rank holds the
full Gaussian set and renders its own camera; after backward ONE collective averages the flat
59 N-float gradient buffer (RCCL over xGMI; `nccl` backend == RCCL on ROCm), then every rank takes the
identical fused Adam step.  The reference itself is single-GPU, one camera per step
(/root/reference/qed_splatter/model.py:210-212); there is no collective to translate.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

# Rehearsal hook (bench.py QED_BENCH_RCCL_SELF=1, tests): issue the collectives even in a process group of ONE rank, so
# that a one-GPU box drives the real RCCL calls (AVG all-reduce, all_gather_into_tensor, async works waited on the
# compute stream) of the N > 1 path.  Off in production: a single rank has nothing to exchange.
FORCE_COLLECTIVES = False


def _single(world_size: int) -> bool:
    return world_size <= 1 and not FORCE_COLLECTIVES


def allreduce_flat_grad(model, world_size: int, group=None) -> torch.Tensor:
    """Average the model's six parameter gradients across ranks with a single all-reduce.

    The gradients already alias one contiguous allocation (``model.flat_grad()``), so nothing is
    packed or copied.  Loss is normalised by the global camera count, i.e. gradients are averaged.
    """
    g = model.flat_grad()
    if g is None:
        raise RuntimeError("no gradients to reduce: call backward() first")
    _refuse_compact(model, "allreduce_flat_grad")
    if _single(world_size):
        model._dp_skip = None
        return g
    collective_skip_flag(model, world_size, group)
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(g, op=dist.ReduceOp.AVG, group=group)      # ncclAvg: no extra scaling pass
    else:                                                            # gloo (CPU tests) has no AVG
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        g.mul_(1.0 / max(world_size, 1))
    return g


def _refuse_compact(model, who: str) -> None:
    """After fused_loss(compact_sh_grad=True) features_dc.grad holds colour gradients and features_rest.grad is
    unwritten memory: only exchange_grads_compact / FlatAdam.step(fused_sh=True) may consume them."""
    if getattr(model, "last_compact", False):
        raise RuntimeError(f"{who}: the last backward wrote compact SH gradients (fused_loss(compact_sh_grad=True)); "
                           "use exchange_grads_compact() / FlatAdam.step(fused_sh=True), or render without the flag")


def exchange_grads_compact(model, world_size: int, group=None, views=None, rebuild: bool = True) -> torch.Tensor:
    """Data-parallel gradient exchange that moves ~2.6x fewer bytes than all-reducing the flat gradient.

    The 48 SH-coefficient gradients of a Gaussian are b_k(direction to the camera) x (3 colour gradients): a
    rank-1 product per view.  After ``model.fused_loss(..., compact_sh_grad=True)`` + backward, this
      1. all-reduces (AVG) the 11 N geometry floats (means, scales, quats, opacities),
      2. all-gathers ONE message per rank: its 3 N clamp-masked colour gradients + its 4x4 view matrix,
      3. rebuilds the averaged features_dc / features_rest gradients locally (qed_sh_grad_from_views),
    leaving ``model.flat_grad()`` exactly as ``allreduce_flat_grad`` would (up to fp32 summation order).
    At 8 ranks / 500 k Gaussians a rank receives 42 + 19 MB instead of moving 206 MB.

    ``views`` (tests): a list of (colour-gradient [N,3], viewmat [1,4,4]) pairs standing in for the gather.

    ``rebuild=False`` skips step 3 and leaves the gathered views in ``model.sh_views`` for
    ``FlatAdam.step(fused_sh=True)``, which evaluates the coefficient gradients inside the optimiser pass
    (the message buffers are persistent, so a captured Adam graph keeps reading the right memory)."""
    from . import _lib as L
    g = model.flat_grad()
    if g is None:
        raise RuntimeError("no gradients to exchange: call backward() first")
    if not getattr(model, "last_compact", False):
        raise RuntimeError("exchange_grads_compact needs gradients from fused_loss(..., compact_sh_grad=True)")
    names, begin, N = model.group_names, model.group_begin, model.num_points
    i_dc, i_rest = names.index("features_dc"), names.index("features_rest")
    n_rows = len(views) if views is not None else max(world_size, 1)
    geo, send, recv_buf, nv, row = _compact_buffers(model, g, n_rows)
    v_local = g[begin[i_dc]:begin[i_dc + 1]]
    bufs = (send, recv_buf)
    if views is not None:
        n_views = len(views)
        recv = bufs[1]
        for c, (v, m) in enumerate(views):
            recv[c, :nv] = v.reshape(-1)
            recv[c, nv:nv + 16] = m.reshape(-1)
        model._dp_skip = None
    elif not _single(world_size):
        n_views = max(world_size, 1)
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(geo, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(geo, op=dist.ReduceOp.SUM, group=group)
            geo.mul_(1.0 / n_views)
        recv = bufs[1]
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(recv.view(-1), send, group=group)
        else:                                                     # gloo (rehearsal / CPU tests)
            dist.all_gather(list(recv.unbind(0)), send, group=group)
        _fold_skip_words(model, recv, nv)
    else:
        n_views, recv = 1, send.view(1, row)
        model._dp_skip = None
    if not rebuild:
        model.sh_views = (n_views, recv[:, nv:], row, recv, row, 1.0 / n_views)
        return g
    rest_w = (begin[i_rest + 1] - begin[i_rest]) // max(N, 1)
    deg = int(model.last_sh_degree)
    assert 3 * ((deg + 1) ** 2 - 1) <= rest_w
    L.check(L.load().qed_sh_grad_from_views(N, n_views, L.ptr(model.means), L.ptr(recv[:, nv:]), row, L.ptr(recv), row,
                                            deg, 1.0 / n_views, L.ptr(v_local), 3, L.ptr(g[begin[i_rest]:]), rest_w,
                                            torch.cuda.current_stream().cuda_stream), "qed_sh_grad_from_views")
    model.last_compact = False                     # the flat gradient is complete again
    return g


class CompactExchange:
    """The two collectives of ``exchange_grads_compact_begin`` in flight.  ``wait_views()`` makes the current stream
    wait for the gathered colour gradients (what the SH part of the optimiser reads), ``wait_geometry()`` for the
    averaged geometry gradients (what the leading groups read)."""

    def __init__(self, gather, reduce, geo, scale_after: float, fold=None):
        self._gather, self._reduce, self._geo, self._scale_after, self._fold = gather, reduce, geo, scale_after, fold

    def wait_views(self) -> None:
        if self._gather is not None:
            self._gather.wait()
            self._gather = None
            if self._fold is not None:          # the ranks' overflow words -> the step's collective skip flag
                self._fold()

    def wait_geometry(self) -> None:
        if self._reduce is not None:
            self._reduce.wait()
            self._reduce = None
            if self._scale_after != 1.0:                          # gloo has no AVG
                self._geo.mul_(self._scale_after)


class early_gather:
    """Context manager: while it is active, the backward pass of ``model.fused_loss(..., compact_sh_grad=True)`` issues the
    all-gather of the colour-gradient message ITSELF, between the compositing backward and the projection backward
    (a hook registered for THIS model in rasterization.PRE_PROJECT_BWD): the message is K7's colour gradient under the forward pass's clamp mask
    (qed_pack_color_grad) -- exactly what the projection backward would write -- so the 6 MB per rank are on the links
    while the projection backward (~30 us at 500 k Gaussians) runs.  ``exchange_grads_compact_begin`` then finds the
    gather in flight and only adds the geometry all-reduce.

        with early_gather(model, world):
            model.backward_fused(losses)
        ex = exchange_grads_compact_begin(model, world)
        ...
    """

    def __init__(self, model, world_size: int, group=None):
        self.model, self.world_size, self.group = model, world_size, group

    def _hook(self, vsplat, sh_jac, total):
        from . import _lib as L
        m = self.model
        N = m.num_points
        assert total == N, "one camera per rank in the data-parallel exchange"
        dev = vsplat.device
        nv, row = 3 * N, 3 * N + 16 + 4
        n_rows = max(self.world_size, 1)
        send, recv = _message_buffers(m, row, n_rows, dev)
        L.check(L.load().qed_pack_color_grad(total, L.ptr(vsplat), L.ptr(sh_jac), L.ptr(send),
                                             torch.cuda.current_stream().cuda_stream), "qed_pack_color_grad")
        send[nv:nv + 16] = m.last_viewmat.reshape(-1).to(torch.float32)
        send[nv + 16:nv + 17].view(torch.int32).copy_(_local_overflow_word(dev))
        work = None
        if not _single(self.world_size):
            if dist.get_backend(self.group) == "nccl":
                work = dist.all_gather_into_tensor(recv.view(-1), send, group=self.group, async_op=True)
            else:
                work = dist.all_gather(list(recv.unbind(0)), send, group=self.group, async_op=True)
        m._early_gather = (work, send, recv)

    def __enter__(self):
        from . import rasterization as R
        self._key = self.model.means.untyped_storage().data_ptr()       # this model's hook only (rasterization.PRE_PROJECT_BWD)
        self._prev = R.PRE_PROJECT_BWD.get(self._key)
        R.PRE_PROJECT_BWD[self._key] = self._hook
        self.model._early_gather = None
        return self

    def __exit__(self, *exc):
        from . import rasterization as R
        if self._prev is None:
            R.PRE_PROJECT_BWD.pop(self._key, None)
        else:
            R.PRE_PROJECT_BWD[self._key] = self._prev
        return False


def backward_with_early_gather(model, losses, world_size: int, group=None) -> None:
    """``model.backward_fused(losses)`` in two phases run from THIS thread, with the colour-gradient all-gather issued
    between them: (1) the loss's and the compositing backward (autograd.grad up to the projected splats), (2) pack the
    message from K7's rows + launch the all-gather, (3) the projection backward.  Same gradients as backward_fused; the
    6 MB per rank are on the links while the projection backward runs.

    Why not simply from a hook inside ONE backward pass (``early_gather``): the autograd engine runs the nodes on its own
    worker thread, and ProcessGroupNCCL decides by the CALLING thread's stream state whether a collective is being
    captured into a graph -- issued from the worker thread during a capture, its work object reaches the watchdog, which
    queries an event recorded on a capturing stream (hipErrorCapturedEvent) and aborts the process.  From the capturing
    thread itself the collectives are captured (bench.py --dp-one-graph)."""
    from .model import _unit_grad
    from .rasterization import _VSPLAT_REGISTRY
    info = model.info
    names = ("means2d", "conics", "colors", "opacities", "depths")
    inter = [info[k] for k in names]
    node = inter[0].grad_fn                                        # _ProjectSH's backward node: its saved tensors hold sh_jac
    loss = losses["loss"]
    g = torch.autograd.grad(loss, inter, grad_outputs=_unit_grad(loss.device), allow_unused=True)
    vsplat = _VSPLAT_REGISTRY.get(g[0].untyped_storage().data_ptr())
    from .rasterization import _ProjectSH
    sh_jac = _ProjectSH.saved(node, "sh_jac")
    eg = early_gather(model, world_size, group)
    if vsplat is not None and sh_jac is not None and getattr(model, "last_compact", False):
        model._early_gather = None
        eg._hook(vsplat, sh_jac, vsplat.shape[0])
    keep = [(t, gi) for t, gi in zip(inter, g) if gi is not None]
    torch.autograd.backward([t for t, _ in keep], [gi for _, gi in keep])


# ---- sparse colour-gradient message -------------------------------------------------------------------------------------
# The colour gradient of a Gaussian that a rank's camera does not see (radii == 0) is zero, so a rank need only send the
# rows it has: (index, 3 floats) per VISIBLE Gaussian = 16 B against 12 B per Gaussian of the dense message.  In SURVEY
# 8d's synthetic scene 95 % are visible (the dense form is smaller); a camera of a real capture sees a fraction of the
# scene, and xGMI links, not HBM, bound the exchange.  The message is capacity-bounded like the intersection list: ``cap``
# rows (agreed by all ranks at set-up: sparse_message_capacity), a rank with more visible rows than that marks the step as
# overflowed -- the word every rank's optimiser takes as skip_flag -- and the caller re-calibrates.  Row layout (floats):
#   [0:16] view matrix | [16] overflow word (int32 bits) | [17] visible rows (int32 bits) | [18:20] pad |
#   [20 : 20 + cap] indices (int32 bits) | [20 + cap : 20 + 4 cap] colour gradients
# The receiver scatters the R lists into the dense [R, 3 N + 20] rows the optimiser pass reads (sh_views), header behind the
# colours as in the dense message, so nothing downstream changes.  Plain torch ops without a host read: the same code runs
# on CPU tensors in the gloo tests and on the GPU inside a captured graph.
SPARSE_HEADER = 20


def sparse_row_floats(cap: int) -> int:
    return SPARSE_HEADER + 4 * int(cap)


def sparse_message_capacity(model, world_size: int, group=None, headroom: float = 1.5, visible=None):
    """Set-up time (one host read, one MAX all-reduce): the row capacity every rank will use, or None when the dense
    message is no larger (capacity >= 3/4 N: 16 B per visible row against 12 B per Gaussian)."""
    vis = _visible_rows(model, visible)
    n = vis.numel()
    count = vis.sum().to(torch.int64).reshape(1)
    if not _single(world_size):
        dist.all_reduce(count, op=dist.ReduceOp.MAX, group=group)
    cap = min(n, max(4, (int(int(count) * headroom) + 3) // 4 * 4))          # (a multiple of 4: the row's parts stay 16-byte aligned)
    return None if 4 * cap >= 3 * n else cap


def _visible_rows(model, visible=None) -> torch.Tensor:
    if visible is not None:
        return visible.reshape(-1)
    return (model.info["radii"][0] > 0).reshape(-1)


def pack_sparse_message(v_color: torch.Tensor, visible: torch.Tensor, viewmat: torch.Tensor, overflow_word: torch.Tensor,
                        cap: int, out: torch.Tensor) -> None:
    """This rank's sparse row (layout above) into ``out`` [20 + 4 cap].  ``v_color`` [N,3] or [3 N]; ``visible`` [N] bool.
    A rank with more than ``cap`` visible rows sends the first ``cap`` and raises its overflow word to the count it needed."""
    n = visible.numel()
    vis = visible.reshape(-1)
    pos = torch.cumsum(vis.to(torch.int32), 0, dtype=torch.int32) - 1          # position of a visible row in the list
    count = pos[-1:] + 1 if n > 0 else torch.zeros(1, dtype=torch.int32, device=out.device)
    keep = vis & (pos < cap)
    dst = torch.where(keep, pos, torch.full_like(pos, cap)).to(torch.int64)     # (slot `cap` takes what is not sent)
    idx_buf = torch.zeros(cap + 1, dtype=torch.int32, device=out.device)
    rgb_buf = torch.zeros(cap + 1, 3, dtype=torch.float32, device=out.device)
    idx_buf.scatter_(0, dst, torch.arange(n, dtype=torch.int32, device=out.device))
    rgb_buf.index_copy_(0, dst, v_color.reshape(n, 3).to(torch.float32)) if n > 0 else None
    out[:16] = viewmat.reshape(-1).to(torch.float32)
    over = torch.where(count > cap, count, torch.zeros_like(count))
    out[16:17].view(torch.int32).copy_(torch.maximum(overflow_word.reshape(1).to(torch.int32), over))
    out[17:18].view(torch.int32).copy_(count)
    out[18:20] = 0.0
    out[SPARSE_HEADER:SPARSE_HEADER + cap].view(torch.int32).copy_(idx_buf[:cap])
    out[SPARSE_HEADER + cap:].copy_(rgb_buf[:cap].reshape(-1))


def unpack_sparse_messages(rows: torch.Tensor, n: int, cap: int, dense: torch.Tensor) -> None:
    """The gathered sparse rows [R, 20 + 4 cap] -> dense [R, 3 n + 20]: colour gradients scattered to their Gaussians (zero
    elsewhere), then the 20 header floats (view matrix, overflow word, count) as in the dense message."""
    R = rows.shape[0]
    nv = 3 * n
    dense[:, :nv] = 0.0
    counts = rows[:, 17:18].view(torch.int32).clamp(max=cap)                                     # [R,1]
    idx = rows[:, SPARSE_HEADER:SPARSE_HEADER + cap].view(torch.int32).to(torch.int64)           # [R,cap]
    rgb = rows[:, SPARSE_HEADER + cap:].reshape(R, cap, 3)
    live = torch.arange(cap, device=rows.device).reshape(1, cap) < counts                       # rows beyond the count: unset
    body = torch.zeros(R, n + 1, 3, dtype=torch.float32, device=rows.device)                     # (slot n takes the dead rows)
    dst = torch.where(live, idx, torch.full_like(idx, n))
    body.scatter_(1, dst.unsqueeze(-1).expand(R, cap, 3), rgb)
    dense[:, :nv] = body[:, :n].reshape(R, nv)
    dense[:, nv:nv + SPARSE_HEADER] = rows[:, :SPARSE_HEADER]


def _message_buffers(model, row: int, n_rows: int, device):
    bufs = getattr(model, "_dp_buffers", None)
    if bufs is None or bufs[0].numel() != row or bufs[1].shape != (n_rows, row) or bufs[0].device != device:
        bufs = model._dp_buffers = (torch.zeros(row, dtype=torch.float32, device=device),      # (the row's padding stays 0)
                                    torch.zeros(n_rows, row, dtype=torch.float32, device=device))
    return bufs


def prepare_compact_message(model, world_size: int) -> None:
    """Assemble this rank's message (colour gradients, view matrix, overflow word) in the persistent send buffer.  What
    ``exchange_grads_compact_begin`` does first; as a call of its own it can be CAPTURED at the end of the forward+backward
    graph (three small launches that then cost graph nodes, not eager dispatches) -- pass ``prepared=True`` to the exchange."""
    g = model.flat_grad()
    if g is None or not getattr(model, "last_compact", False):
        raise RuntimeError("prepare_compact_message needs gradients from fused_loss(..., compact_sh_grad=True)")
    _compact_buffers(model, g, max(world_size, 1), fill=True)


def fold_skip_words(model) -> None:
    """The gathered overflow words -> ``model._dp_skip`` (``exchange_grads_compact_begin(..., fold=False)`` leaves this to
    the caller, who can capture it in front of the optimiser launches that read the flag)."""
    bufs = model._dp_buffers
    _fold_skip_words(model, bufs[1], bufs[0].numel() - 20)


def exchange_grads_compact_begin(model, world_size: int, group=None, prepared: bool = False,
                                 fold: bool = True, sparse_cap=None, visible=None) -> CompactExchange:
    """``exchange_grads_compact(..., rebuild=False)`` with both collectives left in flight, the all-gather of the
    colour gradients FIRST: the SH part of the optimiser (88 us at 500 k Gaussians, two thirds of the Adam pass) needs
    only that message and runs while the geometry all-reduce is still on the links:

        ex = exchange_grads_compact_begin(model, world)
        ex.wait_views();    optimizer.step(fused_sh=True, part=1)      # tick + features_dc / features_rest
        ex.wait_geometry(); optimizer.step(fused_sh=True, part=2)      # means, scales, quats, opacities

    xGMI is point to point, so the two messages share the same links and are issued back to back, not concurrently.
    Same result as exchange_grads_compact + step(fused_sh=True).

    ``prepared``: the message is in the send buffer already (prepare_compact_message, e.g. captured behind the backward
    pass).  ``fold=False``: the caller folds the gathered overflow words itself (fold_skip_words) after wait_views() and
    before the optimiser launches -- in a captured optimiser graph that is a graph node instead of an eager launch.

    ``sparse_cap`` (from sparse_message_capacity, the same on every rank): the all-gather carries only the rows of the
    Gaussians this rank's camera saw (index + colour gradient, at most ``sparse_cap`` of them); ``wait_views()`` scatters
    the gathered lists into the dense rows the optimiser reads.  Same ``model.sh_views``, same update as the dense message
    (a Gaussian a rank did not see has a zero colour gradient there either way)."""
    if sparse_cap is not None:
        return _exchange_sparse_begin(model, world_size, group, int(sparse_cap), visible, fold)
    g = model.flat_grad()
    if g is None:
        raise RuntimeError("no gradients to exchange: call backward() first")
    if not getattr(model, "last_compact", False):
        raise RuntimeError("exchange_grads_compact_begin needs gradients from fused_loss(..., compact_sh_grad=True)")
    early = getattr(model, "_early_gather", None)
    model._early_gather = None
    geo, send, recv, nv, row = _compact_buffers(model, g, max(world_size, 1), fill=early is None and not prepared)
    if _single(world_size):
        model.sh_views = (1, send[nv:nv + 16].view(1, 16), row, send.view(1, row), row, 1.0)
        model._dp_skip = None
        return CompactExchange(None, None, geo, 1.0)
    nccl = dist.get_backend(group) == "nccl"
    if early is not None:                                         # the backward pass issued the gather itself (early_gather)
        w_gather = early[0]
    elif nccl:
        w_gather = dist.all_gather_into_tensor(recv.view(-1), send, group=group, async_op=True)
    else:                                                         # gloo (rehearsal / CPU tests)
        w_gather = dist.all_gather(list(recv.unbind(0)), send, group=group, async_op=True)
    if nccl:
        w_reduce = dist.all_reduce(geo, op=dist.ReduceOp.AVG, group=group, async_op=True)
    else:
        w_reduce = dist.all_reduce(geo, op=dist.ReduceOp.SUM, group=group, async_op=True)
    n_views = max(world_size, 1)
    model.sh_views = (n_views, recv[:, nv:], row, recv, row, 1.0 / n_views)
    _dp_skip_word(model, g.device)              # (exists before anything captures an optimiser launch that reads it)
    return CompactExchange(w_gather, w_reduce, geo, 1.0 if nccl else 1.0 / n_views,
                           fold=(lambda: _fold_skip_words(model, recv, nv)) if fold else None)


def _exchange_sparse_begin(model, world_size: int, group, cap: int, visible, fold: bool) -> CompactExchange:
    g = model.flat_grad()
    if g is None:
        raise RuntimeError("no gradients to exchange: call backward() first")
    if not getattr(model, "last_compact", False):
        raise RuntimeError("exchange_grads_compact_begin needs gradients from fused_loss(..., compact_sh_grad=True)")
    names, begin, N = model.group_names, model.group_begin, model.num_points
    i_dc = names.index("features_dc")
    geo, v_local = g[:begin[i_dc]], g[begin[i_dc]:begin[i_dc + 1]]
    nv, row, srow = 3 * N, 3 * N + SPARSE_HEADER, sparse_row_floats(cap)
    n_rows = max(world_size, 1)
    _, dense = _message_buffers(model, row, n_rows, g.device)                 # what sh_views / fold_skip_words read
    sb = getattr(model, "_dp_sparse_buffers", None)
    if sb is None or sb[0].numel() != srow or sb[1].shape != (n_rows, srow) or sb[0].device != g.device:
        sb = model._dp_sparse_buffers = (torch.zeros(srow, dtype=torch.float32, device=g.device),
                                         torch.zeros(n_rows, srow, dtype=torch.float32, device=g.device))
    send, recv = sb
    pack_sparse_message(v_local, _visible_rows(model, visible), model.last_viewmat, _local_overflow_word(g.device), cap, send)
    n_views = n_rows
    model.sh_views = (n_views, dense[:, nv:], row, dense, row, 1.0 / n_views)
    _dp_skip_word(model, g.device)
    if _single(world_size):
        recv[0].copy_(send)
        unpack_sparse_messages(recv, N, cap, dense)
        if fold:
            _fold_skip_words(model, dense, nv)
        return CompactExchange(None, None, geo, 1.0)
    nccl = dist.get_backend(group) == "nccl"
    if nccl:
        w_gather = dist.all_gather_into_tensor(recv.view(-1), send, group=group, async_op=True)
        w_reduce = dist.all_reduce(geo, op=dist.ReduceOp.AVG, group=group, async_op=True)
    else:
        w_gather = dist.all_gather(list(recv.unbind(0)), send, group=group, async_op=True)
        w_reduce = dist.all_reduce(geo, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def after_gather():
        unpack_sparse_messages(recv, N, cap, dense)
        if fold:
            _fold_skip_words(model, dense, nv)
    return CompactExchange(w_gather, w_reduce, geo, 1.0 if nccl else 1.0 / n_views, fold=after_gather)


def _fold_skip_words(model, recv: torch.Tensor, nv: int) -> None:
    """model._dp_skip = max over the gathered rows' overflow words (one small launch; the words are counts >= 0)."""
    words = recv[:, nv + 16:nv + 17].view(torch.int32)            # [ranks, 1]
    torch.amax(words, dim=0, out=_dp_skip_word(model, recv.device))


def _compact_buffers(model, g, n_rows: int, fill: bool = True):
    """(geometry slice of the flat gradient, this rank's message, the receive buffer, 3 N, message length): the
    message buffers are persistent, so captured graphs keep reading the right memory.  ``fill=False``: the message has been
    written already (early_gather)."""
    names, begin = model.group_names, model.group_begin
    i_dc, i_rest = names.index("features_dc"), names.index("features_rest")
    assert i_dc == 4 and i_rest == 5, "group order: geometry groups first, then features_dc, features_rest"
    geo, v_local = g[:begin[i_dc]], g[begin[i_dc]:begin[i_dc + 1]]
    nv = v_local.numel()                                          # 3 N
    # one message per rank: colour gradients + view matrix + (bits of) this rank's intersection-overflow word (4 floats, so
    # that rows stay 16-byte aligned): the optimiser step is skipped on EVERY rank when ANY rank's frame overflowed
    row = nv + 16 + 4
    bufs = _message_buffers(model, row, n_rows, g.device)
    send = bufs[0]
    if not fill:
        return geo, send, bufs[1], nv, row
    send[:nv] = v_local
    send[nv:nv + 16] = model.last_viewmat.reshape(-1).to(torch.float32)
    send[nv + 16:nv + 17].view(torch.int32).copy_(_local_overflow_word(g.device))
    return geo, send, bufs[1], nv, row


def _local_overflow_word(device) -> torch.Tensor:
    """int32[1] view of this device's binning overflow word (non-zero: the frame just rendered overflowed its intersection
    buffer and is empty; what single-GPU optimiser launches take as ``skip_flag``)."""
    if device.type != "cuda":          # gloo rehearsals on CPU tensors (tests): no binning workspace; tests fill the word
        return _CPU_OVERFLOW_WORD
    from .rasterization import _workspace
    return _workspace(device).status[:1]


_CPU_OVERFLOW_WORD = torch.zeros(1, dtype=torch.int32)


def _dp_skip_word(model, device) -> torch.Tensor:
    """The persistent int32[1] the optimiser launches of a data-parallel step take as ``skip_flag`` (FlatAdam._skip): the
    MAXIMUM over all ranks of the overflow words.  A rank whose frame overflowed renders it empty and skips its update on
    the device; unless its peers skip the same step, the replicas' parameters, moments and step counters part for good
    (per-rank cameras make the list length rank dependent, so an overflow need not hit every rank in the same step)."""
    w = getattr(model, "_dp_skip", None)
    if w is None or w.device != device:
        w = model._dp_skip = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def collective_skip_flag(model, world_size: int, group=None) -> torch.Tensor:
    """For the exchanges that gather no message (allreduce_flat_grad, allreduce_and_step): one MAX all-reduce of the
    overflow word into ``model._dp_skip``."""
    g = model.flat_grad()
    w = _dp_skip_word(model, g.device)
    w.copy_(_local_overflow_word(g.device))
    if not _single(world_size):
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
    return w


def allreduce_and_step(model, optimizer, world_size: int, n_chunks: int = 4, group=None) -> None:
    """Gradient all-reduce and Adam step, pipelined: the flat gradient is reduced in ``n_chunks`` contiguous
    pieces issued back to back on the process group's stream, and each piece's range of the flat parameter
    buffer is updated (``FlatAdam.step_range``) as soon as its reduction has finished -- the HBM-bound
    optimiser pass hides behind the link-bound collectives of the later pieces.  Same result as
    ``allreduce_flat_grad`` + ``optimizer.step()``.

    Whether it pays depends on how RCCL's bus bandwidth falls off with message size on the node at hand:
    hiding the ~0.13 ms Adam pass of config B is worth it only if four 30 MB all-reduces cost less than
    0.13 ms more than one 118 MB all-reduce.  bench.py keeps the single collective (not measurable on the
    one-GPU development box); ``scripts/dp_rehearsal.py`` checks the equivalence on two ranks."""
    g = model.flat_grad()
    if g is None:
        raise RuntimeError("no gradients to reduce: call backward() first")
    _refuse_compact(model, "allreduce_and_step")
    total = g.numel()
    if _single(world_size):
        model._dp_skip = None
        optimizer.begin_step()
        optimizer.step_range(0, total)
        return
    collective_skip_flag(model, world_size, group)
    n_chunks = max(1, min(int(n_chunks), total // 4 or 1))
    bounds = [min(total, (total * k // n_chunks) // 4 * 4) for k in range(n_chunks)] + [total]
    avg = dist.get_backend(group) == "nccl"
    works = [dist.all_reduce(g[a:b], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True)
             for a, b in zip(bounds[:-1], bounds[1:])]
    optimizer.begin_step()
    for w, a, b in zip(works, bounds[:-1], bounds[1:]):
        w.wait()                                   # the current stream waits for this piece only
        if not avg:
            g[a:b].mul_(1.0 / max(world_size, 1))
        optimizer.step_range(a, b)


def allreduce_densification_stats(xys_absgrad_norm: torch.Tensor, vis_counts: torch.Tensor,
                                  max_radii: torch.Tensor, group=None) -> None:
    """Reduce what Nerfstudio's densifier accumulates from model.py:289-292 side effects:
    sum of |means2d.absgrad| norms and visibility counts (SUM), largest screen radius (MAX)."""
    dist.all_reduce(xys_absgrad_norm, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vis_counts, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group)
