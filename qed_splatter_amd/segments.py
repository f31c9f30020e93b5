"""``get_outputs`` of a training step as two captured hipGraphs behind ONE autograd node.

The reference-shaped route (Nerfstudio's trainer calling ``get_outputs`` -> ``get_metrics_dict`` -> ``get_loss_dict`` ->
``backward`` -> one optimiser per group; /root/reference/qed_splatter/model.py:199-321, 73-118) enqueues ~25 launches per
step from Python: three autograd nodes, ~40 allocations, one ctypes call per entry point.  On a slow host that bounds the
step.  Nothing those launches are given depends on the step once the shape is fixed: the kernels read the intersection
count from device memory and size their grids by a calibrated capacity (rasterization._Workspace), the parameters are
updated in place.  So once a shape has proved stable AND the host is the slower side (SegmentCache: a capture costs ~35 ms
at 500 k Gaussians, a replay saves ~0.15 ms of host time and nothing while the device is the bottleneck) the device work
behind ``get_outputs`` -- projection, binning, K6, the depth fix-up -- is captured into one graph, and its backward -- the
tile ordering, K7, the projection backward -- into a second one; ``get_outputs`` then costs the host one multi-tensor copy
(camera, intrinsics, background), a graph launch and one autograd node, and ``backward`` one more graph launch.

What the captured form gives up, and how it is guarded:
  * outputs live in STATIC buffers: the tensors ``get_outputs`` returned for step k are overwritten by step k+1's replay.
    Nerfstudio's trainer never reads them past the step; code that does gets an error from ``backward`` (generation
    check) instead of the wrong gradients, and ``config.graph_segments = False`` keeps every call's outputs;
  * the intersection buffer cannot grow inside a graph: the captured binning launch stores {M, overflow} into a pinned
    word the next call looks at (exactly the eager asynchronous path, ``_Workspace.poll_pending``); an overflow renders
    that frame empty, skips the optimiser launches on the device, and drops the segment -- the next calls run eagerly
    (regrown, re-calibrated) and capture again;
  * anything the capture was specialised on -- image size, number of Gaussians, SH degree in use, render / rasterize
    mode, the parameters' storage -- is part of the key: a change (resolution schedule, densification) falls back to
    eager calls and a new capture;
  * a camera optimiser (camera-to-world matrices that require grad), the eval-time crop box and ``no_grad`` calls stay
    eager.
"""
from __future__ import annotations

import warnings
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from .rasterization import _workspace, manual_backward


class _SegmentFn(torch.autograd.Function):
    """One node for the whole of ``get_outputs``' device work: forward = replay of the forward graph, backward = replay
    of the backward graph on the static upstream-gradient buffers."""

    @staticmethod
    def forward(ctx, seg: "OutputsSegment", *params):
        seg.g_fwd.replay()
        seg.generation += 1
        ctx.seg, ctx.generation = seg, seg.generation
        ctx.set_materialize_grads(False)
        outs = [seg.rgb.detach(), seg.alpha.detach()]
        if seg.depth is not None:
            outs.append(seg.depth.detach())
        return tuple(outs)

    @staticmethod
    def backward(ctx, v_rgb, v_alpha, v_depth=None):
        seg = ctx.seg
        if ctx.generation != seg.generation:
            raise RuntimeError(
                "backward through the outputs of an EARLIER get_outputs call: with config.graph_segments the outputs of a "
                "training step live in static buffers that the next get_outputs overwrites.  Take the loss and call "
                "backward before the next get_outputs, or set config.graph_segments = False.")
        if v_rgb is None and v_depth is None and v_alpha is None:
            return (None,) * (1 + len(seg.params))
        with_alpha = v_alpha is not None
        graph, grads = seg.backward_graph(with_alpha)
        pairs = [(v_rgb, seg.v_rgb), (v_depth, seg.v_depth)] + ([(v_alpha, seg.v_alpha)] if with_alpha else [])
        for src, dst in pairs:
            if dst is None:
                continue
            if src is None:
                dst.zero_()
            elif src.data_ptr() != dst.data_ptr() or src.stride() != dst.stride():
                dst.copy_(src)                          # (a loss other than _ImageLosses: its gradient is copied in)
        # Gradients of an EARLIER backward pass that still sit in the fields (gradient accumulation, zero_grad(set_to_none=
        # False), a second backward on retained outputs) are aliases of the static buffers the replay is about to overwrite:
        # the engine adopted them without a copy.  They are moved out first, so that AccumulateGrad adds the new gradients
        # to the old values and not to themselves.  (The flat layout of .grad is given up for that step.)
        from .model import _RAW_GRAD, _raw_grad
        for p, g in zip(seg.params, grads):
            old = _raw_grad(p) if g is not None else None
            if old is not None and old.untyped_storage().data_ptr() == g.untyped_storage().data_ptr():
                owner_m = getattr(seg, "lazy_owner", None)
                m_ = owner_m() if owner_m is not None else None
                if m_ is not None:
                    m_._materialise_sh_grads()      # (a compact SH gradient waiting there is completed before it is copied)
                _RAW_GRAD.__set__(p, old.clone())
        # the accumulator K7 adds into: zeroed by the loss's own backward launch when that was _ImageLosses (StepContext)
        if not (seg.holder and seg.holder[0] is seg.vsplat):
            seg.vsplat.zero_()
        del seg.holder[:]
        graph.replay()
        owner = getattr(seg, "lazy_owner", None)
        if owner is not None:
            # the captured projection backward wrote the SH gradients in compact form (config.lazy_sh_grad): they stay so
            # until somebody reads them -- unless gradients of an earlier pass wait in the fields, which autograd is about
            # to add these to: then the coefficient gradients are written out now, in place
            c_proj = seg.manual[0]
            i_dc = seg.param_names.index("features_dc")
            v_dc, v_rest = grads[i_dc], grads[i_dc + 1]
            from .rasterization import _ProjectSH
            viewmats, deg = _ProjectSH.saved(c_proj, "viewmats"), _ProjectSH.saved(c_proj, "sh_degree")
            m = owner()
            if m is None or not m._lazy_sh_begin(v_dc, v_rest, viewmats, deg):
                from .model import write_sh_grads
                write_sh_grads(seg.params[seg.param_names.index("means")], viewmats, deg, v_dc, v_rest)
        leaf = seg.info["means2d"]
        if leaf.requires_grad:                           # model.py:289-292: xys.grad / xys.absgrad for the densifier
            leaf.grad, leaf.absgrad = seg.leaf_grad, seg.leaf_absgrad
        # fresh aliases of the static gradient views: autograd adopts an incoming gradient it holds the only reference
        # to as .grad without copying it (the views stay pieces of ONE allocation in group order: model.flat_grad())
        return (None,) + tuple(g.detach() if g is not None else None for g in grads)


class OutputsSegment:
    """The captured form of one shape of ``get_outputs`` (see the module docstring).  ``render_fn(c2w, intr, background,
    holder, capture_slot) -> (render, alpha, info)`` makes the eager ``rasterization(...)`` call of get_outputs on the
    tensors it is given."""

    def __init__(self, device, params: List[Tensor], render_fn: Callable, shape_key: Tuple,
                 param_names=("means", "scales", "quats", "opacities", "features_dc", "features_rest")):
        self.device = device
        self.params = list(params)
        self.param_names = tuple(param_names)
        self.render_fn = render_fn
        self.shape_key = shape_key                # the workspace's calibration key of this shape: ((W, H), N, C)
        self.generation = 0
        self.holder: list = []
        self._bwd: Dict[bool, Tuple] = {}
        import ctypes
        host = torch.zeros(4, dtype=torch.int32).pin_memory()
        dptr = ctypes.c_void_p()                  # the device-side alias of the pinned word (rasterization._Workspace.host_slot)
        L.check(L.load().qed_host_device_pointer(host.data_ptr(), ctypes.addressof(dptr)), "qed_host_device_pointer")
        self.slot = (host.numpy(), dptr.value)
        self._keep = host
        self._seen = [None, None, None]           # (source tensor, version) last copied into c2w / intr / bg

    # ---- capture ---------------------------------------------------------------------------------------------
    def capture(self, c2w: Tensor, intr: Tensor, background: Tensor) -> None:
        self.c2w, self.intr = c2w.detach().clone(), intr.detach().clone()
        self.bg = background.detach().to(torch.float32).clone()
        torch.cuda.synchronize(self.device)
        self.g_fwd = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of other threads (a viewer's eval render, RCCL's watchdog) must not break the capture
        self.manual: list = []                    # the two nodes' contexts: the call runs without the autograd engine
        with torch.cuda.graph(self.g_fwd, capture_error_mode="thread_local"):
            render, alpha, info = self.render_fn(self.c2w, self.intr, self.bg, self.holder, self.slot, manual=self.manual)
        self.render, self.alpha, self.info = render, alpha, info
        self.rgb = info.pop("post_rgb")
        self.depth = info.pop("post_depth")
        dev = self.device
        self.v_rgb = torch.zeros_like(self.rgb)
        self.v_depth = torch.zeros_like(self.depth) if self.depth is not None else None
        self.v_alpha = None
        self.vsplat = torch.zeros(info["radii"].numel(), L.VSPLAT_FLOATS, dtype=torch.float32, device=dev)
        self.backward_graph(False)                # the usual backward (nothing downstream used `accumulation`)

    def backward_graph(self, with_alpha: bool):
        """(graph, static parameter gradients) of the backward pass, captured on first use: without / with a gradient
        for ``accumulation`` (a loss on the alpha image sends the compositing backward through its general path)."""
        got = self._bwd.get(with_alpha)
        if got is not None:
            return got
        if with_alpha and self.v_alpha is None:
            self.v_alpha = torch.zeros_like(self.alpha)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.g_fwd.pool(), capture_error_mode="thread_local"):
            self.holder[:] = [self.vsplat]        # _Composite.backward takes its accumulator from the holder
            v_means, v_quats, v_scales, v_opac, v_dc, v_rest, _ = manual_backward(
                self.manual, None, self.v_alpha if with_alpha else None, self.v_rgb, self.v_depth)
        del self.holder[:]
        by_name = dict(means=v_means, scales=v_scales, quats=v_quats, opacities=v_opac, features_dc=v_dc, features_rest=v_rest)
        grads = tuple((by_name[n].view(p.shape) if (by_name[n] is not None and p.requires_grad) else None)
                      for n, p in zip(self.param_names, self.params))
        leaf = self.info["means2d"]
        self.leaf_grad, self.leaf_absgrad = leaf.grad, getattr(leaf, "absgrad", None)
        self._bwd[with_alpha] = (g, grads)
        return self._bwd[with_alpha]

    # ---- replay ----------------------------------------------------------------------------------------------
    def run(self, c2w: Tensor, intr: Tensor, background: Tensor):
        """One training-step forward: (rgb [1,H,W,3], alpha [1,H,W,1], depth [1,H,W,1] | None, info, holder, grad buffers)."""
        ws = _workspace(self.device)
        # the step's camera / background into the static buffers the graph reads: ONE multi-tensor copy launch, and none
        # for a source that is the very tensor (same object, same version) copied last time
        dst, src = [], []
        for i, (d, t) in enumerate(((self.c2w, c2w), (self.intr, intr), (self.bg, background))):
            seen = self._seen[i]
            if seen is not None and seen[0] is t and seen[1] == t._version:
                continue
            self._seen[i] = (t, t._version)
            dst.append(d)
            src.append(t if t.dtype == torch.float32 else t.to(torch.float32))
        if dst:
            torch._foreach_copy_(dst, src)
        self.slot[0][0] = -1                      # (the previous replay's word was read by poll_pending before this call)
        del self.holder[:]
        outs = _SegmentFn.apply(self, *self.params)
        ws.arm_pending(self.slot[0], self.shape_key)
        rgb, alpha = outs[0], outs[1]
        depth = outs[2] if self.depth is not None else None
        return rgb, alpha, depth


class SegmentCache:
    """The segments of one model: at most ``KEEP`` live captures, one per key.

    WHEN a key is captured (``config.graph_segments``):
      * ``"always"``: on its ``WARM_CALLS``-th eager call (the eager calls calibrate the intersection capacity and warm the
        allocator) -- tests, benchmarks;
      * ``True`` (the default, "when it pays"): a capture costs ~35 ms at 500 k Gaussians @ 1080p (measured:
        scripts/segment_capture_time.py) and saves ~0.15 ms of HOST time per step, and nothing at all while the device is
        the slower side.  So (i) ski rental: a key is captured only once it has been called as often as the capture costs
        in savings (~240 calls) -- a shape that densification replaces every 100 steps is never captured, a stable one is
        captured with at most twice the optimal overhead; and (ii) only if the host is the bottleneck: over those calls
        it spent less than ``WAIT_FRAC`` of the wall time waiting for the device (``_Workspace.waited_s``: the host can
        run at most one frame ahead).  A GPU-bound loop therefore stays eager, where a replay would buy nothing."""

    WARM_CALLS = 3
    KEEP = 3
    SAVING_S = 1.5e-4                             # host time one replayed step saves (measured 0.62-0.69 -> 0.45-0.48 ms)
    WAIT_FRAC = 0.10

    def __init__(self):
        self.stats: Dict[Tuple, list] = {}        # key -> [calls, wall clock at the window's start, waited_s at its start]
        self.segments: Dict[Tuple, OutputsSegment] = {}
        self.disabled: Optional[str] = None       # why capturing was given up for this model (the first failure)
        self.capture_s = 0.04                     # cost of a capture: an estimate until one has been timed
        self.decision = "eager: no shape seen often enough yet"

    def drop_all(self) -> None:
        self.segments.clear()
        self.stats.clear()

    def get(self, key: Tuple) -> Optional[OutputsSegment]:
        return self.segments.get(key)

    def should_capture(self, key: Tuple, mode, ws=None) -> bool:
        import time
        now = time.perf_counter()
        waited = ws.waited_s if ws is not None else 0.0
        if len(self.stats) > 16:
            self.stats.clear()
        st = self.stats.get(key)
        if st is None:
            st = self.stats[key] = [0, now, waited]
        st[0] += 1
        if self.disabled is not None:
            return False
        if mode == "always":
            return st[0] > self.WARM_CALLS
        need = max(self.WARM_CALLS, int(self.capture_s / self.SAVING_S))
        if st[0] <= need:
            return False
        wall = now - st[1]
        wait_frac = (waited - st[2]) / wall if wall > 0 else 1.0
        if wait_frac > self.WAIT_FRAC:
            # the device is the slower side: look again after another `need` calls
            self.decision = f"eager: the host waited for the device {100 * wait_frac:.0f} % of the last {st[0]} calls"
            self.stats[key] = [0, now, waited]
            return False
        self.decision = f"captured: host-bound (waited {100 * wait_frac:.0f} % of {st[0]} calls)"
        return True

    def capture(self, key: Tuple, make: Callable[[], OutputsSegment], c2w, intr, background) -> Optional[OutputsSegment]:
        import time
        try:
            t0 = time.perf_counter()
            seg = make()
            seg.capture(c2w, intr, background)
            torch.cuda.synchronize(seg.device)
            self.capture_s = time.perf_counter() - t0
        except Exception as e:                    # capturing is an optimisation of dispatch only: stay eager
            self.disabled = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
            warnings.warn(f"qed_splatter_amd: get_outputs could not be captured into a hipGraph ({self.disabled}); "
                          "staying with eager dispatch for this model", RuntimeWarning, stacklevel=3)
            return None
        while len(self.segments) >= self.KEEP:    # (oldest first: the shapes of an earlier resolution / size)
            self.segments.pop(next(iter(self.segments)))
        self.segments[key] = seg
        return seg
