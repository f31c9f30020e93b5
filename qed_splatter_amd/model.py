"""Host-side mirror of the reference's model-level interface for the render hot path.

Mirrors /root/reference/qed_splatter/model.py:
  * ``get_viewmat``                    model.py:22-38
  * ``QEDSplatterModelConfig``         model.py:41-47 (fields depth_lambda, output_depth_during_training)
  * ``QEDSplatterModel.get_outputs``   model.py:199-321
  * ``QEDSplatterModel.get_loss_dict`` model.py:73-118 (depth-L1 term; the parent's main loss (1 - l) L1 + l (1 - SSIM))

The reference class inherits Nerfstudio's ``SplatfactoModel`` (not installed here);
this mirror is a plain ``nn.Module`` holding the same six parameter groups under the same names
(``means, scales, quats, features_dc, features_rest, opacities``; model.py:227-239) so checkpoints
interchange, and it accepts any camera object with the few attributes model.py:199-250 touches
(``Cameras`` of Nerfstudio or ``PinholeCameras`` below).  INTEGRATION.md shows the two-line change
that makes the real ``QEDSplatterModel`` call this package instead of gsplat.

Two ways to run a training step:
  * the reference's own call sequence: ``get_outputs`` -> ``get_metrics_dict`` -> ``get_loss_dict`` -> sum ->
    ``backward`` -> one optimiser per group (``QedAdam``).  The statements around the operator (model.py:295-306,
    87-116 and the parent's main loss) run inside the compositing kernels and in two fused autograd nodes
    (``rasterization(_post_background=...)``, ``_ImageLosses``);
  * fused: ``fused_loss`` = rasterization + the K8 loss / gradient kernels, ``backward_fused`` and ``FlatAdam`` -- one
    hipGraph-replayable step with the exp / sigmoid / cat of model.py:241,269-271 folded into the projection kernel.
"""
from __future__ import annotations

import math
import os
import weakref
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import torch
from torch import Tensor, nn

from . import _lib as L
from .rasterization import rasterization, _stream, _workspace


_FLIP_CACHE: Dict = {}


def get_viewmat(optimized_camera_to_world: Tensor) -> Tensor:
    """c2w [C,3,4] (OpenGL) -> gsplat world2camera [C,4,4]  (model.py:22-38)."""
    R = optimized_camera_to_world[:, :3, :3]
    T = optimized_camera_to_world[:, :3, 3:4]
    # pre-created per device (the reference keeps _FLIP_GSPLAT at module level, model.py:19-20): a
    # host-to-device copy per call would also be illegal inside a hipGraph capture
    key = (R.device, R.dtype)
    flip = _FLIP_CACHE.get(key)
    if flip is None:
        flip = _FLIP_CACHE[key] = torch.tensor([[[1.0, -1.0, -1.0]]], device=R.device, dtype=R.dtype)
    R = R * flip
    R_inv = R.transpose(1, 2)
    T_inv = -torch.bmm(R_inv, T)
    viewmat = torch.zeros(R.shape[0], 4, 4, device=R.device, dtype=R.dtype)
    viewmat[:, 3, 3] = 1.0
    viewmat[:, :3, :3] = R_inv
    viewmat[:, :3, 3:4] = T_inv
    return viewmat


@dataclass
class QEDSplatterModelConfig:
    """Fields the hot path reads.  The first two are the reference's own (model.py:44,46); the rest
    are the inherited SplatfactoModelConfig fields that model.py:199-321 touches."""
    depth_lambda: float = 0.2
    output_depth_during_training: bool = True
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    rasterize_mode: str = "classic"
    use_bilateral_grid: bool = False
    # the defaults below are SplatfactoModelConfig's (nerfstudio 1.1.x), which the reference's config.py:39-42 leaves
    # untouched: random training background, two halvings of the resolution that end at steps 3000 / 6000
    background_color: str = "random"         # "random" | "black" | "white"
    ssim_lambda: float = 0.2                 # parent's main loss: (1-l) L1 + l (1-SSIM)
    num_downscales: int = 2
    resolution_schedule: int = 3000
    use_scale_regularization: bool = False
    max_gauss_ratio: float = 10.0
    # ---- not reference fields ----
    # list each Gaussian only in the tiles of its 3-sigma square where some pixel can reach alpha >= 1/255
    # (QED_F_TIGHT_TILES): same images and gradients, shorter lists; info["flatten_ids"] etc. become subsets
    tight_tile_lists: bool = True
    # get_outputs(): after the first (calibrating) call do not read the intersection count back every step; a
    # buffer overflow then makes the NEXT call raise (the frame in between rendered empty)
    async_intersection_count: bool = True
    # get_outputs() of a training step as captured hipGraphs behind one autograd node (segments.py): ~0.15-0.2 ms less host
    # work per step.  True = "when it pays": a shape is captured once it has been stable for as many calls as the capture
    # costs (~240) AND the host, not the device, is the slower side; "always" = on the fourth call of a shape (tests,
    # benchmarks); False = never.  A captured step's outputs live in static buffers that the next get_outputs overwrites
    # (using older outputs raises)
    graph_segments: Union[bool, str] = True
    # Training steps whose six groups are stepped by QedAdam keep the SH gradients of features_dc / features_rest in the
    # compact form of the fused step (3 floats per Gaussian + the view; the optimiser evaluates b_k(dir) x colour gradient
    # itself) instead of writing and re-reading 48 N floats: 26 us of 57 in the projection backward at 500 k Gaussians.
    # ``.grad`` of the two Parameters stays correct for every Python reader: reading it materialises the full gradients in
    # place (_LazySHGradParameter); after QedAdam has consumed the compact form it is None (as after zero_grad()).  False:
    # always write the full gradients
    lazy_sh_grad: bool = True

    @classmethod
    def synthetic(cls, **kw) -> "QEDSplatterModelConfig":
        """The configuration of the synthetic benchmark / parity scenes (fixed black background, full
        resolution from step 0); every other field keeps the reference's default."""
        kw.setdefault("background_color", "black")
        kw.setdefault("num_downscales", 0)
        return cls(**kw)


class PinholeCameras:
    """The subset of nerfstudio ``Cameras`` that model.py:199-250 touches."""

    def __init__(self, camera_to_worlds: Tensor, fx: float, fy: float, cx: float, cy: float, width: int, height: int,
                 metadata: Optional[dict] = None):
        self.camera_to_worlds = camera_to_worlds                       # [C,3,4]
        C = camera_to_worlds.shape[0]
        dev = camera_to_worlds.device
        self.fx = torch.full((C, 1), float(fx), device=dev)
        self.fy = torch.full((C, 1), float(fy), device=dev)
        self.cx = torch.full((C, 1), float(cx), device=dev)
        self.cy = torch.full((C, 1), float(cy), device=dev)
        self.width = torch.full((C, 1), int(width), dtype=torch.int64)
        self.height = torch.full((C, 1), int(height), dtype=torch.int64)
        self.metadata = metadata

    @property
    def shape(self):
        return self.camera_to_worlds.shape[:1]

    def intrinsics_fxfycxcy(self) -> Tensor:
        """[C,4] device tensor (fx, fy, cx, cy), cached until the next rescale: the input of
        qed_camera_setup on the fused path."""
        if getattr(self, "_intr", None) is None:
            self._intr = torch.cat([self.fx, self.fy, self.cx, self.cy], dim=1).to(torch.float32).contiguous()
        return self._intr

    def get_intrinsics_matrices(self) -> Tensor:
        K = torch.zeros(self.shape[0], 3, 3, device=self.fx.device)
        K[:, 0, 0] = self.fx[:, 0]
        K[:, 1, 1] = self.fy[:, 0]
        K[:, 0, 2] = self.cx[:, 0]
        K[:, 1, 2] = self.cy[:, 0]
        K[:, 2, 2] = 1.0
        return K

    def rescale_output_resolution(self, s: float) -> None:
        self.fx = self.fx * s
        self.fy = self.fy * s
        self.cx = self.cx * s
        self.cy = self.cy * s
        self.width = (self.width * s).to(torch.int64)
        self.height = (self.height * s).to(torch.int64)
        self._intr = None


GROUP_ORDER = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")

_UNIT_GRADS: Dict = {}


def _unit_grad(device) -> Tensor:
    """One cached 0-dim tensor 1.0 per device: the seed gradient of backward_fused()."""
    one = _UNIT_GRADS.get(device)
    if one is None:
        one = _UNIT_GRADS[device] = torch.ones((), dtype=torch.float32, device=device)
    return one


def _refuse_overwritten(rgb: Tensor, who: str) -> None:
    """Outputs of a captured get_outputs segment live in static buffers: once a later get_outputs has replayed the segment
    they hold THAT step's image.  Using them is an error, not a silently wrong loss (segments.py)."""
    tag = getattr(rgb, "_qed_segment", None)
    if tag is not None and tag[0].generation != tag[1]:
        raise RuntimeError(f"{who}: these outputs belong to an earlier get_outputs call and have been overwritten (with "
                           "config.graph_segments a training step's outputs live in static buffers).  Use them before "
                           "the next get_outputs, or set config.graph_segments = False.")


def _is_camera(obj) -> bool:
    return all(hasattr(obj, a) for a in ("camera_to_worlds", "get_intrinsics_matrices", "width", "height"))


class _SSIM(torch.autograd.Function):
    """SSIM(pred, gt) of two [H,W,3] images with pytorch_msssim semantics (the parent's
    ``self.ssim``), value + gradient w.r.t. pred from ssim.hip."""

    @staticmethod
    def forward(ctx, pred, gt):
        lib = L.load()
        H, W, _ = pred.shape
        pred, gt = pred.contiguous(), gt.contiguous()
        n_maps = lib.qed_ssim_maps_floats(H, W)
        if n_maps < 0:
            raise L.QedSplatError("qed_ssim_maps_floats: image smaller than the 11 x 11 SSIM window")
        maps = torch.empty(n_maps, dtype=torch.float32, device=pred.device)
        ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), dtype=torch.float32, device=pred.device)
        L.check(lib.qed_ssim_fwd(H, W, 3, L.ptr(pred), None, None, L.ptr(gt), None, L.ptr(maps), L.ptr(ssum),
                                 _stream()), "qed_ssim_fwd")
        ctx.save_for_backward(pred, gt, maps)
        return ssum.sum() / (3.0 * (H - 10) * (W - 10))

    @staticmethod
    def backward(ctx, v):
        pred, gt, maps = ctx.saved_tensors
        H, W, _ = pred.shape
        v_pred = torch.empty_like(pred)
        v = v.to(torch.float32).reshape(1).contiguous()            # upstream gradient, multiplied inside the kernel
        L.check(L.load().qed_ssim_bwd(H, W, 3, L.ptr(pred), None, None, L.ptr(gt), None, L.ptr(maps),
                                      1.0 / (3.0 * (H - 10) * (W - 10)), L.ptr(v), L.ptr(v_pred), _stream()),
                "qed_ssim_bwd")
        return v_pred, None


def ssim(pred: Tensor, gt: Tensor) -> Tensor:
    """Mean SSIM of two float32 [H,W,3] images in [0,1] (differentiable in ``pred``)."""
    assert pred.dim() == 3 and pred.shape[-1] == 3 and pred.shape == gt.shape
    return _SSIM.apply(pred.to(torch.float32), gt.to(torch.float32))


def _f32_image(t: Tensor, numel: int, what: str, dev) -> Tensor:
    """A batch tensor as the kernels read it: float32, contiguous, on the model's device, ``numel`` elements
    (bool masks and uint8 images are converted; anything of another size is refused before a launch)."""
    if t.dtype == torch.uint8 and what != "mask":
        t = t.float() / 255.0
    t = t.to(device=dev, dtype=torch.float32).contiguous()
    if t.numel() != numel:
        raise L.QedSplatError(f"{what}: {tuple(t.shape)} holds {t.numel()} values, the render needs {numel}")
    return t


def _ssim_key(pred: Tensor, gt: Tensor):
    """What identifies the inputs of an SSIM forward: address, version and shape of both images."""
    return (pred.data_ptr(), pred._version, tuple(pred.shape), gt.data_ptr(), gt._version, tuple(gt.shape))


class StepContext:
    """What the calls of ONE training step share about it -- created by ``get_outputs`` (or ``fused_loss``), read by
    ``get_metrics_dict`` / ``get_loss_dict`` / the backward pass, and replaced by the next ``get_outputs``.  Everything in
    here is an OPTIMISATION of work that the calls can also do on their own (a conversion of the batch, an SSIM forward
    pass, a zero-filled accumulator), so every consumer falls back to the full computation when the context is not its
    own -- ``outputs`` of an earlier step, a second ``get_loss_dict`` on the same outputs, metrics taken under ``no_grad``:

      * ``rgb``: the very tensor this step's ``get_outputs`` returned as ``outputs["rgb"]`` -- ``owns(outputs)`` is an
        identity test, so ``outputs`` kept from an earlier step never pick up this step's state;
      * ``gt_image(...)``: the last conversion of a batch image (uint8 -> float, downscaling, device), keyed by the source
        tensor's identity, version and the downscale factor -- get_metrics_dict and get_loss_dict prepare the same image;
      * ``ssim``: get_metrics_dict's SSIM forward on (rgb, ground truth) with the maps and sums the loss needs -- taken ONCE
        (``take_ssim``) and checked against the tensors the loss is actually handed before it is used;
      * ``take_accumulator()``: (holder, rows) through which the loss's backward launch hands the compositing backward a
        zero-filled gradient accumulator -- once; a second loss on the same outputs makes the backward pass fill its own."""

    __slots__ = ("rgb", "_holder", "_rows", "_gt", "ssim", "static")

    def __init__(self):
        self.rgb = None
        self._holder, self._rows = None, 0
        self._gt = None
        self.ssim = None
        # a captured segment's static buffers (segments.OutputsSegment): (v_rgb [H,W,3], v_depth [H,W,1] | None, the
        # compositing backward's accumulator) -- the loss's backward launch writes / zeroes THESE instead of fresh ones
        self.static = None

    def owns(self, outputs) -> bool:
        return self.rgb is not None and outputs.get("rgb") is self.rgb

    def bind(self, rgb: Tensor, holder: list, rows: int) -> None:
        self.rgb, self._holder, self._rows = rgb, holder, rows

    def gt_image(self, image: Tensor, d: int, convert) -> Tensor:
        memo = self._gt
        if memo is not None and memo[0] is image and memo[1] == (image._version, d):
            return memo[2]
        out = convert(image)
        # (the source object is held with its conversion, so that its address cannot be recycled under the key)
        self._gt = (image, (image._version, d), out) if out is not image else None
        return out

    def take_ssim(self):
        shared, self.ssim = self.ssim, None
        return shared

    def take_accumulator(self):
        if self._holder is None:
            return None
        pair, self._holder = (self._holder, self._rows, self.static[2] if self.static is not None else None), None
        return pair

    def take_grad_buffers(self):
        """(v_rgb, v_depth) static buffers of a captured segment for the loss's backward launch, once; else None."""
        if self.static is None or self.static[0] is None:
            return None
        bufs, self.static = self.static[:2], (None, None, self.static[2])
        return bufs


class _PostProcess(torch.autograd.Function):
    """model.py:295-297 + 304-306 as ONE node: rgb = clamp(render[..., :3] + (1 - alpha) background, 0, 1) and
    depth = where(alpha > 0, render[..., 3:4], render[..., 3:4].detach().max())."""

    @staticmethod
    def forward(ctx, render, alpha, background):
        lib = L.load()
        ctx.set_materialize_grads(False)
        if not render.is_cuda:
            raise L.QedSplatError("get_outputs needs GPU tensors: there is no CPU path in the product")
        C, H, W, CH = render.shape
        dev = render.device
        render, alpha = render.contiguous(), alpha.contiguous()
        background = background.to(torch.float32).contiguous()
        rgb = torch.empty(C, H, W, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(C, H, W, 1, dtype=torch.float32, device=dev) if CH == 4 else None
        ws = torch.empty(L.LOSS_SUMS_FLOATS, dtype=torch.float32, device=dev) if CH == 4 else None
        L.check(lib.qed_post_process_fwd(C * H * W, CH, L.ptr(render), L.ptr(alpha), L.ptr(background), L.ptr(rgb),
                                         L.ptr(depth), L.ptr(ws), _stream()), "qed_post_process_fwd")
        ctx.save_for_backward(render, alpha, background)
        if depth is None:
            return rgb
        return rgb, depth

    @staticmethod
    def backward(ctx, v_rgb, v_depth=None):
        render, alpha, background = ctx.saved_tensors
        C, H, W, CH = render.shape
        if v_rgb is None and v_depth is None:
            return None, None, None
        v_rgb = v_rgb.to(torch.float32).contiguous() if v_rgb is not None else None
        v_depth = v_depth.to(torch.float32).contiguous() if v_depth is not None else None
        v_render = torch.empty_like(render)
        v_alpha = torch.empty_like(alpha)
        L.check(L.load().qed_post_process_bwd(C * H * W, CH, L.ptr(render), L.ptr(alpha), L.ptr(background), L.ptr(v_rgb),
                                              L.ptr(v_depth), L.ptr(v_render), L.ptr(v_alpha), _stream()),
                "qed_post_process_bwd")
        return v_render, v_alpha, None


class _ImageLosses(torch.autograd.Function):
    """get_loss_dict on the images get_outputs returned: the parent's main loss (1 - l) L1 + l (1 - SSIM) with the
    mask multiplied into both images (behind model.py:83-85) and the masked depth-L1 term (model.py:87-116), as two
    scalars.  The trainer sums the loss dict and differentiates, possibly with weights or a GradScaler: the backward
    multiplies each term's gradient by its upstream gradient, read from device memory."""

    @staticmethod
    def forward(ctx, rgb, depth, gt_rgb, gt_depth, mask, ssim_lambda, depth_lambda, ssim_shared=None, loss_shared=None,
                vsplat=None, grad_out=None):
        lib = L.load()
        ctx.set_materialize_grads(False)
        # (holder, rows, static buffer | None): the compositing backward's accumulator, zeroed by this node's backward launch
        ctx.vsplat = vsplat
        ctx.grad_out = grad_out   # (v_rgb, v_depth) static buffers of a captured get_outputs segment, or None
        if not rgb.is_cuda:
            raise L.QedSplatError("get_loss_dict needs GPU tensors: there is no CPU path in the product")
        H, W, _ = rgb.shape
        dev = rgb.device
        n_pix = H * W
        rgb = rgb.contiguous()
        depth = depth.contiguous() if depth is not None else None
        st = _stream()
        if loss_shared is not None and ssim_shared is not None:
            # get_metrics_dict ran qed_step_metrics on these very images a moment ago: the sums and the two losses exist
            sums, losses = loss_shared
            maps = ssim_shared[0]
            ctx.save_for_backward(rgb, depth, gt_rgb, gt_depth, mask, maps, sums)
            ctx.lams = (float(ssim_lambda), float(depth_lambda))
            return losses[0:1].view(()), losses[1:2].view(())
        sums = torch.empty(L.LOSS_SUMS_FLOATS, dtype=torch.float32, device=dev)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        maps = None
        extra = (None, 0, 0.0, 0.0)
        if ssim_lambda > 0.0:
            n_out = 3.0 * (H - 10) * (W - 10)
            n_maps = lib.qed_ssim_maps_floats(H, W)
            if n_maps < 0:
                raise L.QedSplatError("image smaller than the 11 x 11 SSIM window")
            if ssim_shared is not None:
                # get_metrics_dict ran qed_ssim_fwd on these very images a moment ago (rgb_ssim, model.py:157-166) and
                # kept the coefficient maps: the loss needs the same map sum and the same maps
                maps, ssum = ssim_shared
            else:
                maps = torch.empty(n_maps, dtype=torch.float32, device=dev)
                ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), dtype=torch.float32, device=dev)
                L.check(lib.qed_ssim_fwd(H, W, 3, L.ptr(rgb), None, None, L.ptr(gt_rgb), L.ptr(mask), L.ptr(maps),
                                         L.ptr(ssum), st), "qed_ssim_fwd")
            extra = (L.ptr(ssum), ssum.numel(), -ssim_lambda / n_out, ssim_lambda)
        L.check(lib.qed_image_losses_fwd(n_pix, L.ptr(rgb), L.ptr(depth), L.ptr(gt_rgb), L.ptr(gt_depth), L.ptr(mask),
                                         1.0 - ssim_lambda, depth_lambda, *extra, L.ptr(sums), L.ptr(losses), st),
                "qed_image_losses_fwd")
        ctx.save_for_backward(rgb, depth, gt_rgb, gt_depth, mask, maps, sums)
        ctx.lams = (float(ssim_lambda), float(depth_lambda))
        return losses[0:1].view(()), losses[1:2].view(())

    @staticmethod
    def backward(ctx, g_main, g_depth):
        lib = L.load()
        rgb, depth, gt_rgb, gt_depth, mask, maps, sums = ctx.saved_tensors
        ssim_lambda, depth_lambda = ctx.lams
        H, W, _ = rgb.shape
        st = _stream()

        def scalar(g):
            return None if g is None else g.to(torch.float32).reshape(1).contiguous()
        g_main, g_depth = scalar(g_main), scalar(g_depth)
        sv_rgb, sv_depth = ctx.grad_out if ctx.grad_out is not None else (None, None)

        def out_like(t, static):
            return static if (static is not None and static.shape == t.shape) else torch.empty_like(t)
        v_rgb = out_like(rgb, sv_rgb) if (g_main is not None and ctx.needs_input_grad[0]) else None
        v_depth = out_like(depth, sv_depth) if (g_depth is not None and depth is not None and ctx.needs_input_grad[1]) \
            else None
        if v_rgb is not None and ssim_lambda > 0.0:
            # ONE launch: the L1 term joins the SSIM term inside the SSIM backward pass, the depth term rides along
            n_out = 3.0 * (H - 10) * (W - 10)
            zero = None
            if ctx.vsplat is not None:
                holder, rows, static = ctx.vsplat
                zero = static if static is not None else \
                    torch.empty(rows, L.VSPLAT_FLOATS, dtype=torch.float32, device=rgb.device)
            L.check(lib.qed_image_losses_ssim_bwd(H, W, L.ptr(rgb), L.ptr(depth), L.ptr(gt_rgb), L.ptr(gt_depth),
                                                  L.ptr(mask), L.ptr(maps), L.ptr(sums), 1.0 - ssim_lambda, depth_lambda,
                                                  -ssim_lambda / n_out, L.ptr(g_main), L.ptr(g_depth), L.ptr(v_rgb),
                                                  L.ptr(v_depth), L.ptr(zero), zero.numel() if zero is not None else 0, st),
                    "qed_image_losses_ssim_bwd")
            if zero is not None:
                del holder[:]
                holder.append(zero)
        else:
            L.check(lib.qed_image_losses_bwd(H * W, L.ptr(rgb), L.ptr(depth), L.ptr(gt_rgb), L.ptr(gt_depth), L.ptr(mask),
                                             L.ptr(sums), 1.0 - ssim_lambda, depth_lambda, L.ptr(g_main), L.ptr(g_depth), 0,
                                             L.ptr(v_rgb), L.ptr(v_depth), st), "qed_image_losses_bwd")
        return v_rgb, v_depth, None, None, None, None, None, None, None, None, None


class _FusedImageLoss(torch.autograd.Function):
    """K8: composite + clamp + depth fix-up + L1 RGB + (1 - SSIM) + masked depth-L1, value and gradient
    (model.py:295-297, 304-306, 87-116 and the parent's main loss behind :83-85)."""

    @staticmethod
    def forward(ctx, render, alpha, background, gt_rgb, gt_depth, mask, ssim_lambda, depth_lambda, vsplat_holder=None,
                vsplat_rows=0, tick=None, tile_cost=None, order_buf=None):
        import ctypes
        lib = L.load()
        ctx.set_materialize_grads(False)
        C, H, W, CH = render.shape
        assert C == 1, "one camera per training step (model.py:211)"
        dev = render.device
        n_pix = H * W
        sums = torch.empty(L.LOSS_SUMS_FLOATS, dtype=torch.float32, device=dev)
        losses = torch.empty(3, dtype=torch.float32, device=dev)       # rgb term, depth term, total
        v_render = torch.empty_like(render)
        v_alpha = torch.empty_like(alpha)
        st = _stream()
        args = (n_pix, CH, L.ptr(render), L.ptr(alpha), L.ptr(background), L.ptr(gt_rgb), L.ptr(gt_depth), L.ptr(mask))
        if ssim_lambda > 0.0:
            # main = (1 - l) L1 + l (1 - SSIM): ONE launch forms the SSIM gradient w.r.t. the clamped colour and pushes
            # it, with the L1 part and the depth term, through the clamp / background composite (qed_loss_grad_ssim)
            n_out = 3.0 * (H - 10) * (W - 10)
            # the same launch zeroes the accumulator the compositing backward will add into (no fill launch there)
            vsplat = None
            if vsplat_holder is not None and vsplat_rows > 0:
                vsplat = torch.empty(vsplat_rows, L.VSPLAT_FLOATS, dtype=torch.float32, device=dev)
            maps = torch.empty(lib.qed_ssim_maps_floats(H, W), dtype=torch.float32, device=dev)
            ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), dtype=torch.float32, device=dev)
            # The SSIM forward launch carries two passengers that would otherwise be launches of their own on the step's
            # critical chain: pass 1 of the image loss (qed_loss_reduce: ~9 us) and -- the forward pass's per-tile costs
            # are known by now -- the compositing backward's launch order (~10 us in front of that kernel).
            order_ws = None
            if vsplat_holder is not None and tile_cost is not None:
                # (the camera's persistent launch-order buffer when the caller keeps one: model.fused_loss, frame_key)
                order_ws = order_buf if (order_buf is not None and order_buf.numel() == tile_cost.shape[0] + 1) else \
                    torch.empty(tile_cost.shape[0] + 1, dtype=torch.int32, device=dev)
            if os.environ.get("QED_STEP_PASSENGERS", "1") == "0":          # measurement hook: every job a launch of its own
                L.check(lib.qed_loss_reduce(*args, L.ptr(sums), st), "qed_loss_reduce")
                L.check(lib.qed_ssim_fwd(H, W, CH, L.ptr(render), L.ptr(alpha), L.ptr(background), L.ptr(gt_rgb),
                                         L.ptr(mask), L.ptr(maps), L.ptr(ssum), st), "qed_ssim_fwd")
                order_ws = None
            else:
                L.check(lib.qed_ssim_fwd_step(H, W, CH, L.ptr(render), L.ptr(alpha), L.ptr(background), L.ptr(gt_rgb),
                                            L.ptr(mask), L.ptr(maps), L.ptr(ssum), L.ptr(tile_cost) if order_ws is not None else None,
                                            tile_cost.shape[0] if order_ws is not None else 0, L.ptr(order_ws),
                                            L.ptr(gt_depth), L.ptr(sums), st), "qed_ssim_fwd_step")
            L.check(lib.qed_loss_grad_ssim(H, W, CH, L.ptr(render), L.ptr(alpha), L.ptr(background), L.ptr(gt_rgb),
                                           L.ptr(gt_depth), L.ptr(mask), L.ptr(maps), L.ptr(sums), 1.0 - ssim_lambda,
                                           depth_lambda, -ssim_lambda / n_out, L.ptr(v_render), L.ptr(v_alpha),
                                           L.ptr(losses), L.ptr(ssum), ssum.numel(), ssim_lambda, L.ptr(vsplat),
                                           vsplat.numel() if vsplat is not None else 0,
                                           ctypes.addressof(tick) if tick is not None else None, st), "qed_loss_grad_ssim")
            if vsplat is not None or order_ws is not None:
                del vsplat_holder[:]
                vsplat_holder.append({"vsplat": vsplat, "order_ws": order_ws})
        else:
            L.check(lib.qed_loss_reduce(*args, L.ptr(sums), st), "qed_loss_reduce")
            L.check(lib.qed_loss_grad(*args, L.ptr(sums), 1.0, depth_lambda, L.ptr(v_render), L.ptr(v_alpha),
                                      L.ptr(losses), None, None, 0, 0.0, 0.0, st), "qed_loss_grad")
        ctx.save_for_backward(v_render, v_alpha)
        total = losses[2:3].view(())
        parts = losses[0:2]
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, v_total, _v_parts):
        if v_total is None:
            return (None,) * 13
        v_render, v_alpha = ctx.saved_tensors
        # the kernel wrote d(total)/d(render, alpha).  The usual upstream gradient is the cached unit tensor of
        # backward_fused() and needs no scaling pass; anything else (a weighted loss, a GradScaler) is applied
        if v_total.data_ptr() != _unit_grad(v_total.device).data_ptr():
            v_render, v_alpha = v_render * v_total, v_alpha * v_total
        return v_render, v_alpha, None, None, None, None, None, None, None, None, None, None, None


def write_sh_grads(means: Tensor, viewmat: Tensor, sh_degree: int, v_color: Tensor, v_rest: Tensor) -> None:
    """In place: ``v_color`` [N,3] (the clamp-masked colour gradient qed_project_bwd leaves with QED_F_SH_GRAD_COMPACT)
    becomes the gradient of features_dc, ``v_rest`` [N,KR,3] receives the active coefficients' gradients (one view)."""
    n = v_color.shape[0]
    with torch.no_grad():
        L.check(L.load().qed_sh_grad_from_views(
            n, 1, L.ptr(means), L.ptr(viewmat), 16, L.ptr(v_color), 3 * n, int(sh_degree), 1.0,
            L.ptr(v_color), 3, L.ptr(v_rest), v_rest.numel() // max(n, 1), _stream()), "qed_sh_grad_from_views")


def _reference_key_order(metrics: Dict) -> Dict:
    """The metrics dict in the order the reference fills it (model.py:160-194: the four rgb entries, gaussian_count, the
    seven depth entries when the batch has a depth image, avg_min_scale) -- pinned by tests/golden/reference_kats.npz."""
    order = ("rgb_mse", "rgb_psnr", "rgb_ssim", "rgb_lpips", "gaussian_count", "depth_abs_rel", "depth_sq_rel", "depth_rmse",
             "depth_rmse_log", "depth_a1", "depth_a2", "depth_a3", "avg_min_scale")
    out = {k: metrics[k] for k in order if k in metrics}
    out.update({k: v for k, v in metrics.items() if k not in out})
    return out


def _counted_step(opt, device) -> None:
    """Tell the device's workspace that ``opt`` has counted a step behind the current frame (see _Workspace.counted_step)."""
    if device.type == "cuda":
        from .rasterization import _workspace
        _workspace(device).counted_step(opt)


_RAW_GRAD = torch.Tensor.grad                 # the C-level descriptor: reads / writes the field without the subclass's hooks


def _dist_world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _has_grad_hooks(p: Tensor) -> bool:
    """Tensor hooks (``register_hook``) or post-accumulate-grad hooks on a Parameter: both are handed the raw gradient."""
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def _raw_grad(p: Tensor) -> Optional[Tensor]:
    return _RAW_GRAD.__get__(p)


class _LazySHGradParameter(nn.Parameter):
    """features_dc / features_rest of a flat-buffer model.  A backward pass may leave their gradients in the compact form
    (QEDSplatterModel._lazy_sh: clamp-masked colour gradient + view); the autograd engine then stores the views of the flat
    gradient allocation in the ``.grad`` FIELD as always, but the memory does not hold the coefficient gradients yet.
    ``.grad`` read from Python first writes them there (one qed_sh_grad_from_views launch, in place), so every reader --
    torch.optim.*, clip_grad_norm_, GradScaler.unscale_, logging -- sees what the reference's backward pass produces; only
    QedAdam, which evaluates the product itself, looks at the field without asking for that."""

    @property
    def grad(self):
        owner = self.__dict__.get("_qed_owner")
        if owner is not None:
            m = owner()
            if m is not None and m.__dict__.get("_lazy_sh") is not None:
                m._materialise_sh_grads()
        return _RAW_GRAD.__get__(self)

    @grad.setter
    def grad(self, value):
        owner = self.__dict__.get("_qed_owner")
        m = owner() if owner is not None else None
        if m is not None and m.__dict__.get("_lazy_sh") is not None:
            if value is not None:                    # (somebody assigns one of the two: give the other its values first)
                m._materialise_sh_grads()
            _RAW_GRAD.__set__(self, value)
            m._lazy_sh_dropped()
            return
        _RAW_GRAD.__set__(self, value)


class QEDSplatterModel(nn.Module):
    """Mirror of QEDSplatterModel (model.py:50-321) for the render hot path."""

    def __init__(self, config: Optional[QEDSplatterModelConfig] = None, *, means: Tensor, scales: Tensor,
                 quats: Tensor, opacities: Tensor, features_dc: Tensor, features_rest: Tensor,
                 separate_params: bool = False):
        super().__init__()
        self.config = config or QEDSplatterModelConfig()
        N = means.shape[0]
        srcs = dict(means=means, scales=scales, quats=quats, opacities=opacities.reshape(N, 1),
                    features_dc=features_dc.reshape(N, 3), features_rest=features_rest)
        self.group_names = list(GROUP_ORDER)
        self.step = 0
        self.crop_box = None
        self.camera_optimizer = None
        if separate_params:
            # Six independent tensors, as Nerfstudio's parent class holds them (model.py:12,50-58; one optimiser per
            # group, config.py:44-68).  get_outputs / get_loss_dict / fused_loss run unchanged; what needs the flat
            # layout (FlatAdam, Densifier, the data-parallel exchange) refuses.
            self._flat = None
            self.group_begin = []
            self.gauss_params = nn.ParameterDict(
                {n: nn.Parameter(srcs[n].detach().to(torch.float32).clone().contiguous()) for n in self.group_names})
            return
        # One flat buffer holds all six groups (59 N floats for SH degree 3); the six Parameters are
        # leaf views into it.  _ProjectSH.backward lays the six gradients out in the same order in
        # one allocation, so the data-parallel all-reduce and the fused Adam step each
        # touch a single contiguous range and nothing is ever concatenated.
        total = sum(srcs[n].numel() for n in self.group_names)
        flat = torch.empty(total, dtype=torch.float32, device=means.device)
        self.group_begin: List[int] = [0]
        params = {}
        off = 0
        for name in self.group_names:
            src = srcs[name]
            n = src.numel()
            flat[off:off + n] = src.reshape(-1).to(torch.float32)
            params[name] = self._make_param(name, flat[off:off + n].view(src.shape))
            off += n
            self.group_begin.append(off)
        self._flat = flat
        self.gauss_params = nn.ParameterDict(params)          # same container name as SplatfactoModel

    def _make_param(self, name: str, view: Tensor) -> nn.Parameter:
        """A leaf view of the flat buffer; the two SH groups can hold their gradient in compact form (lazy_sh_grad)."""
        if name not in ("features_dc", "features_rest"):
            return nn.Parameter(view)
        p = _LazySHGradParameter(view)
        p._qed_owner = weakref.ref(self)
        return p

    # ---- lazy SH gradients (config.lazy_sh_grad) ----
    def _lazy_sh_wanted(self, sh_degree_to_use, crop_ids) -> bool:
        """May THIS training step's backward pass leave the SH gradients compact?  Only when all six groups of the flat
        buffer are stepped by QedAdam instances (their fused launch runs the SH groups before it moves the means the SH basis
        is evaluated at), nothing waits in the two ``.grad`` fields and the whole Gaussian set is rendered."""
        attrs = self.__dict__
        if attrs.get("_lazy_sh") is not None and self.training and torch.is_grad_enabled():
            # an earlier backward pass's compact gradients are still in the fields (no zero_grad in between): the pass that
            # follows this forward will be ADDED to them, so they are completed now
            self._materialise_sh_grads()
            return False
        if not (self.config.lazy_sh_grad and self.training and sh_degree_to_use is not None and crop_ids is None
                and attrs.get("_flat") is not None and torch.is_grad_enabled()):
            return False
        dc, rest = self.gauss_params["features_dc"], self.gauss_params["features_rest"]
        if type(dc) is not _LazySHGradParameter or type(rest) is not _LazySHGradParameter \
                or not (dc.requires_grad and rest.requires_grad and self.gauss_params["means"].requires_grad):
            return False
        # Only PYTHON reads of .grad complete a compact gradient.  Readers that take the field in C++ never pass through
        # the property: DistributedDataParallel's reducer copies variable.grad() into its buckets from autograd hooks
        # (Nerfstudio wraps the model in DDP for multi-GPU training), tensor hooks and post-accumulate-grad hooks receive
        # the raw tensor.  With more than one rank in the default process group, or with hooks on either Parameter, the
        # gradients are therefore always written out.  (This package's own data-parallel step, parallel.py, does not come
        # through here: it exchanges the compact form itself.)
        if _dist_world_size() > 1 or _has_grad_hooks(dc) or _has_grad_hooks(rest):
            return False
        st = _FLAT_STATES.get(self._flat.untyped_storage().data_ptr())
        if st is None or len(st.members) != len(self.group_names):
            return False
        return attrs.get("_lazy_sh") is None and _raw_grad(dc) is None and _raw_grad(rest) is None

    def _lazy_sh_begin(self, v_sh0: Tensor, v_shN: Optional[Tensor], viewmats: Tensor, sh_degree: int) -> bool:
        """Asked by the projection backward (rasterization(_lazy_sh=...)) right before its launch: True = write the compact
        form into ``v_sh0`` (and leave ``v_shN``'s active coefficients unwritten), recorded here until it is consumed
        (QedAdam), materialised (a ``.grad`` read) or dropped (``.grad = None``).  False (gradients of an earlier backward
        pass are waiting in the fields: autograd is about to ADD to them): those are completed first and this pass writes
        full gradients."""
        dc, rest = self.gauss_params["features_dc"], self.gauss_params["features_rest"]
        if self.__dict__.get("_lazy_sh") is not None:
            self._materialise_sh_grads()
            return False
        if _raw_grad(dc) is not None or _raw_grad(rest) is not None or v_shN is None or viewmats.shape[0] != 1:
            return False
        means = self.gauss_params["means"]
        # (aliases of the two views, not the objects themselves: the engine adopts an incoming gradient as the .grad field
        # without a copy only while nobody else holds that tensor object -- a reference kept here would make it clone)
        self.__dict__["_lazy_sh"] = {"v_color": v_sh0.detach(), "v_rest": v_shN.detach(), "viewmat": viewmats,
                                     "deg": int(sh_degree),
                                     "n": int(v_sh0.shape[0]), "means_version": means._version,
                                     "means_ptr": means.data_ptr()}
        return True

    def _lazy_sh_dropped(self) -> None:
        """A ``.grad`` field of the two was overwritten: with both empty the compact form is gone."""
        if _raw_grad(self.gauss_params["features_dc"]) is None and _raw_grad(self.gauss_params["features_rest"]) is None:
            self.__dict__["_lazy_sh"] = None

    def _materialise_sh_grads(self) -> None:
        """Write the coefficient gradients the compact form stands for into the very allocation the ``.grad`` fields view
        (features_dc: b_0 x colour gradient in place of the colour gradient; features_rest: the active coefficients)."""
        rec = self.__dict__.get("_lazy_sh")
        if rec is None:
            return
        self.__dict__["_lazy_sh"] = None
        means = self.gauss_params["means"]
        if means._version != rec["means_version"] or means.data_ptr() != rec["means_ptr"] or means.shape[0] != rec["n"]:
            raise RuntimeError(
                "lazy SH gradients: the means were modified after the backward pass and before the gradients of features_dc "
                "/ features_rest were read (the SH basis is evaluated at the means of the forward pass).  Step all six groups "
                "with QedAdam, read the gradients before stepping, or set config.lazy_sh_grad = False")
        write_sh_grads(means, rec["viewmat"], rec["deg"], rec["v_color"], rec["v_rest"])

    def rebind_flat(self, flat: Tensor, n_points: int) -> None:
        """Adopt a new flat parameter buffer (densification changes N): the six Parameters are
        re-created as leaf views into it, in group order, with their per-Gaussian shapes kept."""
        old_n = max(self.num_points, 1)
        shapes = {n: (n_points,) + tuple(self.gauss_params[n].shape[1:]) for n in self.group_names}
        widths = {n: self.gauss_params[n].numel() // old_n for n in self.group_names}
        if self.num_points == 0:
            widths = {n: int(torch.tensor(shapes[n][1:]).prod()) if len(shapes[n]) > 1 else 1 for n in self.group_names}
        assert flat.numel() == n_points * sum(widths.values()) and flat.dtype == torch.float32 and flat.is_contiguous()
        params, begin, off = {}, [0], 0
        for name in self.group_names:
            n = n_points * widths[name]
            params[name] = self._make_param(name, flat[off:off + n].view(shapes[name]))
            off += n
            begin.append(off)
        self._flat = flat
        self.group_begin = begin
        self.__dict__["_lazy_sh"] = None
        self.gauss_params = nn.ParameterDict(params)

    # ---- parameter groups (same names as the reference reads at model.py:227-239) ----
    means = property(lambda self: self.gauss_params["means"])
    scales = property(lambda self: self.gauss_params["scales"])
    quats = property(lambda self: self.gauss_params["quats"])
    opacities = property(lambda self: self.gauss_params["opacities"])
    features_dc = property(lambda self: self.gauss_params["features_dc"])
    features_rest = property(lambda self: self.gauss_params["features_rest"])

    @property
    def device(self):
        return self.gauss_params["means"].device

    @property
    def flat_params(self) -> Tensor:
        """All six groups as one contiguous [59 N] tensor (aliases the Parameters)."""
        if self._flat is None:
            raise RuntimeError("this model holds six separate Parameters (separate_params=True): there is no flat buffer "
                               "(FlatAdam, Densifier and the data-parallel exchange need the default layout)")
        return self._flat

    def flat_grad(self) -> Optional[Tensor]:
        """The six ``.grad`` tensors as one contiguous tensor.  Zero-copy when they alias one
        allocation in group order (what _ProjectSH.backward produces); otherwise concatenated."""
        grads = [self.gauss_params[n].grad for n in self.group_names]
        if any(g is None for g in grads):
            return None
        if self._flat is None:
            self.flat_params                     # (raises: no flat layout)
        g0 = grads[0]
        base, ok = g0.storage_offset(), True
        for g, beg in zip(grads, self.group_begin):
            ok = ok and g.is_contiguous() and g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr() \
                and g.storage_offset() == base + beg
        total = self.group_begin[-1]
        if ok and g0.untyped_storage().nbytes() >= 4 * (base + total):
            return torch.empty(0, dtype=torch.float32, device=g0.device).set_(g0.untyped_storage(), base, (total,))
        flat = torch.cat([g.reshape(-1) for g in grads])
        off = 0
        for n, g in zip(self.group_names, grads):           # re-alias so later steps stay zero-copy
            self.gauss_params[n].grad = flat[off:off + g.numel()].view(g.shape)
            off += g.numel()
        return flat

    @property
    def num_points(self) -> int:
        return self.gauss_params["means"].shape[0]

    def get_param_groups(self) -> Dict[str, List[Tensor]]:
        return {n: [self.gauss_params[n]] for n in self.group_names}

    def _apply(self, fn, *args, **kwargs):
        """model.to() / .cuda() / .float() replace every Parameter's data: gather the six groups into a fresh flat
        buffer on the new device and re-create the Parameters as views of it (optimisers must be rebuilt, as after
        any parameter replacement; FlatAdam / QedAdam detect a stale buffer and raise)."""
        super()._apply(fn, *args, **kwargs)
        if self._flat is None:
            return self
        ps = [self.gauss_params[n] for n in self.group_names]
        if any(p.data_ptr() != self._flat.data_ptr() + 4 * b or p.device != self._flat.device
               for p, b in zip(ps, self.group_begin)):
            flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in ps])
            self.rebind_flat(flat, ps[0].shape[0])
        return self

    # ---- inherited helpers model.py calls: restatements of SplatfactoModel (nerfstudio 1.1.x) ----
    def _get_downscale_factor(self) -> int:
        if self.training:
            return 2 ** max(self.config.num_downscales - self.step // self.config.resolution_schedule, 0)
        return 1

    def _downscale_if_required(self, image: Tensor) -> Tensor:
        """The parent's resize_image: d x d box filter with stride d ("area" downscaling), float32."""
        d = self._get_downscale_factor()
        if d > 1:
            image = image.to(torch.float32)
            weight = torch.full((1, 1, d, d), 1.0 / (d * d), dtype=torch.float32, device=image.device)
            return torch.nn.functional.conv2d(image.permute(2, 0, 1)[:, None, ...], weight, stride=d).squeeze(1).permute(1, 2, 0)
        return image

    def _get_background_color(self) -> Tensor:
        dev = self.device
        if self.config.background_color == "random" and self.training:
            return torch.rand(3, device=dev)
        # the fixed colours are made once per device (a fill launch per step otherwise); nothing writes into them
        if self.config.background_color == "random":
            # the parent's eval colour under "random" (populate_modules: self.background_color)
            key, rgb = "eval", [0.1490, 0.1647, 0.2157]
        elif self.config.background_color == "white":
            key, rgb = "white", [1.0, 1.0, 1.0]
        else:
            key, rgb = "black", [0.0, 0.0, 0.0]
        cache = self.__dict__.setdefault("_bg_consts", {})
        c = cache.get((key, dev))
        if c is None:
            c = cache[(key, dev)] = torch.tensor(rgb, dtype=torch.float32, device=dev)
        return c

    def get_gt_img(self, image: Tensor) -> Tensor:
        """uint8 -> float / 255, downscaled by the current factor, on the model's device (the parent's get_gt_img,
        called at model.py:88,91,94)."""
        def convert(img):
            out = img.float() / 255.0 if img.dtype == torch.uint8 else img
            return self._downscale_if_required(out).to(self.device)

        # get_metrics_dict and get_loss_dict prepare the SAME batch image within one step: the step's context keeps the
        # last conversion (StepContext.gt_image)
        ctx = self.__dict__.get("_step")
        return convert(image) if ctx is None else ctx.gt_image(image, self._get_downscale_factor(), convert)

    def composite_with_background(self, image: Tensor, background: Tensor) -> Tensor:
        """RGBA ground truth composited onto the step's background (the parent does this to the GT image)."""
        if image.shape[2] == 4:
            alpha = image[..., -1].unsqueeze(-1).repeat((1, 1, 3))
            return alpha * image[..., :3] + (1 - alpha) * background
        return image

    def _apply_bilateral_grid(self, rgb: Tensor, cam_idx: int, H: int, W: int) -> Tensor:
        raise NotImplementedError(
            "use_bilateral_grid: the bilateral grid lives in the Nerfstudio parent class (lib_bilagrid) and is out of "
            "scope for this mirror (SURVEY a8); it is reached unchanged when the real QEDSplatterModel uses this "
            "package's rasterization() (INTEGRATION.md)")

    def get_empty_outputs(self, width: int, height: int, background: Tensor) -> Dict[str, Tensor]:
        rgb = background.repeat(height, width, 1)
        depth = background.new_ones(*rgb.shape[:2], 1) * 10
        accumulation = background.new_zeros(*rgb.shape[:2], 1)
        return {"rgb": rgb, "depth": depth, "accumulation": accumulation, "background": background}

    def _outputs_segment(self, W, H, render_mode, deg, flags, render_fn, cam_c2w, background):
        """The captured form of this call's device work (segments.OutputsSegment), or None while the shape is still being
        seen eagerly / after anything the capture was specialised on has changed."""
        from .rasterization import _workspace
        from .segments import OutputsSegment, SegmentCache
        cache = self.__dict__.get("_segments")
        if cache is None:
            cache = self.__dict__["_segments"] = SegmentCache()
        ws = _workspace(self.device)
        ps = [self.gauss_params[n] for n in GROUP_ORDER]
        C = cam_c2w[0].shape[0]
        shape_key = ((W, H), self.num_points, C)
        key = (shape_key, render_mode, deg, flags, self.config.rasterize_mode, tuple(p.data_ptr() for p in ps),
               tuple(bool(p.requires_grad) for p in ps))
        # the count of the previous frame (eager or replayed) comes back here; an overflow drops every capture: their
        # buffers are too small, and the next call has to read M back
        ws.poll_pending()
        if ws.force_sync or not ws.calibrated(shape_key) or not ws.host_words_ok:
            cache.drop_all()
            return None
        seg = cache.get(key)
        if seg is not None:
            return seg
        if not cache.should_capture(key, self.config.graph_segments, ws):
            return None
        def make():
            seg = OutputsSegment(self.device, ps, render_fn, shape_key)
            seg.lazy_owner = weakref.ref(self) if (flags & L.F_SH_GRAD_COMPACT) else None
            return seg

        return cache.capture(key, make, cam_c2w[0], cam_c2w[1], background)

    # ---- a2-a10: get_outputs (model.py:199-321) ----
    def get_outputs(self, camera) -> Dict[str, Union[Tensor, List]]:
        if not _is_camera(camera):
            print("Called get_outputs with not a camera")                     # model.py:206-208
            return {}
        if self.training:
            assert camera.shape[0] == 1, "Only one camera at a time"          # model.py:211
            if self.camera_optimizer is not None:
                optimized_camera_to_world = self.camera_optimizer.apply_to_camera(camera)
            else:
                optimized_camera_to_world = camera.camera_to_worlds
        else:
            optimized_camera_to_world = camera.camera_to_worlds

        if self.crop_box is not None and not self.training:                   # model.py:217-224
            crop_ids = self.crop_box.within(self.means).squeeze()
            if crop_ids.sum() == 0:
                return self.get_empty_outputs(int(camera.width.item()), int(camera.height.item()),
                                              self._get_background_color())
        else:
            crop_ids = None

        if crop_ids is not None:                                              # model.py:226-239
            opacities_crop = self.opacities[crop_ids]
            means_crop = self.means[crop_ids]
            features_dc_crop = self.features_dc[crop_ids]
            features_rest_crop = self.features_rest[crop_ids]
            scales_crop = self.scales[crop_ids]
            quats_crop = self.quats[crop_ids]
        else:
            opacities_crop, means_crop = self.opacities, self.means
            features_dc_crop, features_rest_crop = self.features_dc, self.features_rest
            scales_crop, quats_crop = self.scales, self.quats

        BLOCK_WIDTH = 16                                                      # model.py:243
        camera_scale_fac = self._get_downscale_factor()
        if camera_scale_fac != 1:                     # (x 1.0 and back is exact: eight tiny launches saved per step)
            camera.rescale_output_resolution(1 / camera_scale_fac)
        intr = getattr(camera, "intrinsics_fxfycxcy", None)
        if intr is not None and optimized_camera_to_world.dtype == torch.float32 \
                and not optimized_camera_to_world.requires_grad and optimized_camera_to_world.is_cuda:
            # get_viewmat + get_intrinsics_matrices inside the projection kernel (QED_F_CAMERA_C2W), which fills these
            # two buffers for everything downstream -- instead of ~15 tiny eager launches, or one of qed_camera_setup
            C = optimized_camera_to_world.shape[0]
            viewmat = torch.empty(C, 4, 4, dtype=torch.float32, device=self.device)
            K = torch.empty(C, 3, 3, dtype=torch.float32, device=self.device)
            cam_c2w = (optimized_camera_to_world.contiguous(), intr())
        else:
            viewmat = get_viewmat(optimized_camera_to_world)
            K = camera.get_intrinsics_matrices().to(self.device)
            cam_c2w = None
        W, H = int(camera.width.item()), int(camera.height.item())
        attrs = self.__dict__            # (plain attributes: nn.Module.__setattr__ costs ~5 us apiece, a dozen per step)
        attrs["last_size"] = (H, W)
        # what get_metrics_dict / get_loss_dict / backward share about THIS step lives in one object, replaced here: a
        # loader that refills its batch tensors in place without bumping their version counter must not be served last
        # step's conversion, and nothing of last step's outputs may reach this step's loss
        ctx = attrs["_step"] = StepContext()
        if camera_scale_fac != 1:
            camera.rescale_output_resolution(camera_scale_fac)

        if self.config.rasterize_mode not in ["antialiased", "classic"]:      # model.py:253-254
            raise ValueError("Unknown rasterize_mode: %s", self.config.rasterize_mode)
        if self.config.output_depth_during_training or not self.training:    # model.py:256-259
            render_mode = "RGB+D"
        else:
            render_mode = "RGB"

        flags = L.F_LOG_SCALES | L.F_LOGIT_OPAC                               # exp / sigmoid fused (model.py:270-271)
        if self.config.tight_tile_lists and W <= 16 * 1023 and H <= 16 * 2047:
            flags |= L.F_TIGHT_TILES
        if self.config.sh_degree > 0:                                         # model.py:261-265
            sh_degree_to_use = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)
            colors, sh_rest = features_dc_crop, features_rest_crop            # no torch.cat (model.py:241)
        else:
            sh_degree_to_use = None
            colors, sh_rest = features_dc_crop, None
            flags |= L.F_SIGMOID_COLORS                                       # torch.sigmoid(colors) fused

        # the SH gradients of this step may stay compact until somebody reads them (config.lazy_sh_grad)
        lazy = self._lazy_sh_wanted(sh_degree_to_use, crop_ids)
        if lazy:
            flags |= L.F_SH_GRAD_COMPACT

        background = self._get_background_color()
        holder: list = []         # (get_loss_dict's backward launch leaves the compositing backward's zeroed accumulator here)

        # the camera's launch-order slot (fused_loss: frame_key): the reference's trainer hands the camera index in
        # camera.metadata["cam_idx"]; only on the eager route -- a captured segment is replayed for every camera
        frame_slot = None
        meta = getattr(camera, "metadata", None)
        if self.training and torch.is_grad_enabled() and meta is not None and "cam_idx" in meta and crop_ids is None:
            frame_slot = self._frame_order_slot(meta["cam_idx"], H, W)

        def render_fn(c2w, intr, bg, hold, capture_slot=None, viewmats=None, Ks=None, manual=None):
            """The rasterization(...) call of model.py:267-288 (+ the statements that follow it, inside the compositing
            kernels): here on this call's tensors, and -- once the shape has been seen a few times -- captured on static
            ones (segments.OutputsSegment)."""
            C = 1 if c2w is None else c2w.shape[0]
            if viewmats is None:
                viewmats = torch.empty(C, 4, 4, dtype=torch.float32, device=self.device)
                Ks = torch.empty(C, 3, 3, dtype=torch.float32, device=self.device)
            return rasterization(
                means=means_crop,
                quats=quats_crop,                       # normalised inside the projection kernel (model.py:269)
                scales=scales_crop,
                opacities=opacities_crop,
                colors=colors,
                viewmats=viewmats,
                Ks=Ks,
                width=W,
                height=H,
                tile_size=BLOCK_WIDTH,
                packed=False,
                near_plane=0.01,
                far_plane=1e10,
                render_mode=render_mode,
                sh_degree=sh_degree_to_use,
                sparse_grad=False,
                absgrad=True,
                rasterize_mode=self.config.rasterize_mode,
                _flags=flags,
                _sh_rest=sh_rest,
                _sync=not (self.config.async_intersection_count and self.training),
                _c2w=(c2w, intr) if c2w is not None else None,
                _post_background=bg,
                _vsplat_holder=hold,
                _means2d_leaf=True,     # xys is only retained and read (below; densify.py): its gradient arrives as a view
                _capture_slot=capture_slot,
                _manual=manual,
                # (a captured backward pass always writes the compact form; _SegmentFn.backward asks per replay)
                _lazy_sh=self._lazy_sh_begin if (lazy and manual is None) else None,
                _tile_order=frame_slot if (manual is None and capture_slot is None) else None,
            )

        seg = None
        if (self.training and self.config.graph_segments and cam_c2w is not None and crop_ids is None
                and torch.is_grad_enabled() and self.config.async_intersection_count):
            seg = self._outputs_segment(W, H, render_mode, sh_degree_to_use, flags, render_fn, cam_c2w, background)
        if seg is not None:
            rgb, alpha, depth_im = seg.run(cam_c2w[0], cam_c2w[1], background)
            info, holder, render = seg.info, seg.holder, None
            ctx.static = (seg.v_rgb[0], seg.v_depth[0] if seg.v_depth is not None else None, seg.vsplat)
            background = seg.bg
        else:
            render, alpha, info = render_fn(cam_c2w[0] if cam_c2w else None, cam_c2w[1] if cam_c2w else None, background,
                                            holder, viewmats=viewmat.to(torch.float32), Ks=K.to(torch.float32))
            # model.py:296-297 (composite + clamp) and :304-308 (depth fix-up) ran inside the compositing kernel, and
            # their backward runs inside the compositing backward (rasterization(_post_background=...)): no pass of its
            # own over the image in either direction (_PostProcess above is the stand-alone form of the same statements)
            rgb = info.pop("post_rgb")
            depth_im = info.pop("post_depth")
        # The frame BEFORE this one overflowed its intersection buffer (asynchronous count, one call late): it rendered
        # empty, so its outputs, loss and gradients were those of an empty image.  QedAdam / FlatAdam skipped their update
        # on the device; a trainer that steps torch.optim.* (the reference's own config.py:44-68) sees it here -- in the
        # dict the reference stores as self.info (model.py:267) -- and can drop that iteration's step / restore its state
        info["intersection_overflow_previous_frame"] = _workspace(self.device).take_overflow_flag()
        attrs["info"] = info
        attrs["last_compact"] = False
        if self.training and info["means2d"].requires_grad:                   # model.py:289-290 (a no-op on the leaf)
            info["means2d"].retain_grad()
        attrs["xys"] = info["means2d"]                                        # [1,N,2]
        attrs["radii"] = info["radii"][0]                                     # [N]
        if depth_im is not None:
            depth_im = depth_im.squeeze(0)

        if self.config.use_bilateral_grid and self.training:                  # model.py:300-302 (not built: raises)
            if getattr(camera, "metadata", None) is not None and "cam_idx" in camera.metadata:
                rgb = self._apply_bilateral_grid(rgb, camera.metadata["cam_idx"], H, W)

        # model.py:310-311 `del render; torch.cuda.empty_cache()` is a per-call device sync +
        # allocator flush with no effect on results; deliberately not reproduced.

        if background.shape[0] == 3 and not self.training:                    # model.py:313-314
            background = background.expand(H, W, 3)
        rgb = rgb.squeeze(0)
        if seg is not None:
            rgb._qed_segment = (seg, seg.generation)       # (get_loss_dict / get_metrics_dict refuse them once overwritten)
        ctx.bind(rgb, holder, info["radii"].numel())
        return {
            "rgb": rgb,
            "depth": depth_im,
            "accumulation": alpha.squeeze(0),
            "background": background,
        }

    # ---- a11: get_loss_dict (model.py:73-118) ----
    def _scale_reg(self) -> Tensor:
        """The parent's scale regulariser: 0 unless use_scale_regularization, then every 10th step
        0.1 * mean(max(max_scale / min_scale, max_gauss_ratio) - max_gauss_ratio)."""
        cfg = self.config
        if cfg.use_scale_regularization and self.step % 10 == 0:
            scale_exp = torch.exp(self.scales)
            ratio = scale_exp.amax(dim=-1) / scale_exp.amin(dim=-1)
            reg = torch.maximum(ratio, torch.tensor(cfg.max_gauss_ratio, device=ratio.device)) - cfg.max_gauss_ratio
            return 0.1 * reg.mean()
        zero = getattr(self, "_zero_loss", None)
        if zero is None or zero.device != self.device:
            zero = self._zero_loss = torch.tensor(0.0, device=self.device)
        return zero

    def _loss_mask(self, batch, shape) -> Optional[Tensor]:
        """batch["mask"] [H,W,1] (bool or float; Nerfstudio's are bool) downscaled like the images, as float32."""
        if "mask" not in batch:
            return None
        mask = batch["mask"]
        if mask.dtype == torch.uint8:
            raise TypeError("mask must be bool or floating point (a uint8 mask would scale the images by up to 255 in "
                            "the parent's loss and by 1/255-steps in the depth term, model.py:91-97)")
        mask = self._downscale_if_required(mask).to(self.device)
        assert mask.shape[:2] == tuple(shape[:2]), f"mask {tuple(mask.shape)} vs image {tuple(shape)}"   # model.py:95
        return mask.to(torch.float32).contiguous()

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, Tensor]:
        """Same keys and values as the reference (model.py:73-118 on top of the parent's dict): main_loss =
        (1 - l) L1 + l (1 - SSIM) of the masked images, scale_reg, depth_loss = depth_lambda * masked depth-L1
        (0.0 when no pixel is valid, model.py:111-114).  One fused node instead of ~30 eager launches with boolean
        gathers; each entry stays separately differentiable (the trainer sums and may weight them)."""
        cfg = self.config
        pred_img = outputs["rgb"]
        depth_out = outputs["depth"]
        if depth_out is None:
            raise TypeError("get_loss_dict needs outputs['depth'] (the reference fails the same way with "
                            "output_depth_during_training=False, model.py:87,101)")
        H, W = pred_img.shape[:2]
        _refuse_overwritten(pred_img, "get_loss_dict")
        ctx = self.__dict__.get("_step")
        mine = ctx is not None and ctx.owns(outputs)          # these outputs are this step's (not kept from an earlier one)
        gt_img = self.composite_with_background(self.get_gt_img(batch["image"]), outputs["background"])
        mask = self._loss_mask(batch, pred_img.shape)
        gt_img = _f32_image(gt_img[..., :3] if gt_img.shape[-1] > 3 else gt_img, H * W * 3, "batch['image']", self.device)
        depth_batch = _f32_image(self.get_gt_img(batch["depth_image"]), H * W, "batch['depth_image']", self.device)
        # the SSIM forward get_metrics_dict ran on the same two images (same storage, same version; the cache holds the
        # tensors, so neither address can have been recycled), no mask: not computed a second time
        shared = ctx.take_ssim() if mine else None
        if shared is not None and (mask is not None or cfg.ssim_lambda <= 0.0 or
                                   shared["key"] != _ssim_key(pred_img.contiguous(), gt_img)):
            shared = None
        # ... and its L1 / depth sums, when get_metrics_dict took them along (same depth tensors, same weights)
        loss_shared = None
        if shared is not None and shared.get("loss") is not None:
            dc = depth_out.contiguous()
            if shared["depth_key"] == (dc.data_ptr(), dc._version, depth_batch.data_ptr(), depth_batch._version) \
                    and shared["lambdas"] == (float(cfg.ssim_lambda), float(cfg.depth_lambda)):
                loss_shared = shared["loss"]
        # the accumulator of the compositing backward behind THESE outputs is zeroed by this loss's backward launch (the
        # first loss taken on them: a second one leaves the fill to the backward pass)
        pair = ctx.take_accumulator() if (mine and torch.is_grad_enabled()) else None
        grad_out = ctx.take_grad_buffers() if (mine and torch.is_grad_enabled()) else None
        main, depth = _ImageLosses.apply(pred_img, depth_out, gt_img, depth_batch, mask, float(cfg.ssim_lambda),
                                         float(cfg.depth_lambda), shared["maps_sum"] if shared else None, loss_shared, pair,
                                         grad_out)
        return {"main_loss": main, "scale_reg": self._scale_reg(), "depth_loss": depth}

    # ---- get_metrics_dict (model.py:120-197) ----
    def get_metrics_dict(self, outputs, batch) -> Dict[str, Tensor]:
        """Same keys as the reference, but every value is a 0-dim DEVICE tensor (or an int for
        ``gaussian_count``): the reference's ``float(...)``/``.item()`` per entry (model.py:160-182)
        is a device synchronisation each, which caps iterations/s regardless of kernel speed; the
        caller converts when (and if) it logs.  ``rgb_lpips`` is NaN (no pretrained weights here)."""
        from .metrics import metrics_dict as _image_metrics, nanmean_exp
        d = self._get_downscale_factor()

        def resize(img):                                                       # model.py:131-147 (TF.resize, bilinear)
            if d <= 1:
                return img
            size = (img.shape[0] // d, img.shape[1] // d)
            return torch.nn.functional.interpolate(img.permute(2, 0, 1)[None].float(), size=size, mode="bilinear",
                                                   align_corners=False, antialias=False)[0].permute(1, 2, 0)

        if d <= 1:
            gt_rgb = self.get_gt_img(batch["image"])[..., :3]              # (the conversion get_loss_dict will reuse)
        else:
            # model.py:131-135 resizes the batch image itself -- NOT through get_gt_img, which would halve it again
            img = batch["image"]
            gt_rgb = resize(img.float() / 255.0 if img.dtype == torch.uint8 else img).to(self.device)[..., :3]
        pred_rgb = outputs["rgb"][0] if outputs["rgb"].dim() == 4 else outputs["rgb"]
        _refuse_overwritten(outputs["rgb"], "get_metrics_dict")
        has_depth = "depth_image" in batch and outputs.get("depth") is not None
        gt_depth = resize(batch["depth_image"]).to(self.device) if has_depth else None
        # In training the loss that follows needs the SSIM of the same two images WITH the coefficient maps of its
        # backward pass: compute that form once here and leave it for get_loss_dict (which checks that it is handed the
        # same tensors before using it)
        ctx = self.__dict__.get("_step")
        keep = (self.training and torch.is_grad_enabled() and self.config.ssim_lambda > 0.0 and d <= 1
                and pred_rgb.is_cuda and pred_rgb.dtype == torch.float32 and ctx is not None and ctx.owns(outputs))
        with torch.no_grad():
            if keep and has_depth:
                # a training step: the metrics, and what the loss that follows needs from the same images, in one pass
                from .metrics import step_metrics
                out, shared = step_metrics(pred_rgb.detach(), gt_rgb, outputs["depth"].detach(), gt_depth,
                                           self.scales[..., -1], float(self.config.ssim_lambda), float(self.config.depth_lambda))
                ctx.ssim = shared
                out["gaussian_count"] = self.num_points
                return _reference_key_order(out)
            out = dict(_image_metrics(pred_rgb.detach(), gt_rgb, outputs["depth"].detach() if has_depth else None, gt_depth,
                                      keep_ssim_maps=keep))
            kept = out.pop("_ssim_shared", None)
            if keep:
                ctx.ssim = kept
            out["gaussian_count"] = self.num_points
            out["avg_min_scale"] = nanmean_exp(self.scales[..., -1])                  # model.py:192-194
        return _reference_key_order(out)

    @property
    def intersection_overflows(self) -> int:
        """Frames of this device that overflowed their intersection buffer so far (they rendered empty; QedAdam / FlatAdam
        skipped their updates on the device).  A trainer that steps other optimisers -- torch.optim.Adam is NOT protected
        by the skip flag: the empty frame's zero gradients still make a momentum-only update -- can watch this count and
        drop the step when it moves.  The count of a frame arrives one call late (poll_pending).  (An attribute, not a
        key of the metrics dict: that dict keeps the reference's keys.)"""
        from .rasterization import _workspace
        return _workspace(self.device).overflows

    def frame_overflowed(self) -> bool:
        """Did the frame the LAST get_outputs / fused_loss call enqueued overflow its intersection buffer (it then rendered
        empty: zero loss gradients)?  Asked between ``backward()`` and the optimiser steps by a trainer whose optimisers do
        not take the device-side skip word -- ``torch.optim.Adam``, the reference's own config.py:44-68, would make a
        momentum-only update from the empty frame -- so that it can drop that iteration's step:

            loss.backward()
            if not model.frame_overflowed():
                for o in optimizers.values(): o.step()

        Waits until the device has run that frame's binning (the count lands in pinned memory), not for the frame.  The
        same fact reaches ``self.info["intersection_overflow_previous_frame"]`` at the next get_outputs."""
        ws = _workspace(self.device)
        ws.poll_pending()
        return bool(ws.last_overflow)

    def backward_fused(self, losses: Dict[str, Tensor]) -> None:
        """``losses["loss"].backward()`` without the per-step ``ones_like`` fill autograd would launch for the
        seed gradient (the fused loss kernel has already written d loss / d render for a seed of 1)."""
        losses["loss"].backward(gradient=_unit_grad(losses["loss"].device))

    # ---- fused training step: model.py:199-321 + 73-118 in as few passes as possible ----
    def _frame_order_slot(self, frame_key, H: int, W: int) -> list:
        """[order buffer [tiles + 1] int32, holds-an-order flag] of camera ``frame_key`` at this resolution: the launch order
        a frame's compositing backward is given (from that frame's per-tile work counts), which the NEXT frame of the same
        camera hands its compositing forward.  Persistent, so that a captured step replays against it."""
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        orders = self.__dict__.setdefault("_frame_orders", {})
        slot = orders.get((frame_key, H, W))
        if slot is None or slot[0].device != self.device:
            slot = [torch.empty(tiles + 1, dtype=torch.int32, device=self.device), False]
            orders[(frame_key, H, W)] = slot
        return slot

    def fused_loss(self, camera, batch, background: Optional[Tensor] = None, sync: bool = True,
                   compact_sh_grad: bool = False, optimizer: Optional["FlatAdam"] = None,
                   frame_key=None) -> Dict[str, Tensor]:
        """Forward + K8 fused loss.  Returns {"loss", "main_loss", "depth_loss"}: ``loss`` = main + depth is
        the differentiable total (call ``.backward()`` on it as is: the kernel already wrote its gradient
        for an upstream gradient of 1); the two parts are detached views for logging.  Numerically the
        same quantities as get_outputs + get_loss_dict.

        ``frame_key`` (hashable, optional): identifies the CAMERA this frame is rendered from -- the dataset's camera
        index.  Frames under one key share a launch-order buffer: the order the loss launch computes for this frame's
        compositing backward (from this frame's per-tile work counts) is what the NEXT frame under the same key hands
        its compositing FORWARD kernel, which cannot know its costs in advance (heaviest tiles first: 114-117 us against
        122-126 at config B).  A scheduling hint only: images and gradients do not depend on it.  Leave it None when
        consecutive frames come from unrelated cameras and no index is at hand."""
        assert camera.shape[0] == 1, "Only one camera at a time"
        if self.__dict__.get("_lazy_sh") is not None and torch.is_grad_enabled():
            self._materialise_sh_grads()          # (compact gradients of a get_outputs step nobody consumed: see there)
        cfg = self.config
        self.__dict__["_step"] = StepContext()   # (conversions of the batch are shared within a step, never across steps)
        # the coarse-to-fine schedule of get_outputs (model.py:244-250): render at 1/d of the camera's resolution
        d = self._get_downscale_factor()
        if d > 1:
            camera.rescale_output_resolution(1 / d)
        try:
            intr = getattr(camera, "intrinsics_fxfycxcy", None)
            if intr is not None and camera.camera_to_worlds.dtype == torch.float32 and camera.camera_to_worlds.is_cuda:
                # a1 + a3 inside the projection kernel (QED_F_CAMERA_C2W): it fills viewmat / K for what follows
                viewmat = torch.empty(1, 4, 4, dtype=torch.float32, device=self.device)
                K = torch.empty(1, 3, 3, dtype=torch.float32, device=self.device)
                cam_c2w = (camera.camera_to_worlds.contiguous(), intr())
            else:
                viewmat = get_viewmat(camera.camera_to_worlds).to(self.device, torch.float32)
                K = camera.get_intrinsics_matrices().to(self.device, torch.float32)
                cam_c2w = None
            W, H = int(camera.width[0]), int(camera.height[0])
        finally:
            if d > 1:
                camera.rescale_output_resolution(d)
        self.last_size = (H, W)
        # tight tile lists: info["tiles_per_gauss"/"flatten_ids"/...] become subsets of gsplat's (nothing on the
        # training path reads them); images, alphas and gradients are unchanged
        flags = L.F_LOG_SCALES | L.F_LOGIT_OPAC | (L.F_TIGHT_TILES if cfg.tight_tile_lists else 0)
        if compact_sh_grad and cfg.sh_degree > 0:
            # data parallel: features_dc.grad then holds the clamp-masked colour gradient and features_rest.grad
            # is not written; parallel.exchange_grads_compact() rebuilds both from all ranks' views
            flags |= L.F_SH_GRAD_COMPACT
        self.last_viewmat = None
        self.last_compact = bool(flags & L.F_SH_GRAD_COMPACT)
        self.sh_views = None                  # set by parallel.exchange_grads_compact(rebuild=False)
        if cfg.sh_degree > 0:
            deg = min(self.step // cfg.sh_degree_interval, cfg.sh_degree)
            colors, sh_rest = self.features_dc, self.features_rest
        else:
            deg, colors, sh_rest = None, self.features_dc, None
            flags |= L.F_SIGMOID_COLORS
        # ground truth exactly as get_loss_dict prepares it (uint8 -> float, downscaled, on the device), checked
        # against the render size BEFORE any kernel reads it through a raw pointer
        bg = (background if background is not None else self._get_background_color()).to(self.device, torch.float32)
        gt_rgb = self.composite_with_background(self.get_gt_img(batch["image"]), bg)
        gt_rgb = _f32_image(gt_rgb[..., :3] if gt_rgb.shape[-1] > 3 else gt_rgb, H * W * 3, "batch['image']", self.device)
        gt_depth = _f32_image(self.get_gt_img(batch["depth_image"]), H * W, "batch['depth_image']", self.device)
        mask = self._loss_mask(batch, (H, W))
        holder: list = []         # (the fused loss launch leaves the compositing backward's zeroed accumulator here)
        # the launch-order buffer of this camera (see ``frame_key``): [C T + 1] int32, written by every training frame's
        # loss launch, read by the next frame's compositing forward -- persistent, so that a captured step replays against it
        frame_slot, frame_order, frame_order_valid = None, None, False
        if frame_key is not None:
            frame_slot = self._frame_order_slot(frame_key, H, W)
            frame_order, frame_order_valid = frame_slot[0], frame_slot[1]
        # ``optimizer`` (a FlatAdam stepped with device_state=True, fused_sh=True right after this step's backward): the
        # loss pass's fold launch advances its device step state, so that the optimiser needs no launch of its own for it
        tick = None
        if optimizer is not None and torch.is_grad_enabled() and cfg.ssim_lambda > 0.0:
            tick = optimizer.take_tick()
        try:
            render, alpha, self.info = rasterization(
                means=self.means, quats=self.quats, scales=self.scales, opacities=self.opacities, colors=colors,
                viewmats=viewmat, Ks=K, width=W, height=H, tile_size=16, packed=False, near_plane=0.01, far_plane=1e10,
                render_mode="RGB+D", sh_degree=deg, sparse_grad=False, absgrad=True,
                rasterize_mode=cfg.rasterize_mode, _flags=flags, _sh_rest=sh_rest, _sync=sync, _vsplat_holder=holder,
                _c2w=cam_c2w, _tile_order=frame_order if frame_order_valid else None)
            self.xys = self.info["means2d"]
            self.radii = self.info["radii"][0]
            self.last_viewmat, self.last_sh_degree = viewmat, deg
            # (the compositing node keeps the forward pass's per-tile costs: the loss launch sorts them for its backward)
            tile_cost = getattr(render.grad_fn, "tile_cost", None) if torch.is_grad_enabled() else None
            total, parts = _FusedImageLoss.apply(render, alpha, bg.contiguous(), gt_rgb, gt_depth, mask,
                                                 float(cfg.ssim_lambda), cfg.depth_lambda,
                                                 holder if torch.is_grad_enabled() else None, self.num_points, tick,
                                                 tile_cost, frame_order)
            if frame_slot is not None and tile_cost is not None and float(cfg.ssim_lambda) > 0.0 \
                    and os.environ.get("QED_STEP_PASSENGERS", "1") != "0":
                frame_slot[1] = True              # (the buffer holds an order from here on, in stream order)
        except BaseException:
            # the launch that would have advanced the optimiser's device step state did not happen: the optimiser
            # must tick for itself on the next step (otherwise its counter would be off by one from here on)
            if tick is not None:
                optimizer.drop_tick()
            raise
        return {"loss": total, "main_loss": parts[0], "depth_loss": parts[1]}


def exponential_decay_lr(step: int, lr_init: float, lr_final: float, max_steps: int, warmup_steps: int = 0,
                         lr_pre_warmup: float = 0.0, ramp: str = "cosine") -> float:
    """Nerfstudio's ExponentialDecayScheduler (the schedule config.py:46-51 / :63-67 attach to "means" and
    "camera_opt"): log-linear from lr_init to lr_final over max_steps after an optional warm-up ramp."""
    import math
    if step < warmup_steps:
        if ramp == "cosine":
            return lr_pre_warmup + (lr_init - lr_pre_warmup) * math.sin(0.5 * math.pi * min(max(step / warmup_steps, 0), 1))
        return lr_pre_warmup + (lr_init - lr_pre_warmup) * step / warmup_steps
    t = min(max((step - warmup_steps) / (max_steps - warmup_steps), 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class FlatAdam:
    """Fused multi-tensor Adam over the model's flat parameter buffer, one learning rate per group
    (the six Gaussian groups of config.py:44-68, eps=1e-15) and the reference's exponential decay of
    the "means" rate (config.py:46-51).  SURVEY 8(f) rank 2."""

    MEANS_SCHEDULE = (1.6e-6, 30000)           # lr_final, max_steps (config.py:48-50)

    DEFAULT_LRS = {"means": 1.6e-4, "scales": 0.005, "quats": 0.001, "opacities": 0.05,
                   "features_dc": 0.0025, "features_rest": 0.0025 / 20}

    def __init__(self, model: QEDSplatterModel, lrs: Optional[Dict[str, float]] = None, betas=(0.9, 0.999),
                 eps: float = 1e-15, means_schedule: Optional[tuple] = None):
        """``means_schedule=(lr_final, max_steps)`` turns on the exponential decay of the "means" rate
        (``FlatAdam.MEANS_SCHEDULE`` is the reference's); None keeps every rate constant."""
        import ctypes as C
        self.model = model
        self.means_schedule = means_schedule
        lrs = {**self.DEFAULT_LRS, **(lrs or {})}
        self._means_lr_init = float(lrs["means"])
        self.lr = [float(lrs[n]) for n in model.group_names]
        begins = list(model.group_begin)
        self._begin = (C.c_int64 * len(begins))(*begins)
        self._lr = (C.c_float * len(self.lr))(*self.lr)
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(model.flat_params)
        self.exp_avg_sq = torch.zeros_like(model.flat_params)
        self._flat_ref = model.flat_params         # (held: a freed buffer's address can be handed out again)
        self.t = 0
        # device-resident step state + learning rates: what a captured hipGraph replays against
        self.dev_state = torch.zeros(4, dtype=torch.float32, device=model.device)
        self.dev_lr = torch.zeros(8, dtype=torch.float32, device=model.device)
        self.dev_lr[:len(self.lr)] = torch.tensor(self.lr)
        from .rasterization import _workspace
        _workspace(model.device).steppers.add(self)

    def on_skipped_step(self) -> None:
        """The device skipped the step this optimiser's host counter has already counted (the frame behind it overflowed its
        intersection buffer: _Workspace.poll_pending): take it back, so that the bias corrections of the host-counter path
        stay in step with the moments.  (The device-state path counts on the device, where the tick honours the skip.)
        Not in a data-parallel job: there the skip is collective (parallel.py) but only the rank whose frame overflowed
        hears of it on the host -- taking the step back here alone would make the replicas' bias corrections differ.  All
        ranks' host counters then run one ahead of the moments together; step with device_state=True for exact counts."""
        if getattr(self.model, "_dp_skip", None) is not None:
            return
        self.t = max(self.t - 1, 0)

    def take_tick(self):
        """The qed_adam_tick_t that lets ANOTHER launch of the step advance this optimiser's device step state
        (model.fused_loss(optimizer=...): the loss pass's fold launch does it); the next
        ``step(device_state=True, fused_sh=True)`` then launches no tick of its own.  None while a tick is pending."""
        if getattr(self, "_ticked", False):
            return None
        i = self.model.group_names.index("means")
        t = L.AdamTick()
        t.dev_state, t.beta1, t.beta2 = self.dev_state.data_ptr(), self.betas[0], self.betas[1]
        t.skip_flag = self._skip()
        if self.means_schedule is not None:
            lr_final, max_steps = self.means_schedule
            t.dev_lr_slot = self.dev_lr[i:i + 1].data_ptr()
            t.lr_init, t.lr_final, t.max_steps = self._means_lr_init, float(lr_final), int(max_steps)
        else:
            t.dev_lr_slot, t.lr_init, t.lr_final, t.max_steps = None, 0.0, 0.0, 0
        self._tick_struct, self._ticked = t, True               # (kept alive: the C call reads it through a pointer)
        return t

    def drop_tick(self) -> None:
        """Forget a tick handed out by take_tick() whose launch never happened (a failed graph capture, an exception
        between take_tick() and the loss launch)."""
        self._ticked = False

    def _skip(self) -> int:
        """``skip_flag`` of the Adam entry points: the binning overflow word of this device (a frame whose intersection
        list overflowed renders empty; a step enqueued behind it without a host round trip must be a no-op)."""
        # data parallel: the MAXIMUM of the ranks' words, so that every replica skips the same steps (parallel.py)
        dp = getattr(self.model, "_dp_skip", None)
        if dp is not None:
            return dp.data_ptr()
        from .rasterization import _workspace
        return _workspace(self.model.device).skip_flag_ptr()

    # ---- checkpointing (config.py:29 steps_per_save: the trainer saves every optimiser's state_dict) ----
    def state_dict(self) -> Dict:
        return {"t": self.t, "lr": list(self.lr), "betas": tuple(self.betas), "eps": self.eps,
                "means_schedule": self.means_schedule, "means_lr_init": self._means_lr_init,
                "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "dev_state": self.dev_state.clone(), "dev_lr": self.dev_lr.clone(), "numel": self.exp_avg.numel()}

    def load_state_dict(self, sd: Dict) -> None:
        if int(sd["numel"]) != self.model.flat_params.numel():
            raise ValueError(f"FlatAdam.load_state_dict: the checkpoint holds moments for {sd['numel']} parameters, the "
                             f"model has {self.model.flat_params.numel()} (load the model's Gaussians first)")
        self._check("load_state_dict", compact_ok=True)
        self.t = int(sd["t"])
        self.betas, self.eps = tuple(sd["betas"]), float(sd["eps"])
        self.means_schedule, self._means_lr_init = sd["means_schedule"], float(sd["means_lr_init"])
        for i, lr in enumerate(sd["lr"]):
            self.lr[i] = float(lr)
            self._lr[i] = float(lr)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.dev_state.copy_(sd["dev_state"])
        self.dev_lr.copy_(sd["dev_lr"])
        self._ticked = False

    def set_lr(self, name: str, lr: float) -> None:
        i = self.model.group_names.index(name)
        self.lr[i] = float(lr)
        self._lr[i] = float(lr)
        self.dev_lr[i] = float(lr)

    def rebind(self, exp_avg: Tensor, exp_avg_sq: Tensor) -> None:
        """Adopt new moment buffers after the model adopted a new flat parameter buffer (densification);
        the step count carries on, as torch.optim.Adam's per-parameter ``step`` does in the reference."""
        import ctypes as C
        assert exp_avg.numel() == self.model.flat_params.numel() == exp_avg_sq.numel()
        begins = list(self.model.group_begin)
        self._begin = (C.c_int64 * len(begins))(*begins)
        self.exp_avg, self.exp_avg_sq = exp_avg, exp_avg_sq
        self._flat_ref = self.model.flat_params

    def _check(self, who: str, compact_ok: bool = False) -> None:
        """Refuse to train on garbage: a flat buffer the Parameters no longer alias (model.to() / densification
        without rebind()), or compact SH gradients consumed by a plain step."""
        m = self.model
        if m.flat_params is not self._flat_ref or m.flat_params.device != self.exp_avg.device:
            raise RuntimeError(f"FlatAdam.{who}: the model adopted a new flat parameter buffer (model.to() or "
                               "densification); call rebind(exp_avg, exp_avg_sq) or build a new optimiser")
        if not compact_ok and getattr(m, "last_compact", False):
            raise RuntimeError(f"FlatAdam.{who}: the last backward wrote compact SH gradients "
                               "(fused_loss(compact_sh_grad=True)); step with fused_sh=True")

    # ---- one step in pieces: lets a data-parallel job update a range of the flat buffer as soon as that
    # range of the gradient has been all-reduced, while later ranges are still on the wire ----
    @torch.no_grad()
    def begin_step(self) -> None:
        """Advance the (host-side) step counter and the scheduled rates; follow with step_range() calls that
        together cover [0, numel)."""
        if self.means_schedule is not None:
            lr_final, max_steps = self.means_schedule
            i = self.model.group_names.index("means")
            self.lr[i] = exponential_decay_lr(self.t, self._means_lr_init, lr_final, max_steps)
            self._lr[i] = self.lr[i]
        self.t += 1
        _counted_step(self, self.model.device)

    @torch.no_grad()
    def step_range(self, lo: int, hi: int) -> None:
        """Adam update of flat elements [lo, hi) (lo a multiple of 4) with the current step count."""
        import ctypes as C
        assert 0 <= lo <= hi <= self.model.flat_params.numel() and lo % 4 == 0
        if hi == lo:
            return
        self._check("step_range")
        g = self.model.flat_grad()
        begins = [min(max(b - lo, 0), hi - lo) for b in self.model.group_begin]
        h_begin = (C.c_int64 * len(begins))(*begins)
        p = self.model.flat_params
        L.check(L.load().qed_adam_step(L.ptr(p[lo:hi]), L.ptr(g[lo:hi]), L.ptr(self.exp_avg[lo:hi]),
                                       L.ptr(self.exp_avg_sq[lo:hi]), len(self.lr), C.cast(h_begin, C.c_void_p),
                                       C.cast(self._lr, C.c_void_p), self.betas[0], self.betas[1], self.eps, self.t,
                                       self._skip(), _stream()), "qed_adam_step")

    @torch.no_grad()
    def step(self, device_state: bool = False, fused_sh: bool = False, part: int = 3) -> None:
        """One Adam step.  ``device_state=True`` keeps the step counter / bias corrections in device
        memory (qed_adam_step_dev), which is what makes the step replayable from a hipGraph.

        ``fused_sh=True`` (after ``fused_loss(..., compact_sh_grad=True)`` + backward): the 48 N SH-coefficient
        gradients are never written or read -- qed_adam_step_sh evaluates b_k(dir) x colour gradient while it
        updates features_dc / features_rest (one view: this rank's; data parallel: the views gathered by
        ``parallel.exchange_grads_compact(..., rebuild=False)``).  Same update as the plain step.

        ``part`` (fused_sh only): 1 = the step counter / schedule + the two SH groups (needs the gathered views, not the
        reduced geometry gradients), 2 = the leading groups, 3 = both.  A data-parallel step calls 1 then 2 around the
        wait for the geometry all-reduce (``parallel.exchange_grads_compact_begin``)."""
        import ctypes as C
        assert part in (1, 2, 3) and (fused_sh or part == 3)
        p = self.model.flat_params
        g = self.model.flat_grad()
        if g is None:
            return
        self._check("step", compact_ok=fused_sh)
        lib = L.load()
        sched = (-1, 0.0, 0.0, 0)
        if self.means_schedule is not None and (part & 1):      # the rate of the step about to be taken
            lr_final, max_steps = self.means_schedule
            i = self.model.group_names.index("means")
            if device_state and fused_sh:                        # evaluated by qed_adam_step_sh's own tick launch
                sched = (i, self._means_lr_init, float(lr_final), int(max_steps))
            elif device_state:
                L.check(lib.qed_lr_exp_decay_dev(L.ptr(self.dev_lr[i:i + 1]), L.ptr(self.dev_state), self._means_lr_init,
                                                 float(lr_final), int(max_steps), _stream()), "qed_lr_exp_decay_dev")
            else:
                self.lr[i] = exponential_decay_lr(self.t, self._means_lr_init, lr_final, max_steps)
                self._lr[i] = self.lr[i]
        if part & 1:
            self.t += 1
            _counted_step(self, self.model.device)
        if fused_sh:
            m = self.model
            if not getattr(m, "last_compact", False):
                raise RuntimeError("fused_sh needs gradients from fused_loss(..., compact_sh_grad=True)")
            assert m.group_names[-2:] == ["features_dc", "features_rest"]
            views = m.sh_views
            if views is None:                                    # this rank's view only
                b = m.group_begin
                views = (1, m.last_viewmat, 16, g[b[-3]:b[-2]], 0, 1.0)
            n_views, viewmats, vm_stride, v_views, view_stride, scale = views
            if getattr(self, "_ticked", False) and (part & 1):       # take_tick(): the state is advanced already
                if not device_state:
                    raise RuntimeError("the device step state was advanced by fused_loss(optimizer=...): step with "
                                       "device_state=True")
                part, self._ticked = part | 4, False
            L.check(lib.qed_adam_step_sh(
                L.ptr(p), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), len(self.lr),
                C.cast(self._begin, C.c_void_p), None if device_state else C.cast(self._lr, C.c_void_p),
                L.ptr(self.dev_lr) if device_state else None, self.betas[0], self.betas[1], self.eps, self.t,
                L.ptr(self.dev_state) if device_state else None, *sched, m.num_points, int(m.last_sh_degree or 0),
                L.ptr(m.means), n_views, L.ptr(viewmats), vm_stride, L.ptr(v_views), view_stride, float(scale),
                int(part), self._skip(), _stream()), "qed_adam_step_sh")
            return
        if device_state:
            L.check(lib.qed_adam_step_dev(L.ptr(p), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                          len(self.lr), C.cast(self._begin, C.c_void_p), L.ptr(self.dev_lr),
                                          self.betas[0], self.betas[1], self.eps, L.ptr(self.dev_state), self._skip(),
                                          _stream()), "qed_adam_step_dev")
            return
        L.check(lib.qed_adam_step(L.ptr(p), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                  len(self.lr), C.cast(self._begin, C.c_void_p), C.cast(self._lr, C.c_void_p),
                                  self.betas[0], self.betas[1], self.eps, self.t, self._skip(), _stream()),
                "qed_adam_step")


class _SharedFlatState:
    """What the QedAdam instances of one model share: the moments over the whole flat buffer and this step's
    bookkeeping (which groups have called step(), with which rate)."""

    def __init__(self, flat: Tensor):
        self.flat_ptr = flat.data_ptr()
        self.numel = flat.numel()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.members = weakref.WeakValueDictionary()   # flat offset -> optimiser instance
        self.pending: Dict[int, float] = {}            # flat offset -> lr of the step() call waiting to be launched
        self.t: Dict[int, int] = {}                    # flat offset -> steps taken


class QedAdamSet:
    """The per-group QedAdam instances of one model, seen as ONE optimiser over the flat buffer -- what
    ``densify.Densifier`` needs (it rewrites parameters and both Adam moments in one pass when the number of Gaussians
    changes, as the parent's dup_in_all_optim / remove_from_all_optim do group by group).

        optimizers = {name: QedAdam([model.gauss_params[name]], lr=..., eps=1e-15) for name in model.group_names}
        densifier = Densifier(model, QedAdamSet(model, optimizers), ...)
    """

    def __init__(self, model: "QEDSplatterModel", optimizers: Dict[str, "QedAdam"]):
        assert set(optimizers) >= set(model.group_names), "one QedAdam per parameter group"
        self.model, self.optimizers = model, optimizers

    def _state(self) -> _SharedFlatState:
        return self.optimizers[self.model.group_names[0]]._attach()

    @property
    def exp_avg(self) -> Tensor:
        return self._state().exp_avg

    @property
    def exp_avg_sq(self) -> Tensor:
        return self._state().exp_avg_sq

    def rebind(self, exp_avg: Tensor, exp_avg_sq: Tensor) -> None:
        """After ``model.rebind_flat``: point every instance at its new Parameter and adopt the new moments (the step
        counts carry on, as torch.optim.Adam's per-parameter ``step`` does in the reference)."""
        m = self.model
        old = self._state_or_none()
        steps = {}
        for name, beg in zip(m.group_names, m.group_begin):
            opt = self.optimizers[name]
            old_off = opt._param().storage_offset()
            steps[beg] = old.t.get(old_off, 0) if old is not None else 0
            opt.param_groups[0]["params"] = [m.gauss_params[name]]
            opt._shared = None
            opt.state.clear()                                    # (views of the old moments, keyed by the old Parameter)
        st = self._state()                                       # a fresh shared state over the new flat buffer
        assert exp_avg.numel() == st.numel == exp_avg_sq.numel()
        st.exp_avg, st.exp_avg_sq = exp_avg, exp_avg_sq
        st.t.update(steps)
        for opt in self.optimizers.values():                     # optimizer.state[param]: views of the adopted moments
            opt.state.clear()
            if opt._shared is st:
                opt._expose_views(st)

    def _state_or_none(self) -> Optional[_SharedFlatState]:
        return self.optimizers[self.model.group_names[0]]._shared


_FLAT_STATES: "weakref.WeakValueDictionary[int, _SharedFlatState]" = weakref.WeakValueDictionary()
_ALL_QED_ADAMS: "weakref.WeakSet[QedAdam]" = weakref.WeakSet()


def _skip_flag(device) -> int:
    from .rasterization import _workspace
    return _workspace(device).skip_flag_ptr()


class QedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` semantics (no weight decay, no amsgrad) for ONE parameter group of the Gaussians, as a
    ``torch.optim.Optimizer`` subclass: Nerfstudio's ``AdamOptimizerConfig(_target=QedAdam, lr=..., eps=1e-15)`` builds
    it unchanged for every group of config.py:44-68 and its schedulers / GradScaler / checkpointing keep working
    (``param_groups[0]["lr"]`` is read at every step).  Two layouts, told apart by the Parameter's storage:

    * **separately held Parameters** (what Nerfstudio's parent class keeps: six tensors in a ``ParameterDict``,
      model.py:12,50-58): the moments live in ``self.state[param]`` exactly as ``torch.optim.Adam`` keeps them
      (``step`` / ``exp_avg`` / ``exp_avg_sq``), so code that rewrites them when the number of Gaussians changes (the
      parent's dup / remove-from-optimiser routines, gsplat's strategies) works unchanged; every ``step()`` is one
      ``qed_adam_step`` launch over that tensor.
    * **views of ONE flat buffer** (this package's ``QEDSplatterModel``): the instances of one buffer find each other
      through a registry and the LAST one to be stepped launches a single ``qed_adam_step`` over the whole buffer with
      one rate per group (one pass at HBM speed instead of six times ~10 eager launches).  The shared moments are
      visible as VIEWS under ``self.state[param]``; replacing them there is refused with an error that names
      ``densify.Densifier`` / ``QedAdamSet`` (which rewrite parameters and moments in one pass).  A group that is
      stepped twice before the others, ``zero_grad()``, ``state_dict()`` or ``flush()`` update just the waiting groups'
      ranges -- so a group whose ``.grad`` is None in some iteration never delays the others past that iteration."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if weight_decay != 0.0:
            raise NotImplementedError("QedAdam: weight_decay is not used by the reference (config.py:44-68)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ps = [p for g in self.param_groups for p in g["params"]]
        if len(ps) != 1:
            raise ValueError("QedAdam steps one parameter group per instance "
                             "(Nerfstudio builds one optimiser per group name)")
        self._shared: Optional[_SharedFlatState] = None
        _ALL_QED_ADAMS.add(self)
        p0 = ps[0]
        if p0.is_cuda:
            from .rasterization import _workspace
            _workspace(p0.device).steppers.add(self)

    # -- layout -----------------------------------------------------------------------------------------
    def _param(self) -> Tensor:
        return self.param_groups[0]["params"][0]

    @staticmethod
    def _is_flat_view(p: Tensor) -> bool:
        """A view into a larger allocation (the flat buffer) rather than a tensor that owns its storage."""
        return p.storage_offset() != 0 or p.untyped_storage().nbytes() > 4 * p.numel()

    @staticmethod
    def _require_gpu(p: Tensor) -> None:
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise L.QedSplatError("QedAdam needs contiguous float32 GPU parameters (there is no CPU path)")

    def _attach(self) -> _SharedFlatState:
        p = self._param()
        base = p.untyped_storage().data_ptr()
        st = self._shared
        if st is not None and st.flat_ptr == base and self._views_intact(p):      # the steady state: nothing to do
            return st
        self._require_gpu(p)
        st = _FLAT_STATES.get(base)
        if st is None or self._shared is not st:
            if st is None:
                flat = torch.empty(0, dtype=torch.float32, device=p.device).set_(
                    p.untyped_storage(), 0, (p.untyped_storage().nbytes() // 4,))
                st = _SharedFlatState(flat)
                _FLAT_STATES[base] = st
                st._keepalive_flat = flat
            # every live instance whose Parameter is a view of this buffer is a member from the start: the fused
            # launch waits for all of them, whichever is stepped first
            for inst in list(_ALL_QED_ADAMS):
                q = inst._param()
                if q.is_cuda and q.untyped_storage().data_ptr() == base:
                    inst._shared = st
                    st.members[q.storage_offset()] = inst
                    st.t.setdefault(q.storage_offset(), 0)
        off = p.storage_offset()
        st.members[off] = self
        st.t.setdefault(off, 0)
        self._expose_views(st)
        return st

    def _views_intact(self, p: Tensor) -> bool:
        """``self.state[param]`` still holds the very view objects _expose_views put there (identity, no tensor calls)."""
        held = self.__dict__.get("_views")
        if held is None or held[0] is not p:
            return False
        cur = self.state.get(p)
        return cur is held[1] and cur.get("exp_avg") is held[2] and cur.get("exp_avg_sq") is held[3]

    def _expose_views(self, st: _SharedFlatState) -> None:
        """``self.state[param]`` in torch.optim.Adam's layout, as views of the shared moments."""
        p = self._param()
        if self._views_intact(p) and self._views[2].untyped_storage().data_ptr() == st.exp_avg.untyped_storage().data_ptr():
            return
        off, n = p.storage_offset(), p.numel()
        cur = self.state.get(p)
        m, v = st.exp_avg[off:off + n].view(p.shape), st.exp_avg_sq[off:off + n].view(p.shape)
        if cur is not None and len(cur) and "exp_avg" in cur:
            if cur["exp_avg"].data_ptr() == m.data_ptr() and cur["exp_avg_sq"].data_ptr() == v.data_ptr() \
                    and cur["exp_avg"].shape == p.shape:
                self._views = (p, cur, cur["exp_avg"], cur["exp_avg_sq"])
                return
            raise RuntimeError(
                "QedAdam: optimizer.state[param] of a flat-buffer group was replaced from outside (the parent class's "
                "dup / remove-from-optimiser routines do that).  With Parameters that are views of one flat buffer the "
                "number of Gaussians is changed by qed_splatter_amd.densify.Densifier(model, QedAdamSet(model, "
                "optimizers)), which rewrites parameters and both moments in one pass; or hold the six Parameters as "
                "separate tensors, in which case QedAdam keeps torch.optim.Adam's own per-parameter state.")
        entry = {"step": torch.tensor(float(st.t.get(off, 0))), "exp_avg": m, "exp_avg_sq": v}
        self.state[p] = entry
        self._views = (p, entry, m, v)

    # -- stepping ---------------------------------------------------------------------------------------
    # torch.optim.Optimizer wraps every subclass's step() in a profiler range + hook dispatch (~25 us of Python per call,
    # six calls per iteration on a route whose host cost is what bounds it).  `step.hooked = True` (below the class) tells
    # it not to; registered step hooks are honoured here, so the Optimizer contract stands.
    def step(self, closure=None):
        from torch.optim import optimizer as _O
        pre, post = self._optimizer_step_pre_hooks, self._optimizer_step_post_hooks
        if pre or post or _O._global_optimizer_pre_hooks or _O._global_optimizer_post_hooks:
            args, kwargs = (closure,), {}
            for hook in (*_O._global_optimizer_pre_hooks.values(), *pre.values()):
                result = hook(self, args, kwargs)
                if result is not None:
                    args, kwargs = result
            out = self._step(*args, **kwargs)
            for hook in (*post.values(), *_O._global_optimizer_post_hooks.values()):
                hook(self, args, kwargs)
            return out
        return self._step(closure)

    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        p = self.param_groups[0]["params"][0]
        if _raw_grad(p) is None:               # (the field itself: a compact SH gradient is not materialised by asking)
            return loss
        with torch.no_grad():
            if p.storage_offset() == 0 and not self._is_flat_view(p):
                self._step_own(p)
                return loss
            st = self._attach()
            off = p.storage_offset()
            pending = st.pending
            if off in pending:                     # stepped twice before the others: launch what is waiting first
                self._launch(st, sorted(pending))
            pending[off] = float(self.param_groups[0]["lr"])
            if len(pending) == len(st.members):
                self._launch(st, sorted(pending))
        return loss

    def _step_own(self, p: Tensor) -> None:
        """A Parameter that owns its storage: torch.optim.Adam's state layout, one fused launch."""
        import ctypes as C
        self._require_gpu(p)
        g = p.grad
        grp = self.param_groups[0]
        state = self.state[p]
        if len(state) == 0:
            state["step"] = torch.tensor(0.0)
            state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        m, v = state["exp_avg"], state["exp_avg_sq"]
        if m.shape != p.shape or v.shape != p.shape:
            raise RuntimeError(f"QedAdam: the moments in optimizer.state ({tuple(m.shape)}) do not match the parameter "
                               f"({tuple(p.shape)}): resize both when the number of Gaussians changes")
        if not (m.is_contiguous() and v.is_contiguous()):
            m = state["exp_avg"] = m.contiguous()
            v = state["exp_avg_sq"] = v.contiguous()
        state["step"] += 1
        _counted_step(self, p.device)
        t = int(state["step"])
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        beta1, beta2 = grp["betas"]
        if (p.data_ptr() | m.data_ptr() | v.data_ptr()) % 16 or g.data_ptr() % 4:
            raise L.QedSplatError("QedAdam: parameter / moment tensors must be 16-byte aligned")
        n = p.numel()
        h_begin = (C.c_int64 * 2)(0, n)
        h_lr = (C.c_float * 1)(float(grp["lr"]))
        L.check(L.load().qed_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 1,
                                       C.cast(h_begin, C.c_void_p), C.cast(h_lr, C.c_void_p), float(beta1), float(beta2),
                                       float(grp["eps"]), t, _skip_flag(p.device), _stream()), "qed_adam_step")

    def on_skipped_step(self) -> None:
        """See FlatAdam.on_skipped_step: the launch that this instance's last step() counted did nothing on the device."""
        p = self._param()
        if self._shared is not None and self._is_flat_view(p):
            st, off = self._shared, p.storage_offset()
            if st.t.get(off, 0) > 0:
                st.t[off] -= 1
                held = self.__dict__.get("_views")
                if held is not None:
                    held[1]["step"].fill_(float(st.t[off]))
            return
        state = self.state.get(p)
        if state and float(state["step"]) > 0:
            state["step"] -= 1

    def flush(self) -> None:
        """Launch the update of the groups that have called step() but are still waiting for the others."""
        st = self._shared
        if st is not None and st.pending:
            self._launch(st, sorted(st.pending))

    def zero_grad(self, set_to_none: bool = True) -> None:
        # a waiting update reads .grad at launch time: launch it before the gradients go away (a group that skipped
        # this iteration -- .grad None, or a GradScaler that found an inf for that optimiser only -- must not hold the
        # others' update back into the next iteration's gradients).  One parameter: done here, without the base class's
        # generic (and, per call, several times costlier) walk over groups.
        st = self._shared
        if st is not None and st.pending:
            self._launch(st, sorted(st.pending))
        p = self.param_groups[0]["params"][0]
        g = _raw_grad(p)
        if g is not None:
            if set_to_none:
                p.grad = None                          # (through the subclass's setter: drops a compact form nobody read)
            else:
                if g.grad_fn is not None:
                    g.detach_()
                else:
                    g.requires_grad_(False)
                g.zero_()

    @staticmethod
    def _launch(st: _SharedFlatState, offs) -> None:
        import ctypes as C
        lib = L.load()
        members = [st.members[o] for o in offs]
        ps = [m.param_groups[0]["params"][0] for m in members]
        for mem in members:
            _counted_step(mem, ps[0].device)
        # compact SH gradients (QEDSplatterModel, lazy_sh_grad): consumed as they are when this launch covers all six
        # groups in one run (below); any other launch first has the coefficient gradients written out
        owner = ps[-1].__dict__.get("_qed_owner")
        model = owner() if owner is not None else None
        lazy = model.__dict__.get("_lazy_sh") if model is not None else None
        if lazy is not None and len(offs) != len(st.members):
            model._materialise_sh_grads()
            lazy = None
        gs = [_raw_grad(p) for p in ps]
        for mem, p in zip(members, ps):
            if not mem._views_intact(p):
                mem._expose_views(st)          # (re-creates the views, or refuses moments that were replaced from outside)
        # maximal runs of groups that are adjacent in the flat buffer, share betas / eps / step count and whose
        # gradients are adjacent pieces of one allocation (what _ProjectSH.backward produces): one launch per run
        n_el = [p.numel() for p in ps]
        gptr = [g.data_ptr() for g in gs]
        keys = [m.defaults_key() for m in members]
        runs, cur = [], [0]
        for i in range(1, len(ps)):
            same = (offs[i] == offs[i - 1] + n_el[i - 1] and gptr[i] == gptr[i - 1] + 4 * n_el[i - 1]
                    and keys[i] == keys[i - 1] and st.t[offs[i]] == st.t[offs[i - 1]]
                    and gs[i].dtype == torch.float32 and gs[i].is_contiguous() and gs[i - 1].is_contiguous())
            if same:
                cur.append(i)
            else:
                runs.append(cur)
                cur = [i]
        runs.append(cur)
        stream = L.current_stream()
        skip = _skip_flag(ps[0].device)
        flat_ptr = st.flat_ptr
        m_ptr, v_ptr = st.exp_avg.data_ptr(), st.exp_avg_sq.data_ptr()
        if lazy is not None and not (len(runs) == 1 and offs[0] == 0 and gptr[0] % 16 == 0
                                     and gs[0].is_contiguous() and gs[0].dtype == torch.float32):
            model._materialise_sh_grads()              # (the six groups do not form one run: plain gradients, then)
            lazy = None
        for run in runs:
            first, last = run[0], run[-1]
            lo, hi = offs[first], offs[last] + n_el[last]
            g = gs[first]
            grp = members[first].param_groups[0]
            beta1, beta2 = grp["betas"]
            eps = grp["eps"]
            t = st.t[offs[first]] + 1
            if lo % 4 != 0 or gptr[first] % 4 != 0 or not g.is_contiguous() or g.dtype != torch.float32:
                # a lone group whose range is not 16-byte aligned (only when the groups are stepped out of step with
                # each other and N is not a multiple of 4): the same update with eager torch ops
                for i in run:
                    o, n = offs[i], n_el[i]
                    gi = gs[i].reshape(-1).to(torch.float32)
                    m, v = st.exp_avg[o:o + n], st.exp_avg_sq[o:o + n]
                    m.mul_(beta1).add_(gi, alpha=1 - beta1)
                    v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
                    denom = (v.sqrt() / math.sqrt(1 - beta2 ** t)).add_(eps)
                    ps[i].data.reshape(-1).addcdiv_(m, denom, value=-st.pending[offs[i]] / (1 - beta1 ** t))
                    st.t[offs[i]] = t
                    members[i].state[ps[i]]["step"].fill_(float(t))
                continue
            k = len(run)
            h_begin = (C.c_int64 * (k + 1))(*[offs[i] - lo for i in run], hi - lo)
            h_lr = (C.c_float * k)(*[st.pending[offs[i]] for i in run])
            if lazy is not None:
                # the fused step's optimiser launches: the SH groups from the colour gradient + the view (before the means
                # move), then the leading groups.  The two fields are emptied: the compact form is used up
                n = lazy["n"]
                L.check(lib.qed_adam_step_sh(
                    flat_ptr, gptr[first], m_ptr, v_ptr, k, C.cast(h_begin, C.c_void_p), C.cast(h_lr, C.c_void_p), None,
                    float(beta1), float(beta2), float(eps), t, None, -1, 0.0, 0.0, 0, n, lazy["deg"], flat_ptr + 4 * offs[0],
                    1, L.ptr(lazy["viewmat"]), 16, L.ptr(lazy["v_color"]), 0, 1.0, 3, skip, stream), "qed_adam_step_sh")
                model.__dict__["_lazy_sh"] = None
                _RAW_GRAD.__set__(ps[-1], None)
                _RAW_GRAD.__set__(ps[-2], None)
                lazy = None
            else:
                L.check(lib.qed_adam_step(flat_ptr + 4 * lo, gptr[first], m_ptr + 4 * lo, v_ptr + 4 * lo, k,
                                          C.cast(h_begin, C.c_void_p), C.cast(h_lr, C.c_void_p), float(beta1), float(beta2),
                                          float(eps), t, skip, stream), "qed_adam_step")
            ft = float(t)
            for i in run:
                st.t[offs[i]] = t
                members[i]._views[1]["step"].fill_(ft)
        st.pending.clear()

    def defaults_key(self):
        g = self.param_groups[0]
        return (tuple(g["betas"]), float(g["eps"]))

    step.hooked = True          # (see step(): torch.optim.Optimizer must not wrap it again)

    # -- checkpointing: torch.optim.Adam's layout in both cases ------------------------------------------
    def state_dict(self):
        self.flush()
        p = self._param()
        if not self._is_flat_view(p):
            return super().state_dict()
        sd = super().state_dict()
        st = self._shared
        if st is not None:
            # copies of this group's slice (a view would drag the whole flat buffer into the checkpoint)
            off, n = p.storage_offset(), p.numel()
            sd["state"] = {0: {"step": torch.tensor(float(st.t.get(off, 0))),
                               "exp_avg": st.exp_avg[off:off + n].view(p.shape).clone(),
                               "exp_avg_sq": st.exp_avg_sq[off:off + n].view(p.shape).clone()}}
        return sd

    def load_state_dict(self, state_dict):
        p = self._param()
        if not self._is_flat_view(p):
            super().load_state_dict(state_dict)
            stt = self.state.get(p)
            if stt is not None and "step" in stt and torch.is_tensor(stt["step"]):
                stt["step"] = stt["step"].detach().to("cpu", torch.float32).reshape(())    # host counter, as created
            return
        state = state_dict.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        st = self._attach()
        if state:
            off, n = p.storage_offset(), p.numel()
            s0 = state[0] if 0 in state else next(iter(state.values()))
            st.exp_avg[off:off + n] = s0["exp_avg"].reshape(-1).to(st.exp_avg)
            st.exp_avg_sq[off:off + n] = s0["exp_avg_sq"].reshape(-1).to(st.exp_avg_sq)
            st.t[off] = int(s0["step"])
            self.state[p]["step"].fill_(float(st.t[off]))
