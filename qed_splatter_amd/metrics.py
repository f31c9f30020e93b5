"""Per-step evaluation metrics: the GPU side of the reference's ``metrics.py`` and
``QEDSplatterModel.get_metrics_dict`` (model.py:120-197).

The reference runs torchmetrics PSNR / SSIM / LPIPS plus its own DepthMetrics every iteration and pulls
each scalar to the host with ``.item()`` / ``float()`` (model.py:160-182): about a dozen device
synchronisations per step.  Here one streaming kernel (``qed_image_metrics``) produces the MSE / PSNR and
the seven depth metrics, ``qed_ssim_fwd`` produces SSIM, and everything stays in device memory until the
caller decides to log it.

LPIPS needs the pretrained AlexNet/VGG weights torchmetrics downloads; it is not provided and the
``lpips`` slot is NaN.  The point-cloud metrics of metrics.py:10-63 (cKDTree, CPU, offline) are out
of scope.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L

METRIC_NAMES = ("rgb_mse", "rgb_psnr", "depth_abs_rel", "depth_sq_rel", "depth_rmse", "depth_rmse_log",
                "depth_a1", "depth_a2", "depth_a3", "depth_n_valid")


_stream = L.current_stream


def image_metrics(pred_rgb: Optional[Tensor], gt_rgb: Optional[Tensor], pred_depth: Optional[Tensor] = None,
                  gt_depth: Optional[Tensor] = None, tolerance: float = 0.1) -> Tensor:
    """One pass over [H,W,3] colours and/or [H,W(,1)] depths -> float32[10] on the device, in the
    order of ``METRIC_NAMES``.  No host synchronisation."""
    lib = L.load()
    ref = pred_rgb if pred_rgb is not None else pred_depth
    assert ref is not None, "nothing to measure"
    n_pix = ref.shape[0] * ref.shape[1]

    def prep(t, last):
        if t is None:
            return None
        t = t.to(torch.float32).contiguous()
        assert t.numel() == n_pix * last, "shape mismatch between the images"
        return t

    pr, gr, pd, gd = prep(pred_rgb, 3), prep(gt_rgb, 3), prep(pred_depth, 1), prep(gt_depth, 1)
    work = torch.empty(L.METRICS_WS_DOUBLES, dtype=torch.float64, device=ref.device)
    out = torch.empty(10, dtype=torch.float32, device=ref.device)
    L.check(lib.qed_image_metrics(n_pix, L.ptr(pr), L.ptr(gr), L.ptr(pd), L.ptr(gd), float(tolerance), L.ptr(work),
                                  L.ptr(out), _stream()), "qed_image_metrics")
    return out


_NANS: Dict[torch.device, Tensor] = {}


def _nan(device) -> Tensor:
    """LPIPS needs pretrained weights that are not here: one cached NaN scalar per device instead of a fill per step."""
    t = _NANS.get(device)
    if t is None:
        t = _NANS[device] = torch.full((), float("nan"), device=device)
    return t


def ssim_value(pred_rgb: Tensor, gt_rgb: Tensor, keep_maps: bool = False):
    """StructuralSimilarityIndexMeasure(data_range=1, kernel_size=11) of two [H,W,3] images: torchmetrics
    reflect-pads by 5 and crops the same 5 pixels again, i.e. the mean of the unpadded valid-window SSIM
    map, which is what ssim.hip computes.

    ``keep_maps``: also return what a loss on the same two images needs from this forward pass -- {"key": identity of
    the inputs, "maps_sum": (coefficient maps of the backward pass, per-workgroup map sums), "inputs": the two tensors,
    held so that their addresses stay theirs} -- as a second value."""
    lib = L.load()
    H, W, _ = pred_rgb.shape
    p, g = pred_rgb.to(torch.float32).contiguous(), gt_rgb.to(torch.float32).contiguous()
    n_maps = lib.qed_ssim_maps_floats(H, W)
    if n_maps < 0:
        raise L.QedSplatError("image smaller than the 11 x 11 SSIM window")
    ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), dtype=torch.float32, device=p.device)
    # maps = NULL: the value only (no 75 MB of backward coefficient maps at 1080p)
    maps = torch.empty(n_maps, dtype=torch.float32, device=p.device) if keep_maps else None
    L.check(lib.qed_ssim_fwd(H, W, 3, L.ptr(p), None, None, L.ptr(g), None, L.ptr(maps), L.ptr(ssum), _stream()),
            "qed_ssim_fwd")
    value = ssum.sum() / (3.0 * (H - 10) * (W - 10))
    if not keep_maps:
        return value
    key = (p.data_ptr(), p._version, tuple(p.shape), g.data_ptr(), g._version, tuple(g.shape))
    return value, {"key": key, "maps_sum": (maps, ssum), "inputs": (p, g)}


@torch.no_grad()
def step_metrics(pred_rgb: Tensor, gt_rgb: Tensor, pred_depth: Optional[Tensor], gt_depth: Optional[Tensor],
                 scales_last: Optional[Tensor], ssim_lambda: float, depth_lambda: float, tolerance: float = 0.1):
    """The whole of get_metrics_dict's image arithmetic for one TRAINING step in three launches -- qed_ssim_fwd (with the
    coefficient maps the loss's backward pass needs) + qed_step_metrics (one streaming pass, one fold): MSE / PSNR, the
    seven depth metrics, the SSIM value, avg_min_scale, and the loss sums get_loss_dict would compute from the same images
    a moment later.  Returns (metrics dict, shared) where ``shared`` is what QEDSplatterModel.get_loss_dict takes over:
    {"key", "maps_sum", "inputs", "loss": (sums, losses), "depth_key"}."""
    lib = L.load()
    H, W, _ = pred_rgb.shape
    n_pix = H * W
    dev = pred_rgb.device

    def prep(t, last):
        if t is None:
            return None
        t = t.to(torch.float32).contiguous()
        assert t.numel() == n_pix * last, "shape mismatch between the images"
        return t

    p, g = prep(pred_rgb, 3), prep(gt_rgb, 3)
    pd, gd = prep(pred_depth, 1), prep(gt_depth, 1)
    n_maps = lib.qed_ssim_maps_floats(H, W)
    if n_maps < 0:
        raise L.QedSplatError("image smaller than the 11 x 11 SSIM window")
    st = _stream()
    ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), dtype=torch.float32, device=dev)
    maps = torch.empty(n_maps, dtype=torch.float32, device=dev)
    L.check(lib.qed_ssim_fwd(H, W, 3, L.ptr(p), None, None, L.ptr(g), None, L.ptr(maps), L.ptr(ssum), st), "qed_ssim_fwd")
    work = torch.empty(L.STEP_METRICS_WS_DOUBLES, dtype=torch.float64, device=dev)
    out = torch.empty(12, dtype=torch.float32, device=dev)
    sums = torch.empty(L.LOSS_SUMS_FLOATS, dtype=torch.float32, device=dev)
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    n_sc, sc_stride = (0, 1)
    if scales_last is not None:
        assert scales_last.dim() == 1 and scales_last.dtype == torch.float32
        n_sc, sc_stride = scales_last.numel(), (scales_last.stride(0) if scales_last.numel() > 1 else 1)
    L.check(lib.qed_step_metrics(n_pix, L.ptr(p), L.ptr(g), L.ptr(pd), L.ptr(gd), float(tolerance), L.ptr(ssum), ssum.numel(),
                                 1.0 / (3.0 * (H - 10) * (W - 10)), L.ptr(scales_last) if n_sc else None, n_sc, sc_stride,
                                 None, 1.0 - ssim_lambda, float(depth_lambda), float(ssim_lambda), L.ptr(sums), L.ptr(losses),
                                 L.ptr(work), L.ptr(out), st), "qed_step_metrics")
    md = {"rgb_mse": out[0], "rgb_psnr": out[1], "rgb_ssim": out[10], "rgb_lpips": _nan(dev)}
    if pd is not None:
        md.update({n: out[i] for i, n in enumerate(METRIC_NAMES) if n.startswith("depth_") and n != "depth_n_valid"})
    if n_sc:
        md["avg_min_scale"] = out[11]
    key = (p.data_ptr(), p._version, tuple(p.shape), g.data_ptr(), g._version, tuple(g.shape))
    shared = {"key": key, "maps_sum": (maps, ssum), "inputs": (p, g, pd, gd), "loss": (sums, losses),
              "depth_key": None if pd is None else (pd.data_ptr(), pd._version, gd.data_ptr(), gd._version),
              "lambdas": (float(ssim_lambda), float(depth_lambda))}
    return md, shared


@torch.no_grad()
def nanmean_exp(x: Tensor) -> Tensor:
    """``torch.nanmean(torch.exp(x))`` of a (possibly strided) 1-D view as one pass + a fold (model.py:192-194 applies it
    to ``self.scales[..., -1]`` every step: ~8 eager launches in the reference)."""
    assert x.dim() == 1 and x.dtype == torch.float32
    work = torch.empty(L.METRICS_WS_DOUBLES, dtype=torch.float64, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    L.check(L.load().qed_nanmean_exp(x.numel(), L.ptr(x), x.stride(0) if x.numel() > 1 else 1, L.ptr(work), L.ptr(out),
                                     _stream()), "qed_nanmean_exp")
    return out.view(())


def _to_hwc(img: Tensor) -> Tensor:
    """[1,3,H,W] / [3,H,W] (the layout the reference hands to torchmetrics) or [H,W,3] -> [H,W,3]."""
    if img.dim() == 4:
        img = img[0]
    if img.shape[0] == 3 and img.shape[-1] != 3:
        img = img.permute(1, 2, 0)
    if img.dtype == torch.uint8:                                   # metrics.py:104-105
        img = img.float() / 255.0
    return img


class RGBMetrics(torch.nn.Module):
    """Mirror of metrics.py:84-112: ``forward(pred, gt) -> (psnr, ssim, lpips)`` as 0-dim device tensors."""

    @torch.no_grad()
    def forward(self, pred: Tensor, gt: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        p, g = _to_hwc(pred), _to_hwc(gt)
        m = image_metrics(p, g)
        return m[1], ssim_value(p, g), _nan(p.device)


class DepthMetrics(torch.nn.Module):
    """Mirror of metrics.py:115-156: ``forward(pred, gt) -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)``."""

    def __init__(self, tolerance: float = 0.1):
        super().__init__()
        self.tolerance = tolerance

    @torch.no_grad()
    def forward(self, pred: Tensor, gt: Tensor):
        p = pred.reshape(pred.shape[-2], pred.shape[-1], 1) if pred.dim() == 3 and pred.shape[0] == 1 else pred
        g = gt.reshape(p.shape).to(p.device)
        m = image_metrics(None, None, p, g, self.tolerance)
        return tuple(m[i] for i in range(2, 9))


@torch.no_grad()
def metrics_dict(pred_rgb: Tensor, gt_rgb: Tensor, pred_depth: Optional[Tensor], gt_depth: Optional[Tensor],
                 tolerance: float = 0.1, keep_ssim_maps: bool = False) -> Dict[str, Tensor]:
    """The image part of get_metrics_dict (model.py:152-182) in two launches, values left on the device.
    ``keep_ssim_maps``: the entry "_ssim_shared" carries the SSIM forward for a loss on the same images (ssim_value)."""
    m = image_metrics(pred_rgb, gt_rgb, pred_depth, gt_depth, tolerance)
    ssim = ssim_value(pred_rgb, gt_rgb, keep_maps=keep_ssim_maps)
    out = {"rgb_mse": m[0], "rgb_psnr": m[1], "rgb_ssim": ssim[0] if keep_ssim_maps else ssim,
           "rgb_lpips": _nan(m.device)}
    if keep_ssim_maps:
        out["_ssim_shared"] = ssim[1]
    if pred_depth is not None:
        out.update({n: m[i] for i, n in enumerate(METRIC_NAMES) if n.startswith("depth_") and n != "depth_n_valid"})
    return out
