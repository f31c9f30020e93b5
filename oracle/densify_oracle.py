"""TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.

CPU (PyTorch-eager) restatement of the densification / culling callbacks that consume the side
effects of the reference's ``get_outputs`` (``self.xys`` with ``.absgrad``, ``self.radii``,
``self.last_size``; /root/reference/qed_splatter/model.py:249,289-292) -- SURVEY section 8(f) rank 3.

The callbacks themselves live in the parent class, Nerfstudio's ``SplatfactoModel`` (nerfstudio >= 1.1.0,
pyproject.toml:6; un-vendored), methods ``after_train``, ``refinement_after``, ``cull_gaussians``,
``split_gaussians``, ``dup_gaussians``, ``dup_in_all_optim``, ``remove_from_all_optim``.  Their published
algorithm is restated here with the parent's default thresholds and the two the reference overrides
(``cull_alpha_thresh=0.005``, ``densify_grad_thresh=0.0005``, config.py:40-41).

PARITY UNPINNED: nerfstudio cannot be imported here and the reference holds no fixtures for these
callbacks, so parity for this row is "vs. this restatement".

Restated behaviour worth knowing (all kept, including the quirks):
  * ``vis_counts`` starts at ONE, ``xys_grad_norm`` and ``max_2Dsize`` at zero; they are dropped
    (set to None) after every refinement.
  * ``split_gaussians`` shrinks the scales of the split Gaussians IN PLACE (log(exp(s)/1.6)) before the
    duplicate mask is evaluated, so a split Gaussian whose shrunk scale falls under
    ``densify_size_thresh`` is also duplicated (the copy carries the shrunk scale); the split original
    itself is always culled.
  * the children of one split draw ``randn`` samples scaled by the ORIGINAL scale and rotated by the
    normalised quaternion; ``.repeat(samps, 1)`` orders them sample-major (all splits for sample 0, then
    all splits for sample 1).
  * the culling mask (low opacity, too big in world / screen space) is evaluated on the concatenation
    [old | split children | duplicates]; new Gaussians have ``max_2Dsize`` 0.
  * Adam moments of new Gaussians are zero; culled rows are dropped from both moments.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

NAMES = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")


@dataclass
class DensifyConfig:
    """SplatfactoModelConfig defaults (nerfstudio 1.1.x) with the reference's two overrides."""
    warmup_length: int = 500
    refine_every: int = 100
    cull_alpha_thresh: float = 0.005          # config.py:40 (parent default 0.1)
    cull_scale_thresh: float = 0.5
    continue_cull_post_densification: bool = True
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0005       # config.py:41 (parent default 0.0008)
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    stop_split_at: int = 15000


def quat_to_rotmat(quats: Tensor) -> Tensor:
    w, x, y, z = (quats / quats.norm(dim=-1, keepdim=True)).unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


class DensifyState:
    """xys_grad_norm / vis_counts / max_2Dsize of SplatfactoModel."""

    def __init__(self):
        self.xys_grad_norm: Optional[Tensor] = None
        self.vis_counts: Optional[Tensor] = None
        self.max_2Dsize: Optional[Tensor] = None


def after_train(state: DensifyState, absgrad: Tensor, radii: Tensor, last_size: Tuple[int, int], step: int,
                cfg: DensifyConfig) -> None:
    """SplatfactoModel.after_train with use_absgrad: absgrad = xys.absgrad[0] [N,2], radii [N] int."""
    if step >= cfg.stop_split_at:
        return
    n = radii.shape[0]
    visible = radii > 0
    grads = absgrad[visible].norm(dim=-1)
    if state.xys_grad_norm is None:
        state.xys_grad_norm = torch.zeros(n, dtype=torch.float32)
        state.vis_counts = torch.ones(n, dtype=torch.float32)
    state.vis_counts[visible] += 1
    state.xys_grad_norm[visible] += grads.to(torch.float32)
    if state.max_2Dsize is None:
        state.max_2Dsize = torch.zeros(n, dtype=torch.float32)
    newradii = radii[visible].to(torch.float32)
    state.max_2Dsize[visible] = torch.maximum(state.max_2Dsize[visible], newradii / float(max(last_size[0], last_size[1])))


def refinement_after(params: Dict[str, Tensor], exp_avg: Dict[str, Tensor], exp_avg_sq: Dict[str, Tensor],
                     state: DensifyState, step: int, cfg: DensifyConfig, last_size: Tuple[int, int],
                     num_train_data: int, samples: Optional[Tensor] = None):
    """SplatfactoModel.refinement_after.  Returns (params, exp_avg, exp_avg_sq, info) with new tensors;
    ``samples`` [n_split_samples * n_splits, 3] replaces torch.randn so that runs can be compared."""
    info = {"n_split": 0, "n_dup": 0, "n_culled": 0, "opacity_reset": False, "did_densify": False}
    if step <= cfg.warmup_length:
        return params, exp_avg, exp_avg_sq, info
    params = {k: v.clone() for k, v in params.items()}
    exp_avg = {k: v.clone() for k, v in exp_avg.items()}
    exp_avg_sq = {k: v.clone() for k, v in exp_avg_sq.items()}
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    do_densification = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    max_2Dsize = state.max_2Dsize
    deleted = None
    if do_densification:
        assert state.xys_grad_norm is not None and state.vis_counts is not None and max_2Dsize is not None
        avg_grad_norm = (state.xys_grad_norm / state.vis_counts) * 0.5 * max(last_size[0], last_size[1])
        high_grads = avg_grad_norm > cfg.densify_grad_thresh
        splits = params["scales"].exp().max(dim=-1).values > cfg.densify_size_thresh
        if step < cfg.stop_screen_size_at:
            splits = splits | (max_2Dsize > cfg.split_screen_size)
        splits = splits & high_grads
        nsamps = cfg.n_split_samples
        n_splits = int(splits.sum())
        # ---- split_gaussians ----
        if samples is None:
            samples = torch.randn(nsamps * n_splits, 3)
        assert samples.shape == (nsamps * n_splits, 3)
        sc = params["scales"][splits]
        scaled = torch.exp(sc.repeat(nsamps, 1)) * samples.to(sc.dtype)
        rots = quat_to_rotmat(params["quats"][splits].repeat(nsamps, 1))
        rotated = torch.bmm(rots, scaled[..., None]).squeeze(-1)
        split_params = {
            "means": rotated + params["means"][splits].repeat(nsamps, 1),
            "features_dc": params["features_dc"][splits].repeat(nsamps, 1),
            "features_rest": params["features_rest"][splits].repeat(nsamps, 1, 1),
            "opacities": params["opacities"][splits].repeat(nsamps, 1),
            "scales": torch.log(torch.exp(sc) / 1.6).repeat(nsamps, 1),
            "quats": params["quats"][splits].repeat(nsamps, 1),
        }
        params["scales"][splits] = torch.log(torch.exp(sc) / 1.6)          # in place, before the dup mask
        # ---- dup_gaussians ----
        dups = (params["scales"].exp().max(dim=-1).values <= cfg.densify_size_thresh) & high_grads
        n_dups = int(dups.sum())
        dup_params = {k: params[k][dups] for k in NAMES}
        for k in NAMES:
            params[k] = torch.cat([params[k], split_params[k], dup_params[k]], dim=0)
            z_split = torch.zeros_like(split_params[k])
            z_dup = torch.zeros_like(dup_params[k])
            exp_avg[k] = torch.cat([exp_avg[k], z_split, z_dup], dim=0)             # dup_in_all_optim
            exp_avg_sq[k] = torch.cat([exp_avg_sq[k], z_split, z_dup], dim=0)
        max_2Dsize = torch.cat([max_2Dsize, torch.zeros(nsamps * n_splits + n_dups)], dim=0)
        splits_mask = torch.cat([splits, torch.zeros(nsamps * n_splits + n_dups, dtype=torch.bool)])
        deleted = _cull_mask(params, max_2Dsize, step, cfg, splits_mask)
        info.update(n_split=n_splits, n_dup=n_dups, did_densify=True)
    elif step >= cfg.stop_split_at and cfg.continue_cull_post_densification:
        deleted = _cull_mask(params, max_2Dsize, step, cfg, None)
    if deleted is not None:
        for k in NAMES:
            params[k] = params[k][~deleted]
            exp_avg[k] = exp_avg[k][~deleted]                                       # remove_from_all_optim
            exp_avg_sq[k] = exp_avg_sq[k][~deleted]
        info["n_culled"] = int(deleted.sum())
    if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:
        reset_value = cfg.cull_alpha_thresh * 2.0
        params["opacities"] = torch.clamp(params["opacities"], max=math.log(reset_value / (1 - reset_value)))
        exp_avg["opacities"] = torch.zeros_like(exp_avg["opacities"])
        exp_avg_sq["opacities"] = torch.zeros_like(exp_avg_sq["opacities"])
        info["opacity_reset"] = True
    state.xys_grad_norm = None
    state.vis_counts = None
    state.max_2Dsize = None
    return params, exp_avg, exp_avg_sq, info


def _cull_mask(params, max_2Dsize, step: int, cfg: DensifyConfig, extra: Optional[Tensor]) -> Tensor:
    """SplatfactoModel.cull_gaussians' mask."""
    culls = torch.sigmoid(params["opacities"]).squeeze(-1) < cfg.cull_alpha_thresh
    if extra is not None:
        culls = culls | extra
    if step > cfg.refine_every * cfg.reset_alpha_every:
        toobigs = torch.exp(params["scales"]).max(dim=-1).values > cfg.cull_scale_thresh
        if step < cfg.stop_screen_size_at and max_2Dsize is not None:
            toobigs = toobigs | (max_2Dsize > cfg.cull_screen_size)
        culls = culls | toobigs
    return culls
