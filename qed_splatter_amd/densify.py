"""Densification / culling: the host side of the parent class's callbacks that
consume the side effects of ``get_outputs`` -- ``self.xys`` (with ``.absgrad``), ``self.radii``,
``self.last_size`` (reference model.py:249,289-292).

``Densifier.after_train`` / ``Densifier.refinement_after`` carry the names and the step logic of
Nerfstudio's ``SplatfactoModel.after_train`` / ``refinement_after`` (un-vendored parent; thresholds:
parent defaults + the reference's overrides at config.py:40-41).  The per-Gaussian work runs in
csrc/densify.hip on the flat parameter / Adam-moment buffers; this file only sequences the launches,
reads the four totals the allocation needs (one host synchronisation per refinement -- the reference
does several ``.item()`` calls there) and swaps the buffers in.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib as L


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


_stream = L.current_stream


class Densifier:
    """Owns xys_grad_norm / vis_counts / max_2Dsize and performs the refinement on model + optimiser.

    ``num_train_data`` is the number of training images (the parent only densifies once every image
    has been seen since the last opacity reset).  ``seed`` makes the split samples reproducible and
    identical on every data-parallel rank (all ranks must take the same decisions)."""

    def __init__(self, model, optimizer, config: Optional[DensifyConfig] = None, num_train_data: int = 1, seed: int = 0):
        self.model, self.optimizer = model, optimizer
        self.config = config or DensifyConfig()
        self.num_train_data = num_train_data
        self.xys_grad_norm: Optional[Tensor] = None
        self.vis_counts: Optional[Tensor] = None
        self.max_2Dsize: Optional[Tensor] = None
        self._gen = torch.Generator(device=model.device)
        self._gen.manual_seed(seed)
        self.last_info: Dict[str, int] = {}

    # ---- SplatfactoModel.after_train ----
    @torch.no_grad()
    def after_train(self, step: int) -> None:
        m = self.model
        if step >= self.config.stop_split_at:
            return
        n = m.num_points
        if self.xys_grad_norm is None:
            self.xys_grad_norm = torch.zeros(n, dtype=torch.float32, device=m.device)
            self.vis_counts = torch.ones(n, dtype=torch.float32, device=m.device)
        if self.max_2Dsize is None:
            self.max_2Dsize = torch.zeros(n, dtype=torch.float32, device=m.device)
        absgrad = m.xys.absgrad                                   # [1,N,2], a strided view of the vsplat rows
        assert absgrad.shape == (1, n, 2) and absgrad.stride(2) == 1, "absgrad must be [1,N,2] with unit inner stride"
        radii = m.radii.to(torch.int32).contiguous()
        L.check(L.load().qed_densify_accumulate(n, L.ptr(absgrad), absgrad.stride(1), L.ptr(radii),
                                                1.0 / float(max(m.last_size)), L.ptr(self.xys_grad_norm),
                                                L.ptr(self.vis_counts), L.ptr(self.max_2Dsize), _stream()),
                "qed_densify_accumulate")

    def all_reduce_stats(self, group=None) -> None:
        """Data-parallel ranks see different cameras: sum / max the statistics before refining."""
        from .parallel import allreduce_densification_stats
        if self.xys_grad_norm is not None:
            # vis_counts starts at one on every rank: keep a single "one" after the sum
            import torch.distributed as dist
            ws = dist.get_world_size(group)
            allreduce_densification_stats(self.xys_grad_norm, self.vis_counts, self.max_2Dsize, group)
            self.vis_counts -= float(ws - 1)

    # ---- SplatfactoModel.refinement_after ----
    @torch.no_grad()
    def refinement_after(self, step: int, samples: Optional[Tensor] = None) -> Dict[str, int]:
        cfg, m, opt = self.config, self.model, self.optimizer
        info = {"n_split": 0, "n_dup": 0, "n_culled": 0, "opacity_reset": False, "did_densify": False,
                "n_before": m.num_points, "n_after": m.num_points}
        self.last_info = info
        if step <= cfg.warmup_length:
            return info
        lib = L.load()
        reset_interval = cfg.reset_alpha_every * cfg.refine_every
        do_densify = step < cfg.stop_split_at and step % reset_interval > self.num_train_data + cfg.refine_every
        do_cull_only = (not do_densify) and step >= cfg.stop_split_at and cfg.continue_cull_post_densification
        if do_densify or do_cull_only:
            if do_densify:
                assert self.xys_grad_norm is not None and self.vis_counts is not None and self.max_2Dsize is not None, \
                    "after_train() must have run since the last refinement"
            self._refine(step, do_densify, samples, info)
        if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:
            n = m.num_points
            reset_value = cfg.cull_alpha_thresh * 2.0
            b = m.group_begin[m.group_names.index("opacities")]
            L.check(lib.qed_densify_reset_opacity(n, L.ptr(m.flat_params[b:b + n]), L.ptr(opt.exp_avg[b:b + n]),
                                                  L.ptr(opt.exp_avg_sq[b:b + n]),
                                                  math.log(reset_value / (1.0 - reset_value)), _stream()),
                    "qed_densify_reset_opacity")
            info["opacity_reset"] = True
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None
        info["n_after"] = m.num_points
        return info

    def _refine(self, step: int, densify: bool, samples: Optional[Tensor], info: Dict[str, int]) -> None:
        cfg, m, opt = self.config, self.model, self.optimizer
        lib = L.load()
        n, dev = m.num_points, m.device
        if n == 0:
            return
        H, W = m.last_size
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        pos = torch.empty(lib.qed_densify_pos_ints(n), dtype=torch.int32, device=dev)
        totals = torch.empty(4, dtype=torch.int32, device=dev)
        cull_big = step > cfg.refine_every * cfg.reset_alpha_every
        screen = step < cfg.stop_screen_size_at
        L.check(lib.qed_densify_classify(
            n, L.ptr(m.scales), L.ptr(m.opacities), L.ptr(self.xys_grad_norm), L.ptr(self.vis_counts),
            L.ptr(self.max_2Dsize), int(densify), 0.5 * float(max(H, W)), cfg.densify_grad_thresh,
            cfg.densify_size_thresh, cfg.split_screen_size if screen else -1.0, cfg.cull_alpha_thresh,
            cfg.cull_scale_thresh if cull_big else -1.0, cfg.cull_screen_size if screen else -1.0,
            L.ptr(flags), L.ptr(pos), L.ptr(totals), _stream()), "qed_densify_classify")
        n_split, k_old, k_child, k_dup = (int(v) for v in totals.tolist())      # the one host sync of a refinement
        ns = cfg.n_split_samples
        n_new = k_old + ns * k_child + k_dup
        if densify and n_split > 0:
            if samples is None:
                samples = torch.randn(ns * n_split, 3, device=dev, generator=self._gen)   # split_gaussians' randn
            assert samples.shape == (ns * n_split, 3)
            samples = samples.to(device=dev, dtype=torch.float32).contiguous()
        widths = [(m.group_begin[g + 1] - m.group_begin[g]) // n for g in range(6)]
        new_begin = [0]
        for w in widths:
            new_begin.append(new_begin[-1] + w * n_new)
        new_p = torch.empty(new_begin[-1], dtype=torch.float32, device=dev)
        new_m = torch.empty_like(new_p)
        new_v = torch.empty_like(new_p)
        h_tot = (C.c_int32 * 4)(n_split, k_old, k_child, k_dup)
        h_old = (C.c_int64 * 7)(*m.group_begin)
        h_new = (C.c_int64 * 7)(*new_begin)
        L.check(lib.qed_densify_emit(n, ns, L.ptr(flags), L.ptr(pos), C.cast(h_tot, C.c_void_p), L.ptr(samples),
                                     L.ptr(m.flat_params), L.ptr(opt.exp_avg), L.ptr(opt.exp_avg_sq),
                                     C.cast(h_old, C.c_void_p), L.ptr(new_p), L.ptr(new_m), L.ptr(new_v),
                                     C.cast(h_new, C.c_void_p), _stream()), "qed_densify_emit")
        m.rebind_flat(new_p, n_new)
        opt.rebind(new_m, new_v)
        n_dup = int((flags & 2).ne(0).sum()) if densify else 0
        info.update(n_split=n_split, n_dup=n_dup, did_densify=bool(densify),
                    n_culled=n + ns * n_split + n_dup - n_new)
