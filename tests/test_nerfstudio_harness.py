"""The drop-in route behind a Nerfstudio-SHAPED parent (VERDICT round 2, items 1-2 of "What's missing").

Nerfstudio's ``SplatfactoModel`` -- the reference's parent class, /root/reference/qed_splatter/model.py:12,50-58 -- keeps
six SEPARATE ``nn.Parameter``s in a ``ParameterDict``, the trainer builds one optimiser per group from a config object
with a ``_target`` (config.py:44-68) plus one for the camera optimiser (config.py:69-74), and ``get_outputs`` renders
through ``camera_optimizer.apply_to_camera`` (model.py:212).  Neither Nerfstudio nor gsplat is installed here, so this
file stands a minimal parent in: separate tensors, a ``_target``-style optimiser factory, a loss scale, a pose-offset
camera optimiser whose camera-to-world matrices require grad.  Two integration depths are driven end to end against the
fp64 oracle:

  (i)  the import swap alone: the reference's own eager statements around ``rasterization`` + ``torch.optim.Adam``;
  (ii) the fused nodes (``_PostProcess`` / ``_ImageLosses`` inside this package's model mirror, held with
       ``separate_params=True``) + ``QedAdam`` built by the same factory.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Tuple

import pytest
import torch
from torch import nn

from oracle import splat_oracle as O
from tests.util import PARAM_NAMES, REL_TOL, assert_close, assert_close_elem, scene, threshold_pixel_mask

pytestmark = pytest.mark.gpu
LRS = {"means": 1.6e-4, "scales": 0.005, "quats": 0.001, "opacities": 0.05, "features_dc": 0.0025,
       "features_rest": 0.0025 / 20, "camera_opt": 1e-4}                                       # config.py:44-74
LOSS_SCALE = 4.0


# ---- the stand-in parent's pieces ------------------------------------------------------------------------------------
class PoseOffsets(nn.Module):
    """Stand-in for Nerfstudio's CameraOptimizer (mode "SO3xR3" upstream): ``apply_to_camera`` returns camera-to-world
    matrices that depend on a learnable adjustment, i.e. REQUIRE GRAD (model.py:212)."""

    def __init__(self, n_cameras: int, device, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.pose_adjustment = nn.Parameter((1e-3 * torch.randn(n_cameras, 3, 4, generator=g)).to(device))

    def apply_to_camera(self, camera) -> torch.Tensor:
        return camera.camera_to_worlds + self.pose_adjustment


@dataclass
class OptimizerSpec:
    """The shape of Nerfstudio's AdamOptimizerConfig: a ``_target`` class and its keyword arguments."""
    _target: Any
    lr: float
    eps: float = 1e-15
    extra: Dict[str, Any] = field(default_factory=dict)

    def setup(self, params):
        return self._target(params, lr=self.lr, eps=self.eps, **self.extra)


def build_optimizers(param_groups: Dict[str, list], target) -> Dict[str, torch.optim.Optimizer]:
    """What Nerfstudio's ``Optimizers`` does: one optimiser instance per group name."""
    return {name: OptimizerSpec(target if name != "camera_opt" else torch.optim.Adam, LRS[name]).setup(params)
            for name, params in param_groups.items()}


def _separate_model(sc, dev, **cfg_kw):
    from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
    cfg_kw.setdefault("sh_degree_interval", 1)
    cfg = QEDSplatterModelConfig.synthetic(**cfg_kw)
    m = QEDSplatterModel(cfg, separate_params=True, **{k: sc[k].to(dev) for k in PARAM_NAMES})
    m.step = 100
    m.camera_optimizer = PoseOffsets(1, dev)
    K = sc["Ks"][0]
    h, w = sc["gt_rgb"].shape[:2]
    cam = PinholeCameras(sc["camera_to_worlds"][:1].to(dev), K[0, 0], K[1, 1], K[0, 2], K[1, 2], w, h)
    batch = {"image": sc["gt_rgb"].to(dev), "depth_image": sc["gt_depth"].to(dev)}
    # six tensors that own their storage, as the parent holds them
    ptrs = {p.untyped_storage().data_ptr() for p in m.gauss_params.values()}
    assert len(ptrs) == 6 and all(p.storage_offset() == 0 for p in m.gauss_params.values())
    groups = {n: [m.gauss_params[n]] for n in m.group_names}
    groups["camera_opt"] = list(m.camera_optimizer.parameters())
    return m, cam, batch, groups


def import_swap_outputs(m, cam, rasterization):
    """Route (i): what model.py:199-321 does in training with ONLY its import swapped -- the parent's eager torch
    statements (concatenated SH coefficients, exp / sigmoid / normalise, composite, clamp, depth fix-up) around the
    operator, which is handed activated tensors exactly as gsplat is."""
    from qed_splatter_amd.model import get_viewmat
    assert cam.shape[0] == 1
    c2w = m.camera_optimizer.apply_to_camera(cam)
    sh = torch.cat((m.features_dc[:, None, :], m.features_rest), dim=1)
    W, H = int(cam.width.item()), int(cam.height.item())
    bg = m._get_background_color()
    render, alpha, info = rasterization(
        means=m.means, quats=m.quats / m.quats.norm(dim=-1, keepdim=True), scales=torch.exp(m.scales),
        opacities=torch.sigmoid(m.opacities).squeeze(-1), colors=sh, viewmats=get_viewmat(c2w),
        Ks=cam.get_intrinsics_matrices().to(m.device), width=W, height=H, tile_size=16, packed=False, near_plane=0.01,
        far_plane=1e10, render_mode="RGB+D", sh_degree=min(m.step // m.config.sh_degree_interval, m.config.sh_degree),
        sparse_grad=False, absgrad=True, rasterize_mode=m.config.rasterize_mode)
    if info["means2d"].requires_grad:
        info["means2d"].retain_grad()
    m.info, m.xys, m.radii = info, info["means2d"], info["radii"][0]
    rgb = torch.clamp(render[..., :3] + (1 - alpha) * bg, 0.0, 1.0)
    d = render[..., 3:4]
    depth = torch.where(alpha > 0, d, d.detach().max()).squeeze(0)
    return {"rgb": rgb.squeeze(0), "depth": depth, "accumulation": alpha.squeeze(0), "background": bg}


def eager_loss_dict(m, outputs, batch):
    """Route (i): the parent's main loss and the reference's depth term (model.py:87-116) as plain torch ops (boolean
    gathers and all); the SSIM module of the parent is stood in for by the conv2d restatement."""
    lam = m.config.ssim_lambda
    pred, gt = outputs["rgb"], batch["image"]
    d_out, d_gt = outputs["depth"], batch["depth_image"]
    if "mask" in batch:
        mask = batch["mask"].to(pred.dtype)
        pred, gt, d_out, d_gt = pred * mask, gt * mask, d_out * mask, d_gt * mask
    main = (1 - lam) * (gt - pred).abs().mean() + lam * (1 - O.ssim(pred, gt))
    valid = torch.isfinite(d_out) & torch.isfinite(d_gt) & (d_gt > 0.0)
    vo, vg = d_out[valid], d_gt[valid]
    dl = (vo - vg).abs().mean() if vo.numel() > 0 else torch.tensor(0.0, device=pred.device)
    return {"main_loss": main, "scale_reg": torch.tensor(0.0, device=pred.device), "depth_loss": m.config.depth_lambda * dl}


def _oracle_grads(sc, pose0, w, h, cfg, mask64, radii):
    ps = {k: sc[k].double().requires_grad_(True) for k in PARAM_NAMES}
    pose = pose0.double().cpu().requires_grad_(True)
    out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                               ps["features_rest"], sc["camera_to_worlds"][:1].double() + pose, sc["Ks"][:1].double(), w, h,
                               sc["background"].double(), rasterize_mode=cfg.rasterize_mode, radii_override=radii,
                               return_margin=True)
    l_rgb = O.main_loss(out["rgb"], sc["gt_rgb"].double(), cfg.ssim_lambda, mask64)
    l_d = O.depth_l1_loss(out["depth"], sc["gt_depth"].double(), mask64, cfg.depth_lambda)
    (LOSS_SCALE * (l_rgb + l_d)).backward()
    return out, l_rgb, l_d, ps, pose


@pytest.mark.parametrize("route", ["import_swap", "fused_nodes"])
def test_separate_parameters_and_camera_optimizer_against_the_oracle(cuda, route):
    """Six separate Parameters + a requires-grad c2w through the whole training-mode call sequence; EVERY Gaussian's
    gradient (and the pose gradient) against the fp64 oracle -- threshold pixels are taken out through batch["mask"]
    on both sides, not by leaving Gaussians out."""
    from qed_splatter_amd.rasterization import rasterization
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=77)
    m, cam, batch, groups = _separate_model(sc, cuda)
    m.train()
    # pass 1 (no gradient): radii + the oracle's per-pixel margins -> the pixel mask both sides will use
    with torch.no_grad():
        m.get_outputs(cam)
    radii = m.info["radii"].cpu()
    pose0 = m.camera_optimizer.pose_adjustment.detach().clone()
    with torch.no_grad():
        ref0 = O.splatfacto_outputs(*(sc[k].double() for k in PARAM_NAMES),
                                    sc["camera_to_worlds"][:1].double() + pose0.double().cpu(), sc["Ks"][:1].double(), w, h,
                                    sc["background"].double(), radii_override=radii, return_margin=True)
    mask64 = threshold_pixel_mask(ref0, sc["gt_rgb"], sc["gt_depth"], margin_tol=1e-4)
    n_out = int((mask64 == 0).sum())
    print(f"[parity] {route}: {n_out} of {w * h} pixels masked out (threshold / clamp-edge / kink pixels)")
    assert n_out < 0.002 * w * h
    batch["mask"] = mask64.to(cuda, torch.float32)
    # pass 2: the trainer's sequence with a loss scale
    if route == "import_swap":
        out = import_swap_outputs(m, cam, rasterization)
        ld = eager_loss_dict(m, out, batch)
    else:
        out = m.get_outputs(cam)
        m.get_metrics_dict(out, batch)
        ld = m.get_loss_dict(out, batch)
    (LOSS_SCALE * sum(ld.values())).backward()
    ref, l_rgb, l_d, ps, pose = _oracle_grads(sc, pose0, w, h, m.config, mask64, radii)
    assert float(ld["main_loss"]) == pytest.approx(float(l_rgb), rel=1e-4)
    assert float(ld["depth_loss"]) == pytest.approx(float(l_d), rel=1e-4)
    for name in PARAM_NAMES:                                     # kept = 1.0: all n Gaussians, element by element
        g = m.gauss_params[name].grad
        assert g is not None and g.shape[0] == n
        assert_close(g.cpu(), ps[name].grad, REL_TOL, f"{route} grad {name}")
        assert_close_elem(g.cpu(), ps[name].grad, f"{route} grad {name}", atol_frac=1e-5)
    g_pose = m.camera_optimizer.pose_adjustment.grad
    assert g_pose is not None and float(g_pose.abs().max()) > 0.0
    assert_close(g_pose.cpu(), pose.grad, REL_TOL, f"{route} grad of the camera optimiser's pose adjustment")
    assert m.xys.grad is not None and m.xys.absgrad.shape == (1, n, 2)          # what the densifier reads


def _train(m, cam, batch, opts, steps, route_fused=True, shadow=None):
    """``steps`` iterations of the trainer's sequence.  ``shadow`` = (tensors, optimisers): a second set of separately
    held tensors that receives a COPY of every unscaled gradient and is stepped by its own optimisers -- two optimisers
    compared on identical gradients (two separate training runs differ by the summation order of the atomics)."""
    from qed_splatter_amd.rasterization import rasterization
    for _ in range(steps):
        for o in opts.values():
            o.zero_grad()
        if route_fused:
            out = m.get_outputs(cam)
            m.get_metrics_dict(out, batch)
            ld = m.get_loss_dict(out, batch)
        else:
            out = import_swap_outputs(m, cam, rasterization)
            ld = eager_loss_dict(m, out, batch)
        (LOSS_SCALE * sum(ld.values())).backward()
        for name, o in opts.items():                              # GradScaler.step: unscale in place, then step
            for p in o.param_groups[0]["params"]:
                p.grad.mul_(1.0 / LOSS_SCALE)
            if shadow is not None and name in shadow[0]:
                shadow[0][name].grad = o.param_groups[0]["params"][0].grad.clone()
                shadow[1][name].step()
            o.step()
        m.step += 1


@pytest.mark.parametrize("n", [3000, 2999])                       # 2999: gradient views at odd offsets of the flat buffer
def test_qedadam_on_separate_tensors_equals_torch_adam(cuda, n):
    """Five iterations of route (ii) with one QedAdam per separately held group, against torch.optim.Adam stepping a
    copy of the tensors with the same gradients: parameters, moments and step counts; ``optimizer.state[param]`` is
    torch.optim.Adam's own layout; checkpoints round-trip."""
    from qed_splatter_amd.model import QedAdam
    w, h = 160, 112
    sc = scene(n, w, h, seed=78)
    m, cam, batch, groups = _separate_model(sc, cuda)
    m.train()
    opts = build_optimizers(groups, QedAdam)
    twin = {k: nn.Parameter(p.detach().clone()) for k, p in m.gauss_params.items()}
    twin_opts = {k: OptimizerSpec(torch.optim.Adam, LRS[k]).setup([p]) for k, p in twin.items()}
    _train(m, cam, batch, opts, 5, shadow=(twin, twin_opts))
    for k in PARAM_NAMES:
        p = m.gauss_params[k]
        assert_close_elem(p, twin[k], f"params {k} after 5 steps", atol_frac=1e-6)
        sq, st = opts[k].state[p], twin_opts[k].state[twin[k]]
        assert set(sq) >= {"step", "exp_avg", "exp_avg_sq"} and float(sq["step"]) == float(st["step"]) == 5.0
        assert sq["exp_avg"].shape == p.shape and sq["exp_avg_sq"].shape == p.shape
        assert_close(sq["exp_avg"], st["exp_avg"], 1e-6, f"exp_avg {k}")
        assert_close(sq["exp_avg_sq"], st["exp_avg_sq"], 1e-6, f"exp_avg_sq {k}")
    # checkpoint: state_dict -> fresh optimisers -> load_state_dict: the same state, and the same next update from the
    # same gradients
    import copy
    saved = {k: copy.deepcopy(o.state_dict()) for k, o in opts.items()}     # (what torch.save / torch.load amount to)
    m2, cam2, batch2, groups2 = _separate_model(sc, cuda)
    m2.train()
    with torch.no_grad():
        for k, p in m2.gauss_params.items():
            p.copy_(m.gauss_params[k])
        m2.camera_optimizer.pose_adjustment.copy_(m.camera_optimizer.pose_adjustment)
    o2 = build_optimizers(groups2, QedAdam)
    for k, o in o2.items():
        o.load_state_dict(saved[k])
    for k in PARAM_NAMES:
        s1, s2 = opts[k].state[m.gauss_params[k]], o2[k].state[m2.gauss_params[k]]
        assert float(s2["step"]) == 5.0 and torch.equal(s1["exp_avg"], s2["exp_avg"]) and torch.equal(s1["exp_avg_sq"], s2["exp_avg_sq"])
        g = torch.randn_like(m.gauss_params[k])
        m.gauss_params[k].grad, m2.gauss_params[k].grad = g, g.clone()
        opts[k].step()
        o2[k].step()
        assert torch.equal(m.gauss_params[k], m2.gauss_params[k]), f"{k}: the resumed optimiser steps differently"
        assert float(o2[k].state[m2.gauss_params[k]]["step"]) == 6.0


def _cb(sc, dev) -> Tuple:
    from qed_splatter_amd.model import PinholeCameras
    K = sc["Ks"][0]
    h, w = sc["gt_rgb"].shape[:2]
    cam = PinholeCameras(sc["camera_to_worlds"][:1].to(dev), K[0, 0], K[1, 1], K[0, 2], K[1, 2], w, h)
    return cam, {"image": sc["gt_rgb"].to(dev), "depth_image": sc["gt_depth"].to(dev)}


def test_parent_style_state_surgery_on_separate_tensors(cuda):
    """What the parent's dup / remove-from-optimiser routines do when N changes -- take ``optimizer.state[param]``,
    resize ``exp_avg`` / ``exp_avg_sq``, re-key it under the new Parameter -- works on QedAdam with separately held
    tensors; on flat-buffer views the same surgery is refused with an error that names the supported way."""
    from qed_splatter_amd.model import QedAdam, QEDSplatterModel, QEDSplatterModelConfig
    n, w, h = 1500, 96, 64
    sc = scene(n, w, h, seed=79)
    m, cam, batch, groups = _separate_model(sc, cuda)
    m.train()
    opts = build_optimizers(groups, QedAdam)
    _train(m, cam, batch, opts, 2)
    keep = torch.rand(n, device=cuda) > 0.25
    dup = torch.rand(int(keep.sum()), device=cuda) > 0.5
    new = {}
    for name in m.group_names:                                    # cull, then duplicate -- group by group, as the parent does
        opt, p = opts[name], m.gauss_params[name]
        state = opt.state[p]
        del opt.state[p]
        kept = p.detach()[keep]
        q = nn.Parameter(torch.cat([kept, kept[dup]], dim=0))
        for key in ("exp_avg", "exp_avg_sq"):
            s = state[key][keep]
            state[key] = torch.cat([s, torch.zeros_like(s[dup])], dim=0)
        opt.param_groups[0]["params"] = [q]
        opt.state[q] = state
        new[name] = q
    m.gauss_params = nn.ParameterDict(new)
    n2 = m.num_points
    assert n2 == int(keep.sum()) + int(dup.sum()) and n2 != n
    before = {k: p.detach().clone() for k, p in m.gauss_params.items()}
    _train(m, cam, batch, opts, 2)
    for name in m.group_names:
        st = opts[name].state[m.gauss_params[name]]
        assert float(st["step"]) == 4.0 and st["exp_avg"].shape[0] == n2
        assert not torch.equal(m.gauss_params[name], before[name]) and bool(torch.isfinite(m.gauss_params[name]).all())
    # flat-buffer views: the state is visible, replacing it is refused
    cfg = QEDSplatterModelConfig.synthetic(sh_degree_interval=1)
    mf = QEDSplatterModel(cfg, **{k: sc[k].to(cuda) for k in PARAM_NAMES})
    mf.step = 100
    mf.train()
    of = {k: QedAdam([mf.gauss_params[k]], lr=LRS[k], eps=1e-15) for k in mf.group_names}
    camf, batchf = _cb(sc, cuda)
    _train(mf, camf, batchf, of, 1)
    p = mf.gauss_params["scales"]
    st = of["scales"].state[p]
    assert float(st["step"]) == 1.0 and st["exp_avg"].shape == p.shape and float(st["exp_avg"].abs().max()) > 0.0
    assert st["exp_avg"].untyped_storage().data_ptr() == of["means"].state[mf.gauss_params["means"]]["exp_avg"] \
        .untyped_storage().data_ptr()                                                    # views of ONE shared buffer
    st["exp_avg"] = st["exp_avg"][: n // 2].clone()
    with pytest.raises(RuntimeError, match="Densifier"):
        _train(mf, camf, batchf, of, 1)
