"""GPU tests of the reference-shaped route -- get_outputs() -> get_loss_dict() -> sum -> backward() -> one optimiser per
parameter group -- and of the parent-class semantics behind it (mask on both images, scale_reg, the
coarse-to-fine resolution schedule), against the fp64 oracle and plain torch.  Reference-generated known answers
(tests/golden/reference_kats.npz) are fed to the KERNELS here, not to a torch mirror."""
from __future__ import annotations

import math
import os

import numpy as np
import pytest
import torch

from oracle import splat_oracle as O
from tests.util import PARAM_NAMES, REL_TOL, assert_close, assert_close_elem, scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _model(sc, dev, step=100, **cfg_kw):
    from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
    cfg_kw.setdefault("sh_degree_interval", 1)
    cfg = QEDSplatterModelConfig.synthetic(**cfg_kw)
    m = QEDSplatterModel(cfg, **{k: sc[k].to(dev) for k in PARAM_NAMES})
    m.step = step
    K = sc["Ks"][0]
    h, w = sc["gt_rgb"].shape[:2]
    cam = PinholeCameras(sc["camera_to_worlds"][:1].to(dev), K[0, 0], K[1, 1], K[0, 2], K[1, 2], w, h)
    batch = {"image": sc["gt_rgb"].to(dev), "depth_image": sc["gt_depth"].to(dev)}
    return m, cam, batch


# ---- the two fused nodes against plain torch (the statements of model.py:295-306 / the parent's loss) --------------
@pytest.mark.parametrize("ch", [3, 4])
def test_post_process_matches_torch_statements(cuda, ch):
    from qed_splatter_amd.model import _PostProcess
    g = torch.Generator().manual_seed(3)
    H, W = 37, 53
    render = (torch.rand(1, H, W, ch, generator=g) * 1.6 - 0.3).to(cuda).requires_grad_(True)
    alpha = torch.rand(1, H, W, 1, generator=g)
    alpha[0, :5] = 0.0                                                      # empty pixels take the max depth
    alpha = alpha.to(cuda).requires_grad_(True)
    bg = torch.tensor([0.2, 0.7, 0.4], device=cuda)
    w_rgb = torch.rand(1, H, W, 3, generator=g).to(cuda)
    w_d = torch.rand(1, H, W, 1, generator=g).to(cuda)
    out = _PostProcess.apply(render, alpha, bg)
    rgb, depth = (out, None) if ch == 3 else out
    loss = (rgb * w_rgb).sum() + ((depth * w_d).sum() if depth is not None else 0.0)
    loss.backward()
    g_render, g_alpha = render.grad.clone(), alpha.grad.clone()
    render.grad = alpha.grad = None
    # model.py:296-297, 304-306 verbatim
    rgb_t = torch.clamp(render[:, ..., :3] + (1 - alpha) * bg, 0.0, 1.0)
    loss_t = (rgb_t * w_rgb).sum()
    if ch == 4:
        depth_t = render[:, ..., 3:4]
        depth_t = torch.where(alpha > 0, depth_t, depth_t.detach().max())
        loss_t = loss_t + (depth_t * w_d).sum()
        assert torch.equal(depth, depth_t)
    loss_t.backward()
    assert float((rgb - rgb_t).abs().max()) <= 1e-6
    assert float((g_render - render.grad).abs().max()) <= 1e-6
    assert float((g_alpha - alpha.grad).abs().max()) <= 1e-5


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("lam", [0.0, 0.2])
def test_image_losses_value_and_weighted_gradients(cuda, masked, lam):
    """main / depth losses vs the fp64 oracle, and the backward under DIFFERENT upstream gradients per term."""
    from qed_splatter_amd.model import _ImageLosses
    g = torch.Generator().manual_seed(11)
    H, W = 45, 70
    rgb = torch.rand(H, W, 3, generator=g)
    gt = torch.rand(H, W, 3, generator=g)
    depth = torch.rand(H, W, 1, generator=g) * 9 + 1
    gtd = torch.rand(H, W, 1, generator=g) * 9 + 1
    gtd[torch.rand(H, W, 1, generator=g) < 0.1] = 0.0
    gtd[3, 4] = float("nan")
    gtd[5, 6] = float("inf")
    mask = (torch.rand(H, W, 1, generator=g) > 0.3).float() if masked else None
    a = rgb.to(cuda).requires_grad_(True)
    d = depth.to(cuda).requires_grad_(True)
    main, dl = _ImageLosses.apply(a, d, gt.to(cuda), gtd.to(cuda), mask.to(cuda) if masked else None, lam, 0.2)
    (2.0 * main + 3.0 * dl).backward()
    ar = rgb.double().requires_grad_(True)
    dr = depth.double().requires_grad_(True)
    main_r = O.main_loss(ar, gt.double(), lam, mask.double() if masked else None)
    dl_r = O.depth_l1_loss(dr, gtd.double(), mask.double() if masked else None, 0.2)
    (2.0 * main_r + 3.0 * dl_r).backward()
    assert float(main) == pytest.approx(float(main_r), rel=2e-6)
    assert float(dl) == pytest.approx(float(dl_r), rel=2e-6)
    # |x - y| has a kink: exclude elements within fp32 rounding of it
    ok = ((rgb - gt).abs() > 1e-6)
    if masked:
        ok = ok | (mask == 0)
    assert_close_elem(a.grad.cpu()[ok], ar.grad[ok], f"v_rgb masked={masked} lambda={lam}", atol_frac=2e-6)
    assert_close_elem(d.grad.cpu(), dr.grad, "v_depth", atol_frac=1e-6)


def test_image_losses_no_valid_depth_and_errors(cuda):
    from qed_splatter_amd._lib import QedSplatError
    from qed_splatter_amd.model import _ImageLosses
    H, W = 16, 20
    rgb = torch.rand(H, W, 3, device=cuda, requires_grad=True)
    d = torch.rand(H, W, 1, device=cuda, requires_grad=True)
    main, dl = _ImageLosses.apply(rgb, d, torch.rand(H, W, 3, device=cuda), torch.zeros(H, W, 1, device=cuda), None, 0.2, 0.2)
    assert float(dl) == 0.0                                              # model.py:111-114: 0.0, not NaN
    (main + dl).backward()
    assert float(d.grad.abs().max()) == 0.0 and bool(torch.isfinite(rgb.grad).all())
    with pytest.raises(QedSplatError):                                    # smaller than the SSIM window
        _ImageLosses.apply(torch.rand(8, 8, 3, device=cuda), None, torch.rand(8, 8, 3, device=cuda), None, None, 0.2, 0.2)


# ---- reference-generated known answers through the kernels --------------------------------------------
def test_reference_depth_l1_kats_through_both_kernel_paths(cuda, lib):
    """The five depth-L1 cases the reference itself produced (masks, NaN, inf, zero GT, no valid pixel) and SURVEY's
    0.0653 through (i) get_loss_dict's kernels and (ii) K8 (qed_loss_reduce / qed_loss_grad with alpha = 1 and the
    depth in channel 3)."""
    from qed_splatter_amd import _lib as L
    from qed_splatter_amd.model import _ImageLosses
    kats = np.load(os.path.join(GOLD, "reference_kats.npz"))
    cases = [(kats[f"dl{i}_depth_out"], kats[f"dl{i}_depth_gt"], kats[f"dl{i}_mask"], float(kats[f"dl{i}_lambda"]),
              float(kats[f"dl{i}_loss"])) for i in kats["dl_cases"]]
    cases.append((kats["survey_depth_out"], kats["survey_depth_gt"], np.zeros(0), 0.2, float(kats["survey_loss"])))
    st = torch.cuda.current_stream().cuda_stream
    for d_out, d_gt, m, lam, want in cases:
        H, W = d_out.shape[:2]
        d = torch.from_numpy(d_out).float().reshape(H, W, 1).to(cuda)
        g = torch.from_numpy(d_gt).float().reshape(H, W, 1).to(cuda)
        mask = torch.from_numpy(m).float().reshape(H, W, 1).to(cuda) if m.size else None
        rgb = torch.zeros(H, W, 3, device=cuda)
        _, dl = _ImageLosses.apply(rgb, d, rgb, g, mask, 0.0, lam)
        assert float(dl) == pytest.approx(want, rel=1e-6, abs=1e-9)
        # K8: render = (0, 0, 0, depth), alpha = 1 (so the depth fix-up keeps every pixel)
        render = torch.cat([torch.zeros(H, W, 3, device=cuda), d], dim=-1).contiguous()
        alpha = torch.ones(H, W, 1, device=cuda)
        bg = torch.zeros(3, device=cuda)
        sums = torch.empty(L.LOSS_SUMS_FLOATS, device=cuda)
        losses = torch.empty(3, device=cuda)
        v_r, v_a = torch.empty_like(render), torch.empty_like(alpha)
        args = (H * W, 4, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(rgb), L.ptr(g), L.ptr(mask))
        L.check(lib.qed_loss_reduce(*args, L.ptr(sums), st), "reduce")
        L.check(lib.qed_loss_grad(*args, L.ptr(sums), 1.0, lam, L.ptr(v_r), L.ptr(v_a), L.ptr(losses), None, None, 0, 0.0,
                                  0.0, st), "grad")
        assert float(losses[1]) == pytest.approx(want, rel=1e-6, abs=1e-9)
        # NaN renders must not poison the gradient of the other pixels
        assert bool(torch.isfinite(v_r).all())


def test_reference_viewmat_kats_through_camera_setup(cuda, lib):
    """The six rigid poses of reference_kats.npz (random rotations, SURVEY's known answer) through qed_camera_setup.
    Bit-exact against the reference's own output except where R^T t is rounded differently: the kernel may contract
    the 3-term dot product into FMAs (<= 1 ulp of the largest product), so the translation column gets 2 ulp."""
    from qed_splatter_amd import _lib as L
    kats = np.load(os.path.join(GOLD, "reference_kats.npz"))
    c2w = torch.from_numpy(kats["viewmat_c2w"]).float().to(cuda).contiguous()
    want = torch.from_numpy(kats["viewmat_out"]).float()
    C = c2w.shape[0]
    intr = torch.tensor([[500.0, 510.0, 320.0, 240.0]] * C, device=cuda)
    vm = torch.empty(C, 4, 4, device=cuda)
    Ks = torch.empty(C, 3, 3, device=cuda)
    L.check(lib.qed_camera_setup(C, L.ptr(c2w), L.ptr(intr), L.ptr(vm), L.ptr(Ks), torch.cuda.current_stream().cuda_stream),
            "qed_camera_setup")
    vm = vm.cpu()
    assert torch.equal(vm[:, :3, :3], want[:, :3, :3]) and torch.equal(vm[:, 3], want[:, 3])
    scale = c2w[:, :3, 3].abs().max().item()
    assert float((vm[:, :3, 3] - want[:, :3, 3]).abs().max()) <= 2 * np.spacing(np.float32(scale)) * 3
    K0 = Ks.cpu()[0]
    assert K0.tolist() == [[500.0, 0.0, 320.0], [0.0, 510.0, 240.0], [0.0, 0.0, 1.0]]


def test_projection_derives_the_view_matrix_itself(cuda, lib):
    """QED_F_CAMERA_C2W: rasterization(..., _c2w=(camera-to-world, intrinsics)) -- get_viewmat inside the projection kernel
    -- against the same call with view matrices from qed_camera_setup, for the six random rigid poses of the reference's
    known answers at once (C = 6): the view matrices it leaves behind match the reference's to 2 ulp of the translation,
    everything projected from them agrees to fp32 rounding, the integer outputs exactly."""
    from qed_splatter_amd import _lib as L
    from qed_splatter_amd.rasterization import rasterization
    kats = np.load(os.path.join(GOLD, "reference_kats.npz"))
    c2w = torch.from_numpy(kats["viewmat_c2w"]).float()
    want = torch.from_numpy(kats["viewmat_out"]).float()
    C = c2w.shape[0]
    # look at the origin region from a few units away so that the random poses see something
    w, h, n = 96, 64, 1500
    sc = scene(n, w, h, seed=31)
    g = torch.Generator().manual_seed(3)
    means = (torch.rand(n, 3, generator=g) - 0.5) * 8.0
    c2w = c2w.clone()
    c2w[:, :3, 3] = c2w[:, :3, 2] * 6.0                    # camera on its own viewing axis (OpenGL: looks down -z)
    c2w = c2w.to(cuda).contiguous()
    intr = torch.tensor([[80.0, 82.0, w / 2, h / 2]] * C, device=cuda)
    vm_ref = torch.empty(C, 4, 4, device=cuda)
    K_ref = torch.empty(C, 3, 3, device=cuda)
    L.check(lib.qed_camera_setup(C, L.ptr(c2w), L.ptr(intr), L.ptr(vm_ref), L.ptr(K_ref),
                                 torch.cuda.current_stream().cuda_stream), "qed_camera_setup")
    kw = dict(means=means.to(cuda), quats=sc["quats"].to(cuda), scales=sc["scales"].exp().to(cuda),
              opacities=torch.sigmoid(sc["opacities"]).reshape(-1).to(cuda),
              colors=torch.cat([sc["features_dc"][:, None], sc["features_rest"]], 1).to(cuda), width=w, height=h,
              sh_degree=3, render_mode="RGB+D")
    r0, a0, i0 = rasterization(viewmats=vm_ref, Ks=K_ref, **kw)
    vm, Ks = torch.full((C, 4, 4), 7.0, device=cuda), torch.full((C, 3, 3), 7.0, device=cuda)
    r1, a1, i1 = rasterization(viewmats=vm, Ks=Ks, _c2w=(c2w, intr), **kw)
    torch.cuda.synchronize()
    assert int(i0["n_isects"]) > 1000
    assert torch.equal(Ks, K_ref)
    assert torch.equal(vm[:, :3, :3], vm_ref[:, :3, :3]) and torch.equal(vm[:, 3], vm_ref[:, 3])
    scale = float(c2w[:, :3, 3].abs().max())
    assert float((vm[:, :3, 3] - vm_ref[:, :3, 3]).abs().max()) <= 2 * np.spacing(np.float32(scale)) * 3
    same_t = torch.equal(vm, vm_ref)
    for k in ("radii", "tiles_per_gauss"):
        assert same_t is False or torch.equal(i1[k], i0[k])
        assert float((i1[k] != i0[k]).float().mean()) < 2e-3            # (a rounding-level shift can move a tile edge)
    assert_close(i1["means2d"], i0["means2d"], 1e-5, "means2d")
    assert_close(i1["depths"], i0["depths"], 1e-5, "depths")
    if same_t:
        assert torch.equal(r1, r0) and torch.equal(a1, a0)
    else:
        assert float((r1 - r0).abs().max()) < 5e-3 and float((a1 - a0).abs().max()) < 5e-3


# ---- parent-class semantics -----------------------------------------------------------------------------
def test_masked_rgb_loss_api_fused_and_oracle_agree(cuda):
    """The mask multiplies BOTH images before L1 and SSIM (parent) and both depths (model.py:93-97): API route, fused
    route and the oracle agree to the north_star's 1e-4, and masked-out pixels pass no colour gradient.  (The band
    mask also carries the threshold pixels of this scene, tests/util.py::threshold_pixel_mask: no Gaussian is left
    out of the comparison.)"""
    from tests.util import threshold_pixel_mask
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=5)
    m1, cam, batch = _model(sc, cuda)
    with torch.no_grad():
        m1.get_outputs(cam)
    ps = {k: sc[k].double().requires_grad_(True) for k in PARAM_NAMES}
    ref = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                               ps["features_rest"], sc["camera_to_worlds"][:1].double(), sc["Ks"][:1].double(), w, h,
                               sc["background"].double(), radii_override=m1.info["radii"].cpu(), return_margin=True)
    mask = threshold_pixel_mask(ref, sc["gt_rgb"], sc["gt_depth"], 1e-4).float()
    mask[:, : w // 3] = 0.0                                               # a masked-out band
    batch["mask"] = (mask > 0).to(cuda)                                   # bool, as Nerfstudio delivers it
    out = m1.get_outputs(cam)
    out["rgb"].retain_grad()
    ld = m1.get_loss_dict(out, batch)
    assert set(ld) == {"main_loss", "scale_reg", "depth_loss"} and float(ld["scale_reg"]) == 0.0
    sum(ld.values()).backward()
    assert float(out["rgb"].grad[:, : w // 3].abs().max()) == 0.0
    m2, cam2, batch2 = _model(sc, cuda)
    batch2["mask"] = mask.to(cuda)
    lf = m2.fused_loss(cam2, batch2)
    lf["loss"].backward()
    l_rgb = O.main_loss(ref["rgb"], sc["gt_rgb"].double(), 0.2, mask.double())
    l_d = O.depth_l1_loss(ref["depth"], sc["gt_depth"].double(), mask.double(), 0.2)
    for got in (ld, lf):
        assert float(got["main_loss"]) == pytest.approx(float(l_rgb), rel=1e-4)
        assert float(got["depth_loss"]) == pytest.approx(float(l_d), rel=1e-4)
    (l_rgb + l_d).backward()
    for name in PARAM_NAMES:
        assert_close(m2.gauss_params[name].grad, m1.gauss_params[name].grad, 2e-5, f"fused vs api grad {name}")
        assert_close(m1.gauss_params[name].grad.cpu(), ps[name].grad, REL_TOL, f"api vs oracle grad {name}")
        assert_close_elem(m1.gauss_params[name].grad.cpu(), ps[name].grad, f"api vs oracle grad {name}", atol_frac=1e-5)


def test_scale_regularization_matches_oracle(cuda):
    w, h, n = 64, 48, 500
    sc = scene(n, w, h, seed=9)
    sc["scales"][:50, 0] += 3.0                                           # some very anisotropic Gaussians
    m, cam, batch = _model(sc, cuda, step=20, use_scale_regularization=True)
    out = m.get_outputs(cam)
    ld = m.get_loss_dict(out, batch)
    want = O.scale_reg(sc["scales"].double(), 20, True, 10.0)
    assert float(want) > 0 and float(ld["scale_reg"]) == pytest.approx(float(want), rel=1e-5)
    m.step = 21
    assert float(m.get_loss_dict(m.get_outputs(cam), batch)["scale_reg"]) == 0.0      # only every 10th step


def test_resolution_schedule_api_and_fused(cuda):
    """num_downscales = 2 (the parent's default): renders at 1/4, 1/2, full resolution as the step passes the
    schedule; the ground truth is box-filtered (the parent's resize_image); API and fused routes agree."""
    w, h, n = 256, 192, 4000
    sc = scene(n, w, h, seed=21)
    for step, d in [(0, 4), (3000, 2), (6000, 1)]:
        m1, cam, batch = _model(sc, cuda, step=step, num_downscales=2, resolution_schedule=3000, sh_degree_interval=1000)
        assert m1._get_downscale_factor() == d
        out = m1.get_outputs(cam)
        assert out["rgb"].shape == (h // d, w // d, 3) and m1.last_size == (h // d, w // d)
        assert int(cam.width[0]) == w                                     # the camera is restored (model.py:250)
        gt = m1.get_gt_img(batch["image"])
        assert_close(gt.cpu(), O.resize_image(sc["gt_rgb"], d), 1e-6, f"resize_image d={d}")
        ld = m1.get_loss_dict(out, batch)
        sum(ld.values()).backward()
        m2, cam2, batch2 = _model(sc, cuda, step=step, num_downscales=2, resolution_schedule=3000,
                                  sh_degree_interval=1000)
        lf = m2.fused_loss(cam2, batch2)
        lf["loss"].backward()
        assert float(lf["main_loss"]) == pytest.approx(float(ld["main_loss"]), rel=1e-5)
        assert float(lf["depth_loss"]) == pytest.approx(float(ld["depth_loss"]), rel=1e-5)
        for name in PARAM_NAMES:
            assert_close(m2.gauss_params[name].grad, m1.gauss_params[name].grad, 5e-5, f"d={d} grad {name}")
    m1.eval()
    assert m1._get_downscale_factor() == 1                                # eval renders at full resolution


def test_background_colors_and_rgba_ground_truth(cuda):
    from qed_splatter_amd.model import QEDSplatterModelConfig
    sc = scene(200, 48, 32, seed=2)
    m, cam, batch = _model(sc, cuda)
    m.config = QEDSplatterModelConfig(sh_degree_interval=1)              # the parent's defaults: "random", 2 downscales
    assert m.config.background_color == "random" and m.config.num_downscales == 2
    m.train()
    b1, b2 = m._get_background_color(), m._get_background_color()
    assert b1.shape == (3,) and not torch.equal(b1, b2)
    m.eval()
    assert m._get_background_color().tolist() == pytest.approx([0.1490, 0.1647, 0.2157])
    # RGBA ground truth is composited onto the step's background (parent's composite_with_background)
    rgba = torch.rand(32, 48, 4, device=cuda)
    bg = torch.tensor([0.3, 0.6, 0.9], device=cuda)
    got = m.composite_with_background(rgba, bg)
    want = rgba[..., 3:] * rgba[..., :3] + (1 - rgba[..., 3:]) * bg
    assert torch.allclose(got, want)
    m.train()
    m.config.use_bilateral_grid = True
    cam.metadata = {"cam_idx": 0}
    with pytest.raises(NotImplementedError):
        m.get_outputs(cam)


# ---- input validation before raw pointers reach a kernel (ADVICE) -----------------------------------------------------
def test_batch_validation_bool_mask_uint8_image_wrong_size(cuda):
    from qed_splatter_amd._lib import QedSplatError
    w, h, n = 96, 64, 800
    sc = scene(n, w, h, seed=4)
    m, cam, batch = _model(sc, cuda)
    ref = m.fused_loss(cam, dict(batch, mask=(torch.rand(h, w, 1, device=cuda) > 0.2).float()))
    torch.manual_seed(0)
    mask = torch.rand(h, w, 1, device=cuda) > 0.2
    a = m.fused_loss(cam, dict(batch, mask=mask))                         # bool mask: converted, not reinterpreted
    b = m.fused_loss(cam, dict(batch, mask=mask.float()))
    assert float(a["loss"]) == float(b["loss"]) and math.isfinite(float(ref["loss"]))
    img8 = (batch["image"] * 255).round().to(torch.uint8)                 # cached uint8 images (config.py:38)
    c = m.fused_loss(cam, dict(batch, image=img8.cpu()))                  # ... that live on the host
    d = m.fused_loss(cam, dict(batch, image=img8.float() / 255.0))
    assert float(c["loss"]) == float(d["loss"])
    with pytest.raises(QedSplatError):
        m.fused_loss(cam, dict(batch, image=batch["image"][: h // 2]))    # wrong size: refused before any launch
    with pytest.raises(QedSplatError):
        m.fused_loss(cam, dict(batch, depth_image=batch["depth_image"][:, : w - 1]))
    with pytest.raises(TypeError):
        m.fused_loss(cam, dict(batch, mask=(mask.to(torch.uint8) * 255)))
    with pytest.raises(AssertionError):
        m.fused_loss(cam, dict(batch, mask=mask[: h - 1]))


def test_scaled_loss_backward_through_the_fused_node(cuda):
    """(2 * loss).backward() gives 2 x the gradients (a weighted loss or a GradScaler upstream of the fused node)."""
    w, h, n = 96, 64, 800
    sc = scene(n, w, h, seed=4)
    m1, cam1, batch1 = _model(sc, cuda)
    m1.backward_fused(m1.fused_loss(cam1, batch1))
    m2, cam2, batch2 = _model(sc, cuda)
    (2.0 * m2.fused_loss(cam2, batch2)["loss"]).backward()
    for name in PARAM_NAMES:
        g1, g2 = m1.gauss_params[name].grad, m2.gauss_params[name].grad
        assert_close(g2, 2.0 * g1, 2e-5, f"scaled grad {name}")       # atomics: summation order differs between runs


# ---- optimiser ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1000, 1003])                               # 1003: group offsets not 16-byte aligned
@pytest.mark.parametrize("flat_grads", [True, False])
def test_qed_adam_six_instances_match_torch_adam(cuda, n, flat_grads):
    """One QedAdam per parameter group (how Nerfstudio builds optimisers from config.py:44-68) == six torch.optim.Adam
    on IDENTICAL gradients (two renders differ by atomic summation order, which Adam with eps = 1e-15 turns into
    +-lr for near-zero gradients), with a scheduler on the means' rate.  flat_grads: the six gradients are adjacent
    views of one allocation (what the projection backward produces -> one fused launch) or separate tensors."""
    from qed_splatter_amd.model import FlatAdam, QedAdam
    sc = scene(n, 96, 64, seed=8)
    lrs = FlatAdam.DEFAULT_LRS
    runs = []
    for kind in ("qed", "torch"):
        m, _, _ = _model(sc, cuda)
        cls = QedAdam if kind == "qed" else torch.optim.Adam
        opts = {k: cls([m.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in m.group_names}
        sched = torch.optim.lr_scheduler.LambdaLR(opts["means"], lambda s: 0.9 ** s)
        gen = torch.Generator().manual_seed(5)
        for _ in range(5):
            flat = (torch.randn(m.flat_params.numel(), generator=gen) * 1e-3).to(cuda)
            for k, b in zip(m.group_names, m.group_begin):
                p = m.gauss_params[k]
                g = flat[b:b + p.numel()].view(p.shape)
                p.grad = g if flat_grads else g.clone()
            for o in opts.values():
                o.step()
            sched.step()
        runs.append((m, opts))
    (mq, oq), (mt, ot) = runs
    for k in PARAM_NAMES:
        assert_close_elem(mq.gauss_params[k], mt.gauss_params[k], f"params {k} after 5 steps (n={n})", atol_frac=1e-6)
    # checkpoint layout of torch.optim.Adam: per-group exp_avg / exp_avg_sq / step
    sd_q, sd_t = oq["scales"].state_dict(), ot["scales"].state_dict()
    assert float(sd_q["state"][0]["step"]) == float(sd_t["state"][0]["step"]) == 5.0
    # (1 - beta reaches the kernel formed in double and rounded once, as torch's Python scalars are)
    assert_close(sd_q["state"][0]["exp_avg"], sd_t["state"][0]["exp_avg"], 1e-6, "exp_avg")
    assert_close(sd_q["state"][0]["exp_avg_sq"], sd_t["state"][0]["exp_avg_sq"], 1e-6, "exp_avg_sq")
    assert sd_q["param_groups"][0]["lr"] == sd_t["param_groups"][0]["lr"]
    assert oq["means"].param_groups[0]["lr"] == pytest.approx(lrs["means"] * 0.9 ** 5)
    # a group stepped on its own (out of step with the others) is updated on its own, with its own step count
    for m_, o_ in ((mq, oq), (mt, ot)):
        o_["quats"].step()
    oq["quats"].flush()
    assert_close_elem(mq.gauss_params["quats"], mt.gauss_params["quats"], "quats stepped alone", atol_frac=1e-6)
    assert float(oq["quats"].state_dict()["state"][0]["step"]) == 6.0
    oq["scales"].load_state_dict(sd_q)
    assert float(oq["scales"].state_dict()["state"][0]["step"]) == 5.0


def test_qed_adam_in_the_reference_call_sequence(cuda):
    """zero_grad -> get_outputs -> get_metrics_dict -> get_loss_dict -> sum -> backward -> six optimisers: runs, lowers
    the loss, and the six gradients arrive as adjacent views (one fused optimiser launch)."""
    from qed_splatter_amd.model import FlatAdam, QedAdam
    sc = scene(3000, 160, 112, seed=12)
    m, cam, batch = _model(sc, cuda)
    opts = {k: QedAdam([m.gauss_params[k]], lr=FlatAdam.DEFAULT_LRS[k], eps=1e-15) for k in m.group_names}
    losses = []
    for _ in range(12):
        for o in opts.values():
            o.zero_grad(set_to_none=True)
        out = m.get_outputs(cam)
        md = m.get_metrics_dict(out, batch)
        ld = m.get_loss_dict(out, batch, md)
        loss = sum(ld.values())
        loss.backward()
        assert m.flat_grad().data_ptr() == m.gauss_params["means"].grad.data_ptr()
        for o in opts.values():
            o.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] and all(math.isfinite(x) for x in losses)
    assert "rgb_psnr" in md and "depth_abs_rel" in md


def test_flat_adam_refuses_stale_buffers_and_compact_gradients(cuda):
    from qed_splatter_amd.model import FlatAdam
    from qed_splatter_amd.parallel import allreduce_flat_grad
    sc = scene(500, 64, 48, seed=3)
    m, cam, batch = _model(sc, cuda)
    opt = FlatAdam(m)
    m.backward_fused(m.fused_loss(cam, batch, compact_sh_grad=True))
    with pytest.raises(RuntimeError, match="compact"):
        opt.step()                                                        # features_rest.grad is unwritten memory
    with pytest.raises(RuntimeError, match="compact"):
        allreduce_flat_grad(m, 1)
    opt.step(fused_sh=True)
    # model.to() replaces every Parameter's data: the model re-creates its flat buffer, the optimiser notices
    m.cpu()
    m.to(cuda)
    assert m.gauss_params["scales"].data_ptr() == m.flat_params.data_ptr() + 4 * m.group_begin[1]
    m.backward_fused(m.fused_loss(cam, batch))
    with pytest.raises(RuntimeError, match="new flat parameter buffer"):
        opt.step()


@pytest.mark.parametrize("uint8_image", [False, True])
def test_ssim_forward_shared_between_metrics_and_loss(cuda, uint8_image):
    """get_metrics_dict computes rgb_ssim of (rgb, gt) and get_loss_dict needs the SSIM of the same two images: in training
    the second forward pass is not run (one qed_ssim_fwd per step instead of two), with identical loss values and rgb
    gradients; a mask, another image, eval mode or an in-place change of the render all fall back to the full computation."""
    from qed_splatter_amd import _lib as L
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=9)

    def run(with_metrics, tweak=None):
        m, cam, batch = _model(sc, cuda)
        m.train()
        if uint8_image:
            batch["image"] = (batch["image"] * 255).round().to(torch.uint8)
        out = m.get_outputs(cam)
        out["rgb"].retain_grad()
        L.TIMER.reset()
        L.TIMER.active = True
        try:
            md = m.get_metrics_dict(out, batch) if with_metrics else None
            if tweak is not None:
                tweak(m, out, batch)
            ld = m.get_loss_dict(out, batch, md)
            torch.cuda.synchronize()
        finally:
            L.TIMER.active = False
        n_fwd = len(L.TIMER.events.get("qed_ssim_fwd", []))
        calls = {k: len(v) for k, v in L.TIMER.events.items()}
        sum(ld.values()).backward()
        assert m._step.ssim is None                                  # consumed (or never made)
        run.last = (m, out, batch, calls)
        return md, ld, out["rgb"].grad.clone(), n_fwd

    _, ld0, g0, n0 = run(False)
    md1, ld1, g1, n1 = run(True)
    assert n0 == 1 and n1 == 1, (n0, n1)                              # metrics + loss: still ONE SSIM forward
    # ... and ONE reduction pass: qed_step_metrics took the loss sums along, get_loss_dict's forward launched nothing
    m1, out1, batch1, calls = run.last
    assert calls.get("qed_step_metrics") == 1 and "qed_image_losses_fwd" not in calls and "qed_image_metrics" not in calls \
        and "qed_nanmean_exp" not in calls, calls
    assert float(ld1["main_loss"]) == pytest.approx(float(ld0["main_loss"]), rel=2e-6)      # (fp64 fold vs fp32 fold)
    assert float(ld1["depth_loss"]) == pytest.approx(float(ld0["depth_loss"]), rel=2e-6)
    assert torch.equal(g1, g0)
    # rgb_ssim is the value the loss used: main = 0.8 L1 + 0.2 (1 - ssim)
    assert 0.0 < float(md1["rgb_ssim"]) < 1.0
    # the one-pass metrics equal the separate entry points' (the eval-mode route of get_metrics_dict)
    from qed_splatter_amd.metrics import metrics_dict, nanmean_exp
    ref = metrics_dict(out1["rgb"].detach(), m1.get_gt_img(batch1["image"])[..., :3], out1["depth"].detach(), batch1["depth_image"])
    for k, v in ref.items():
        if k != "rgb_lpips":
            assert float(md1[k]) == pytest.approx(float(v), rel=1e-5), k
    assert float(md1["avg_min_scale"]) == pytest.approx(float(nanmean_exp(m1.scales[..., -1].detach())), rel=1e-6)
    assert md1["gaussian_count"] == n and set(md1) == set(ref) | {"gaussian_count", "avg_min_scale"}

    def add_mask(m, out, batch):
        batch["mask"] = torch.ones(h, w, 1, device=cuda)
    assert run(True, add_mask)[3] == 2                                # masked loss: SSIM of other images

    def touch(m, out, batch):
        with torch.no_grad():
            out["rgb"].mul_(1.0)                                      # same values, new version: not trusted
    assert run(True, touch)[3] == 2

    def other_image(m, out, batch):
        batch["image"] = batch["image"].clone()
    assert run(True, other_image)[3] == 2

    m, cam, batch = _model(sc, cuda)
    m.eval()
    with torch.no_grad():
        out = m.get_outputs(cam)
        m.get_metrics_dict(out, batch)
    assert m._step.ssim is None                                       # nothing kept outside training


@pytest.mark.parametrize("ch,masked,size", [(4, False, (75, 101)), (4, True, (64, 96)), (3, False, (45, 70)), (3, True, (33, 33))])
def test_fused_ssim_backward_and_loss_gradient_equals_the_two_passes(cuda, lib, ch, masked, size):
    """qed_loss_grad_ssim (SSIM backward + loss gradient in one launch) against qed_ssim_bwd followed by qed_loss_grad on
    random buffers: colours on both sides of the clamp, alpha == 0 pixels (depth fix-up), invalid ground-truth depths,
    image sizes that are not multiples of the 32 x 32 tile."""
    from qed_splatter_amd import _lib as L
    H, W = size
    g = torch.Generator().manual_seed(H * 1000 + W + ch)
    render = (torch.rand(H, W, ch, generator=g) * 1.6 - 0.3)
    if ch == 4:
        render[..., 3] = torch.rand(H, W, generator=g) * 10.0
    alpha = torch.rand(H, W, 1, generator=g)
    alpha[torch.rand(H, W, 1, generator=g) < 0.1] = 0.0
    bg = torch.tensor([0.2, 0.5, 0.9])
    gt = torch.rand(H, W, 3, generator=g)
    gd = torch.rand(H, W, 1, generator=g) * 10.0
    gd[torch.rand(H, W, 1, generator=g) < 0.1] = 0.0
    gd[0, 0, 0] = float("nan")
    mask = (torch.rand(H, W, 1, generator=g) > 0.3).float() if masked else None
    render, alpha, bg, gt, gd = (t.to(cuda).contiguous() for t in (render, alpha, bg, gt, gd))
    mask = mask.to(cuda).contiguous() if masked else None
    lam, dl = 0.2, 0.2
    n_out = 3.0 * (H - 10) * (W - 10)
    st = torch.cuda.current_stream().cuda_stream
    maps = torch.empty(lib.qed_ssim_maps_floats(H, W), device=cuda)
    ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), device=cuda)
    L.check(lib.qed_ssim_fwd(H, W, ch, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(gt), L.ptr(mask), L.ptr(maps),
                             L.ptr(ssum), st), "qed_ssim_fwd")
    args = (H * W, ch, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(gt), L.ptr(gd), L.ptr(mask))
    out = []
    zall = torch.full((16 * 1237 + 64,), 3.0, device=cuda)             # the buffer the fused launch also zeroes + a guard
    zbuf, guard = zall[: 16 * 1237], zall[16 * 1237:]
    for fused in (False, True):
        sums = torch.full((L.LOSS_SUMS_FLOATS,), float("nan"), device=cuda)
        losses = torch.empty(3, device=cuda)
        v_r, v_a = torch.full_like(render, 7.0), torch.full_like(alpha, 7.0)
        L.check(lib.qed_loss_reduce(*args, L.ptr(sums), st), "qed_loss_reduce")
        if fused:
            L.check(lib.qed_loss_grad_ssim(H, W, ch, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(gt), L.ptr(gd),
                                           L.ptr(mask), L.ptr(maps), L.ptr(sums), 1.0 - lam, dl, -lam / n_out, L.ptr(v_r),
                                           L.ptr(v_a), L.ptr(losses), L.ptr(ssum), ssum.numel(), lam, L.ptr(zbuf),
                                           zbuf.numel(), None, st), "qed_loss_grad_ssim")
            torch.cuda.synchronize()
            assert float(zbuf.abs().max()) == 0.0 and float(guard.min()) == 3.0     # zeroed, and nothing beyond it
        else:
            v_rgb = torch.empty(H, W, 3, device=cuda)
            L.check(lib.qed_ssim_bwd(H, W, ch, L.ptr(render), L.ptr(alpha), L.ptr(bg), L.ptr(gt), L.ptr(mask), L.ptr(maps),
                                     -lam / n_out, None, L.ptr(v_rgb), st), "qed_ssim_bwd")
            L.check(lib.qed_loss_grad(*args, L.ptr(sums), 1.0 - lam, dl, L.ptr(v_r), L.ptr(v_a), L.ptr(losses), L.ptr(v_rgb),
                                      L.ptr(ssum), ssum.numel(), -lam / n_out, lam, st), "qed_loss_grad")
        torch.cuda.synchronize()
        out.append((v_r, v_a, losses))
    (r0, a0, l0), (r1, a1, l1) = out
    assert torch.isfinite(r1).all() and torch.isfinite(a1).all()
    # same expressions, evaluated by another kernel: at most a fused-multiply-add of difference
    assert float((r1 - r0).abs().max()) <= 1e-9 + 2e-7 * float(r0.abs().max())
    assert float((a1 - a0).abs().max()) <= 1e-9 + 2e-7 * float(a0.abs().max())
    assert torch.allclose(l1, l0, rtol=2e-6, atol=0.0), (l1, l0)
    if ch == 3:
        assert float(l1[1]) == 0.0


@pytest.mark.parametrize("masked,with_depth", [(False, True), (True, True), (False, False)])
def test_fused_image_losses_backward_equals_the_two_passes(cuda, lib, masked, with_depth):
    """qed_image_losses_ssim_bwd == qed_ssim_bwd (scaled by the device-resident upstream gradient) followed by
    qed_image_losses_bwd(accumulate): upstream gradients 2 and 3, invalid depths, a size that is not a multiple of 32."""
    from qed_splatter_amd import _lib as L
    H, W = 70, 107
    g = torch.Generator().manual_seed(17)
    rgb, gt = torch.rand(H, W, 3, generator=g), torch.rand(H, W, 3, generator=g)
    depth, gd = torch.rand(H, W, 1, generator=g) * 9 + 0.5, torch.rand(H, W, 1, generator=g) * 10
    gd[torch.rand(H, W, 1, generator=g) < 0.15] = 0.0
    gd[3, 4, 0] = float("inf")
    mask = (torch.rand(H, W, 1, generator=g) > 0.25).float().to(cuda) if masked else None
    rgb, gt, depth, gd = (t.to(cuda).contiguous() for t in (rgb, gt, depth, gd))
    lam, dl = 0.2, 0.2
    n_out = 3.0 * (H - 10) * (W - 10)
    st = torch.cuda.current_stream().cuda_stream
    maps = torch.empty(lib.qed_ssim_maps_floats(H, W), device=cuda)
    ssum = torch.empty(lib.qed_ssim_sum_floats(H, W), device=cuda)
    sums = torch.empty(L.LOSS_SUMS_FLOATS, device=cuda)
    losses = torch.empty(3, device=cuda)
    L.check(lib.qed_ssim_fwd(H, W, 3, L.ptr(rgb), None, None, L.ptr(gt), L.ptr(mask), L.ptr(maps), L.ptr(ssum), st), "ssim_fwd")
    L.check(lib.qed_image_losses_fwd(H * W, L.ptr(rgb), L.ptr(depth), L.ptr(gt), L.ptr(gd), L.ptr(mask), 1.0 - lam, dl,
                                     L.ptr(ssum), ssum.numel(), -lam / n_out, lam, L.ptr(sums), L.ptr(losses), st), "fwd")
    g_main, g_depth = torch.tensor([2.0], device=cuda), torch.tensor([3.0], device=cuda)
    a_rgb, a_d = torch.empty_like(rgb), torch.empty_like(depth)
    L.check(lib.qed_ssim_bwd(H, W, 3, L.ptr(rgb), None, None, L.ptr(gt), L.ptr(mask), L.ptr(maps), -lam / n_out,
                             L.ptr(g_main), L.ptr(a_rgb), st), "qed_ssim_bwd")
    L.check(lib.qed_image_losses_bwd(H * W, L.ptr(rgb), L.ptr(depth), L.ptr(gt), L.ptr(gd), L.ptr(mask), L.ptr(sums),
                                     1.0 - lam, dl, L.ptr(g_main), L.ptr(g_depth), 1, L.ptr(a_rgb), L.ptr(a_d), st), "bwd")
    b_rgb, b_d = torch.full_like(rgb, 5.0), torch.full_like(depth, 5.0)
    zbuf = torch.full((1000, 16), 3.0, device=cuda)
    L.check(lib.qed_image_losses_ssim_bwd(H, W, L.ptr(rgb), L.ptr(depth), L.ptr(gt), L.ptr(gd), L.ptr(mask), L.ptr(maps),
                                          L.ptr(sums), 1.0 - lam, dl, -lam / n_out, L.ptr(g_main), L.ptr(g_depth),
                                          L.ptr(b_rgb), L.ptr(b_d) if with_depth else None, L.ptr(zbuf), zbuf.numel(), st), "fused")
    torch.cuda.synchronize()
    assert float(zbuf.abs().max()) == 0.0                              # the launch also zeroes the buffer it is handed
    assert float((b_rgb - a_rgb).abs().max()) <= 1e-9 + 2e-7 * float(a_rgb.abs().max())
    if with_depth:
        assert torch.equal(b_d, a_d) and float(a_d.abs().max()) > 0.0
    else:
        assert float((b_d - 5.0).abs().max()) == 0.0                   # untouched


def test_optimizer_tick_riding_on_the_loss_launch(cuda):
    """fused_loss(optimizer=opt): the loss pass's fold launch advances the optimiser's device step state and the Adam call
    launches no tick of its own -- same parameters, moments, step counter and scheduled rate as the plain sequence over
    several steps; a second fused_loss before the step does not tick twice; a host-state step after a taken tick raises."""
    from qed_splatter_amd.model import FlatAdam
    w, h, n = 128, 96, 2000
    sc = scene(n, w, h, seed=41)
    runs = []
    for hook in (False, True):
        m, cam, batch = _model(sc, cuda)
        opt = FlatAdam(m, means_schedule=(1.6e-6, 50))
        saved = []
        for it in range(4):
            for p in m.parameters():
                p.grad = None
            losses = m.fused_loss(cam, batch, compact_sh_grad=True, optimizer=opt if hook else None)
            if hook and it == 2:                                   # a repeated forward pass before the step
                losses = m.fused_loss(cam, batch, compact_sh_grad=True, optimizer=opt)
            m.backward_fused(losses)
            # identical gradients in both runs (the compositing backward's atomics are not order-deterministic)
            if hook:
                m.flat_grad().copy_(runs[0][2][it])
            else:
                saved.append(m.flat_grad().clone())
            opt.step(device_state=True, fused_sh=True)
        torch.cuda.synchronize()
        runs.append((m, opt, saved))
    (m0, o0, _), (m1, o1, _) = runs
    assert float(o0.dev_state[0]) == float(o1.dev_state[0]) == 4.0
    assert torch.equal(o1.dev_state, o0.dev_state) and torch.equal(o1.dev_lr, o0.dev_lr)
    assert torch.equal(m1.flat_params, m0.flat_params)
    assert torch.equal(o1.exp_avg, o0.exp_avg) and torch.equal(o1.exp_avg_sq, o0.exp_avg_sq)
    m, cam, batch = _model(sc, cuda)
    opt = FlatAdam(m)
    m.backward_fused(m.fused_loss(cam, batch, compact_sh_grad=True, optimizer=opt))
    with pytest.raises(RuntimeError, match="device_state=True"):
        opt.step(fused_sh=True)


# ---- get_outputs' post-processing inside the compositing kernels (rasterization(_post_background=...)) ---------------------
@pytest.mark.parametrize("mode", ["RGB+D", "RGB"])
def test_post_processing_inside_the_compositing_kernels_equals_the_standalone_node(cuda, mode):
    """model.py:295-297 / 304-306 folded into the compositing forward (rgb, depth written beside render; one fix-up pass
    for the empty pixels' depth) and backward (v_render / v_alpha derived from v_rgb / v_depth in the tile prologue) against
    the stand-alone node _PostProcess on the plain operator's outputs: images equal, gradients equal up to the order of the
    atomics -- with saturated colours on both sides of the clamp, a coloured background and empty pixels."""
    from qed_splatter_amd.model import _PostProcess, get_viewmat
    from qed_splatter_amd.rasterization import rasterization
    w, h, n = 176, 120, 2500
    sc = scene(n, w, h, seed=12)
    sc["features_dc"] = sc["features_dc"] * 3.0                       # many colours beyond [0, 1] before the clamp
    sc["scales"] = sc["scales"] + 1.5                                 # (large, fairly opaque splats: most covered pixels saturate)
    sc["opacities"] = sc["opacities"] + 2.0
    sc["means"][:, 0] = sc["means"][:, 0].abs()                       # the left half of the image stays empty (alpha = 0)
    bg = torch.tensor([0.9, 0.2, 0.6], device=cuda)
    g = torch.Generator().manual_seed(8)
    w_rgb = torch.randn(1, h, w, 3, generator=g).to(cuda)
    w_d = torch.randn(1, h, w, 1, generator=g).to(cuda)
    w_r = torch.randn(1, h, w, 4 if mode == "RGB+D" else 3, generator=g).to(cuda)
    vm = get_viewmat(sc["camera_to_worlds"][:1].to(cuda))

    def run(fused, also_render=False):
        ps = {k: sc[k].to(cuda).requires_grad_(True) for k in PARAM_NAMES}
        render, alpha, info = rasterization(
            means=ps["means"], quats=torch.nn.functional.normalize(ps["quats"], dim=-1), scales=ps["scales"].exp(),
            opacities=torch.sigmoid(ps["opacities"]).squeeze(-1),
            colors=torch.cat([ps["features_dc"][:, None, :], ps["features_rest"]], dim=1), viewmats=vm,
            Ks=sc["Ks"][:1].to(cuda), width=w, height=h, render_mode=mode, sh_degree=3, absgrad=True,
            _post_background=bg if fused else None)
        if fused:
            rgb, depth = info["post_rgb"], info["post_depth"]
        elif mode == "RGB+D":
            rgb, depth = _PostProcess.apply(render, alpha, bg)
        else:
            rgb, depth = _PostProcess.apply(render, alpha, bg), None
        loss = (rgb * w_rgb).sum()
        if depth is not None:
            loss = loss + (depth * w_d).sum()
        if also_render:
            loss = loss + (render * w_r).sum() + alpha.sum()
        grads = torch.autograd.grad(loss, [ps[k] for k in PARAM_NAMES])
        return rgb.detach(), None if depth is None else depth.detach(), alpha.detach(), grads

    rgb0, d0, a0, g0 = run(False)
    rgb1, d1, a1, g1 = run(True)
    assert float((a0 == 0).float().mean()) > 0.2                       # empty pixels exist: they show the max depth
    assert float(((rgb0 == 0) | (rgb0 == 1)).float().mean()) > 0.05    # ... and saturated colours the clamp mask
    assert torch.equal(a1, a0) and torch.equal(rgb1, rgb0)
    if mode == "RGB+D":
        assert torch.equal(d1, d0) and float(d1[a1 == 0].min()) == float(d1.max())
    else:
        assert d1 is None
    for k, x, y in zip(PARAM_NAMES, g1, g0):
        assert_close(x, y, 2e-5, f"fused vs stand-alone post-processing: grad {k}")
    # rgb / depth AND render / alpha used downstream: both gradient routes are added
    _, _, _, g2 = run(False, also_render=True)
    _, _, _, g3 = run(True, also_render=True)
    for k, x, y in zip(PARAM_NAMES, g3, g2):
        assert_close(x, y, 2e-5, f"mixed use: grad {k}")


def test_flat_adam_state_dict_round_trip(cuda, tmp_path):
    """The fused route's optimiser checkpoints like any other (config.py:29 steps_per_save): state_dict -> torch.save ->
    a fresh model + optimiser -> load_state_dict continues exactly where the first left off (moments, step counts, the
    device-resident step state and rates a captured graph replays against, the scheduled rate)."""
    from qed_splatter_amd.model import FlatAdam
    sc = scene(1500, 96, 64, seed=6)

    def step(m, cam, batch, opt, device_state):
        for p in m.parameters():
            p.grad = None
        m.backward_fused(m.fused_loss(cam, batch, compact_sh_grad=True))
        opt.step(device_state=device_state, fused_sh=True)

    for device_state in (False, True):
        m1, cam, batch = _model(sc, cuda)
        o1 = FlatAdam(m1, means_schedule=(1.6e-6, 50))
        for _ in range(3):
            step(m1, cam, batch, o1, device_state)
        path = tmp_path / f"ckpt{int(device_state)}.pt"
        torch.save({"model": m1.state_dict(), "optim": o1.state_dict()}, path)
        ck = torch.load(path)
        m2, cam2, batch2 = _model(sc, cuda)
        m2.load_state_dict(ck["model"])
        assert m2.gauss_params["means"].data_ptr() == m2.flat_params.data_ptr()      # still views of the flat buffer
        o2 = FlatAdam(m2)
        o2.load_state_dict(ck["optim"])
        assert o2.t == 3 and o2.means_schedule == (1.6e-6, 50)
        assert torch.equal(o2.exp_avg, o1.exp_avg) and torch.equal(o2.dev_state, o1.dev_state)
        # the same gradients into both: the next update is bit-identical
        g = torch.randn_like(m1.flat_params)
        for m, o in ((m1, o1), (m2, o2)):
            off = 0
            for name in m.group_names:
                p = m.gauss_params[name]
                p.grad = g[off:off + p.numel()].view(p.shape)
                off += p.numel()
            m.last_compact = False
            o.step(device_state=device_state)
        assert torch.equal(m1.flat_params, m2.flat_params) and torch.equal(o1.exp_avg_sq, o2.exp_avg_sq)
    with pytest.raises(ValueError, match="load the model"):
        m3, _, _ = _model(scene(700, 96, 64, seed=6), cuda)
        FlatAdam(m3).load_state_dict(ck["optim"])


def test_qedadam_group_without_a_gradient_does_not_delay_the_others(cuda):
    """The six QedAdam instances of a flat buffer launch one fused update when the LAST of them is stepped.  A group whose
    .grad is None in some iteration (or whose step a GradScaler skipped) must not hold the others' update back into the next
    iteration's gradients: zero_grad() -- and state_dict(), flush() -- launch what is waiting, with THIS iteration's
    gradients, exactly as six torch.optim.Adam would have stepped."""
    from qed_splatter_amd.model import QedAdam, QEDSplatterModel, QEDSplatterModelConfig
    n = 800
    sc = scene(n, 64, 48, seed=13)
    lrs = {"means": 1.6e-4, "scales": 0.005, "quats": 0.001, "opacities": 0.05, "features_dc": 0.0025, "features_rest": 1.25e-4}
    cfg = QEDSplatterModelConfig.synthetic()
    mq = QEDSplatterModel(cfg, **{k: sc[k].to(cuda) for k in PARAM_NAMES})
    mt = QEDSplatterModel(cfg, separate_params=True, **{k: sc[k].to(cuda) for k in PARAM_NAMES})
    oq = {k: QedAdam([mq.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in mq.group_names}
    ot = {k: torch.optim.Adam([mt.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in mt.group_names}
    g = torch.Generator().manual_seed(1)
    for it in range(3):
        grads = {k: torch.randn(mq.gauss_params[k].shape, generator=g).to(cuda) for k in PARAM_NAMES}
        skip = "quats" if it == 1 else None                       # iteration 1: the quats group has no gradient
        for m, opts in ((mq, oq), (mt, ot)):
            for k in PARAM_NAMES:
                m.gauss_params[k].grad = None if k == skip else grads[k].clone()
            for k in PARAM_NAMES:
                opts[k].step()
        if it == 1:
            # five groups are waiting for the sixth; the trainer's zero_grad() at the top of the next iteration launches them
            assert len(oq["means"]._shared.pending) == 5
        for opts in (oq, ot):
            for o in opts.values():
                o.zero_grad()
        assert len(oq["means"]._shared.pending) == 0
    for k in PARAM_NAMES:
        assert_close_elem(mq.gauss_params[k], mt.gauss_params[k], f"{k} after an iteration without a quats gradient", atol_frac=1e-6)
    assert float(oq["quats"].state[mq.gauss_params["quats"]]["step"]) == 2.0 == float(ot["quats"].state[mt.gauss_params["quats"]]["step"])
    # state_dict() flushes as well
    for k in PARAM_NAMES:
        mq.gauss_params[k].grad = torch.randn(mq.gauss_params[k].shape, generator=g).to(cuda)
    for k in PARAM_NAMES[:3]:
        oq[k].step()
    assert len(oq["means"]._shared.pending) == 3
    sd = oq["means"].state_dict()
    assert len(oq["means"]._shared.pending) == 0 and float(sd["state"][0]["step"]) == 4.0


# ---- the eval-time crop box (model.py:217-239) ----------------------------------------------------------------------
class _Box:
    """Stands in for Nerfstudio's OrientedBox: ``within(points) -> [N,1] bool`` (the reference squeezes it, model.py:218)."""

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def within(self, pts):
        return ((pts[:, 2] >= self.lo) & (pts[:, 2] < self.hi))[:, None]


def test_eval_crop_box_renders_the_kept_subset_and_none_kept_gives_the_empty_outputs(cuda):
    """get_outputs in eval mode with a crop box (model.py:217-239): the six groups are indexed by the box's mask and only
    the kept Gaussians are rendered -- outputs equal the oracle's on that subset, `radii` / `xys` have the subset's length;
    in training mode the box is ignored (model.py:217 `and not self.training`); a box that keeps nothing returns
    get_empty_outputs (background colour image, depth 10, zero accumulation, model.py:219-222)."""
    from tests.test_gpu_parity import MARGIN_E2E
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=41)
    m, cam, batch = _model(sc, cuda, background_color="white")
    z = sc["means"][:, 2]
    cut = float(z.sort().values[int(0.6 * n)])                       # keeps the ~40 % nearest the camera (z in (-12, -2))
    m.crop_box = _Box(cut, 0.0)
    keep = ((z >= cut) & (z < 0.0))
    n_kept = int(keep.sum())
    assert 0.35 * n < n_kept < 0.45 * n
    m.eval()
    with torch.no_grad():
        out = m.get_outputs(cam)
    assert m.radii.shape == (n_kept,) and m.xys.shape == (1, n_kept, 2)            # "radii: [N]" of the cropped set
    assert out["rgb"].shape == (h, w, 3) and out["depth"].shape == (h, w, 1) and out["accumulation"].shape == (h, w, 1)
    assert out["background"].shape == (h, w, 3)                                    # expanded in eval (model.py:313-314)
    sub = {k: sc[k][keep].double() for k in PARAM_NAMES}
    bg = torch.ones(3, dtype=torch.float64)
    ref = O.splatfacto_outputs(sub["means"], sub["scales"], sub["quats"], sub["opacities"], sub["features_dc"],
                               sub["features_rest"], sc["camera_to_worlds"][:1].double(), sc["Ks"][:1].double(), w, h, bg,
                               radii_override=m.info["radii"].cpu(), return_margin=True)
    safe = ref["info"]["margin"][0] > MARGIN_E2E
    assert float(safe.float().mean()) > 0.999
    assert_close(out["rgb"].cpu()[safe], ref["rgb"][safe], REL_TOL, "rgb (cropped)")
    assert_close(out["depth"].cpu()[safe], ref["depth"][safe], REL_TOL, "depth (cropped)")
    assert_close(out["accumulation"].cpu()[safe], ref["accumulation"][safe], REL_TOL, "accumulation (cropped)")
    # the crop really removed something: the full set renders a different image
    m.crop_box = None
    with torch.no_grad():
        full = m.get_outputs(cam)
    assert m.radii.shape == (n,)
    assert float((full["accumulation"] - out["accumulation"]).abs().max()) > 0.05
    # training ignores the box
    m.crop_box = _Box(cut, 0.0)
    m.train()
    out_t = m.get_outputs(cam)
    assert m.radii.shape == (n,)
    assert torch.equal(out_t["accumulation"].detach(), full["accumulation"])
    # a box that keeps nothing: the parent's get_empty_outputs
    m.eval()
    m.crop_box = _Box(5.0, 6.0)
    with torch.no_grad():
        empty = m.get_outputs(cam)
    assert set(empty) == {"rgb", "depth", "accumulation", "background"}
    assert empty["rgb"].shape == (h, w, 3) and empty["depth"].shape == (h, w, 1) and empty["accumulation"].shape == (h, w, 1)
    assert bool((empty["rgb"] == 1.0).all()) and bool((empty["depth"] == 10.0).all()) and bool((empty["accumulation"] == 0.0).all())
    assert torch.equal(empty["background"], torch.ones(3, device=cuda))


# ---- the per-step context (StepContext): calls out of order must still give the right values ------------------------
def test_step_context_misuse_gives_correct_values_each_time(cuda):
    """get_outputs leaves a StepContext that get_metrics_dict / get_loss_dict / backward share (an SSIM forward, the batch
    conversions, a zero-filled accumulator for the compositing backward).  Every consumer must fall back to the full
    computation when the context is not its own: (B) two losses on one `outputs`; (C) a loss on the PREVIOUS step's
    outputs after a new get_outputs, then the loss of the new ones; (D) metrics taken under no_grad, then a training
    loss; (E) metrics of stale outputs must not reach the fresh outputs' loss.  Values and gradients are compared with a
    fresh model that makes the plain sequence of calls."""
    from qed_splatter_amd.model import PinholeCameras
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=12, n_cameras=2)
    K = sc["Ks"][0]

    def fresh():
        m, cam, batch = _model(sc, cuda)
        m.train()
        cam2 = PinholeCameras(sc["camera_to_worlds"][1:2].to(cuda), K[0, 0], K[1, 1], K[0, 2], K[1, 2], w, h)
        return m, cam, cam2, batch

    def grads(m):
        return {k: m.gauss_params[k].grad.clone() for k in PARAM_NAMES}

    def same(ld, ref):
        return all(float(ld[k]) == pytest.approx(float(ref[k]), rel=3e-6, abs=1e-9) for k in ref)

    def close(g, ref, scale=1.0):
        for k in PARAM_NAMES:
            assert_close(g[k], scale * ref[k].double().cpu(), 2e-5, f"grad {k}")

    # (A) the plain sequence, per camera
    ref = {}
    for which in (0, 1):
        m, cam, cam2, batch = fresh()
        out = m.get_outputs((cam, cam2)[which])
        ld = m.get_loss_dict(out, batch)
        sum(ld.values()).backward()
        ref[which] = ({k: v.detach().clone() for k, v in ld.items()}, grads(m))
    # (B) metrics, then the loss twice on the same outputs
    m, cam, cam2, batch = fresh()
    out = m.get_outputs(cam)
    md = m.get_metrics_dict(out, batch)
    ld1 = m.get_loss_dict(out, batch, md)
    ld2 = m.get_loss_dict(out, batch, md)
    assert same(ld1, ref[0][0]) and same(ld2, ref[0][0])
    (sum(ld1.values()) + sum(ld2.values())).backward()
    close(grads(m), ref[0][1], 2.0)
    # (C) + (E): stale outputs
    m, cam, cam2, batch = fresh()
    out_old = m.get_outputs(cam)
    out_new = m.get_outputs(cam2)
    m.get_metrics_dict(out_old, batch)                     # (E) stale metrics: nothing of them may be kept
    assert m._step.ssim is None
    md_new = m.get_metrics_dict(out_new, batch)
    assert m._step.ssim is not None
    ld_old = m.get_loss_dict(out_old, batch)               # (C) the previous step's outputs: the context is not theirs
    assert m._step.ssim is not None                        #     ... and they did not consume the new outputs' SSIM
    ld_new = m.get_loss_dict(out_new, batch, md_new)
    assert m._step.ssim is None
    assert same(ld_old, ref[0][0]) and same(ld_new, ref[1][0])
    sum(ld_old.values()).backward()
    close(grads(m), ref[0][1])
    for p in m.parameters():
        p.grad = None
    sum(ld_new.values()).backward()
    close(grads(m), ref[1][1])
    # (D) metrics under no_grad, then a training loss
    m, cam, cam2, batch = fresh()
    out = m.get_outputs(cam)
    with torch.no_grad():
        m.get_metrics_dict(out, batch)
    assert m._step.ssim is None                            # an SSIM forward without the maps is of no use to the loss
    ld = m.get_loss_dict(out, batch)
    assert same(ld, ref[0][0])
    sum(ld.values()).backward()
    close(grads(m), ref[0][1])


# ---- get_outputs as captured hipGraphs behind one autograd node (segments.py) ---------------------------------------
def _reference_sequence(m, cam, batch, opts, sched=None):
    """One iteration of the trainer's call sequence (bench.api_path_ms)."""
    import functools
    for o in opts.values():
        o.zero_grad(set_to_none=True)
    out = m.get_outputs(cam)
    md = m.get_metrics_dict(out, batch)
    ld = m.get_loss_dict(out, batch, md)
    functools.reduce(torch.add, ld.values()).backward()
    for o in opts.values():
        o.step()
    return out, ld


@pytest.mark.parametrize("background,separate", [("black", False), ("random", False), ("black", True)])
def test_graphed_get_outputs_trains_like_the_eager_route(cuda, background, separate):
    """config.graph_segments: after a few eager calls of a shape, get_outputs replays a captured forward graph and its
    backward replays a captured backward graph (projection / binning / K6 | ordering / K7 / projection backward).  Same
    kernels on the same data: losses, xys.grad / absgrad and the parameters after ten steps equal the eager route's up to
    the order of the float atomics.  Outputs of an earlier step are refused once overwritten."""
    from qed_splatter_amd import rasterization as R
    from qed_splatter_amd.model import FlatAdam, QedAdam
    w, h, n = 200, 136, 6000
    sc = scene(n, w, h, seed=23)
    runs = {}
    for graphed in (False, True):
        R._WORKSPACES.clear()
        torch.manual_seed(5)                                  # the "random" training background draws from the global generator
        m, cam, batch = _model(sc, cuda, background_color=background, graph_segments="always" if graphed else False)
        if separate:                                          # six tensors that own their storage, as Nerfstudio's parent holds them
            from qed_splatter_amd.model import QEDSplatterModel
            m = QEDSplatterModel(m.config, separate_params=True, **{k: sc[k].to(cuda) for k in PARAM_NAMES})
            m.step = 100
        m.train()
        lrs = FlatAdam.DEFAULT_LRS
        opts = {k: QedAdam([m.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in PARAM_NAMES}
        losses, outs = [], None
        for step in range(10):
            out, ld = _reference_sequence(m, cam, batch, opts)
            losses.append(torch.stack([v.detach() for v in ld.values()]))
            if step == 5:
                kept = out                                    # an earlier step's outputs, used too late below
        torch.cuda.synchronize()
        cache = m.__dict__.get("_segments")
        assert (cache is not None and len(cache.segments) == 1 and cache.disabled is None) == graphed
        flat = torch.cat([m.gauss_params[k].detach().reshape(-1) for k in m.group_names])
        runs[graphed] = (torch.stack(losses).cpu(), flat, m.xys.grad.clone(), m.xys.absgrad.clone(),
                         m.radii.clone(), out["rgb"].detach().clone())
        if graphed:
            with pytest.raises(RuntimeError, match="overwritten"):
                m.get_loss_dict(kept, batch)
            assert m.info["means2d"] is m.xys and m.xys.grad is not None
    (l0, p0, g0, a0, r0, i0), (l1, p1, g1, a1, r1, i1) = runs[False], runs[True]
    assert torch.equal(r0, r1)
    assert_close(l1, l0.double(), 2e-5, "losses over ten steps")
    assert_close(i1, i0.double().cpu(), 2e-5, "rgb of the tenth step")
    assert_close(g1, g0.double().cpu(), 2e-4, "xys.grad")
    assert_close(a1, a0.double().cpu(), 2e-4, "xys.absgrad")
    assert float((p1 - p0).abs().max()) <= 2e-3 * float(p0.abs().max())       # ten Adam steps amplify the last bits
    assert float((p1 - p0).abs().mean()) <= 1e-5 * float(p0.abs().max())


def test_graphed_get_outputs_follows_shape_changes_and_other_losses(cuda):
    """What a capture was specialised on is part of its key: another resolution / SH degree runs eagerly until it has been
    seen often enough; a loss that is not _ImageLosses (its gradients arrive in tensors of its own, and `accumulation` gets
    one too) goes through the copy-in path and the general backward graph -- against the eager route's gradients."""
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=29)
    g = torch.Generator().manual_seed(3)
    w_rgb = torch.rand(h, w, 3, generator=g).to(cuda)
    w_acc = torch.rand(h, w, 1, generator=g).to(cuda)
    w_d = torch.rand(h, w, 1, generator=g).to(cuda)
    grads = {}
    for graphed in (False, True):
        m, cam, batch = _model(sc, cuda, graph_segments="always" if graphed else False)
        m.train()
        for it in range(6):
            for p in m.parameters():
                p.grad = None
            out = m.get_outputs(cam)
            loss = (out["rgb"] * w_rgb).sum() + (out["accumulation"] * w_acc).sum() + (out["depth"] * w_d).sum()
            loss.backward()
        grads[graphed] = {k: m.gauss_params[k].grad.clone() for k in PARAM_NAMES}
        if graphed:
            seg = next(iter(m._segments.segments.values()))
            assert set(seg._bwd) == {False, True}            # the usual graph (captured with the segment) + the general one
            m.step = 0                                        # SH degree 0 in use: another key -> eager, then its own capture
            for it in range(5):
                out = m.get_outputs(cam)
                (out["rgb"] * w_rgb).sum().backward()
            assert len(m._segments.segments) == 2
    for k in PARAM_NAMES:
        assert_close(grads[True][k], grads[False][k].double().cpu(), 2e-5, f"grad {k} (custom loss through the segment)")


@pytest.mark.parametrize("training", [False, True])
def test_reference_get_metrics_dict_kats_keys_values_and_writer_shape(cuda, training):
    """get_metrics_dict against the reference's own (model.py:120-197, d = 1 branch, executed by
    tests/golden/make_reference_kats.py): the same KEYS in the same order, rgb_mse on the first three channels of an RGBA
    ground truth, the seven depth metrics, avg_min_scale (nanmean), gaussian_count -- with and without a depth image in
    the batch.  The mirror returns 0-dim device tensors where the reference returns Python floats (a deliberate
    deviation: no sync per entry); what a writer does with them must work: ``float(v)`` on every entry and a JSON dump."""
    import json
    from qed_splatter_amd.model import QEDSplatterModel, QEDSplatterModelConfig
    k = np.load(os.path.join(GOLD, "reference_kats.npz"))
    scales = torch.from_numpy(k["md_scales"])
    n = scales.shape[0]
    g = torch.Generator().manual_seed(1)
    m = QEDSplatterModel(QEDSplatterModelConfig.synthetic(), means=torch.randn(n, 3, generator=g).to(cuda),
                         scales=scales.to(cuda), quats=torch.randn(n, 4, generator=g).to(cuda),
                         opacities=torch.zeros(n, 1).to(cuda), features_dc=torch.zeros(n, 3).to(cuda),
                         features_rest=torch.zeros(n, 15, 3).to(cuda))
    m.train(training)
    outputs = {"rgb": torch.from_numpy(k["md_rgb"]).to(cuda), "depth": torch.from_numpy(k["md_depth"]).to(cuda)}
    image = torch.from_numpy(k["md_image"]).to(cuda)
    for tag, batch in (("md", {"image": image, "depth_image": torch.from_numpy(k["md_gt_depth"]).to(cuda)}),
                       ("mdn", {"image": image})):
        got = m.get_metrics_dict(outputs, batch)
        assert list(got.keys()) == json.loads(str(k[f"{tag}_keys"])), tag
        as_floats = {key: float(v) for key, v in got.items()}            # what a writer does with every entry
        json.dumps(as_floats)
        assert as_floats["gaussian_count"] == float(k[f"{tag}_gaussian_count"]) == n
        for key in got:
            if key in ("rgb_psnr", "rgb_ssim", "rgb_lpips", "gaussian_count"):   # torchmetrics' / no weights offline
                continue
            assert as_floats[key] == pytest.approx(float(k[f"{tag}_{key}"]), rel=2e-5), (tag, key)
        # PSNR follows from the pinned MSE (data range 1)
        assert as_floats["rgb_psnr"] == pytest.approx(-10.0 * math.log10(float(k[f"{tag}_rgb_mse"])), rel=1e-5)
        assert -1.0 <= as_floats["rgb_ssim"] <= 1.0 and math.isnan(as_floats["rgb_lpips"])    # (random images: SSIM ~ 0)


@pytest.mark.parametrize("how", ["accumulate", "zero_in_place", "retain_graph"])
def test_graphed_backward_adds_to_gradients_that_wait_in_the_fields(cuda, how):
    """The captured backward pass hands autograd ALIASES of its static gradient buffers, which the engine adopts as .grad
    without a copy.  A second backward pass while those are still set (gradient accumulation over two get_outputs calls,
    zero_grad(set_to_none=False), a second backward on retained outputs) must add to the OLD values: the waiting gradients
    are moved out of the static buffers before the replay overwrites them.  Against the eager route, which allocates."""
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=31)
    g = torch.Generator().manual_seed(4)
    w1, w2 = torch.rand(h, w, 3, generator=g).to(cuda), torch.rand(h, w, 3, generator=g).to(cuda)
    got = {}
    for graphed in (False, True):
        m, cam, batch = _model(sc, cuda, graph_segments="always" if graphed else False)
        m.train()
        for it in range(6):                                   # (the shape is captured on the fourth call)
            for p in m.parameters():
                p.grad = None
            (m.get_outputs(cam)["rgb"] * w1).sum().backward()
        if graphed:
            assert len(m._segments.segments) == 1
        if how == "accumulate":                               # two forward / backward passes, no zero_grad in between
            for p in m.parameters():
                p.grad = None
            (m.get_outputs(cam)["rgb"] * w1).sum().backward()
            (m.get_outputs(cam)["rgb"] * w2).sum().backward()
        elif how == "zero_in_place":                          # the fields keep (zeroed) tensors: still the static buffers
            for p in m.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            (m.get_outputs(cam)["rgb"] * w2).sum().backward()
        else:                                                 # two backward passes through ONE step's outputs
            for p in m.parameters():
                p.grad = None
            out = m.get_outputs(cam)
            (out["rgb"] * w1).sum().backward(retain_graph=True)
            (out["rgb"] * w2).sum().backward()
        got[graphed] = {k: m.gauss_params[k].grad.detach().clone() for k in PARAM_NAMES}
    for k in PARAM_NAMES:
        assert_close(got[True][k], got[False][k].double().cpu(), 2e-5, f"grad {k} ({how}, through the segment)")


# ---- SH gradients kept compact until somebody reads them (config.lazy_sh_grad) --------------------------------------
def _six_qed_adams(m):
    from qed_splatter_amd.model import FlatAdam, QedAdam
    return {k: QedAdam([m.gauss_params[k]], lr=FlatAdam.DEFAULT_LRS[k], eps=1e-15) for k in
            ("means", "features_dc", "features_rest", "opacities", "scales", "quats")}        # config.py's order


def _flat(m):
    return torch.cat([m.gauss_params[k].detach().reshape(-1) for k in m.group_names])


def _same_training(x1, x0, what):
    """Two runs of the same steps: equal up to what the run-to-run order of K7's float atomics does to an early Adam step
    (an update of +-lr wherever a gradient sits at the noise floor): loose on the largest element, tight on the mean."""
    d, ref = (x1 - x0).abs(), float(x0.abs().max())
    assert float(d.max()) <= 2e-3 * ref and float(d.mean()) <= 1e-5 * ref, \
        f"{what}: max {float(d.max()) / ref:.2e}, mean {float(d.mean()) / ref:.2e} of the largest element"


@pytest.mark.parametrize("graphed,deg", [(False, 3), (True, 3), (False, 0), (False, 1), (True, 2)])
def test_lazy_sh_gradients_train_like_written_out_ones(cuda, graphed, deg):
    """With six QedAdam instances on the flat buffer the backward pass of the reference-shaped route leaves the SH gradients
    in the fused step's compact form (the projection backward neither writes nor the optimiser reads 48 N floats): from the
    second step on (the first registers the optimisers) the two ``.grad`` fields are consumed by the step and read None
    afterwards; parameters and moments after eight steps equal those of config.lazy_sh_grad = False."""
    from qed_splatter_amd import rasterization as R
    from qed_splatter_amd.model import _raw_grad
    w, h, n = 200, 136, 6000
    sc = scene(n, w, h, seed=31)
    runs = {}
    for lazy in (False, True):
        R._WORKSPACES.clear()
        m, cam, batch = _model(sc, cuda, step=deg, lazy_sh_grad=lazy, graph_segments="always" if graphed else False)
        m.train()                                               # (sh_degree_interval = 1: the active SH degree is `deg`)
        opts = _six_qed_adams(m)
        consumed = 0
        for step in range(8):
            _reference_sequence(m, cam, batch, opts)
            dc, rest = m.gauss_params["features_dc"], m.gauss_params["features_rest"]
            if _raw_grad(dc) is None and _raw_grad(rest) is None:
                consumed += 1
                assert dc.grad is None and rest.grad is None and m.means.grad is not None
        torch.cuda.synchronize()
        assert consumed == (7 if lazy else 0)                   # (the first step runs before any optimiser has registered)
        if graphed:
            cache = m.__dict__["_segments"]
            assert len(cache.segments) >= 1 and cache.disabled is None
        st = opts["means"]._shared
        runs[lazy] = (_flat(m), st.exp_avg.clone(), st.exp_avg_sq.clone())
    b = m.group_begin
    # (eight Adam steps amplify the last bits of the gradients -- the order of K7's float atomics differs from run to run --
    # so the bound on the largest element is loose and the one on the mean tight; a wrong SH gradient is off by O(1))
    for name, x1, x0 in zip(("params", "exp_avg", "exp_avg_sq"), runs[True], runs[False]):
        for part, lo, hi in (("geometry", 0, b[4]), ("features_dc", b[4], b[5]), ("features_rest", b[5], b[6])):
            d, ref = (x1[lo:hi] - x0[lo:hi]).abs(), float(x0[lo:hi].abs().max())
            assert float(d.max()) <= 2e-3 * ref and float(d.mean()) <= 1e-5 * ref, \
                f"{part} {name}: max {float(d.max()) / ref:.2e}, mean {float(d.mean()) / ref:.2e} of the largest element"
    assert bool((runs[True][1][b[5]:] != 0).any()) == (deg > 0)    # (degree 0: no coefficient beyond features_dc is active)


def test_lazy_sh_gradients_read_as_the_full_gradients(cuda):
    """Every Python reader of ``.grad`` sees the reference's gradients: reading either field between backward and step
    writes the coefficient gradients in place (then the step is the plain one); two backward passes without zero_grad add
    up; zero_grad drops an unread compact form; a group stepped out of turn, or torch.optim.Adam on some group, never meets
    the compact form."""
    from qed_splatter_amd import rasterization as R
    from qed_splatter_amd.model import QedAdam, _raw_grad
    w, h, n = 160, 112, 3000
    sc = scene(n, w, h, seed=32)

    def one_backward(m, cam, batch):
        out = m.get_outputs(cam)
        ld = m.get_loss_dict(out, batch, m.get_metrics_dict(out, batch))
        sum(ld.values()).backward()

    def prepared(lazy):
        R._WORKSPACES.clear()
        m, cam, batch = _model(sc, cuda, lazy_sh_grad=lazy, graph_segments=False)
        m.train()
        opts = _six_qed_adams(m)
        _reference_sequence(m, cam, batch, opts)                 # registers the optimisers (and moves the parameters)
        for o in opts.values():
            o.zero_grad(set_to_none=True)
        return m, cam, batch, opts

    m0, cam, batch, opts0 = prepared(False)
    one_backward(m0, cam, batch)
    want = {k: m0.gauss_params[k].grad.clone() for k in m0.group_names}

    # (1) a read between backward and step
    m1, cam, batch, opts1 = prepared(True)
    one_backward(m1, cam, batch)
    assert m1.__dict__["_lazy_sh"] is not None                   # compact: nobody has asked yet
    got_rest = m1.gauss_params["features_rest"].grad             # ... now somebody has
    assert m1.__dict__["_lazy_sh"] is None
    assert_close(got_rest, want["features_rest"].double(), 1e-5, "features_rest.grad (materialised)")
    assert_close(m1.gauss_params["features_dc"].grad, want["features_dc"].double(), 1e-5, "features_dc.grad (materialised)")
    assert_close(m1.means.grad, want["means"].double(), 1e-5, "means.grad")
    for o in opts0.values():
        o.step()
    for o in opts1.values():
        o.step()
    _same_training(_flat(m1), _flat(m0), "parameters after a step on materialised gradients")
    assert m1.gauss_params["features_rest"].grad is not None     # (a plain step leaves the fields alone)

    # (2) accumulation over two backward passes
    for o in (*opts0.values(), *opts1.values()):
        o.zero_grad(set_to_none=True)
    one_backward(m0, cam, batch)
    one_backward(m0, cam, batch)
    one_backward(m1, cam, batch)
    assert m1.__dict__["_lazy_sh"] is not None
    one_backward(m1, cam, batch)                                 # completes the first pass's gradients, then adds its own
    assert m1.__dict__["_lazy_sh"] is None
    for k in ("features_dc", "features_rest", "means"):
        assert_close(m1.gauss_params[k].grad, m0.gauss_params[k].grad.double(), 2e-5, f"{k}.grad summed over two passes")

    # (3) zero_grad drops a compact form nobody read; model.zero_grad() (nn.Module's) too
    for o in opts1.values():
        o.zero_grad(set_to_none=True)
    one_backward(m1, cam, batch)
    assert m1.__dict__["_lazy_sh"] is not None
    for o in opts1.values():
        o.zero_grad(set_to_none=True)
    assert m1.__dict__["_lazy_sh"] is None and _raw_grad(m1.gauss_params["features_dc"]) is None
    one_backward(m1, cam, batch)
    m1.zero_grad()
    assert m1.__dict__["_lazy_sh"] is None and all(p.grad is None for p in m1.parameters())

    # (4) a group stepped out of turn: the waiting groups are launched alone -> plain gradients first
    for o in (*opts0.values(), *opts1.values()):
        o.zero_grad(set_to_none=True)
    one_backward(m0, cam, batch)
    one_backward(m1, cam, batch)
    for opts in (opts0, opts1):
        opts["means"].step()
        opts["means"].step()                                     # second call: launches the first alone
        for k in ("features_dc", "features_rest", "opacities", "scales", "quats"):
            opts[k].step()
        opts["means"].flush()
    _same_training(_flat(m1), _flat(m0), "parameters after an out-of-turn step")

    # (5) torch.optim.Adam on one group: fewer than six QedAdam members -> the backward pass writes full gradients
    R._WORKSPACES.clear()
    m2, cam, batch = _model(sc, cuda, lazy_sh_grad=True, graph_segments=False)
    m2.train()
    opts2 = _six_qed_adams(m2)
    del opts2["features_rest"]
    import gc
    gc.collect()
    opts2["features_rest"] = torch.optim.Adam([m2.gauss_params["features_rest"]], lr=1e-3, eps=1e-15)
    for _ in range(3):
        _reference_sequence(m2, cam, batch, opts2)
        assert m2.__dict__.get("_lazy_sh") is None and _raw_grad(m2.gauss_params["features_rest"]) is not None


def test_camera_index_keeps_a_launch_order_per_camera_and_changes_no_result(cuda):
    """The reference's trainer hands the camera index in camera.metadata["cam_idx"]: on the eager route the model keeps one
    launch-order slot per camera -- the compositing backward writes its costliest-first order into it, the next frame of
    that camera hands it to the compositing forward.  A scheduling hint: outputs, losses and gradients are those of a
    camera without an index."""
    sc = scene(6000, 320, 208, seed=12)
    m, cam, batch = _model(sc, cuda, graph_segments=False)
    m.train()

    def step(camera):
        for p in m.parameters():
            p.grad = None
        out = m.get_outputs(camera)
        ld = m.get_loss_dict(out, batch)
        (ld["main_loss"] + ld["depth_loss"]).backward()
        return out["rgb"].detach().clone(), float(ld["main_loss"]), {k: m.gauss_params[k].grad.detach().clone() for k in PARAM_NAMES}

    rgb0, l0, g0 = step(cam)
    assert not m.__dict__.get("_frame_orders")
    cam.metadata = {"cam_idx": 3}
    T = ((320 + 15) // 16) * ((208 + 15) // 16)
    for rep in range(3):
        rgb1, l1, g1 = step(cam)
        slot = m._frame_orders[(3, 208, 320)]
        assert slot[1] and sorted(slot[0][:T].cpu().tolist()) == list(range(T))       # written by the backward pass
        assert torch.equal(rgb1, rgb0) and l1 == l0                                   # (rep >= 1: the forward pass used it)
        for k in PARAM_NAMES:
            assert_close(g1[k], g0[k], 2e-5, f"rep {rep}: grad {k}")
