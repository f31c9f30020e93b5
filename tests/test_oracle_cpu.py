"""CPU tests (no GPU): the oracle against the reference's own known answers and the committed
golden vectors; fp64 gradcheck; structural properties of the algorithm; the product's host-side
(pure torch) mirrors of the reference functions."""
from __future__ import annotations

import math
import os

import numpy as np
import pytest
import torch

from oracle import splat_oracle as O
from tests.util import PARAM_NAMES, activated, scene

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kats():
    return np.load(os.path.join(GOLD, "reference_kats.npz"))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "oracle_small.npz"))


# ---- reference-owned arithmetic: pinned by vectors generated from the reference itself ------------
def test_get_viewmat_matches_reference(kats):
    c2w = torch.from_numpy(kats["viewmat_c2w"])
    want = torch.from_numpy(kats["viewmat_out"])
    assert torch.equal(O.get_viewmat(c2w), want)
    from qed_splatter_amd.model import get_viewmat            # product mirror (host-side torch)
    assert torch.equal(get_viewmat(c2w), want)
    # SURVEY 8c known answer: c2w = [I | (1,2,3)] -> [[1,0,0,-1],[0,-1,0,2],[0,0,-1,3],[0,0,0,1]]
    ka = torch.tensor([[1.0, 0, 0, -1], [0, -1, 0, 2], [0, 0, -1, 3], [0, 0, 0, 1]])
    assert torch.equal(O.get_viewmat(c2w[-1:])[0], ka)


def test_depth_l1_matches_reference(kats):
    for i in kats["dl_cases"]:
        d_out = torch.from_numpy(kats[f"dl{i}_depth_out"])
        d_gt = torch.from_numpy(kats[f"dl{i}_depth_gt"])
        m = kats[f"dl{i}_mask"]
        mask = torch.from_numpy(m) if m.size else None
        lam = float(kats[f"dl{i}_lambda"])
        got = float(O.depth_l1_loss(d_out, d_gt, mask, lam))
        assert got == pytest.approx(float(kats[f"dl{i}_loss"]), rel=1e-6, abs=1e-9), f"case {i}"
    got = float(O.depth_l1_loss(torch.from_numpy(kats["survey_depth_out"]), torch.from_numpy(kats["survey_depth_gt"])))
    assert got == pytest.approx(0.0653, abs=5e-5)            # the figure quoted in SURVEY.md 8c
    assert got == pytest.approx(float(kats["survey_loss"]), rel=1e-6)


def test_product_get_loss_dict_has_no_cpu_path(kats):
    """get_loss_dict runs fused HIP kernels (tests/test_api_path.py feeds the reference's vectors to them on the
    GPU); handed CPU tensors it must fail loudly instead of quietly computing something else."""
    from qed_splatter_amd._lib import QedSplatError
    from qed_splatter_amd.model import QEDSplatterModel, QEDSplatterModelConfig
    sc = scene(8, 16, 16, seed=1)
    m = QEDSplatterModel(QEDSplatterModelConfig.synthetic(ssim_lambda=0.0), **{k: sc[k] for k in PARAM_NAMES})
    d = torch.rand(16, 16, 1)
    with pytest.raises(QedSplatError):
        m.get_loss_dict({"depth": d, "rgb": torch.zeros(16, 16, 3), "background": torch.zeros(3)},
                        {"depth_image": d, "image": torch.zeros(16, 16, 3)})


def test_parent_class_helpers_on_the_host():
    """The restated SplatfactoModel helpers that run on the host: resolution schedule, box-filter
    downscaling of the ground truth (== oracle.resize_image), background colours, defaults."""
    from qed_splatter_amd.model import QEDSplatterModel, QEDSplatterModelConfig
    sc = scene(8, 16, 16, seed=1)
    cfg = QEDSplatterModelConfig()
    assert (cfg.num_downscales, cfg.resolution_schedule, cfg.background_color) == (2, 3000, "random")
    assert (cfg.depth_lambda, cfg.output_depth_during_training) == (0.2, True)          # model.py:44,46
    m = QEDSplatterModel(cfg, **{k: sc[k] for k in PARAM_NAMES})
    for step, d in [(0, 4), (2999, 4), (3000, 2), (5999, 2), (6000, 1), (30000, 1)]:
        m.step = step
        assert m._get_downscale_factor() == d
    m.step = 0
    img = torch.rand(37, 50, 3)
    assert torch.allclose(m.get_gt_img(img), O.resize_image(img, 4), atol=1e-6) and m.get_gt_img(img).shape == (9, 12, 3)
    u8 = (img * 255).to(torch.uint8)
    assert torch.allclose(m.get_gt_img(u8), O.resize_image(u8.float() / 255.0, 4), atol=1e-6)
    m.eval()
    assert m._get_downscale_factor() == 1 and m.get_gt_img(img).shape == img.shape
    assert m._get_background_color().tolist() == pytest.approx([0.1490, 0.1647, 0.2157])
    m.train()
    assert m._get_background_color().shape == (3,)
    with pytest.raises(NotImplementedError):
        m._apply_bilateral_grid(img, 0, 37, 50)


def test_oracle_masked_main_loss_and_scale_reg():
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(24, 30, 3, generator=g, dtype=torch.float64), torch.rand(24, 30, 3, generator=g, dtype=torch.float64)
    mask = (torch.rand(24, 30, 1, generator=g) > 0.4).double()
    # the parent multiplies both images by the mask, then takes plain L1 / SSIM of the products
    assert float(O.main_loss(a, b, 0.2, mask)) == pytest.approx(float(O.main_loss(a * mask, b * mask, 0.2)), rel=1e-14)
    assert float(O.main_loss(a, b, 0.0, mask)) == pytest.approx(float(((a - b).abs() * mask).mean()), rel=1e-12)
    s = torch.log(torch.tensor([[1.0, 1.0, 1.0], [1.0, 2.0, 30.0], [0.1, 0.1, 5.0]], dtype=torch.float64))
    assert float(O.scale_reg(s, 10, True, 10.0)) == pytest.approx(0.1 * (0.0 + 20.0 + 40.0) / 3, rel=1e-12)
    assert float(O.scale_reg(s, 11, True, 10.0)) == 0.0 and float(O.scale_reg(s, 10, False)) == 0.0


def test_depth_metrics_cross_check(kats):
    """oracle.depth_metrics (metrics.py:126-156) on the reference's own input/output vector."""
    pred, gt = torch.from_numpy(kats["dm_pred"]), torch.from_numpy(kats["dm_gt"])
    vals = O.depth_metrics(pred, gt, 0.1)
    np.testing.assert_allclose([float(v) for v in vals], kats["dm_out"], rtol=1e-6)
    assert all(math.isnan(float(v)) for v in O.depth_metrics(pred, torch.zeros_like(gt)))   # metrics.py:134-143


# ---- oracle regression against the committed golden vectors --------------------------------------
def _load_case(gold, name):
    pre = name + "/"
    return {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}


@pytest.mark.parametrize("name", ["classic_deg3", "antialiased_deg2", "rgb_only_colors"])
def test_oracle_reproduces_golden(gold, name):
    c = _load_case(gold, name)
    w, h = (int(v) for v in c["in_size"])
    deg = int(c["in_deg"])
    ps = {k: torch.from_numpy(c[f"in_{k}"]).double().requires_grad_(True) for k in PARAM_NAMES}
    out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                               ps["features_rest"], torch.from_numpy(c["in_camera_to_worlds"]).double(),
                               torch.from_numpy(c["in_Ks"]).double(), w, h, torch.zeros(3, dtype=torch.float64),
                               sh_degree_to_use=None if deg < 0 else deg, rasterize_mode=str(c["in_mode"]))
    l_rgb = O.main_loss(out["rgb"], torch.from_numpy(c["in_gt_rgb"]).double(), 0.2)
    l_d = O.depth_l1_loss(out["depth"], torch.from_numpy(c["in_gt_depth"]).double(), None, 0.2)
    (l_rgb + l_d).backward()
    info = out["info"]
    assert np.array_equal(info["radii"].numpy(), c["radii"])
    assert np.array_equal(info["isect_ids"].numpy(), c["isect_ids"])
    assert np.array_equal(info["flatten_ids"].numpy(), c["flatten_ids"])
    assert np.array_equal(info["isect_offsets"].numpy(), c["isect_offsets"])
    assert np.array_equal(info["last_ids"].numpy(), c["last_ids"])
    np.testing.assert_allclose(out["render"].detach().numpy(), c["render"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out["depth"].detach().numpy(), c["depth"], rtol=1e-12, atol=1e-14)
    assert float(l_rgb) == pytest.approx(float(c["loss_rgb"]), rel=1e-12)
    assert float(l_d) == pytest.approx(float(c["loss_depth"]), rel=1e-12)
    for k in PARAM_NAMES:
        np.testing.assert_allclose(ps[k].grad.numpy(), c[f"grad_{k}"], rtol=1e-9, atol=1e-14)


def test_oracle_fp32_close_to_fp64(gold):
    """The oracle is dtype generic; its fp32 run (what the CPU baseline times) agrees with fp64."""
    c = _load_case(gold, "classic_deg3")
    w, h = (int(v) for v in c["in_size"])
    ps = {k: torch.from_numpy(c[f"in_{k}"]).float() for k in PARAM_NAMES}
    out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                               ps["features_rest"], torch.from_numpy(c["in_camera_to_worlds"]),
                               torch.from_numpy(c["in_Ks"]), w, h, torch.zeros(3),
                               radii_override=torch.from_numpy(c["radii"]))
    safe = torch.from_numpy(c["margin"][0] > 1e-4)
    err = (out["render"][0].double() - torch.from_numpy(c["render"][0])).abs().amax(-1)
    assert float(err[safe].max()) < 1e-4


# ---- gradients: autograd of the oracle is itself checked numerically ------------------------------
def test_oracle_gradcheck_fp64():
    w, h, n = 16, 16, 5
    sc = scene(n, w, h, seed=42)
    sc["scales"] = sc["scales"] + 3.0
    a = activated(sc)
    a["colors"] = a["colors"][:, :4, :]

    def f(means, quats, scales, opacities, colors):
        render, alpha, _ = O.rasterization(means=means, quats=quats, scales=scales, opacities=opacities, colors=colors,
                                           viewmats=a["viewmats"], Ks=a["Ks"], width=w, height=h,
                                           render_mode="RGB+D", sh_degree=1)
        return render.sum() + 0.5 * (alpha ** 2).sum()

    ins = [a[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
    assert torch.autograd.gradcheck(f, ins, eps=1e-6, atol=1e-5, rtol=1e-3, nondet_tol=0.0)


# ---- structural properties ----------------------------------------------
@pytest.fixture(scope="module")
def small_run():
    w, h, n = 100, 70, 1500
    sc = scene(n, w, h, seed=9, n_cameras=2)
    sc["scales"] = sc["scales"] + 1.5
    a = activated(sc)
    render, alpha, info = O.rasterization(**a, width=w, height=h, render_mode="RGB+D", sh_degree=3, return_margin=True)
    return sc, a, render, alpha, info, (w, h, n)


def test_keys_sorted_and_offsets_partition(small_run):
    _, _, _, _, info, (w, h, n) = small_run
    keys = info["isect_ids"]
    M = keys.numel()
    assert M > 0 and bool((keys[1:] >= keys[:-1]).all())
    offs = info["isect_offsets"].reshape(-1)
    assert int(offs[0]) == 0 and bool((offs[1:] >= offs[:-1]).all()) and int(offs[-1]) <= M
    assert int(info["tiles_per_gauss"].sum()) == M
    # each tile's run holds exactly the keys of that (camera, tile), depth ascending
    T = info["tile_width"] * info["tile_height"]
    tb = O.tile_bits(T)
    ends = torch.cat([offs[1:], torch.tensor([M], dtype=offs.dtype)])
    for t in (0, 7, T - 1, T, T + 5, 2 * T - 1):
        s, e = int(offs[t]), int(ends[t])
        kk = keys[s:e] >> 32
        assert bool(((kk >> tb) == t // T).all()) and bool(((kk & ((1 << tb) - 1)) == t % T).all())


def test_alpha_range_and_weights(small_run):
    _, _, render, alpha, info, _ = small_run
    assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0 - 1e-4 + 1e-12
    assert float(render[..., :3].min()) >= 0.0
    # accumulated depth <= alpha * max depth (weights sum to alpha)
    assert bool((render[..., 3] <= alpha[..., 0] * info["depths"].max() + 1e-9).all())


def test_permutation_invariance(small_run):
    sc, a, render, alpha, _, (w, h, n) = small_run
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    b = dict(a)
    for k in ("means", "quats", "scales", "opacities", "colors"):
        b[k] = a[k][perm]
    r2, a2, _ = O.rasterization(**b, width=w, height=h, render_mode="RGB+D", sh_degree=3)
    torch.testing.assert_close(r2, render, rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(a2, alpha, rtol=1e-9, atol=1e-12)


def test_ssim_closed_form_cases():
    """Pins of the SSIM restatement that need no library: identical images give exactly 1, constant
    images reduce to the luminance term, the window sums to 1 and is symmetric, the map covers the
    valid (H-10) x (W-10) region only, and SSIM is symmetric in its arguments."""
    g = torch.Generator().manual_seed(3)
    a = torch.rand(40, 37, 3, generator=g, dtype=torch.float64)
    b = torch.rand(40, 37, 3, generator=g, dtype=torch.float64)
    w = O.ssim_window()
    assert w.shape == (11,) and float(w.sum()) == pytest.approx(1.0, abs=1e-15) and torch.equal(w, w.flip(0))
    assert float(w[5] / w[4]) == pytest.approx(math.exp(1 / 4.5), rel=1e-12)
    v, smap = O.ssim(a, a, return_map=True)
    assert smap.shape == (30, 27, 3)
    assert float(v) == pytest.approx(1.0, abs=1e-12)
    assert float(O.ssim(a, b)) == pytest.approx(float(O.ssim(b, a)), rel=1e-12)
    ca, cb = torch.full((16, 16, 3), 0.25, dtype=torch.float64), torch.full((16, 16, 3), 0.75, dtype=torch.float64)
    lum = (2 * 0.25 * 0.75 + 1e-4) / (0.25 ** 2 + 0.75 ** 2 + 1e-4)      # variances vanish -> cs = C2/C2 = 1
    assert float(O.ssim(ca, cb)) == pytest.approx(lum, rel=1e-9)
    # direct (non-separable) evaluation of one map pixel
    x, y = a[3:14, 5:16, 0], b[3:14, 5:16, 0]
    W2 = w[:, None] * w[None, :]
    mx, my = (W2 * x).sum(), (W2 * y).sum()
    vx, vy, cxy = (W2 * x * x).sum() - mx * mx, (W2 * y * y).sum() - my * my, (W2 * x * y).sum() - mx * my
    direct = (2 * mx * my + 1e-4) / (mx * mx + my * my + 1e-4) * (2 * cxy + 9e-4) / (vx + vy + 9e-4)
    _, m2 = O.ssim(a, b, return_map=True)
    assert float(m2[3, 5, 0]) == pytest.approx(float(direct), rel=1e-12)
    assert float(O.main_loss(a, b, 0.2)) == pytest.approx(0.8 * float((a - b).abs().mean()) + 0.2 * (1 - float(O.ssim(a, b))), rel=1e-12)


def test_empty_cases():
    w, h = 40, 24
    sc = scene(16, w, h, seed=3)
    a = activated(sc)
    a["means"] = a["means"] * torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)   # all behind the camera
    render, alpha, info = O.rasterization(**a, width=w, height=h, render_mode="RGB+D", sh_degree=3)
    assert info["flatten_ids"].numel() == 0 and float(render.abs().max()) == 0 and float(alpha.max()) == 0
    # no valid ground-truth depth -> loss 0, not NaN (model.py:111-114)
    assert float(O.depth_l1_loss(torch.rand(4, 4, 1), torch.zeros(4, 4, 1))) == 0.0


def test_non_multiple_of_16_sizes():
    for (w, h) in [(33, 17), (1, 1), (16, 16), (47, 95)]:
        sc = scene(40, w, h, seed=w + h)
        sc["scales"] = sc["scales"] + 2.5
        a = activated(sc)
        render, alpha, info = O.rasterization(**a, width=w, height=h, render_mode="RGB", sh_degree=0)
        assert render.shape == (1, h, w, 3) and alpha.shape == (1, h, w, 1)
        assert info["tile_width"] == math.ceil(w / 16) and info["tile_height"] == math.ceil(h / 16)


# ---- product host logic that needs no GPU ----------------------------------------------------------
def test_model_parameter_layout_and_flat_grad():
    from qed_splatter_amd.model import GROUP_ORDER, QEDSplatterModel
    sc = scene(10, 16, 16, seed=1)
    m = QEDSplatterModel(None, **{k: sc[k] for k in PARAM_NAMES})
    assert set(m.gauss_params.keys()) == set(GROUP_ORDER) and tuple(m.group_names) == GROUP_ORDER
    # ... laid out back to back in GROUP_ORDER inside the flat buffer
    off = 0
    for name in GROUP_ORDER:
        p = m.gauss_params[name]
        assert p.data_ptr() == m.flat_params.data_ptr() + 4 * off
        off += p.numel()
    assert m.flat_params.numel() == 59 * 10 and m.num_points == 10
    # the parameters alias the flat buffer
    m.flat_params.zero_()
    assert float(m.means.abs().sum()) == 0.0
    # gradients that do not alias one buffer are re-packed once, then alias
    for k in PARAM_NAMES:
        m.gauss_params[k].grad = torch.ones_like(m.gauss_params[k])
    fg = m.flat_grad()
    assert fg.numel() == 590 and float(fg.sum()) == 590.0
    assert m.flat_grad().data_ptr() == fg.data_ptr() == m.means.grad.data_ptr()
    # state_dict keys interchange with SplatfactoModel's gauss_params.* naming
    assert set(m.state_dict().keys()) == {f"gauss_params.{k}" for k in PARAM_NAMES}


def test_rasterization_has_no_cpu_fallback():
    from qed_splatter_amd._lib import QedSplatError
    from qed_splatter_amd.rasterization import rasterization
    sc = scene(8, 16, 16, seed=1)
    a = activated(sc, torch.float32)
    with pytest.raises(QedSplatError):
        rasterization(**a, width=16, height=16, sh_degree=3)
