#!/usr/bin/env python3
"""Generate known-answer vectors for the REFERENCE-OWNED arithmetic of the hot path by importing the
reference itself (this container only; /root/reference never travels to the GPU box).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_reference_kats.py

Follows SURVEY.md Appendix B: the reference's third-party imports that are not installed here
(nerfstudio, torchvision, torchmetrics; gsplat is wrapped in try/except by the reference itself)
are satisfied with empty placeholder modules so that `import qed_splatter.model` succeeds; only
functions whose bodies are entirely the reference's own torch code are then executed:
  * get_viewmat                         /root/reference/qed_splatter/model.py:22-38
  * QEDSplatterModel.get_loss_dict      model.py:73-118   (depth-L1 term; parent loss stubbed to {})
  * DepthMetrics.forward                /root/reference/qed_splatter/metrics.py:126-156
  * _opengl_c2w_to_opencv_w2c, _frame_intrinsics   /root/reference/qed_splatter/create_init_pointcloud.py:49-70
                                        (open3d / tyro / PIL satisfied with empty placeholders; nothing of them runs)
  * QEDSplatterModel.get_metrics_dict   model.py:120-197, the d = 1 branch: ground-truth selection ([..., :3] of an RGBA
                                        image), rgb_mse, the seven depth metrics through the reference's own DepthMetrics,
                                        avg_min_scale, gaussian_count and the KEY SET.  PSNR / SSIM / LPIPS come from
                                        torchmetrics (absent): a stand-in returns constants, which are not recorded
The rasterizer (gsplat) cannot be exercised this way -> rasterizer parity stays "unpinned".
Output: tests/golden/reference_kats.npz (data only: inputs and the reference's outputs).
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.npz")


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_placeholders():
    class SplatfactoModelConfig:          # noqa: D401 - empty stand-in for an absent third-party class
        pass

    class SplatfactoModel(torch.nn.Module):
        def get_loss_dict(self, outputs, batch, metrics_dict=None):
            return {}

        def get_gt_img(self, image):
            return image

    _placeholder("nerfstudio")
    _placeholder("nerfstudio.models")
    _placeholder("nerfstudio.models.splatfacto", SplatfactoModelConfig=SplatfactoModelConfig,
                 SplatfactoModel=SplatfactoModel)
    _placeholder("nerfstudio.cameras")
    _placeholder("nerfstudio.cameras.cameras", Cameras=object)
    _placeholder("nerfstudio.utils")
    _placeholder("nerfstudio.utils.misc", torch_compile=lambda f=None, **k: f)
    _placeholder("torchvision")
    _placeholder("torchvision.transforms")
    _placeholder("torchvision.transforms.functional")
    dummy = type("Dummy", (torch.nn.Module,), {"__init__": lambda self, *a, **k: torch.nn.Module.__init__(self)})
    _placeholder("torchmetrics")
    _placeholder("torchmetrics.image", PeakSignalNoiseRatio=dummy, StructuralSimilarityIndexMeasure=dummy)
    _placeholder("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=dummy)
    # create_init_pointcloud.py:24-27 (annotations are strings there: nothing of these is touched at import)
    _placeholder("open3d")
    _placeholder("tyro", cli=lambda *a, **k: None)
    _placeholder("PIL", Image=types.SimpleNamespace())
    sys.modules["PIL.Image"] = sys.modules["PIL"].Image


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    import qed_splatter.model as M          # prints "Please install gsplat>=1.0.0" (model.py:9)
    from qed_splatter.metrics import DepthMetrics

    out = {}
    g = torch.Generator().manual_seed(20260101)

    # ---- get_viewmat: identity + translation (the SURVEY 8c known answer) and random rigid poses ----
    c2w0 = torch.cat([torch.eye(3), torch.tensor([[1.0], [2.0], [3.0]])], dim=1)[None]
    q = torch.randn(5, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(5, 3, 3)
    c2w = torch.cat([torch.cat([R, torch.randn(5, 3, 1, generator=g) * 3], dim=2), c2w0], dim=0)
    out["viewmat_c2w"] = c2w.numpy()
    out["viewmat_out"] = M.get_viewmat(c2w).numpy()

    # ---- depth-L1 term of get_loss_dict ----
    model = M.QEDSplatterModel.__new__(M.QEDSplatterModel)
    torch.nn.Module.__init__(model)
    cases = []
    for idx, (h, w_, use_mask, lam) in enumerate([(4, 5, False, 0.2), (16, 12, True, 0.2), (9, 7, False, 0.3),
                                                   (6, 6, True, 0.25), (3, 3, False, 0.2)]):
        model.config = types.SimpleNamespace(depth_lambda=lam)
        d_out = torch.rand(h, w_, 1, generator=g) * 10
        d_gt = torch.rand(h, w_, 1, generator=g) * 10
        d_gt[torch.rand(h, w_, 1, generator=g) < 0.2] = 0.0           # invalid (zero) ground truth
        d_gt[0, 0, 0] = float("nan")
        if idx == 2:
            d_gt[1, 1, 0] = float("inf")
            d_out[2, 2, 0] = float("nan")
        if idx == 4:
            d_gt = torch.zeros_like(d_gt)                              # no valid pixel -> loss 0.0 (model.py:111-114)
        batch = {"depth_image": d_gt, "image": torch.rand(h, w_, 3, generator=g)}
        if use_mask:
            batch["mask"] = (torch.rand(h, w_, 1, generator=g) > 0.3).float()
        outputs = {"depth": d_out, "rgb": torch.rand(h, w_, 3, generator=g)}
        loss = model.get_loss_dict(outputs, batch)["depth_loss"]
        out[f"dl{idx}_depth_out"] = d_out.numpy()
        out[f"dl{idx}_depth_gt"] = d_gt.numpy()
        out[f"dl{idx}_mask"] = batch["mask"].numpy() if use_mask else np.zeros(0, np.float32)
        out[f"dl{idx}_lambda"] = np.float32(lam)
        out[f"dl{idx}_loss"] = np.float32(float(loss))
        cases.append(idx)
    out["dl_cases"] = np.array(cases)

    # the SURVEY 8c known answer: manual_seed(0), rand(4,5,1) x2, gt[0,0,0]=0, gt[1,1,0]=nan -> 0.0653
    torch.manual_seed(0)
    d_out = torch.rand(4, 5, 1)
    d_gt = torch.rand(4, 5, 1)
    d_gt[0, 0, 0] = 0
    d_gt[1, 1, 0] = float("nan")
    model.config = types.SimpleNamespace(depth_lambda=0.2)
    out["survey_depth_out"] = d_out.numpy()
    out["survey_depth_gt"] = d_gt.numpy()
    out["survey_loss"] = np.float32(float(model.get_loss_dict({"depth": d_out, "rgb": torch.zeros(4, 5, 3)},
                                                              {"depth_image": d_gt, "image": torch.zeros(4, 5, 3)})
                                          ["depth_loss"]))

    # ---- DepthMetrics (metrics.py:126-156): usable as a cross-check of rendered depth ----
    dm = DepthMetrics()
    pred = torch.rand(1, 24, 20, generator=g) * 8 + 0.5
    gt = torch.rand(1, 24, 20, generator=g) * 8
    gt[gt < 0.8] = 0.0
    out["dm_pred"] = pred.numpy()
    out["dm_gt"] = gt.numpy()
    out["dm_out"] = np.array([float(v) for v in dm(pred, gt)], dtype=np.float64)

    # ---- qed-init-pc's pose / intrinsics helpers (create_init_pointcloud.py:49-70) ----
    import json
    import qed_splatter.create_init_pointcloud as CP
    qq = torch.randn(6, 4, generator=g, dtype=torch.float64)
    qq = qq / qq.norm(dim=-1, keepdim=True)
    w, x, y, z = qq.unbind(-1)
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                      2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(6, 3, 3)
    poses = torch.eye(4, dtype=torch.float64)[None].repeat(6, 1, 1)
    poses[:, :3, :3] = Rm
    poses[:, :3, 3] = torch.randn(6, 3, generator=g, dtype=torch.float64) * 4
    poses[0] = torch.eye(4, dtype=torch.float64)
    poses[0, :3, 3] = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    out["ip_c2w_opengl"] = poses.numpy()
    out["ip_w2c_opencv"] = np.stack([CP._opengl_c2w_to_opencv_w2c(p_) for p_ in poses.numpy()])
    # (contents, frame) pairs: frame-level values win, fl_y falls back to the FRAME's fl_x, then the file's
    intr_cases = [
        ({"fl_x": 500.0, "fl_y": 510.0, "cx": 320.0, "cy": 240.0}, {}),
        ({"fl_x": 500.0, "cx": 320.0, "cy": 240.0}, {}),
        ({"fl_x": 500.0, "fl_y": 510.0, "cx": 320.0, "cy": 240.0}, {"fl_x": 777.5, "cx": 300.25}),
        ({"fl_x": 500.0, "cx": 320.0, "cy": 240.0}, {"fl_x": 640.0}),
        ({"fl_x": 500.0, "fl_y": 510.0, "cx": 320.0, "cy": 240.0}, {"fl_x": 600.0, "fl_y": 601.0, "cx": 1.5, "cy": 2.5}),
    ]
    out["ip_intrinsics_cases"] = np.array(json.dumps(intr_cases))
    out["ip_intrinsics_out"] = np.stack([CP._frame_intrinsics(c, f) for c, f in intr_cases])

    # ---- get_metrics_dict, d = 1 (model.py:120-197) ----
    class _RGBStandIn(torch.nn.Module):                     # torchmetrics is absent: constants, never recorded
        def forward(self, pred, gt):
            return torch.tensor(1.0), torch.tensor(2.0), torch.tensor(3.0)
    mm = M.QEDSplatterModel.__new__(M.QEDSplatterModel)
    torch.nn.Module.__init__(mm)
    mm.rgb_metrics, mm.depth_metrics, mm.mse_loss = _RGBStandIn(), DepthMetrics(), torch.nn.MSELoss()
    mm._get_downscale_factor = lambda: 1
    n_pts = 37
    mm.gauss_params = torch.nn.ParameterDict({"scales": torch.nn.Parameter(torch.randn(n_pts, 3, generator=g) - 3.0),
                                              "means": torch.nn.Parameter(torch.zeros(n_pts, 3))})
    type(mm).scales = property(lambda self: self.gauss_params["scales"])
    type(mm).num_points = property(lambda self: self.gauss_params["means"].shape[0])
    type(mm).device = property(lambda self: torch.device("cpu"))
    with torch.no_grad():
        mm.gauss_params["scales"][3, 2] = float("nan")       # nanmean (model.py:193)
    Hh, Ww = 20, 28
    md_rgb = torch.rand(Hh, Ww, 3, generator=g)
    md_depth = torch.rand(Hh, Ww, 1, generator=g) * 8 + 0.5
    md_img = torch.rand(Hh, Ww, 4, generator=g)                # RGBA ground truth: [..., :3] is compared
    md_gt_depth = torch.rand(Hh, Ww, 1, generator=g) * 8
    md_gt_depth[md_gt_depth < 0.8] = 0.0
    for tag, batch in (("md", {"image": md_img, "depth_image": md_gt_depth}), ("mdn", {"image": md_img})):
        got = mm.get_metrics_dict({"rgb": md_rgb, "depth": md_depth}, batch)
        out[f"{tag}_keys"] = np.array(json.dumps(list(got.keys())))
        out[f"{tag}_types"] = np.array(json.dumps({k: type(v).__name__ for k, v in got.items()}))
        for k, v in got.items():
            if k not in ("rgb_psnr", "rgb_ssim", "rgb_lpips"):
                out[f"{tag}_{k}"] = np.float64(float(v))
    out["md_rgb"], out["md_depth"], out["md_image"], out["md_gt_depth"] = (md_rgb.numpy(), md_depth.numpy(), md_img.numpy(),
                                                                           md_gt_depth.numpy())
    out["md_scales"] = mm.gauss_params["scales"].detach().numpy()

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {sorted(out)}")
    print("survey_loss =", out["survey_loss"])


if __name__ == "__main__":
    main()
