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

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {sorted(out)}")
    print("survey_loss =", out["survey_loss"])


if __name__ == "__main__":
    main()
