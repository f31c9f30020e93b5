import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.util import PARAM_NAMES, scene, sweep_nonsmooth_pixels
from oracle import splat_oracle as O
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
cuda = torch.device("cuda:0")
w, h, n = 176, 112, 2600
for kind in ("opaque", "needles", "needles_short", "all_general", "plain"):
    sc = scene(n, w, h, seed=51)
    if kind == "opaque": sc["opacities"][0::5] = 9.0
    if kind == "needles": sc["scales"][1::7] = torch.log(torch.tensor([1.0, 0.0004, 0.0004]))
    if kind == "needles_short": sc["scales"][1::7] = torch.log(torch.tensor([0.25, 0.0004, 0.0004]))
    if kind == "all_general": sc["opacities"][:] = 7.5 + sc["opacities"] * 0.25
    cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1)
    K = sc["Ks"][0]
    cam = PinholeCameras(sc["camera_to_worlds"].to(cuda), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
    batch = {"image": sc["gt_rgb"].to(cuda), "depth_image": sc["gt_depth"].to(cuda)}
    model = QEDSplatterModel(cfg, **{k: sc[k].to(cuda) for k in PARAM_NAMES}); model.step = 3
    with torch.no_grad(): model.fused_loss(cam, batch)
    radii = model.info["radii"].cpu()
    def oracle_step(dt, mask=None):
        ps = {k: sc[k].detach().clone().to(dt).requires_grad_(True) for k in PARAM_NAMES}
        out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"], ps["features_rest"],
                                   sc["camera_to_worlds"].to(dt), sc["Ks"].to(dt), w, h, sc["background"].to(dt),
                                   sh_degree_to_use=3, radii_override=radii, return_margin=True)
        if mask is not None:
            (O.main_loss(out["rgb"], sc["gt_rgb"].to(dt), cfg.ssim_lambda, mask.to(dt)) + O.depth_l1_loss(out["depth"], sc["gt_depth"].to(dt), mask.to(dt), cfg.depth_lambda)).backward()
        return out, ps
    out, _ = oracle_step(torch.float64)
    bad_px, _ = sweep_nonsmooth_pixels(out, sc)
    mask = (~bad_px)[..., None].double()
    _, ps64 = oracle_step(torch.float64, mask)
    _, ps32 = oracle_step(torch.float32, mask)
    batch["mask"] = mask.to(cuda, torch.float32)
    model.backward_fused(model.fused_loss(cam, batch))
    res = []
    for name in PARAM_NAMES:
        b = ps64[name].grad
        e = float((model.gauss_params[name].grad.detach().cpu().double() - b).abs().max() / b.abs().max())
        band = float((ps32[name].grad.double() - b).abs().max() / b.abs().max())
        res.append(f"{name}:{e:.1e}/{band:.1e}")
    print(os.environ.get("QED_SPLAT_LIB", "product")[-12:], kind, "masked", f"{1-float(mask.mean()):.4f}", " ".join(res), flush=True)
