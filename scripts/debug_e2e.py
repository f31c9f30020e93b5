import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import splat_oracle as O
from tests.test_gpu_parity import _model, _oracle_step
from tests.util import scene
w, h, n = 256, 256, 10000
dev = torch.device("cuda:0")
sc = scene(n, w, h, seed=1234)
m, cam, batch = _model(sc, dev)
out = m.get_outputs(cam)
info = m.info
ref, l_rgb, l_d, ps = _oracle_step(sc, w, h, m.config, radii=info["radii"].cpu())
ri = ref["info"]
print("M gpu", info["n_isects"], "oracle", ri["flatten_ids"].numel())
print("tiles_per_gauss equal:", torch.equal(info["tiles_per_gauss"].cpu(), ri["tiles_per_gauss"]))
d = (info["tiles_per_gauss"].cpu() != ri["tiles_per_gauss"]).nonzero()
print("tpg diffs:", d[:10].tolist())
if info["n_isects"] == ri["flatten_ids"].numel():
    neq = (info["flatten_ids"].cpu() != ri["flatten_ids"]).nonzero().flatten()
    print("flatten_ids diffs:", neq.numel(), neq[:10].tolist())
err = (out["rgb"].detach().cpu().double() - ref["rgb"].detach()).abs().amax(-1)
idx = err.flatten().argsort(descending=True)[:8]
for i in idx.tolist():
    y, x = divmod(i, w)
    print(f"pix ({x},{y}) err {err[y,x]:.3e} margin {ri['margin'][0,y,x]:.3e} last gpu {int(info['last_ids'][0,y,x])} ref {int(ri['last_ids'][0,y,x])}")
md = (info["means2d"].detach().cpu().double() - ri["means2d"].detach()).abs().max()
print("means2d maxdiff", float(md), "depth maxdiff", float((info["depths"].detach().cpu().double() - ri["depths"].detach()).abs().max()))
print("conics rel", float(((info["conics"].detach().cpu().double() - ri["conics"].detach()).abs().max())/ri["conics"].abs().max()))
print("opac", float((info["opacities"].detach().cpu().double() - ri["opacities"].detach()).abs().max()))
print("colors", float((info["colors"].detach().cpu().double() - ri["colors"].detach()[...,:3]).abs().max()))
