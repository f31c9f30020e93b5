import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.util import PARAM_NAMES, scene
from qed_splatter_amd.model import get_viewmat
from qed_splatter_amd.rasterization import rasterization
cuda = torch.device("cuda:0")
w, h, n = 200, 136, 8000
sc = scene(n, w, h, seed=31)
ps = {k: sc[k].to(cuda) for k in PARAM_NAMES}
def run(which):
    if which is None: os.environ.pop("QED_COMPOSITE_WAVES", None)
    else: os.environ["QED_COMPOSITE_WAVES"] = which
    with torch.no_grad():
        r, a, info = rasterization(means=ps["means"], quats=torch.nn.functional.normalize(ps["quats"], dim=-1), scales=ps["scales"].exp(),
            opacities=torch.sigmoid(ps["opacities"]).squeeze(-1), colors=torch.cat([ps["features_dc"][:, None, :], ps["features_rest"]], dim=1),
            viewmats=get_viewmat(sc["camera_to_worlds"][:1].to(cuda)), Ks=sc["Ks"][:1].to(cuda), width=w, height=h, render_mode="RGB+D", sh_degree=3)
    return r.clone(), a.clone()
r0, a0 = run("tile"); r1, a1 = run("quadrant")
d = (r0 != r1).any(-1)[0]
ys, xs = torch.nonzero(d, as_tuple=True)
print(os.environ.get("QED_SPLAT_LIB", "product")[-10:], "differing pixels", int(d.sum()), "of", d.numel(), "max abs", float((r0 - r1).abs().max()))
if len(ys):
    q = ((xs % 16) // 8 + 2 * ((ys % 16) // 8))
    print("by quadrant", [int((q == k).sum()) for k in range(4)])
    print("first", [(int(y), int(x)) for y, x in zip(ys[:8], xs[:8])])
