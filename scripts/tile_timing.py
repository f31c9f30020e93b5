import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qed_splatter_amd import _lib as L
from qed_splatter_amd.model import PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
from qed_splatter_amd.scene import synthetic_scene
dev = torch.device("cuda:0"); L.load()
n, w, h = 500000, 1920, 1080
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(n, w, h, seed=1235).items()}
model = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree_interval=1), **{k: sc[k] for k in ("means","scales","quats","opacities","features_dc","features_rest")})
model.step = 30000
K = sc["Ks"][0].cpu()
cam = PinholeCameras(sc["camera_to_worlds"], float(K[0,0]), float(K[1,1]), float(K[0,2]), float(K[1,2]), w, h)
model.eval()
with torch.no_grad():
    for _ in range(3):
        out = model.get_outputs(cam)
torch.cuda.synchronize()
a = out["accumulation"][..., 0]
dur = a[::16, 0::16].flatten().cpu().double()
t0 = a[::16, 1::16].flatten().cpu().double()
offs = model.info["isect_offsets"].flatten().cpu()
lens = torch.diff(torch.cat([offs, torch.tensor([model.info["n_isects"]], dtype=offs.dtype)])).double()
print("tiles", dur.numel(), "dur mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f (cycles)" % (dur.mean(), dur.median(), dur.quantile(0.9), dur.quantile(0.99), dur.max()))
print("list len mean %.0f p99 %.0f max %.0f" % (lens.mean(), lens.quantile(0.99), lens.max()))
t0 = (t0 - t0.min()) % (1 << 28)
print("start-time spread: p50 %.0f p90 %.0f p99 %.0f max %.0f ; end max %.0f" % (t0.median(), t0.quantile(0.9), t0.quantile(0.99), t0.max(), (t0 + dur).max()))
# correlation of duration with list length
print("corr(dur, len) = %.3f" % torch.corrcoef(torch.stack([dur, lens]))[0, 1])
import numpy as np
hw = a[::16, 2::16].flatten().cpu().long()
xcc = a[::16, 3::16].flatten().cpu().long()
t0r = a[::16, 1::16].flatten().cpu().double()
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("distinct SIMDs seen:", key.unique().numel(), " distinct CUs:", (key // 4).unique().numel(), "xcc:", xcc.unique().tolist())
mx = []
for k in key.unique().tolist()[:4000]:
    m = key == k
    st = t0r[m].numpy(); en = st + dur[m].numpy()
    ev = sorted([(x, 1) for x in st] + [(x, -1) for x in en])
    c = best = 0
    for _, d in ev:
        c += d; best = max(best, c)
    mx.append(best)
mx = np.array(mx)
print("tiles per SIMD: mean %.2f" % (len(key) / key.unique().numel()), " max concurrent waves per SIMD: mean %.2f max %d hist %s" % (mx.mean(), mx.max(), np.bincount(mx).tolist()))
# per-XCD activity profile (clocks are per XCD)
for x in range(8):
    m = xcc == x
    st = t0r[m].numpy().copy()
    med = np.median(st)
    st[st - med > 2 ** 23] -= 2 ** 24                  # unwrap the 24-bit start stamps around the median
    st[med - st > 2 ** 23] += 2 ** 24
    en = st + dur[m].numpy()
    base = st.min()
    span = en.max() - base
    bins = np.linspace(base, en.max(), 13)
    act = [int(((st < b1) & (en > b0)).sum()) for b0, b1 in zip(bins[:-1], bins[1:])]
    print(f"xcc {x}: tiles {int(m.sum())} span {span:.0f} cycles; active waves per 1/12 of the span: {act}")
# per-CU profile: spans within one CU share a clock for sure
cu_key = key // 4
spans = []; occ = []
for k in cu_key.unique().tolist():
    m = cu_key == k
    st = t0r[m].numpy().copy(); med = np.median(st)
    st[st - med > 2 ** 23] -= 2 ** 24; st[med - st > 2 ** 23] += 2 ** 24
    en = st + dur[m].numpy()
    spans.append(en.max() - st.min()); occ.append(dur[m].sum().item() / (en.max() - st.min()))
spans = np.array(spans); occ = np.array(occ)
print("per-CU span: mean %.0f min %.0f max %.0f cycles; mean concurrent waves per CU %.2f (min %.2f max %.2f)" % (spans.mean(), spans.min(), spans.max(), occ.mean(), occ.min(), occ.max()))
k0 = cu_key.unique().tolist()[5]
m = cu_key == k0
st = t0r[m].numpy().copy(); med = np.median(st); st[st - med > 2 ** 23] -= 2 ** 24; st[med - st > 2 ** 23] += 2 ** 24
o = np.argsort(st)
print("one CU: starts (relative)", (st[o] - st.min()).astype(int).tolist()[:40])
print("one CU: durs", dur[m].numpy()[o].astype(int).tolist()[:40])
