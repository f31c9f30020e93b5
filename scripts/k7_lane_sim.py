"""CPU model of K7's lane utilisation at config B under different pixel -> lane layouts (no GPU, no kernels).

For a sample of tiles of the SURVEY 8(d) scene: which pixels of the tile each listed Gaussian really reaches in the
backward pass (alpha >= 1/255, sigma >= 0, index <= the pixel's last contributing index), then how many 64-lane
"visit bodies" and reductions each scheme needs for those same (pixel, Gaussian) pairs:

  quadrants      today's kernel: a visit = one 8x8 quadrant, one Gaussian per wave instruction
  2 streams      the two halves of the wave walk their own ordered list over their own half of the tile (units of 32
                 pixels); a body runs unit u when either half's current Gaussian reaches its unit u
  4 streams      the same with 16-lane rows (units of 16 pixels)

Usage: python scripts/k7_lane_sim.py [n_tiles] [seed]
"""
import math
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from qed_splatter_amd.scene import synthetic_scene           # noqa: E402
from oracle import splat_oracle as O                          # noqa: E402


def tile_valid_sets(n_tiles, seed):
    W, H, N = 1920, 1080, 500_000
    sc = synthetic_scene(N, W, H, 1235)
    vm = O.get_viewmat(sc["camera_to_worlds"])
    radii, m2d, depths, conics, _ = O.project_gaussians(sc["means"], sc["quats"] / sc["quats"].norm(dim=-1, keepdim=True),
                                                         sc["scales"].exp(), vm, sc["Ks"], W, H)
    radii, m2d, depths, conics = radii[0].numpy(), m2d[0].numpy(), depths[0].numpy(), conics[0].numpy()
    op = torch.sigmoid(sc["opacities"][:, 0]).numpy()
    vis = radii > 0
    idx = np.nonzero(vis)[0]
    x, y, r = m2d[idx, 0], m2d[idx, 1], radii[idx].astype(np.float32)
    rng = np.random.default_rng(seed)
    tw, th = W // 16, (H + 15) // 16
    tiles = rng.choice(tw * th, size=n_tiles, replace=False)
    out = []
    for t in tiles:
        ty, tx = divmod(int(t), tw)
        x0, y0 = tx * 16, ty * 16
        cand = (x + r > x0) & (x - r < x0 + 16) & (y + r > y0) & (y - r < y0 + 16)
        g = idx[cand]
        if g.size == 0:
            continue
        g = g[np.argsort(depths[g], kind="stable")]
        px = x0 + np.arange(16) + 0.5
        py = y0 + np.arange(16) + 0.5
        dx = m2d[g, 0][:, None, None] - px[None, None, :]
        dy = m2d[g, 1][:, None, None] - py[None, :, None]
        sig = 0.5 * (conics[g, 0][:, None, None] * dx * dx + conics[g, 2][:, None, None] * dy * dy) + conics[g, 1][:, None, None] * dx * dy
        a = np.minimum(0.999, op[g][:, None, None] * np.exp(-sig))
        ok = (sig >= 0) & (a >= 1.0 / 255.0)
        inside = (py[None, :, None] < H) & (px[None, None, :] < W)
        ok &= inside
        keep = ok.reshape(len(g), -1).any(1)                     # the exact tile test of the projection kernel
        g, a, ok = g[keep], a[keep], ok[keep]
        # forward compositing -> last contributing index per pixel
        T = np.ones((16, 16), np.float32)
        done = np.zeros((16, 16), bool)
        last = np.full((16, 16), -1)
        for i in range(len(g)):
            act = ok[i] & ~done
            nT = T * (1 - a[i])
            term = act & (nT <= 1e-4)
            done |= term
            acc = act & ~term
            T = np.where(acc, nT, T)
            last = np.where(acc, i, last)
            if done.all():
                break
        order = np.arange(len(g))[:, None, None]
        V = ok & (order <= last[None])
        n_list = len(g)
        tile_last = last.max()
        V = V[: tile_last + 1][::-1]                             # backward order: back to front
        out.append((n_list, V))
    return out


def units_any(V, blocks):
    """V [G,16,16] bool; blocks: list of (ys, xs) slices -> [G, n_blocks] bool"""
    return np.stack([V[:, ys, xs].reshape(V.shape[0], -1).any(1) for ys, xs in blocks], 1)


def sl(a, b):
    return slice(a, b)


def scheme_quadrants(V):
    blocks = [(sl(0, 8), sl(0, 8)), (sl(0, 8), sl(8, 16)), (sl(8, 16), sl(0, 8)), (sl(8, 16), sl(8, 16))]
    U = units_any(V, blocks)
    red = U.any(1).sum()
    return U.sum(), red, red


def streams(V, stream_blocks, batch=64):
    """stream_blocks[s] = list of unit blocks of stream s (same count per stream).  Streams walk their own lists inside
    each staged batch of 64 list entries (the batch is shared: all streams finish it before the next is staged)."""
    S = len(stream_blocks)
    Us = [units_any(V, b) for b in stream_blocks]                 # [G, nu] per stream
    G = V.shape[0]
    bodies = steps = rows = 0
    for b0 in range(0, G, batch):
        lists = []
        for s in range(S):
            u = Us[s][b0:b0 + batch]
            lists.append(u[u.any(1)])
        n = max(len(l) for l in lists)
        steps += n
        rows += sum(len(l) for l in lists)
        for i in range(n):
            un = np.zeros(Us[0].shape[1], bool)
            for l in lists:
                if i < len(l):
                    un |= l[i]
            bodies += un.sum()
    return bodies, steps, rows


def layouts():
    L = {}
    # two streams, halves top / bottom, units 2x2 of 8w x 4h, the bottom half mirrored vertically
    top = [(sl(0, 4), sl(0, 8)), (sl(0, 4), sl(8, 16)), (sl(4, 8), sl(0, 8)), (sl(4, 8), sl(8, 16))]
    bot = [(sl(12, 16), sl(0, 8)), (sl(12, 16), sl(8, 16)), (sl(8, 12), sl(0, 8)), (sl(8, 12), sl(8, 16))]
    L["2s top/bottom, units 8x4 (2x2, mirrored)"] = [top, bot]
    bot_n = [(sl(8, 12), sl(0, 8)), (sl(8, 12), sl(8, 16)), (sl(12, 16), sl(0, 8)), (sl(12, 16), sl(8, 16))]
    L["2s top/bottom, units 8x4 (2x2, not mirrored)"] = [top, bot_n]
    L["2s top/bottom, units 4x8 columns"] = [[(sl(0, 8), sl(4 * k, 4 * k + 4)) for k in range(4)],
                                             [(sl(8, 16), sl(4 * k, 4 * k + 4)) for k in range(4)]]
    L["2s left/right, units 8x4 rows"] = [[(sl(4 * k, 4 * k + 4), sl(0, 8)) for k in range(4)],
                                          [(sl(4 * k, 4 * k + 4), sl(8, 16)) for k in range(4)]]
    # the priced "half-quadrant" form: stream = top / bottom half of every quadrant, unit = quadrant
    q = [(0, 0), (0, 8), (8, 0), (8, 8)]
    L["2s half-quadrants (top/bottom 8x4 of each quadrant)"] = [[(sl(y, y + 4), sl(x, x + 8)) for y, x in q],
                                                               [(sl(y + 4, y + 8), sl(x, x + 8)) for y, x in q]]
    # four streams: stream = quadrant, units 2x2 of 4x4, mirrored towards the tile centre
    def quad_units(y0, x0, my, mx):
        ys = [0, 4] if not my else [4, 0]
        xs = [0, 4] if not mx else [4, 0]
        return [(sl(y0 + a, y0 + a + 4), sl(x0 + b, x0 + b + 4)) for a in ys for b in xs]
    L["4s quadrants, units 4x4 (mirrored)"] = [quad_units(0, 0, True, True), quad_units(0, 8, True, False),
                                               quad_units(8, 0, False, True), quad_units(8, 8, False, False)]
    L["4s quadrants, units 4x4 (plain)"] = [quad_units(0, 0, False, False), quad_units(0, 8, False, False),
                                            quad_units(8, 0, False, False), quad_units(8, 8, False, False)]
    # four streams: stream = 16x4 strip, units 4x4
    L["4s strips 16x4, units 4x4"] = [[(sl(4 * s, 4 * s + 4), sl(4 * k, 4 * k + 4)) for k in range(4)] for s in range(4)]
    return L


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    data = tile_valid_sets(n_tiles, seed)
    pairs = sum(int(V.sum()) for _, V in data)
    listed = sum(n for n, _ in data)
    staged = sum(V.shape[0] for _, V in data)
    print(f"tiles {len(data)}  listed entries {listed}  staged (<= tile's last index) {staged}  valid (pixel, Gaussian) pairs {pairs}")
    BODY, RED = 38, 29
    b, r, _ = map(sum, zip(*[scheme_quadrants(V) for _, V in data]))
    cost0 = BODY * b + RED * r
    print(f"{'quadrants (today)':58s} bodies {b:8d} lanes/body {pairs / b:5.1f}  reductions {r:7d}  rows {r:7d}  cost {cost0 / 1e6:7.2f} M  1.000")
    for name, sb in layouts().items():
        tot = [streams(V, sb) for _, V in data]
        b, s, rows = map(sum, zip(*tot))
        cost = BODY * b + RED * s
        print(f"{name:58s} bodies {b:8d} lanes/body {pairs / b:5.1f}  reductions {s:7d}  rows {rows:7d}  cost {cost / 1e6:7.2f} M  {cost / cost0:.3f}")


if __name__ == "__main__":
    main()
