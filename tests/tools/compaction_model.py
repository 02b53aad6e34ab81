"""CPU model: phase-2 trip counts of the compositing kernel per tile (C2 recipe) under three lane assignments:
(i) fixed 8x8 quadrant per wave (render2), (ii) 4x4 blocks repacked into waves at round boundaries, (iii) single pixels
repacked. Whole-round (256-entry) windows. Needs the oracle; test infrastructure only."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import make_scene, run_oracle

sc = make_scene(P=196608, res=(256, 256), s0=0.01, view="oblique"); o = run_oracle(sc)
W = H = 256; f32 = np.float32
fx = float(f32(W) / (f32(2.0) * f32(sc["tanfovx"])))
v64 = o["view2gaussian"].astype(np.float64); opac = o["conic_opacity"][:, 3]
ranges, pl = o["ranges"], o["point_list"]; nc = o["n_contrib"][0]
rng = np.random.default_rng(0); tiles = rng.choice(256, 32, replace=False)
R = 256
tot = dict(fixed=0, block=0, pixel=0, ideal=0.0, pairs=0, fixed64=0)
for tile in tiles:
    r0, r1 = ranges[tile]; ids = pl[r0:r1]; n = len(ids)
    ty, tx = divmod(tile, 16)
    ys, xs = np.meshgrid(np.arange(ty * 16, ty * 16 + 16), np.arange(tx * 16, tx * 16 + 16), indexing="ij")
    rx = ((xs + 0.5 - 128) / fx).reshape(-1, 1); ry = ((ys + 0.5 - 128) / fx).reshape(-1, 1)
    v = v64[ids][None]
    n0 = v[..., 0] * rx + v[..., 1] * ry + v[..., 2]; n1 = v[..., 1] * rx + v[..., 3] * ry + v[..., 4]; n2 = v[..., 2] * rx + v[..., 4] * ry + v[..., 5]
    a = rx * n0 + ry * n1 + n2; b = v[..., 6] * rx + v[..., 7] * ry + v[..., 8]
    p = -0.5 * (v[..., 9] - b * b / a)
    hit = (opac[ids][None] * np.exp(np.minimum(p, 0)) >= 1 / 255)            # [256 px, n]
    last = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64).reshape(-1)
    doneidx = np.full(256, n, dtype=np.int64)
    for px in range(256):
        h = np.nonzero(hit[px, last[px]:])[0]
        if len(h): doneidx[px] = last[px] + h[0]
    proc = hit & (np.arange(n)[None, :] <= doneidx[:, None])                  # pairs phase 2 executes
    # pixel -> (wave, group) of render2
    py, px_ = np.divmod(np.arange(256), 16)
    blk = (py // 4) * 4 + (px_ // 4)                                           # 16 blocks
    wave = (py // 8) * 2 + (px_ // 8)
    for e0 in range(0, n, R):
        alive = doneidx >= e0
        if not alive.any(): break
        cnt = proc[:, e0:e0 + R].sum(1)
        tot["pairs"] += cnt.sum(); tot["ideal"] += cnt.sum() / 64.0
        tot["fixed"] += sum(cnt[wave == w].max() for w in range(4))
        for w in range(4):       # 64-entry windows (approximation: windows over the tile list, not the per-block lists)
            for s0 in range(e0, min(e0 + R, n), 64):
                tot["fixed64"] += proc[wave == w, s0:s0 + 64].sum(1).max()
        ab = [bb for bb in range(16) if alive[blk == bb].any()]
        bm = [cnt[blk == bb].max() for bb in ab]
        # greedy: sort blocks by max desc, pack 4 per wave
        bm.sort(reverse=True)
        tot["block"] += sum(bm[i] for i in range(0, len(bm), 4))
        pm = np.sort(cnt[alive])[::-1]
        tot["pixel"] += sum(pm[i] for i in range(0, len(pm), 64))
print({k: float(v) for k, v in tot.items()})
print("util fixed %.3f (64-windows %.3f) block %.3f pixel %.3f" % (tot["ideal"] / tot["fixed"], tot["ideal"] / tot["fixed64"], tot["ideal"] / tot["block"], tot["ideal"] / tot["pixel"]))
