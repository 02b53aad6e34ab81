"""CPU model of the one-wave (8x8 quadrant) compositing kernel: phase-2 trip counts, staged entries and phase-1 tests per
view of the C2 recipe, for several window sizes, next to render2's shape (4 waves / tile, per-4x4-block lists, 64-entry
windows inside 192-entry rounds). The conservative ellipse is modelled as alpha >= 1/255 with the threshold exponent
scaled by 1.11 (its measured area ratio). Needs the oracle; test infrastructure only.

  python tests/tools/wave1_model.py [n_tiles] [s0]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import make_scene, run_oracle

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 48
S0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
if os.environ.get("REAL"):
    # the merged set of the real image (tests/real_data.py: needs a HIP device for the predictor and the cycle aggregation), one
    # view of its final orbit: how do the schedules fare on pixel-aligned splats on a real depth map?
    import torch
    from f3dgaus_amd import synthetic
    dump = os.path.join(ROOT, "gpurun_out", "real_set.npz")       # written on a GPU box by tools/dump_real_set.py (colour is irrelevant here)
    if not torch.cuda.is_available() and os.path.exists(dump):
        z = np.load(dump)
        g = {k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}
        g["features_rest"] = torch.zeros(g["xyz"].shape[0], 3, 3)
    else:
        from real_data import real_merged_set
        g = {k: v.cpu() for k, v in real_merged_set(torch.device("cuda:0")).items()}
    cams = synthetic.orbit_cameras(128, resolution=256)
    vi = int(os.environ.get("VIEW", "40"))
    sc = dict(P=g["xyz"].shape[0], W=256, H=256, sh_degree=1, kernel_size=0.0, scale_modifier=1.0, tanfovx=cams["tanfovx"],
              tanfovy=cams["tanfovy"], bg=torch.zeros(3), viewmatrix=cams["viewmatrix"][vi:vi + 1], projmatrix=cams["projmatrix"][vi:vi + 1],
              campos=cams["campos"][vi:vi + 1], means3D=g["xyz"], opacities=g["opacity"], scales=g["scaling"], rotations=g["rotation"],
              shs=torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous(), colors_precomp=None)
else:
    sc = make_scene(P=196608, res=(256, 256), s0=S0, view="oblique")
o = run_oracle(sc)
W = H = 256
f32 = np.float32
fx = float(f32(W) / (f32(2.0) * f32(sc["tanfovx"])))
v64 = o["view2gaussian"].astype(np.float64)
opac = o["conic_opacity"][:, 3].astype(np.float64)
ranges, pl = o["ranges"], o["point_list"]
nc = o["n_contrib"][0]
rng = np.random.default_rng(0)
tiles = rng.choice(256, NT, replace=False)

WINS = (64, 128, 192, 256)
tot = {("w1", w): dict(trips=0, staged=0, p1=0) for w in WINS}
tot["r2"] = dict(trips=0, staged=0)
pairs_total = 0
tile_entries = 0
quad_entries = 0
for tile in tiles:
    r0, r1 = ranges[tile]
    ids = pl[r0:r1]
    n = len(ids)
    tile_entries += n
    ty, tx = divmod(tile, 16)
    ys, xs = np.meshgrid(np.arange(ty * 16, ty * 16 + 16), np.arange(tx * 16, tx * 16 + 16), indexing="ij")
    rx = ((xs + 0.5 - 128) / fx).reshape(-1, 1)
    ry = ((ys + 0.5 - 128) / fx).reshape(-1, 1)
    v = v64[ids][None]
    n0 = v[..., 0] * rx + v[..., 1] * ry + v[..., 2]
    n1 = v[..., 1] * rx + v[..., 3] * ry + v[..., 4]
    n2 = v[..., 2] * rx + v[..., 4] * ry + v[..., 5]
    a = rx * n0 + ry * n1 + n2
    b = v[..., 6] * rx + v[..., 7] * ry + v[..., 8]
    p = np.minimum(-0.5 * (v[..., 9] - b * b / a), 0)
    op = np.maximum(opac[ids][None], 1e-12)
    thr = np.log(1.0 / (255.0 * op))
    hit = p >= thr                                # alpha >= 1/255             [256 px, n]
    ell = p >= 1.11 * np.minimum(thr, 0) - 1e-3   # conservative ellipse (model)
    last = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64).reshape(-1)
    doneidx = np.full(256, n, dtype=np.int64)     # tile-list position whose test saturates the pixel (n: never)
    for px in range(256):
        h = np.nonzero(hit[px, last[px]:])[0]
        if len(h):
            doneidx[px] = last[px] + h[0]
    pos = np.arange(n)[None, :]
    proc = ell & (pos <= doneidx[:, None])        # pairs phase 2 walks
    pairs_total += int(proc.sum())
    py, px_ = np.divmod(np.arange(256), 16)
    wave = (py // 8) * 2 + (px_ // 8)
    blk = (py // 4) * 4 + (px_ // 4)
    # box of the ellipse inside the tile, from the passing pixel centres
    anyx = np.zeros((16, n), bool)
    anyy = np.zeros((16, n), bool)
    for c in range(16):
        anyx[c] = ell[px_ == c].any(0)
        anyy[c] = ell[py == c].any(0)
    has = anyx.any(0)
    x0 = np.where(has, anyx.argmax(0), 99)
    x1 = np.where(has, 15 - anyx[::-1].argmax(0), -1)
    y0 = np.where(has, anyy.argmax(0), 99)
    y1 = np.where(has, 15 - anyy[::-1].argmax(0), -1)

    # ---- render2 shape: rounds of 192 tile entries, per-block lists by box, 64-entry windows of the block lists
    R2 = 192
    for e0 in range(0, n, R2):
        if not (doneidx >= e0).any():
            break
        tot["r2"]["staged"] += min(R2, n - e0)
        e1 = min(e0 + R2, n)
        for w in range(4):
            lanes = np.nonzero(wave == w)[0]
            # per block list positions
            trip_w = {}
            for bb in np.unique(blk[lanes]):
                bx, by = (bb % 4) * 4, (bb // 4) * 4
                inb = (x0[e0:e1] <= bx + 3) & (x1[e0:e1] >= bx) & (y0[e0:e1] <= by + 3) & (y1[e0:e1] >= by)
                lst = np.nonzero(inb)[0] + e0
                bl = np.nonzero(blk == bb)[0]
                if not (doneidx[bl] >= e0).any():
                    continue
                for k, s in enumerate(range(0, len(lst), 64)):
                    c = proc[np.ix_(bl, lst[s:s + 64])].sum(1).max() if len(lst) else 0
                    trip_w[k] = max(trip_w.get(k, 0), int(c))
            tot["r2"]["trips"] += sum(trip_w.values())

    # ---- one wave per quadrant
    for w in range(4):
        lanes = np.nonzero(wave == w)[0]
        qx, qy = (w % 2) * 8, (w // 2) * 8
        inq = (x0 <= qx + 7) & (x1 >= qx) & (y0 <= qy + 7) & (y1 >= qy)
        lst = np.nonzero(inq)[0]
        quad_entries += len(lst)
        dq = doneidx[lanes].max()                 # the quadrant is finished after this tile position
        for wn in WINS:
            t = tot[("w1", wn)]
            for s in range(0, len(lst), wn):
                if lst[s] > dq:
                    break
                sub = lst[s:s + wn]
                t["staged"] += len(sub)
                t["p1"] += len(sub) * 64
                cnts = proc[np.ix_(lanes, sub)].sum(1)
                mx = int(cnts.max())
                t["trips"] += mx
                dead = doneidx[lanes] < sub[0]
                for k_ in ("lost_dead", "lost_alive", "lost_block"):
                    t.setdefault(k_, 0)
                t["lost_dead"] += int(mx * dead.sum())
                t["lost_alive"] += int((mx - cnts[~dead]).sum())
                bm = 0                            # if each 4x4 block could run at its own maximum
                for bb in np.unique(blk[lanes]):
                    bm += int(cnts[blk[lanes] == bb].max()) * 16
                t["lost_block"] += mx * 64 - bm

        # sliding window: 64 entries resident, slides by STEP when every lane has consumed the oldest STEP entries
        for STEP, RES in ((32, 64), (16, 64), (32, 96), (32, 128), (64, 128), (32, 256), (32, 512), (32, 1 << 20)):
            key = ("slide", STEP, RES)
            t = tot.setdefault(key, dict(trips=0, staged=0, slides=0))
            if len(lst) == 0:
                continue
            pm = proc[np.ix_(lanes, lst)]                     # [64, n_q] passing & before saturation
            nq = pm.shape[1]
            nxt = [np.nonzero(pm[l])[0] for l in range(64)]   # per lane: positions (in quadrant list) it must process, in order
            ptr = np.zeros(64, dtype=np.int64)
            s = 0
            while s < nq and lst[s] <= dq:
                hi = min(s + RES, nq)
                t["staged"] += min(STEP, nq - s) if s > 0 else min(RES, nq)
                t["slides"] += 1
                # trips until every lane has no unprocessed entry < s + STEP
                while True:
                    cur = np.array([nxt[l][ptr[l]] if ptr[l] < len(nxt[l]) else 1 << 30 for l in range(64)])
                    if not (cur < min(s + STEP, nq)).any():
                        break
                    can = cur < hi
                    ptr[can] += 1
                    t["trips"] += 1
                s += STEP

scale = 256.0 / NT
print("tiles %d  tile entries/view %.3g  quadrant entries/view %.3g (x%.2f)  phase-2 pairs/view %.3g" %
      (NT, tile_entries * scale, quad_entries * scale, quad_entries / tile_entries, pairs_total * scale))
ideal = pairs_total / 64.0
t = tot["r2"]
print("render2 shape: staged %.3g/view, trips %.3g, lane utilisation %.3f" % (t["staged"] * scale, t["trips"] * scale, ideal / t["trips"]))
for STEP, RES in ((32, 64), (16, 64), (32, 96), (32, 128), (64, 128), (32, 256), (32, 512), (32, 1 << 20)):
    t = tot[("slide", STEP, RES)]
    print("one wave, %d resident entries sliding by %d: staged %.3g/view, trips %.3g, lane utilisation %.3f, slides %.3g" %
          (RES, STEP, t["staged"] * scale, t["trips"] * scale, ideal / t["trips"], t["slides"] * scale))
for wn in WINS:
    t = tot[("w1", wn)]
    print("one wave, %3d-entry windows: staged %.3g/view (gather x%.2f of render2), trips %.3g, lane utilisation %.3f, "
          "phase-1 tests %.3g; lost lane-trips: finished pixels %.3f, live pixels %.3f (of all: between blocks %.3f)" % (wn, t["staged"] * scale, t["staged"] / tot["r2"]["staged"], t["trips"] * scale, ideal / t["trips"], t["p1"] * scale,
           t["lost_dead"] / (64 * t["trips"]), t["lost_alive"] / (64 * t["trips"]), t["lost_block"] / (64 * t["trips"])))
