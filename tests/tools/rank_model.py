"""CPU model of the rank-packed compositing schedule (render4, f3dg_render.hip): replays the sliding half-window schedule of the
one-wave kernel on one view (C2 recipe or the dumped real merged set) and records, for every phase-2 trip ("rank": every pixel with a
pending entry pops its next one), how many of the quadrant's 64 pixels take part. From those populations it prices
  * the fused schedule (render3s): every trip = stateless part + blend in the pixel's own lane;
  * the packed schedule: trips with more than TH active pixels stay fused; the remaining ranks of a slide are packed, several ranks
    per dense trip of <= 64 (pixel, entry) pairs (stateless part, one pair per lane), followed by one short blend trip per rank.
Costs are VALU issue slots (transcendentals = 4). Needs the oracle; test infrastructure only.

  [REAL=1 VIEW=40] python tests/tools/rank_model.py [n_tiles] [s0]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import make_scene, run_oracle

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
if os.environ.get("REAL"):
    import torch
    from f3dgaus_amd import synthetic
    z = np.load(os.path.join(ROOT, "gpurun_out", "real_set.npz"))       # tools/dump_real_set.py on a GPU box
    g = {k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}
    g["features_rest"] = torch.zeros(g["xyz"].shape[0], 3, 3)
    cams = synthetic.orbit_cameras(128, resolution=256)
    vi = int(os.environ.get("VIEW", "40"))
    sc = dict(P=g["xyz"].shape[0], W=256, H=256, sh_degree=1, kernel_size=0.0, scale_modifier=1.0, tanfovx=cams["tanfovx"],
              tanfovy=cams["tanfovy"], bg=torch.zeros(3), viewmatrix=cams["viewmatrix"][vi:vi + 1], projmatrix=cams["projmatrix"][vi:vi + 1],
              campos=cams["campos"][vi:vi + 1], means3D=g["xyz"], opacities=g["opacity"], scales=g["scaling"], rotations=g["rotation"],
              shs=torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous(), colors_precomp=None)
elif os.environ.get("PIXEL"):        # the drop-in loop's one-view scene: 65,536 pixel-ordered Gaussians (bench.py --workload dropin)
    import torch
    from f3dgaus_amd import synthetic
    g = synthetic.make_pixel_gaussians(256, s0=S0, seed=0, device="cpu")
    cams = synthetic.orbit_cameras(60, resolution=256)
    vi = int(os.environ.get("VIEW", "7"))
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

slides = []          # per slide: list of active-lane counts per trip
n_slides = 0
n_waves = 0
for tile in tiles:
    r0, r1 = ranges[tile]
    ids = pl[r0:r1]
    n = len(ids)
    if n == 0:
        continue
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
    hit = p >= thr
    ell = p >= 1.11 * np.minimum(thr, 0) - 1e-3   # conservative ellipse (model)
    last = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64).reshape(-1)
    doneidx = np.full(256, n, dtype=np.int64)
    for px in range(256):
        h = np.nonzero(hit[px, last[px]:])[0]
        if len(h):
            doneidx[px] = last[px] + h[0]
    pos = np.arange(n)[None, :]
    proc = ell & (pos <= doneidx[:, None])
    py, px_ = np.divmod(np.arange(256), 16)
    wave = (py // 8) * 2 + (px_ // 8)
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
    for w in range(4):
        lanes = np.nonzero(wave == w)[0]
        qx, qy = (w % 2) * 8, (w // 2) * 8
        inq = (x0 <= qx + 7) & (x1 >= qx) & (y0 <= qy + 7) & (y1 >= qy)
        lst = np.nonzero(inq)[0]
        if len(lst) == 0:
            continue
        dq = doneidx[lanes].max()
        n_waves += 1
        pm = proc[np.ix_(lanes, lst)]
        nq = pm.shape[1]
        nxt = [np.nonzero(pm[l])[0] for l in range(64)]
        ptr = np.zeros(64, dtype=np.int64)
        ln = np.array([len(x) for x in nxt])
        s = 0
        STEP, RES = 32, 64
        while s < nq and lst[s] <= dq:
            hi = min(s + RES, nq)
            trips = []
            while True:
                cur = np.array([nxt[l][ptr[l]] if ptr[l] < ln[l] else 1 << 30 for l in range(64)])
                if not (cur < min(s + STEP, nq)).any():
                    break
                can = cur < hi
                ptr[can] += 1
                trips.append(int(can.sum()))
            slides.append(trips)
            s += STEP

print("quadrant waves %d: slides per wave %.1f" % (n_waves, len(slides) / max(n_waves, 1)))
slides_n = len(slides)
all_trips = np.array([t for s in slides for t in s])
print("slides %d, trips %d (%.2f per slide), lane-trips %d, utilisation %.3f" %
      (slides_n, len(all_trips), len(all_trips) / slides_n, all_trips.sum(), all_trips.sum() / (64.0 * len(all_trips))))
hist = np.bincount(np.minimum(all_trips, 64) // 8, minlength=9)
print("trips by active pixels (1-7, 8-15, ..., 56-63, 64):", (hist / len(all_trips)).round(3))

FUSED, DENSE, SLIDE = 95.0, 72.0, 270.0


def cost(TH, serial, maxranks, cap=64):
    c = 0.0
    dense = ranks = fused = 0
    for s in slides:
        k = 0
        while k < len(s) and s[k] > TH:
            c += FUSED
            fused += 1
            k += 1
        while k < len(s):
            fill = 0
            r = 0
            while k < len(s) and fill + s[k] <= cap and r < maxranks:
                fill += s[k]
                k += 1
                r += 1
            if r == 0:                      # a rank above the capacity: fused
                c += FUSED
                fused += 1
                k += 1
                continue
            c += DENSE + r * serial
            dense += 1
            ranks += r
    return c, fused, dense, ranks


base = FUSED * len(all_trips)
print("fused schedule: %.0f slots per slide in phase 2 (+ %.0f per slide outside)" % (base / slides_n, SLIDE))
for serial in (40.0, 48.0, 56.0):
    for maxranks in (4, 8, 64):
        for TH in (16, 24, 32, 40, 48):
            c, fu, de, ra = cost(TH, serial, maxranks)
            print("packed: blend trip %2.0f, <= %2d ranks per dense trip, fused above %2d active: %.0f per slide = %.3f of fused "
                  "(with the slide overhead %.3f); fused trips %.2f, dense %.2f, blend %.2f per slide" %
                  (serial, maxranks, TH, c / slides_n, c / base, (c + SLIDE * slides_n) / (base + SLIDE * slides_n), fu / slides_n, de / slides_n, ra / slides_n))
