"""CPU model of a TWO-quadrants-per-wave compositing schedule: lane l owns pixel l of quadrant A and pixel l of quadrant B of one tile;
each quadrant keeps its own sliding half-window (32 + 32 entries) and slides when every lane has finished ITS older half; in a trip a
lane takes a pending entry of A's older half, else of B's older half, else of A's newer, else of B's newer. Counts wave trips against
the one-quadrant-per-wave schedule of render3s on the same data. Needs the oracle; test infrastructure only.

  [REAL=1 VIEW=40] python tests/tools/pair_model.py [n_tiles]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import make_scene, run_oracle

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if os.environ.get("REAL"):
    import torch
    from f3dgaus_amd import synthetic
    z = np.load(os.path.join(ROOT, "gpurun_out", "real_set.npz"))
    g = {k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}
    g["features_rest"] = torch.zeros(g["xyz"].shape[0], 3, 3)
    cams = synthetic.orbit_cameras(128, resolution=256)
    vi = int(os.environ.get("VIEW", "40"))
    sc = dict(P=g["xyz"].shape[0], W=256, H=256, sh_degree=1, kernel_size=0.0, scale_modifier=1.0, tanfovx=cams["tanfovx"],
              tanfovy=cams["tanfovy"], bg=torch.zeros(3), viewmatrix=cams["viewmatrix"][vi:vi + 1], projmatrix=cams["projmatrix"][vi:vi + 1],
              campos=cams["campos"][vi:vi + 1], means3D=g["xyz"], opacities=g["opacity"], scales=g["scaling"], rotations=g["rotation"],
              shs=torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous(), colors_precomp=None)
else:
    sc = make_scene(P=196608, res=(256, 256), s0=0.01, view="oblique")
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


class Quad:
    """One quadrant's sliding window: per lane the list positions it must process, in order."""
    def __init__(self, nxt, nq, last_needed):
        self.nxt, self.nq, self.ln = nxt, nq, np.array([len(x) for x in nxt])
        self.ptr = np.zeros(64, dtype=np.int64)
        self.s = 0                      # start of the older half
        self.dq = last_needed           # the quadrant is finished after this list position
        self.done = nq == 0

    def cur(self):
        return np.array([self.nxt[l][self.ptr[l]] if self.ptr[l] < self.ln[l] else 1 << 30 for l in range(64)])

    def needs_slide(self, cur):
        return not (cur < min(self.s + 32, self.nq)).any()

    def slide(self):
        self.s += 32
        if self.s >= self.nq or self.s > self.dq:
            self.done = True


single_trips = pair_trips = pairs = 0
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
    ell = p >= 1.11 * np.minimum(thr, 0) - 1e-3
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

    def quad(w):
        lanes = np.nonzero(wave == w)[0]
        qx, qy = (w % 2) * 8, (w // 2) * 8
        inq = (x0 <= qx + 7) & (x1 >= qx) & (y0 <= qy + 7) & (y1 >= qy)
        lst = np.nonzero(inq)[0]
        pm = proc[np.ix_(lanes, lst)] if len(lst) else np.zeros((64, 0), bool)
        nxt = [np.nonzero(pm[l])[0] for l in range(64)]
        dq = int(np.searchsorted(lst, doneidx[lanes].max(), side="right")) if len(lst) else 0
        return nxt, pm.shape[1], dq, int(pm.sum())

    for wa, wb in ((0, 3), (1, 2)):          # diagonal quadrants of the tile share a wave
        qs = []
        for w in (wa, wb):
            nxt, nq, dq, npairs = quad(w)
            pairs += npairs
            # --- the single-quadrant schedule (render3s)
            q = Quad(nxt, nq, dq)
            while not q.done:
                hi = min(q.s + 64, q.nq)
                while True:
                    cur = q.cur()
                    if q.needs_slide(cur):
                        break
                    can = cur < hi
                    q.ptr[can] += 1
                    single_trips += 1
                q.slide()
            qs.append(Quad(nxt, nq, dq))
        A, B = qs
        # --- two quadrants, one wave
        while not (A.done and B.done):
            ca = A.cur() if not A.done else np.full(64, 1 << 30)
            cb = B.cur() if not B.done else np.full(64, 1 << 30)
            if not A.done and A.needs_slide(ca):
                A.slide()
                continue
            if not B.done and B.needs_slide(cb):
                B.slide()
                continue
            ha = min(A.s + 64, A.nq) if not A.done else 0
            hb = min(B.s + 64, B.nq) if not B.done else 0
            olda = ca < (A.s + 32 if not A.done else 0)
            oldb = cb < (B.s + 32 if not B.done else 0)
            ina, inb = ca < ha, cb < hb
            take_a = olda | (~oldb & ina)
            take_b = ~take_a & inb
            A.ptr[take_a] += 1
            B.ptr[take_b] += 1
            pair_trips += 1

print("pairs %d; one quadrant per wave: %d trips (utilisation %.3f); two quadrants per wave: %d trips (utilisation %.3f) = %.3f of the trips" %
      (pairs, single_trips, pairs / (64.0 * single_trips), pair_trips, pairs / (64.0 * pair_trips), pair_trips / single_trips))
for name, fused, two in (("nine channels", 95.0, 136.0), ("rgb + depth + alpha", 70.0, 91.0)):
    print("  %s: %.0f slots per trip -> %.0f with two pixel states per lane: phase 2 at %.3f of today's" % (name, fused, two, two * pair_trips / (fused * single_trips)))
