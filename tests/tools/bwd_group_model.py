"""CPU model of the GROUPED lock-step backward walk: render3_bwd_kernel steps through every entry of a quadrant's list whose conservative
ellipse reaches a pixel that still composites it (one step = the alpha evaluation + the gradient terms + the wave reduction, whatever the
number of pixels taking part). Consecutive entries whose ellipses reach DISJOINT pixel sets could share the first two parts of a step (every
pixel evaluates the one entry that reaches it); this model counts, on one view (C2 recipe or the dumped real merged set), the steps of the
plain walk and of the grouped walk with groups of <= K entries, and prices both (EVAL issue slots per shared part, RED per member's reduction).
Needs the oracle; test infrastructure only.   [REAL=1 VIEW=40] python tests/tools/bwd_group_model.py [n_tiles] [s0]"""
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

KS = (1, 2, 3, 4, 6, 8)
steps = {k: 0 for k in KS}          # shared parts (alpha + gradient terms)
members = 0                          # reductions (one per entry that reaches a live pixel)
pairs = 0
size_hist = np.zeros(65, np.int64)   # pixels an entry's ellipse reaches (live pixels only)
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
    last = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64).reshape(-1)       # entries [0, last) are walked by the pixel
    pos = np.arange(n)[None, :]
    live = ell & (pos < last[:, None])
    py, px_ = np.divmod(np.arange(256), 16)
    wave = (py // 8) * 2 + (px_ // 8)
    for w in range(4):
        lanes = np.nonzero(wave == w)[0]
        m = live[lanes]                        # [64 pixels, n entries]
        idx = np.nonzero(m.any(0))[0][::-1]    # back to front
        if len(idx) == 0:
            continue
        pairs += int((hit[lanes] & m).sum())
        members += len(idx)
        cnt = m[:, idx].sum(0)
        size_hist += np.bincount(cnt, minlength=65)
        bits = [sum(1 << int(l) for l in np.nonzero(m[:, e])[0]) for e in idx]
        for K in KS:
            i = 0
            s = 0
            while i < len(bits):
                u = bits[i]
                k = 1
                while i + k < len(bits) and k < K and (u & bits[i + k]) == 0:
                    u |= bits[i + k]
                    k += 1
                i += k
                s += 1
            steps[K] += s

print("entries walked %d, contributing pairs %d (%.1f per entry)" % (members, pairs, pairs / members))
cs = np.cumsum(size_hist) / size_hist.sum()
print("pixels reached by an entry's ellipse: <=4 %.2f, <=8 %.2f, <=16 %.2f, <=32 %.2f; mean %.1f" % (cs[4], cs[8], cs[16], cs[32], (size_hist * np.arange(65)).sum() / size_hist.sum()))
EVAL, RED, RED1 = 250.0, 95.0, 70.0
base = members * (EVAL + RED1)
for K in KS:
    c = steps[K] * EVAL + members * (RED if K > 1 else RED1) + (members * 10 if K > 1 else 0)
    print("groups of <= %d: %d steps (%.2f entries per step): %.3f of the plain walk's issue slots" % (K, steps[K], members / steps[K], c / base))
