"""How many (Gaussian, tile) instances would an EXACT ellipse-against-tile test save over the axis-aligned box of the conservative ellipse
that tile culling uses (csrc/f3dg_preprocess.hip)? One C2-recipe view: the conservative ellipse of every visible Gaussian restated in
numpy (as tests/tools/ellipse_margin_model.py), its box's tiles counted, and for each of them the minimum of the ellipse's quadratic form
over the tile's pixel-centre square (convex: clamp the centre, then the four edges) compared with 1. Needs the oracle; test infrastructure.

  python tests/tools/tile_exact_model.py [sigma0]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_scene, run_oracle

S0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
sc = make_scene(P=196608, res=(256, 256), s0=S0, view="oblique")
o = run_oracle(sc)
W = H = 256
fx = fy = W / (2.0 * float(sc["tanfovx"]))
X = Y = float(sc["tanfovx"])
vis = o["radii"] > 0
vg = o["view2gaussian"][vis].astype(np.float64)
opac = o["conic_opacity"][vis, 3].astype(np.float64)
ok = opac > 0
vg, opac = vg[ok], opac[ok]
thr = -np.log(255.0 * opac) - 1e-4
u = 5.9604644775390625e-08
C = vg[:, 9]
k = -2.0 * thr + 2e-3
cK = C - k
A = np.abs(vg[:, 0]) * X * X + 2 * np.abs(vg[:, 1]) * X * Y + 2 * np.abs(vg[:, 2]) * X + np.abs(vg[:, 3]) * Y * Y + 2 * np.abs(vg[:, 4]) * Y + np.abs(vg[:, 5])
Bn = np.abs(vg[:, 6]) * X + np.abs(vg[:, 7]) * Y + np.abs(vg[:, 8])
D = 1.1 * u * (6.0 * cK * A + 7.0 * Bn * Bn)
B0, B1, B2 = vg[:, 6], vg[:, 7], vg[:, 8]
m00, m01, m02 = cK * vg[:, 0] - B0 * B0, cK * vg[:, 1] - B0 * B1, cK * vg[:, 2] - B0 * B2
m11, m12, m22 = cK * vg[:, 3] - B1 * B1, cK * vg[:, 4] - B1 * B2, cK * vg[:, 5] - B2 * B2 - D
D22 = m00 * m11 - m01 * m01
good = (cK > 0) & (m00 > 0) & (m11 > 0) & (D22 > 1e-9 * np.abs(m00 * m11))
cxr, cyr = (m01 * m12 - m02 * m11) / D22, (m01 * m02 - m00 * m12) / D22
Qc = m22 + m02 * cxr + m12 * cyr
good &= Qc < 0
kk = -1.0 / Qc
a, b, c = m00 * kk / (fx * fx), 2 * m01 * kk / (fx * fy), m11 * kk / (fy * fy)
tr, det = a + c, a * c - 0.25 * b * b
lmax = 0.5 * tr + np.sqrt(np.maximum(0.25 * tr * tr - det, 0))
s = 1.001 + 0.05 * np.sqrt(lmax)
a, b, c = a / (s * s), b / (s * s), c / (s * s)
px, py = cxr * fx + W / 2.0 - 0.5, cyr * fy + H / 2.0 - 0.5
a, b, c, px, py = (t[good] for t in (a, b, c, px, py))
det = a * c - 0.25 * b * b
hx, hy = np.sqrt(c / det) * 1.0005 + 2e-3, np.sqrt(a / det) * 1.0005 + 2e-3
T = 16
x0 = np.clip(np.ceil((px - hx - (T - 1)) / T), 0, W // T).astype(int); x1 = np.clip(np.floor((px + hx) / T) + 1, 0, W // T).astype(int)
y0 = np.clip(np.ceil((py - hy - (T - 1)) / T), 0, H // T).astype(int); y1 = np.clip(np.floor((py + hy) / T) + 1, 0, H // T).astype(int)
box = np.maximum(x1 - x0, 0) * np.maximum(y1 - y0, 0)


def qmin_on_segment(ax, ay, bx, by, a, b, c):
    """minimum of a x^2 + b x y + c y^2 on the segment A + t (B - A), t in [0, 1] (offsets from the ellipse's centre)"""
    dx, dy = bx - ax, by - ay
    qa = a * dx * dx + b * dx * dy + c * dy * dy
    qb = 2 * a * ax * dx + b * (ax * dy + ay * dx) + 2 * c * ay * dy
    t = np.clip(np.where(qa > 0, -qb / (2 * np.maximum(qa, 1e-300)), 0.0), 0.0, 1.0)
    x, y = ax + t * dx, ay + t * dy
    return a * x * x + b * x * y + c * y * y


exact = 0
n = len(a)
for i in range(n):
    if box[i] <= 0:
        continue
    if box[i] == 1:
        exact += 1
        continue
    txs, tys = np.meshgrid(np.arange(x0[i], x1[i]), np.arange(y0[i], y1[i]))
    lx, ly = txs.ravel() * T - px[i], tys.ravel() * T - py[i]            # the tile's pixel centres span [l, l + 15]
    ux, uy = lx + (T - 1), ly + (T - 1)
    inside = (lx <= 0) & (ux >= 0) & (ly <= 0) & (uy >= 0)               # centre inside the tile's square
    q = np.minimum.reduce([qmin_on_segment(lx, ly, ux, ly, a[i], b[i], c[i]), qmin_on_segment(lx, uy, ux, uy, a[i], b[i], c[i]),
                           qmin_on_segment(lx, ly, lx, uy, a[i], b[i], c[i]), qmin_on_segment(ux, ly, ux, uy, a[i], b[i], c[i])])
    exact += int((inside | (q <= 1.0)).sum())
print("sigma0 = %g, one view: %d Gaussians with an ellipse; instances by the ellipse's box %d (%.2f per Gaussian); by the exact ellipse-tile "
      "test %d (%.2f): x%.3f" % (S0, n, int(box.sum()), box.sum() / n, exact, exact / n, exact / box.sum()))
