"""How much of the conservative ellipse (csrc/f3dg_preprocess.hip: cull_conic + conservative_ellipse) is margin? For one C2-recipe view the
area of every visible Gaussian's ellipse -- the number of (pixel, entry) pairs phase 1 lets into phase 2 is proportional to it -- with
the margins as built and with each of them reduced, against the exact level set alpha >= 1/255 (no margin at all). numpy float64
restatement of the device code's formulas; needs the oracle. Test infrastructure only.

  python tests/tools/ellipse_margin_model.py [sigma0]
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


def areas(k_extra, d_scale, s_rel, s_px, local=False):
    C = vg[:, 9]
    k = -2.0 * thr + k_extra
    cK = C - k
    if local:       # bound |x|, |y| by the Gaussian's own neighbourhood instead of the image corner (model: centre ray + 25 % of the field)
        B0, B1, B2 = vg[:, 6], vg[:, 7], vg[:, 8]
        m00, m01, m02 = cK * vg[:, 0] - B0 * B0, cK * vg[:, 1] - B0 * B1, cK * vg[:, 2] - B0 * B2
        m11, m12 = cK * vg[:, 3] - B1 * B1, cK * vg[:, 4] - B1 * B2
        D22 = m00 * m11 - m01 * m01
        cx0, cy0 = (m01 * m12 - m02 * m11) / D22, (m01 * m02 - m00 * m12) / D22
        Xl, Yl = np.minimum(np.abs(cx0) + 0.1 * X, X), np.minimum(np.abs(cy0) + 0.1 * Y, Y)
    else:
        Xl, Yl = X, Y
    A = np.abs(vg[:, 0]) * Xl * Xl + 2 * np.abs(vg[:, 1]) * Xl * Yl + 2 * np.abs(vg[:, 2]) * Xl + np.abs(vg[:, 3]) * Yl * Yl + 2 * np.abs(vg[:, 4]) * Yl + np.abs(vg[:, 5])
    Bn = np.abs(vg[:, 6]) * Xl + np.abs(vg[:, 7]) * Yl + np.abs(vg[:, 8])
    D = d_scale * 1.1 * u * (6.0 * cK * A + 7.0 * Bn * Bn)
    B0, B1, B2 = vg[:, 6], vg[:, 7], vg[:, 8]
    m00, m01, m02 = cK * vg[:, 0] - B0 * B0, cK * vg[:, 1] - B0 * B1, cK * vg[:, 2] - B0 * B2
    m11, m12, m22 = cK * vg[:, 3] - B1 * B1, cK * vg[:, 4] - B1 * B2, cK * vg[:, 5] - B2 * B2 - D
    D22 = m00 * m11 - m01 * m01
    good = (cK > 0) & (m00 > 0) & (m11 > 0) & (D22 > 1e-9 * np.abs(m00 * m11))
    cx, cy = (m01 * m12 - m02 * m11) / D22, (m01 * m02 - m00 * m12) / D22
    Qc = m22 + m02 * cx + m12 * cy
    good &= Qc < 0
    kk = -1.0 / Qc
    a, b, c = m00 * kk / (fx * fx), 2 * m01 * kk / (fx * fy), m11 * kk / (fy * fy)
    det, tr = a * c - 0.25 * b * b, a + c
    lmax = 0.5 * tr + np.sqrt(np.maximum(0.25 * tr * tr - det, 0))
    s = s_rel + s_px * np.sqrt(lmax)
    area = np.pi / np.sqrt(np.maximum(det, 1e-300)) * s * s
    return area, good


exact, g0 = areas(0.0, 0.0, 1.0, 0.0)
rows = [("as built (k + 2e-3, D at the image corner, s = 1.001 + 0.05 px / semi-minor axis)", (2e-3, 1.0, 1.001, 0.05)),
        ("  s = 1.001 + 0.005 px", (2e-3, 1.0, 1.001, 0.005)),
        ("  s = 1.0003 + 0.002 px", (2e-3, 1.0, 1.0003, 0.002)),
        ("  D = 0 (the float32 error bound of a, b dropped: NOT valid, shows its share)", (2e-3, 0.0, 1.001, 0.05)),
        ("  D with |x|, |y| bounded near the Gaussian (model)", (2e-3, 1.0, 1.001, 0.05, True)),
        ("  both: s = 1.001 + 0.005 px and the local D", (2e-3, 1.0, 1.001, 0.005, True))]
print("sigma0 = %g: %d visible Gaussians; total exact level-set area %.4g px^2" % (S0, int(g0.sum()), exact[g0].sum()))
for name, args in rows:
    ar, g = areas(*args)
    g &= g0
    print("%-92s area / exact = %.4f" % (name, ar[g].sum() / exact[g].sum()))
