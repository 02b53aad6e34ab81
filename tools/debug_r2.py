import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_scene, run_hip
from f3dgaus_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
sc = make_scene(P=2000, res=(64, 64), s0=0.05, view="canonical")
L.f3dg_set_option(b"render_fast", 0)
L.f3dg_set_option(b"render_kernel", 1); a = run_hip(sc, dev)
L.f3dg_set_option(b"render_kernel", int(sys.argv[1]) if len(sys.argv) > 1 else 3); b = run_hip(sc, dev)
d = np.abs(a["out_color"][0] - b["out_color"][0]).max(0)
bad = d > 0
print("pixels differing", bad.sum(), "of", bad.size)
ys, xs = np.nonzero(bad)
print("x&15 hist", np.bincount(xs & 15, minlength=16)); print("y&15 hist", np.bincount(ys & 15, minlength=16))
nc1, nc2 = a["n_contrib"][0], b["n_contrib"][0]
print("n_contrib last differ", (nc1[0] != nc2[0]).sum(), "max differ", (nc1[1] != nc2[1]).sum())
for y, x in list(zip(ys, xs))[:10]:
    print(y, x, "k1 last", nc1[0][y, x], "k2 last", nc2[0][y, x], "alpha", a["out_color"][0][7, y, x], b["out_color"][0][7, y, x])
