"""How far the distortion channel of the split-pixel mode (and of the default fast kernel) lies from the oracle where it is well conditioned
(fixture F12: values above 1e-4): quantiles of the relative deviation."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import make_scene, run_oracle
from test_raster_forward_gpu import SCENES
from test_scan_mode_gpu import _render
dev = torch.device("cuda:0")
for name in ("F12_depth_spread", "F9_long_tile_lists", "F2_oblique_aniso"):
    scene = make_scene(**SCENES[name])
    o = run_oracle(scene, view=0)["out_color"]
    big = np.abs(o[8]) > 1e-4
    for label, scan, th in (("fast render3s", False, None), ("scan th=64", True, 64), ("scan th=12", True, 12)):
        out, k = _render(scene, dev, scan=scan, th=th)
        d = np.abs(out[0, 8] - o[8])
        rel = d[big] / np.abs(o[8][big]) if big.any() else np.array([0.0])
        print(name, label, "big px", int(big.sum()), "rel q50 %.2e q99 %.2e q99.9 %.2e max %.2e" % tuple(np.quantile(rel, [0.5, 0.99, 0.999, 1.0])),
              "| all px abs q99.9 %.2e max %.2e" % tuple(np.quantile(d, [0.999, 1.0])), "| rgb max %.2e" % np.abs(out[0, :3] - o[:3]).max())
