import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_baseline_configs_gpu import _scene, _render, _oracle_scene
from helpers import run_oracle
from f3dgaus_amd import _lib
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
dev = torch.device("cuda:0")
P, res, V = 1000000, 512, 32
g, cams, shs = _scene(P, res, V, dev)
gen = torch.Generator(device="cpu").manual_seed(11)
dpix = torch.randn(V, 9, res, res, generator=gen).to(dev); dpix[:, 7] = 0
v = 20
o = run_oracle(_oracle_scene(g, cams, shs, res, v))
go = o["oracle"].backward(dpix[v].cpu().numpy())
go2 = o["oracle"].backward(dpix[v].cpu().numpy())
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
print("oracle run-to-run:", {k: rel(go2[k], go[k]) for k in ("dL_dview2gaussian", "dL_dopacity", "dL_dcolor", "dL_dmean2D", "dL_dsh")})
bg = torch.zeros(3, device=dev)
for fast in (2, 0):
    _lib.lib().f3dg_set_option(b"render_fast", fast)
    o1, r1, w1 = _render(g, cams, shs, res, slice(v, v + 1), dev, save_aux=True)
    g1 = rasterize_backward_raw(w1, g["xyz"], shs, None, g["scaling"], g["rotation"], r1, dpix[v:v+1], 1, cams["viewmatrix"][v:v+1], cams["projmatrix"][v:v+1], cams["campos"][v:v+1], bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)
    print("fast" if fast else "exact", {"v2g": rel(g1["dL_dview2gaussian"][0].cpu().numpy(), go["dL_dview2gaussian"]), "opac": rel(g1["dL_dopacity"].cpu().numpy(), go["dL_dopacity"]),
          "col": rel(g1["dL_dcolors"][0].cpu().numpy(), go["dL_dcolor"]), "m2d": rel(g1["dL_dmeans2D"][0].cpu().numpy(), go["dL_dmean2D"]), "sh": rel(g1["dL_dsh"].cpu().numpy(), go["dL_dsh"])})
    a = g1["dL_dview2gaussian"][0].cpu().numpy().astype(np.float64); b = go["dL_dview2gaussian"].astype(np.float64)
    d = np.abs(a - b); i = np.unravel_index(d.argmax(), d.shape); print(" worst", i, a[i], b[i], "max", np.abs(b).max(), "per-column rel", (d.max(0) / np.abs(b).max(0)))
