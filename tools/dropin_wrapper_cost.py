"""What the drop-in wrapper's own torch kernels cost per one-view call (deferred status on): the loop as it is, with the SH concatenation
taken out (precomputed), and with the per-call zeros tensor taken out as well.   python tools/dropin_wrapper_cost.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras, gaussian_renderer as gr, synthetic  # noqa: E402

dev = torch.device("cuda:0")
RES, V = 256, 60
cfg = cameras.default_cfg(RES)
g = synthetic.make_pixel_gaussians(RES, s0=0.01, seed=0, device=dev)
pc = {k: g[k][None] for k in ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest")}
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
wv, fp, cc = cams["viewmatrix"].unsqueeze(1), cams["projmatrix"].unsqueeze(1), cams["campos"].unsqueeze(1)
bg = torch.zeros(1, 3, device=dev)
frames = torch.empty((V, 3, RES, RES), device=dev)


def loop():
    with torch.no_grad():
        for th in range(V):
            o = f3d.render_predicted_more_v2_gof(pc, 0, wv[th:th + 1].contiguous(), fp[th:th + 1].contiguous(), cc[th:th + 1].contiguous(), bg, cfg)
            frames[th] = o["render"]


def timed(label):
    f3d.set_deferred_status(True)
    for _ in range(3):
        loop()
    f3d.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        loop()
    f3d.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * V)
    f3d.set_deferred_status(False)
    print("%-60s %.1f us per call, %.0f views/s" % (label, 1e6 * dt, 1.0 / dt), flush=True)


timed("as shipped")
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
orig_cat = gr._cat_sh
gr._cat_sh = lambda dc, rest: shs
timed("SH concatenation precomputed")
orig_zeros = torch.zeros_like
z = torch.zeros_like(g["xyz"])
torch.zeros_like = lambda t, **kw: z if t.shape == z.shape else orig_zeros(t, **kw)
timed("... and the per-call zeros tensor shared")
torch.zeros_like = orig_zeros
gr._cat_sh = orig_cat
