"""C5-shaped stress run (BASELINE config 5): 1,000,000 Gaussians, V views @512x512, forward + backward with random dL_dpix
on channels 0-6 and 8. Checks finiteness / known answers and prints timings (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import synthetic  # noqa: E402
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw  # noqa: E402

P = int(os.environ.get("P", 1000000)); V = int(os.environ.get("V", 8)); RES = 512
dev = torch.device("cuda:0")
g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.zeros(3, device=dev)
kw = dict(image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"],
          rotations=g["rotation"], sh_degree=1, save_aux=True)
out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, **kw)
print("instances", ws.num_rendered, "R/P/view", ws.num_rendered / P / V, "workspace GB", ws.nbytes / 1e9)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, workspace=ws, check=False, **kw)
torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 3
dpix = torch.randn(V, 9, RES, RES, device=dev); dpix[:, 7] = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
gr = rasterize_backward_raw(ws, g["xyz"], shs, None, g["scaling"], g["rotation"], radii, dpix, 1, cams["viewmatrix"], cams["projmatrix"],
                            cams["campos"], bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)
torch.cuda.synchronize(); bwd = time.perf_counter() - t0
assert torch.isfinite(out).all()
for k, v in gr.items():
    assert torch.isfinite(v).all(), k
assert float(gr["dL_dconic"].abs().max()) == 0 and float(gr["dL_dcov3D"].abs().max()) == 0
print(f"forward {fwd * 1e3:.1f} ms for {V} views ({V / fwd:.0f} views/s), backward {bwd * 1e3:.1f} ms; alpha mean {float(out[:, 7].mean()):.3f}")
print({k: float(v.abs().max()) for k, v in gr.items()})
