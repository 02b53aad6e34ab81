"""Per-call cost of small batched renders (the cycle-aggregation regime: 8 views per call): wall time per call in a
back-to-back loop (check=False: no host sync) against the GPU time of the same calls from HIP events. A gap means the
host (Python + ~25 kernel launches per call) cannot keep the GPU busy."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import synthetic  # noqa: E402

dev = torch.device("cuda:0")
RES = 256
bg = torch.zeros(3, device=dev)
for V, P in ((1, 65536), (1, 196608), (3, 196608), (8, 65536), (8, 196608), (8, 589824), (12, 196608)):
    cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
    g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    kw = dict(image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"],
              rotations=g["rotation"], sh_degree=1)
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, **kw)
    n = 200
    for _ in range(10):
        f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, workspace=ws, out=out, radii=radii, check=False, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, workspace=ws, out=out, radii=radii, check=False, **kw)
    t_issue = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    print(f"P={P} V={V}: host issue {t_issue / n * 1e6:.0f} us/call, wall {t_wall / n * 1e6:.0f} us/call, GPU span {e0.elapsed_time(e1) / n * 1e3:.0f} us/call "
          f"-> {V * n / t_wall:.0f} views/s")
