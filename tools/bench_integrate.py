"""Timing of GaussianRasterizer_GOF.integrate on the GPU box at mesh-extraction-like sizes (visualize.py:449-505 calls it
once per view with ~1e6 tetrahedra vertices): P Gaussians of the C2 recipe at 256x256, PN points scattered around them.
Prints the whole-call time (blocking call, like the reference) and per-kernel times from HIP events."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras, synthetic  # noqa: E402
from f3dgaus_amd.diff_gof_rasterization import GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF  # noqa: E402

dev = torch.device("cuda:0")
RES = 256
for P, PN, s0 in ((196608, 1_000_000, 0.01), (589824, 1_000_000, 0.01), (196608, 1_000_000, 0.05)):
    g = synthetic.make_gaussians(P, s0=s0, seed=0)
    cams = synthetic.orbit_cameras(8, resolution=RES, include_canonical=True)
    gen = torch.Generator().manual_seed(1)
    pts = (g["xyz"][torch.randint(0, P, (PN,), generator=gen)] + 0.03 * torch.randn(PN, 3, generator=gen)).to(dev)
    shs = torch.cat([g["features_dc"], g["features_rest"]], dim=1).to(dev)
    rs = GaussianRasterizationSettings_GOF(
        image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], kernel_size=0.0,
        subpixel_offset=torch.empty(0, device=dev), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
        viewmatrix=cams["viewmatrix"][3].to(dev), projmatrix=cams["projmatrix"][3].to(dev), sh_degree=1,
        campos=cams["campos"][3].to(dev), prefiltered=False, debug=False)
    r = GaussianRasterizer_GOF(rs)
    args = dict(points3D=pts, means3D=g["xyz"].to(dev), means2D=None, opacities=g["opacity"].to(dev), shs=shs,
                scales=g["scaling"].to(dev), rotations=g["rotation"].to(dev))
    for _ in range(2):
        color, ai, ci, radii = r.integrate(**args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        color, ai, ci, radii = r.integrate(**args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"integrate P={P} sigma0={s0} PN={PN} @{RES}^2: {dt * 1e3:.2f} ms/call | points in image {int(color[8].sum().item())}, "
          f"max points/pixel {int(color[8].max().item())}, mean alpha_integrated {ai.mean().item():.4f}")
