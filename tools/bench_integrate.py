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
CONFIGS = ((196608, 1_000_000, 0.01), (589824, 1_000_000, 0.01), (196608, 1_000_000, 0.05))
if os.environ.get("CFG"):            # one configuration only (so that rocprofv3 --stats averages belong to it), no sweep
    CONFIGS = (CONFIGS[int(os.environ["CFG"])],)
for P, PN, s0 in CONFIGS:
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
    # algorithmic bytes of the per-pixel pass (integrate_pass1_kernel), SURVEY 8d's compositing formula on this call's instances:
    # 72 B per (Gaussian, tile) instance of the reference's lists + 36 B per pixel written + 2 B per contributor id it records
    f3d.set_option("tile_cull", 0)
    R = f3d.rasterize_views(args["means3D"], args["opacities"], rs.viewmatrix[None], rs.projmatrix[None], rs.campos[None], rs.bg,
                            image_height=RES, image_width=RES, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, sh=shs, scales=args["scales"],
                            rotations=args["rotations"], sh_degree=1)[2].num_rendered
    f3d.set_option("tile_cull", 1)
    print(f"  pass 1 of that call: R = {R} instances -> {72 * R + 36 * RES * RES} algorithmic bytes + the contributor lists")

if os.environ.get("CFG"):
    sys.exit(0)
# ---- the mesh-extraction sweep (visualize.py:449-507): 9 point sets x V cameras of the merged 589,824 Gaussians
from f3dgaus_amd import cameras as _cams  # noqa: E402
P, PN, V, SETS = 589824, 1_000_000, int(os.environ.get("V", 16)), 9
cfg = _cams.default_cfg(RES)
g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
oc = synthetic.orbit_cameras(V, resolution=RES, device=dev)
pc = {"xyz": g["xyz"][None], "opacity": g["opacity"][None], "scaling": g["scaling"][None], "rotation": g["rotation"][None],
      "features_dc": g["features_dc"][None], "features_rest": g["features_rest"][None]}
gen = torch.Generator().manual_seed(2)
sets = [(g["xyz"][torch.randint(0, P, (PN,), generator=gen).to(dev)] + 0.03 * torch.randn(PN, 3, generator=gen).to(dev)) for _ in range(2)]
bg = torch.zeros(3, device=dev)
for rep in range(2):                    # second repetition is timed (workspace sized, caches warm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ref = torch.ones(PN, device=dev)
    for v in range(V):                  # reference-shaped: the whole integrate per (camera, point set)
        o = f3d.render_predicted_more_v2_gof_in(sets[0], pc, 0, oc["viewmatrix"][v], oc["projmatrix"][v], oc["campos"][v], bg, cfg)
        ref = torch.min(ref, o["alpha_integrated"])
    torch.cuda.synchronize(); t_loop = (time.perf_counter() - t0) / V
for rep in range(2):
    for k in (1, 4, 16):                # cameras prepared per launch sequence (f3dg_integrate_prepare_batched); 16 is the default
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sweep = f3d.AlphaSweep(pc, 0, oc["viewmatrix"], oc["projmatrix"], oc["campos"], bg, cfg, max_points=PN, cameras_per_call=k)
        torch.cuda.synchronize(); t_prep = (time.perf_counter() - t0) / V
        if rep:
            print(f"AlphaSweep of {V} cameras, {k} per call: {t_prep * 1e3:.2f} ms per camera")
a = sweep(sets[0]); torch.cuda.synchronize()
assert torch.equal(a, ref)
t0 = time.perf_counter()
for k in range(SETS):
    a = sweep(sets[k % 2])
torch.cuda.synchronize(); t_pts = (time.perf_counter() - t0) / (SETS * V)
print(f"mesh-extraction sweep, P={P} PN={PN} @{RES}^2, {V} cameras resident ({sweep.nbytes / 1e9:.2f} GB): reference-shaped call "
      f"{t_loop * 1e3:.2f} ms per (camera, point set); prepared: {t_prep * 1e3:.2f} ms per camera once + {t_pts * 1e3:.2f} ms per "
      f"(camera, point set) -> {SETS} sets x 129 cameras: {SETS * 129 * t_loop:.2f} s vs {129 * t_prep + SETS * 129 * t_pts:.2f} s")
