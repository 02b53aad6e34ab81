"""One view per call WITH a backward (the shape of the reference's training loop: one render_predicted_more_v2_gof + loss.backward() per
view): forward with auxiliary planes + f3dg_backward on 65,536 pixel-ordered Gaussians, with the forward on the general path
(small_path_aux 0: 27 launches) and on the small-call path (default).   python tools/bench_one_view_train.py"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import _lib, synthetic  # noqa: E402
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw  # noqa: E402

dev = torch.device("cuda:0")
RES, V = 256, 60
g = synthetic.make_pixel_gaussians(RES, s0=0.01, seed=0, device=dev)
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.zeros(3, device=dev)
dpix = torch.randn(1, 9, RES, RES, generator=torch.Generator().manual_seed(11)).to(dev)
dpix[:, 7] = 0
kw = dict(image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"], rotations=g["rotation"],
          sh_degree=1, save_aux=True)
L = _lib.lib()
for item in filter(None, os.environ.get("F3DG_OPTIONS", "").split(",")):
    _lib.check(L.f3dg_set_option(item.split("=")[0].strip().encode(), int(item.split("=")[1])), "f3dg_set_option")
for aux_small in (0, 1):
    L.f3dg_set_option(b"small_path", 2)
    L.f3dg_set_option(b"small_path_aux", aux_small)
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"][:1], cams["projmatrix"][:1], cams["campos"][:1], bg, **kw)

    def step(v):
        vm, pm, cp = cams["viewmatrix"][v:v + 1], cams["projmatrix"][v:v + 1], cams["campos"][v:v + 1]
        f3d.rasterize_views(g["xyz"], g["opacity"], vm, pm, cp, bg, workspace=ws, out=out, radii=radii, check=False, **kw)
        return rasterize_backward_raw(ws, g["xyz"], shs, None, g["scaling"], g["rotation"], radii, dpix, 1, vm, pm, cp, bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)
    for v in range(10):
        step(v)
    torch.cuda.synchronize()
    L.f3dg_debug_launch_count(1)
    t0 = time.perf_counter()
    for v in range(V):
        step(v)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / V
    launches = int(L.f3dg_debug_launch_count(1)) / V
    L.f3dg_profile_enable(1)
    for v in range(V):
        step(v)
    torch.cuda.synchronize()
    L.f3dg_profile_enable(0)
    st = (C.c_double * 5)()
    nc = C.c_int(0)
    L.f3dg_profile_collect(st, C.byref(nc))
    print("small_path_aux %d (%s): %.0f us per forward + backward, %.1f launches; stages (us): projection %.1f, binning %.1f, compositing %.1f, compositing backward %.1f, "
          "per-Gaussian backward %.1f" % (aux_small, L.f3dg_debug_last_render_kernel().decode().split("<")[0], 1e6 * dt, launches,
                                         *(1e3 * st[k] / V for k in range(5))), flush=True)
L.f3dg_set_option(b"small_path_aux", 1)
