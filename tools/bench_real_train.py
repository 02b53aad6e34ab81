"""Forward with auxiliary planes (the call a backward follows: the reference's arithmetic) + backward on the real image's merged set, V views of
its orbit per call: stage times with render3s (render_pack 0) and with the rank-packed kernel (the default).  python tools/bench_real_train.py [V]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import _lib, synthetic  # noqa: E402
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw  # noqa: E402
from real_data import real_merged_set  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
g = real_merged_set(dev)
cams = synthetic.orbit_cameras(128, resolution=256, device=dev)
sel = torch.arange(0, 128, 128 // V, device=dev)[:V]
vm, pm, cp = cams["viewmatrix"][sel].contiguous(), cams["projmatrix"][sel].contiguous(), cams["campos"][sel].contiguous()
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.zeros(3, device=dev)
gen = torch.Generator().manual_seed(11)
dpix = torch.randn(V, 9, 256, 256, generator=gen).to(dev)
dpix[:, 7] = 0
kw = dict(image_height=256, image_width=256, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"], rotations=g["rotation"],
          sh_degree=1, save_aux=True)
L = _lib.lib()
for item in filter(None, os.environ.get("F3DG_OPTIONS", "").split(",")):
    _lib.check(L.f3dg_set_option(item.split("=")[0].strip().encode(), int(item.split("=")[1])), "f3dg_set_option")
for pack in ((-1,) if os.environ.get("F3DG_OPTIONS") else (0, -1)):
    L.f3dg_set_option(b"render_pack", pack)
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], vm, pm, cp, bg, **kw)

    def step():
        f3d.rasterize_views(g["xyz"], g["opacity"], vm, pm, cp, bg, workspace=ws, out=out, radii=radii, check=False, **kw)
        rasterize_backward_raw(ws, g["xyz"], shs, None, g["scaling"], g["rotation"], radii, dpix, 1, vm, pm, cp, bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    L.f3dg_profile_enable(1)
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    L.f3dg_profile_enable(0)
    st = (C.c_double * 5)()
    nc = C.c_int(0)
    L.f3dg_profile_collect(st, C.byref(nc))
    print("render_pack %2d (%s): per %d-view step: projection %.2f, binning %.2f, compositing forward %.2f, compositing backward %.2f, per-Gaussian backward %.2f ms" % (
        pack, L.f3dg_debug_last_render_kernel().decode().split("<")[0], V, st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n), flush=True)
L.f3dg_set_option(b"render_pack", -1)
