"""Role clocks of the small-launch pipeline kernel (render3q) on the drop-in loop's one-view scene:  python tools/r3q_clocks.py [split] [exact]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import _lib, synthetic  # noqa: E402

split = int(sys.argv[1]) if len(sys.argv) > 1 else 2
exact = len(sys.argv) > 2 and sys.argv[2] == "exact"
dev = torch.device("cuda:0")
L = _lib.lib()
g = synthetic.make_pixel_gaussians(256, s0=0.01, seed=0, device=dev)
cams = synthetic.orbit_cameras(60, resolution=256, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.zeros(3, device=dev)
kw = dict(image_height=256, image_width=256, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1, exact=exact)
L.f3dg_set_option(b"render_split", split)
out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"][:1], cams["projmatrix"][:1], cams["campos"][:1], bg, **kw)
L.f3dg_set_option(b"render_count", 1)
L.f3dg_debug_render4_counts(None, 1)
N = 60
for v in range(N):
    f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"][v:v + 1], cams["projmatrix"][v:v + 1], cams["campos"][v:v + 1], bg, workspace=ws, out=out, radii=radii, check=False, **kw)
torch.cuda.synchronize()
h = (C.c_ulonglong * 64)()
_lib.check(L.f3dg_debug_render3q_clocks(h), "clocks")
L.f3dg_set_option(b"render_count", 0)
L.f3dg_set_option(b"render_split", -1)
print(L.f3dg_debug_last_render_kernel().decode())
for r, name in enumerate(("consumer", "evaluators", "producer")):
    tot, bar, spin, win, rnd, waves = (h[16 * r + k] for k in range(6))
    if waves:
        print("%-10s per wave: %8.0f clocks, %5.1f %% waiting for a window, %5.1f %% waiting for a counter; %.1f windows, %.1f rounds" % (
            name, tot / waves, 100.0 * bar / tot, 100.0 * spin / tot, win / waves, rnd / waves))
print("longest wave of any workgroup: consumer %d, evaluator %d, producer %d clocks" % (h[48], h[49], h[50]))
