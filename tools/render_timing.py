"""Phase timing of the compositing kernel generations on the C2 workload (needs a library built with -DF3DG_TIMING:
tools/render_timing.sh). Prints shader-clock cycles per wave and phase."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import f3dgaus_amd as f3d
from f3dgaus_amd import _lib, synthetic
L = _lib.lib()
dev = torch.device("cuda:0")
P, V, RES = 196608, 120, 256
g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.zeros(3, device=dev)
ws = None
names = ["barrier", "staging", "lists", "phase1", "phase2", "unused", "total", "waves"]
for kern in [int(a) for a in sys.argv[1:]] or [2]:
    L.f3dg_set_option(b"render_kernel", kern)
    for rep in range(2):
        out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg,
                                             image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"],
                                             sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1, workspace=ws)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8)()
        L.f3dg_debug_timing(buf, 1)
    w = max(buf[7], 1)
    print("kernel", kern, " ".join("%s=%.0f" % (n, buf[i] / w) for i, n in enumerate(names[:7])), "waves", buf[7])
