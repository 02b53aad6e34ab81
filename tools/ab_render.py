"""A/B harness of the C2 forward (or another shape): per-stage milliseconds from the library's HIP events (min / median / max over the
steps) and a hash of the rendered frames, for the library as it is built now. Options come from the environment (F3DG_OPT_<name>=<int>
-> f3dg_set_option(name, int)). Used by tools/ab_build.sh, which rebuilds one translation unit with extra -D flags per variant.

  python tools/ab_render.py [--gaussians P] [--views V] [--res R] [--sigma0 S] [--steps K] [--mode fast|exact] [--label text]
"""
import argparse
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import f3dgaus_amd as f3d
from f3dgaus_amd import _lib, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=196608)
ap.add_argument("--views", type=int, default=120)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--sigma0", type=float, default=0.01)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--mode", default="fast")
ap.add_argument("--label", default="")
ap.add_argument("--pixel-ordered", action="store_true")
ap.add_argument("--counts", action="store_true", help="one more call with the compositing kernel's work counters on (option render_count)")
ap.add_argument("--real", action="store_true", help="the merged set of the real image (tests/real_data.py), 589,824 Gaussians")
a = ap.parse_args()
L = _lib.lib()
dev = torch.device("cuda:0")
for k, v in os.environ.items():
    if k.startswith("F3DG_OPT_"):
        _lib.check(L.f3dg_set_option(k[9:].lower().encode(), int(v)), "f3dg_set_option " + k)
L.f3dg_set_option(b"render_fast", 1 if a.mode == "fast" else 0)
P, V, RES = a.gaussians, a.views, a.res
if a.real:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from real_data import real_merged_set
    g = real_merged_set(dev)
else:
    g = synthetic.make_pixel_gaussians(RES, s0=a.sigma0, device=dev) if a.pixel_ordered else synthetic.make_gaussians(P, s0=a.sigma0, seed=0, device=dev)
P = g["xyz"].shape[0]
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.zeros(3, device=dev)
ws = None
out = None


def call(check):
    global ws, out
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg,
                                         image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"],
                                         sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1, workspace=ws, check=check)


call(True)
call(True)
R = ws.num_rendered
torch.cuda.synchronize()
rows = []
for _ in range(a.steps):
    L.f3dg_profile_enable(1)
    call(False)
    L.f3dg_profile_enable(0)
    st = (C.c_double * 5)()
    nc = C.c_int(0)
    _lib.check(L.f3dg_profile_collect(st, C.byref(nc)), "collect")
    rows.append([st[0], st[1], st[2]])
rows = np.array(rows)
h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
T = ((RES + 15) // 16) ** 2
b = 72.0 * R + (36.0 * RES * RES + 8.0 * T) * V
fmt = lambda c: "%.3f/%.3f/%.3f" % (rows[:, c].min(), np.median(rows[:, c]), rows[:, c].max())
print("%-28s P=%d V=%d R=%d mode=%s | pre %s | bin %s | comp %s ms (min/med/max) | frac %.3f | sha %s" %
      (a.label, P, V, R, a.mode, fmt(0), fmt(1), fmt(2), b / (np.median(rows[:, 2]) * 1e-3) / 8e12, h), flush=True)
if a.counts:
    cb = (C.c_ulonglong * 16)()
    L.f3dg_set_option(b"render_count", 1)
    L.f3dg_debug_render_counts(cb, 1)
    call(False)
    torch.cuda.synchronize()
    L.f3dg_debug_render_counts(cb, 1)
    L.f3dg_set_option(b"render_count", 0)
    c = [int(x) for x in cb]
    print("    counters: staged %.3e scanned %.3e wave-trips %.3e slides %.3e lane-util %.3f | slides<=8 live %.3e (trips %.3e) <=24 %.3e (trips %.3e) | "
          "tail steps %.3e trips %.3e tested %.3e" % (c[0], c[1], c[2], c[3], c[4] / (64.0 * max(c[2], 1)), c[8], c[6], c[9], c[7], c[10], c[11], c[12]), flush=True)
