"""Large-size robustness run: 4 M Gaussians, 1024x1024 (4096 tiles: two tile passes, long lists), 2 views, forward with and
without the filters (must be bit-identical) + backward (finite)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import _lib, synthetic  # noqa: E402
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw  # noqa: E402

P, V, RES = int(os.environ.get("P", 4_000_000)), 2, int(os.environ.get("RES", 1024))
dev = torch.device("cuda:0")
g = synthetic.make_gaussians(P, s0=0.004, seed=0, device=dev)
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
kw = dict(image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"],
          rotations=g["rotation"], sh_degree=1, save_aux=True)
L = _lib.lib()
outs = []
for on in (1, 0):
    for k in (b"render_pretest", b"render_cull", b"render_queue"):
        L.f3dg_set_option(k, on)
    t0 = time.perf_counter()
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, **kw)
    torch.cuda.synchronize()
    print(f"filters={on}: {1e3 * (time.perf_counter() - t0):.1f} ms (first call), instances {ws.num_rendered}, workspace {ws.nbytes / 1e9:.2f} GB")
    outs.append(out.clone())
for k in (b"render_pretest", b"render_cull", b"render_queue"):
    L.f3dg_set_option(k, 1)
assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), "filtered and plain compositing differ"
a = outs[0][:, 7]
assert float(a.min()) >= 0 and float(a.max()) <= 1.0 + 1e-5
out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, **kw)
dpix = torch.randn(V, 9, RES, RES, device=dev)
gr = rasterize_backward_raw(ws, g["xyz"], shs, None, g["scaling"], g["rotation"], radii, dpix, 1, cams["viewmatrix"], cams["projmatrix"],
                            cams["campos"], bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)
torch.cuda.synchronize()
for k, v in gr.items():
    assert torch.isfinite(v).all(), k
print("ok: bit-identical with / without filters, alpha mean %.3f, backward finite" % float(a.mean()))
