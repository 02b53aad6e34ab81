"""One small-call configuration in a loop (for tools/kstats.sh): P, V from argv. `graph` as third argument replays a HIP graph."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import synthetic  # noqa: E402
P, V = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
dev = torch.device("cuda:0")
RES = 256
bg = torch.zeros(3, device=dev)
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
kw = dict(image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"],
          rotations=g["rotation"], sh_degree=1)
out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, **kw)
call = lambda: f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, workspace=ws, out=out,
                                   radii=radii, check=False, **kw)
for _ in range(5):
    call()
torch.cuda.synchronize()
if os.environ.get("F3DG_RENDER_LOWOCC"):
    from f3dgaus_amd import _lib
    _lib.lib().f3dg_set_option(b"render_lowocc", int(os.environ["F3DG_RENDER_LOWOCC"]))
if os.environ.get("F3DG_SMALL_DEBUG"):
    from f3dgaus_amd import _lib
    _lib.lib().f3dg_set_option(b"small_debug", int(os.environ["F3DG_SMALL_DEBUG"]))
n = 100
if mode == "graph":
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            call()
    torch.cuda.synchronize()
    run = gr.replay
else:
    run = call
for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    run()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
if mode == "sites":
    from f3dgaus_amd import _lib
    L = _lib.lib()
    L.f3dg_set_option(b"time_launches", 1)
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    L.f3dg_debug_launch_times(1)
    L.f3dg_set_option(b"time_launches", 0)
from f3dgaus_amd import _lib as _l
_l.lib().f3dg_debug_launch_count(1)
run()
torch.cuda.synchronize()
print(f"P={P} V={V} {mode}: host issue {t_issue / n * 1e6:.0f} us/call, wall {t_wall / n * 1e6:.0f} us/call, "
      f"{_l.lib().f3dg_debug_launch_count(1)} kernel launches per call -> {V * n / t_wall:.0f} views/s")
