import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import f3dgaus_amd as f3d
from f3dgaus_amd import synthetic
from helpers import run_hip
P, V, RES = 1000000, 2, 512
dev = torch.device("cuda:0")
g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], torch.zeros(3, device=dev),
    image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1, save_aux=True)
import ctypes as C
from f3dgaus_amd import _lib
T = (RES // 16) ** 2
ranges = torch.empty((V * T, 2), dtype=torch.int32, device=dev)
L = _lib.lib()
L.f3dg_debug_export(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), P, RES, RES, V, ws.max_rendered,
                    None, None, None, None, None, None, None, None, C.c_void_p(ranges.data_ptr()), None, None, None)
torch.cuda.synchronize()
r = ranges.cpu().numpy().astype(np.int64); n = r[:, 1] - r[:, 0]
print("instances", ws.num_rendered, "segments", len(n), "mean", n.mean(), "max", n.max(), "pct>4032", (n > 4032).mean(), "pct>8192", (n > 8192).mean(), "pct>16384", (n > 16384).mean())
print("quantiles", np.quantile(n, [0.5, 0.9, 0.99, 0.999]))
print("instances in long segments", n[n > 4032].sum() / n.sum())
