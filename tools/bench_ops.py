"""Micro-benchmarks of the other kernels of the path (run on the GPU box): splat head, render epilogue, and a C3-shaped
end-to-end pass (predictor with random weights + cycle aggregation + orbit rendering) with its time split."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


cfg = cameras.default_cfg(256)
rig = cameras.OrbitRig(cfg)
B, res = 64, 256
HW = res * res
net = torch.randn(B, 23, res, res, device=dev) * 0.5
depth = torch.rand(B, 1, res, res, device=dev) * 2 + 6.667
ob = rig.orbit(8)
v2w = ob.view_to_world_transforms[:, 0][torch.arange(B) % 8].to(dev)
quat = ob.source_cv2wT_quat[:, 0][torch.arange(B) % 8].to(dev)
pred = f3d.GaussianSplatPredictor_gtunet(cfg).to(dev).eval()
out = f3d.gaussian_predictor.allocate_gaussians(B, HW, dev)
t = timeit(lambda: f3d.splat_head(net, depth, pred.ray_dirs, v2w, quat, out=out, n_offset=0))
print(f"splat_head: B={B} x {HW} Gaussians: {t * 1e6:.0f} us -> {192.0 * B * HW / t / 1e9:.0f} GB/s of algorithmic bytes (192 B/Gaussian), "
      f"{B * HW / t / 1e9:.2f} G Gaussians/s")

V = 120
raster = torch.rand(V, 9, res, res, device=dev)
raster[:, 6] = raster[:, 6] * 2 + 6.6
wv = rig.orbit(V).world_view_transforms[:, 0].to(dev)
from f3dgaus_amd.gaussian_renderer import _epilogue  # noqa: E402
fov = 13.164 * 3.141592653589793 / 180
t = timeit(lambda: _epilogue(raster, wv, res, res, fov, fov))
print(f"render_epilogue (+4x4 inverses): V={V}: {t * 1e6:.0f} us -> {(36 + 24) * V * HW / t / 1e9:.0f} GB/s (reads 9 ch, writes 6 ch per pixel)")

# C3-shaped end to end (random weights): B images -> cycle aggregation (1 + 8 predictor passes, 8 renders per image) -> 16-view orbit
Bc = int(os.environ.get("B", 8))
torch.manual_seed(0)
model = f3d.Unet_GS_gtunet(cfg, renderer=None).to(dev).eval()
images = torch.rand(Bc, 3, res, res, device=dev)
dep = torch.rand(Bc, 1, res, res, device=dev) * 2 + 6.667
t_cycle = timeit(lambda: f3d.cycle.cycle_aggregate(model, images, dep, cfg, rig=rig), n=3, warm=1)
merged = f3d.cycle.cycle_aggregate(model, images, dep, cfg, rig=rig)
t_orbit = timeit(lambda: f3d.cycle.render_orbit(merged, cfg, rig=rig, num_views=16, views_per_call=16, epilogue=True), n=3, warm=1)
with torch.no_grad():
    x0 = torch.cat([images, torch.ones_like(images[:, :1])], 1)
    t_unet = timeit(lambda: model.gaussian_predictor.network_with_offset(x0), n=3, warm=1)
print(f"C3-shaped, B={Bc}: cycle_aggregate {t_cycle * 1e3:.1f} ms (of which 9 U-Net passes ~ {9 * t_unet * 1e3:.1f} ms), "
      f"16-view orbit of the merged 589,824 Gaussians {t_orbit * 1e3:.1f} ms ({16 * Bc / t_orbit:.0f} views/s)")
