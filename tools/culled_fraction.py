"""Fraction of (view, Gaussian) pairs that are in no tile list -- the pairs whose 80 bytes of record + ellipse the projection kernel no
longer writes (inference calls) -- on the C2 recipe, at sigma0 = 0.05 and on the merged set of the real image (tests/golden/
real_image_256.npz through the build's own predictor and cycle aggregation). One SAVE_AUX call each, tiles_touched exported.

  python tools/culled_fraction.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import f3dgaus_amd as f3d
from f3dgaus_amd import _lib, synthetic

L = _lib.lib()
dev = torch.device("cuda:0")


def fraction(g, V, res, label):
    cams = synthetic.orbit_cameras(V, resolution=res, device=dev)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    P = g["xyz"].shape[0]
    rows = []
    for cull in (1, 0):
        L.f3dg_set_option(b"tile_cull", cull)
        out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"],
                                             torch.zeros(3, device=dev), image_height=res, image_width=res, tanfovx=cams["tanfovx"],
                                             tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1,
                                             save_aux=True)
        tiles = torch.zeros(V * P, dtype=torch.int32, device=dev)
        rc = L.f3dg_debug_export(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), P, res, res, V,
                                 ws.max_rendered, None, None, None, C.c_void_p(tiles.data_ptr()), None, None, None, None, None, None, None, None)
        assert rc == 0
        torch.cuda.synchronize()
        T = ((res + 15) // 16) ** 2
        rng = torch.zeros(V * T * 2, dtype=torch.int32, device=dev)
        L.f3dg_debug_export(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), P, res, res, V,
                            ws.max_rendered, None, None, None, None, None, None, None, None, C.c_void_p(rng.data_ptr()), None, None, None)
        r = rng.cpu().numpy().reshape(-1, 2)
        lens = r[:, 1] - r[:, 0]
        rows.append((cull, float((tiles == 0).float().mean()), ws.num_rendered, float(lens.mean()), int(lens.max())))
    L.f3dg_set_option(b"tile_cull", 1)
    for cull, frac, R, lm, lx in rows:
        print("| %s | %d | %d | %d | %.4f | %d | %.2f | %.0f | %d |" % (label, P, V, cull, frac, R, R / (V * P), lm, lx), flush=True)


print("| workload | Gaussians | views | tile_cull | pairs in no list | instances R | R / (V P) | tile list mean | max |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
fraction(synthetic.make_gaussians(196608, s0=0.01, seed=0, device=dev), 120, 256, "C2 (synthetic, sigma0 = 0.01)")
fraction(synthetic.make_gaussians(196608, s0=0.05, seed=0, device=dev), 120, 256, "synthetic, sigma0 = 0.05")
try:
    from real_data import real_merged_set
    g = real_merged_set(dev)
    fraction(g, 128, 256, "real image, merged set (9 x 65,536)")
    first = {k: v[:65536].contiguous() for k, v in g.items()}
    fraction(first, 8, 256, "real image, first prediction (65,536)")
except Exception as ex:      # pragma: no cover
    print("real set unavailable:", repr(ex))
