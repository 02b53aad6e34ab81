"""Deterministic synthetic Gaussians / cameras for tests and bench (SURVEY section 8d "Concrete synthetic inputs").

Gaussians live in the frustum of the canonical F3D-Gaus camera (fov 13.164 deg, depth 6.667..8.667):
  xy ~ U(-0.8, 0.8) * (z / 7.667), z ~ U(6.667, 8.667), scale = exp(N(log s0, 0.5^2)), rot = normalised N(0,1)^4,
  opacity ~ U(0.05, 0.95), features_dc ~ N(0,1), features_rest ~ N(0, 0.1^2) [P,3,3].
Generated on the CPU generator (bit-reproducible everywhere) and moved to the requested device.
"""
import math

import torch

from . import cameras


def make_gaussians(P, s0=0.01, seed=0, sh_rest=3, device="cpu", behind_fraction=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    z = torch.rand(P, generator=g) * 2.0 + 6.667
    xy = (torch.rand(P, 2, generator=g) * 1.6 - 0.8) * (z / 7.667).unsqueeze(1)
    xyz = torch.cat([xy, z.unsqueeze(1)], 1)
    if behind_fraction > 0:       # some Gaussians behind the near plane / far off-screen (fixture F2)
        nb = int(P * behind_fraction)
        xyz[:nb, 2] = torch.rand(nb, generator=g) * 0.4 - 0.1
        xyz[nb:2 * nb, 0] += 5.0
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.5 + math.log(s0))
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opacity = torch.rand(P, 1, generator=g) * 0.9 + 0.05
    dc = torch.randn(P, 1, 3, generator=g)
    rest = torch.randn(P, sh_rest, 3, generator=g) * 0.1
    out = dict(xyz=xyz, scaling=scales, rotation=rot, opacity=opacity, features_dc=dc, features_rest=rest)
    return {k: v.float().contiguous().to(device) for k, v in out.items()}


def make_pixel_gaussians(res=256, s0=0.01, seed=0, sh_rest=3, device="cpu", fov_deg=13.164):
    """res * res PIXEL-ORDERED Gaussians, id = y * res + x, as the splat head predicts them (reference src/gaussian_predictor.py:
    one Gaussian per input pixel, back-projected along the pixel's ray of the canonical camera to a smooth depth map + noise).
    Everything but the positions follows ``make_gaussians``. The id order is the point: a contiguous id range is a band of image
    rows, which is what the one-view-per-call loops of the reference feed the rasterizer (visualize.py:293-314, 387-416)."""
    P = res * res
    g = make_gaussians(P, s0=s0, seed=seed, sh_rest=sh_rest, device="cpu")
    gen = torch.Generator(device="cpu").manual_seed(seed + 101)
    ys, xs = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing="ij")
    u = (xs.reshape(-1) + 0.5) / res * 2 - 1
    v = (ys.reshape(-1) + 0.5) / res * 2 - 1
    t = math.tan(fov_deg * math.pi / 360)
    z = 7.667 + 0.6 * torch.sin(2.5 * u) * torch.cos(2.0 * v) + 0.02 * torch.randn(P, generator=gen)
    g["xyz"] = torch.stack([u * t * z, v * t * z, z], 1).float().contiguous()
    return {k: w.to(device) for k, w in g.items()}


def orbit_cameras(n_views, resolution=256, device="cpu", include_canonical=False):
    """world_view [V,4,4], full_proj [V,4,4], centers [V,3] of the n-view F3D-Gaus orbit (+ canonical first)."""
    cfg = cameras.default_cfg(resolution)
    rig = cameras.OrbitRig(cfg)
    cs = rig.orbit(n_views)
    wv, fp, cc = cs.world_view_transforms[:, 0], cs.full_proj_transforms[:, 0], cs.camera_centers[:, 0]
    if include_canonical:
        c0 = rig.canonical
        wv = torch.cat([c0.world_view_transforms[:, 0] if c0.world_view_transforms.ndim == 4 else c0.world_view_transforms, wv], 0)
        fp = torch.cat([c0.full_proj_transforms[:, 0] if c0.full_proj_transforms.ndim == 4 else c0.full_proj_transforms, fp], 0)
        cc = torch.cat([c0.camera_centers.reshape(1, 3), cc], 0)
    tanfov = math.tan(cfg["model"]["fov"] * math.pi / 360)
    return dict(viewmatrix=wv.contiguous().to(device), projmatrix=fp.contiguous().to(device),
                campos=cc.contiguous().to(device), tanfovx=tanfov, tanfovy=tanfov, cfg=cfg)
