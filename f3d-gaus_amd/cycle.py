"""Cycle-aggregation loop and orbit rendering of the inference path (reference visualize.py:221-416), batched.

Reference shape of the work per batch of B images (visualize.py):
  :283       predict Gaussians of the input image from the canonical camera          (B x 65,536 Gaussians)
  :293-314   render them from 8 orbit cameras: 8 x B separate rasterizer calls, each with a host sync and .cpu()
  :326-340   feed every render (clamped RGB + alpha, rendered median depth) back through the predictor with ITS camera
             and torch.cat the 9 Gaussian sets per key                              (B x 589,824 Gaussians)
  :387-416   render the merged set along a 128-view orbit: 128 x B rasterizer calls
MI355X shape of the same work here:
  * the 8 (resp. 128) cameras of an image go through ONE launch sequence (render_views), nothing leaves the GPU;
  * the 8 re-predictions run as ONE predictor batch of 8*B images, and the splat head writes every pass directly
    into the preallocated merged buffers (no torch.cat chain, no O(n^2) re-copies);
  * the order of the merged set is the reference's: canonical view first, then orbit views 0..7.
"""
import torch

from . import cameras
from .gaussian_predictor import GAUSSIAN_KEYS, allocate_gaussians
from .gaussian_renderer import render_views


@torch.no_grad()
def cycle_aggregate(model, images, depth, cfg, rig=None, num_views=8, yaw_diff=0.25, pitch_diff=0.15,
                    return_renders=False):
    """images [B,3,H,W] in [0,1], depth [B,1,H,W] (z in camera space). Returns the merged Gaussian dict
    (every value [B, (1+num_views)*H*W, ...]) -- what visualize.py calls ``gaussian_splat_batch_merge``."""
    device = images.device
    B, _, H, W = images.shape
    HW = H * W
    rig = rig or cameras.OrbitRig(cfg)
    cano = rig.canonical
    orbit = rig.orbit(num_views, yaw_diff, pitch_diff)
    background = torch.zeros(3, dtype=torch.float32, device=device)
    squre_clip = cfg['opt']['squre_clip']

    merged = allocate_gaussians(B, (1 + num_views) * HW, device)

    # first forward: input image + alpha 1, canonical camera (visualize.py:282-283)
    x0 = torch.cat([images, torch.ones_like(images[:, :1])], 1).unsqueeze(1)
    v2w0 = cano.view_to_world_transforms.reshape(1, 1, 4, 4).expand(B, 1, 4, 4).to(device)
    q0 = cano.source_cv2wT_quat.reshape(1, 1, 4).expand(B, 1, 4).to(device)
    model(x0, background, v2w0, q0, return_3d_features=True, render=False, squre_clip=squre_clip, unet_depth=depth,
          out=merged, n_offset=0)
    first = {k: merged[k][:, :HW] for k in GAUSSIAN_KEYS}

    # 8 novel views of every image in one launch sequence per image (visualize.py:293-314)
    wv, fp, cc = (orbit.world_view_transforms.to(device), orbit.full_proj_transforms.to(device),
                  orbit.camera_centers.to(device))
    rgb = torch.empty((B, num_views, 3, H, W), dtype=torch.float32, device=device)
    alpha = torch.empty((B, num_views, 1, H, W), dtype=torch.float32, device=device)
    zmed = torch.empty((B, num_views, 1, H, W), dtype=torch.float32, device=device)
    ws = None
    for b in range(B):
        r = render_views(first, b, wv, fp, cc, background, cfg, workspace=ws, epilogue=False)
        ws = r["workspace"]
        rgb[b] = r["render"].clamp(0, 1)              # visualize.py:311
        alpha[b] = r["rendered_alpha"]
        zmed[b] = r["rendered_depth"]

    # re-predict from every novel view with its own camera and merge in place (visualize.py:326-340)
    v2w = orbit.view_to_world_transforms.to(device)       # [V,1,4,4]
    quat = orbit.source_cv2wT_quat.to(device)              # [V,1,4]
    for v in range(num_views):
        xin = torch.cat([rgb[:, v], alpha[:, v]], 1).unsqueeze(1)          # [B,1,4,H,W]
        model(xin, background, v2w[v:v + 1].expand(B, 1, 4, 4), quat[v:v + 1].expand(B, 1, 4),
              return_3d_features=True, render=False, squre_clip=squre_clip, unet_depth=zmed[:, v],
              out=merged, n_offset=(1 + v) * HW)
    if return_renders:
        return merged, dict(rgb=rgb, alpha=alpha, depth=zmed)
    return merged


@torch.no_grad()
def render_orbit(gaussians, cfg, rig=None, num_views=128, yaw_diff=0.25, pitch_diff=0.15, views_per_call=32,
                 epilogue=True):
    """visualize.py:343-416: the frontal camera + num_views orbit cameras are built exactly as the reference does,
    and views 1..num_views are rendered. Returns dict of [B, num_views, C, H, W] tensors (render, rendered_depth,
    rendered_alpha, depth_normal). The reference's background[th:th+1] out-of-range read is NOT
    reproduced: the intended background (0,0,0) is used."""
    rig = rig or cameras.OrbitRig(cfg)
    cams = rig.orbit_with_frontal(num_views, yaw_diff, pitch_diff)
    device = gaussians["xyz"].device
    B = gaussians["xyz"].shape[0]
    res = int(cfg['model']['training_resolution'])
    wv = cams.world_view_transforms[1:].to(device)
    fp = cams.full_proj_transforms[1:].to(device)
    cc = cams.camera_centers[1:].to(device)
    bg = torch.zeros(3, dtype=torch.float32, device=device)
    out = {"render": torch.empty((B, num_views, 3, res, res), device=device),
           "rendered_depth": torch.empty((B, num_views, 1, res, res), device=device),
           "rendered_alpha": torch.empty((B, num_views, 1, res, res), device=device)}
    if epilogue:
        out["depth_normal"] = torch.empty((B, num_views, 3, res, res), device=device)
    workspaces = {}
    for b in range(B):
        for a in range(0, num_views, views_per_call):
            e = min(a + views_per_call, num_views)
            r = render_views(gaussians, b, wv[a:e], fp[a:e], cc[a:e], bg, cfg, workspace=workspaces.get(e - a),
                             epilogue=epilogue)
            workspaces[e - a] = r["workspace"]
            out["render"][b, a:e] = r["render"]
            out["rendered_depth"][b, a:e] = r["rendered_depth"]
            out["rendered_alpha"][b, a:e] = r["rendered_alpha"]
            if epilogue:
                out["depth_normal"][b, a:e] = r["depth_normal"]
    return out
