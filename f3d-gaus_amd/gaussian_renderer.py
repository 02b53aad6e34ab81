"""Renderer operator of the hot path: drop-in ``render_predicted_more_v2_gof`` / ``_v3_gof`` and helpers.

Mirrors reference src/gaussian_renderer/__init__.py: ``focal2fov`` :15-19, ``depths_to_points`` :881-896,
``depth_to_normal`` :898-909, ``render_predicted_more_v2_gof`` :915-1067, ``render_predicted_more_v3_gof``
:1232-1380 (same, but ``pc[bs][key]`` list-of-dicts). Same signature, same returned dict keys / shapes / dtypes;
the rasterizer is ``GaussianRasterizer_GOF`` of this package (HIP), and the post-processing (normal
normalisation + rotation to world space, finite-difference normal of the median depth) is one fused HIP kernel
(``f3dg_render_epilogue``) instead of ~10 small torch kernels per call.

``render_views`` is the batched MI355X-first form the loops use: all cameras of one image in one launch sequence.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .diff_gof_rasterization import (GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF, _stream, add_redo, deferred_status,
                                     integrate_points, integrate_prepare, integrate_prepare_batched, rasterize_nograd,
                                     rasterize_views)


def _raster_exact(cfg):
    """cfg['model']['raster_exact'] -> True / False / None (absent: the process default of f3dg_set_option("render_fast"))."""
    try:
        m = cfg['model']
        v = m.get('raster_exact', None) if hasattr(m, 'get') else (m['raster_exact'] if 'raster_exact' in m else None)
    except (KeyError, TypeError):
        return None
    return None if v is None else bool(v)


def _raster_scan(cfg):
    """cfg['model']['raster_scan'] (this build's key, default absent = off): inference calls composite with the split-pixel schedule
    (F3DG_FLAG_SCAN: same blended entries, sums associated as a scan; within the fast arithmetic's 1e-4, not bit-identical to it)."""
    try:
        m = cfg['model']
        v = m.get('raster_scan', None) if hasattr(m, 'get') else (m['raster_scan'] if 'raster_scan' in m else None)
    except (KeyError, TypeError):
        return None
    return None if v is None else bool(v)


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def focal2fov_torch(focal, pixels):
    return 2 * torch.atan(pixels / (2 * focal))


def _epilogue(raster, world_view, W, H, FoVx, FoVy, want_normal=True, want_depth_normal=True, out=None):
    """raster [V,9,H,W], world_view [V,4,4] (row-vector convention). Returns (normal_world, depth_normal) [V,3,H,W]; `out`: a pair of
    such tensors to fill instead of new ones."""
    V = raster.shape[0]
    device = raster.device
    wv = world_view.reshape(V, 16)
    if wv.dtype != torch.float32 or not wv.is_contiguous():
        wv = wv.float().contiguous()
    fx = W / (2 * math.tan(FoVx / 2.))
    fy = H / (2 * math.tan(FoVy / 2.))
    if out is not None:
        nw, dn = out
    else:
        nw = torch.empty((V, 3, H, W), dtype=torch.float32, device=device) if want_normal else None
        dn = torch.empty((V, 3, H, W), dtype=torch.float32, device=device) if want_depth_normal else None
    raster = raster.contiguous()
    # (c2w = inverse(world_view^T) is formed inside the kernel: no torch.linalg.inv per call)
    rc = _lib.lib().f3dg_render_epilogue_view(_stream(), V, H, W, _lib.ptr(raster), _lib.ptr(wv), float(fx), float(fy),
                                              _lib.ptr(nw), _lib.ptr(dn))
    _lib.check(rc, "f3dg_render_epilogue_view")
    return nw, dn


def _epilogue_autograd(rendered_image, world_view_transform, W, H, FoVx, FoVy):
    """Differentiable counterpart of ``_epilogue`` for one [9,H,W] raster: the reference's own formulation (gr.py:898-909,
    :1043-1053) in torch ops. Used only when gradients are required; inference takes the fused kernel."""
    render_normal = torch.nn.functional.normalize(rendered_image[3:6], p=2, dim=0)
    c2w = (world_view_transform.reshape(4, 4).T).inverse()
    normal_world = (c2w[:3, :3] @ render_normal.reshape(3, -1)).reshape(3, *render_normal.shape[1:])
    depth = rendered_image[6:7]
    points = depths_to_points(world_view_transform.reshape(4, 4), W, H, FoVx, FoVy, depth).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    output[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return normal_world, output.permute(2, 0, 1)


def pack_frames(raster, out=None, max_workgroups=0):
    """raster [n,C>=3,H,W] float32 on the HIP device -> uint8 [n,H,W,3] = (255 * clip(raster[:, :3], 0, 1)).astype(uint8),
    the frame format of visualize.py:416, in one kernel (f3dg_pack_frames). ``out``: a contiguous uint8 [n,H,W,3] tensor to fill --
    on the same device, or a PINNED host tensor: the kernel then writes the frames straight into host memory
    (f3dg_pack_frames_host, at most ``max_workgroups`` workgroups, 0 = 64; no staging buffer and no copy command -- HIP's device ->
    host copy is a whole-chip shader copy that competes with the next launch). The host sees them once the stream is synchronised."""
    if raster.device.type != "cuda":
        raise RuntimeError("pack_frames needs a tensor on a HIP device (no CPU fallback)")
    r = raster.contiguous().float()
    n, Cc, H, W = r.shape
    if out is None:
        out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=r.device)
    elif out.dtype != torch.uint8 or not out.is_contiguous() or out.numel() != n * H * W * 3:
        raise RuntimeError(f"out must be a contiguous uint8 tensor of shape ({n}, {H}, {W}, 3)")
    elif out.device.type == "cpu":
        if not out.is_pinned():
            raise RuntimeError("a host `out` must be pinned (tensor.pin_memory()): the kernel writes into it")
        rc = _lib.lib().f3dg_pack_frames_host(_stream(), n, H, W, Cc, _lib.ptr(r), _lib.ptr(out), int(max_workgroups))
        _lib.check(rc, "f3dg_pack_frames_host")
        return out
    elif out.device != r.device:
        raise RuntimeError(f"out must be on {r.device} or in pinned host memory")
    rc = _lib.lib().f3dg_pack_frames(_stream(), n, H, W, Cc, _lib.ptr(r), _lib.ptr(out))
    _lib.check(rc, "f3dg_pack_frames")
    return out


def depths_to_points(world_view_transform, image_width, image_height, FoVx, FoVy, depthmap):
    """Back-projects a depth map to world-space points [H*W,3] (gaussian_renderer/__init__.py:881-896).
    Small torch helper kept for API parity; the render path itself uses the fused epilogue kernel."""
    dev = depthmap.device
    c2w = (world_view_transform.T).inverse()
    W, H = image_width, image_height
    fx = W / (2 * math.tan(FoVx / 2.))
    fy = H / (2 * math.tan(FoVy / 2.))
    intrins = torch.tensor([[fx, 0., W / 2.], [0., fy, H / 2.], [0., 0., 1.0]]).float().to(dev)
    grid_x, grid_y = torch.meshgrid(torch.arange(W, device=dev).float(), torch.arange(H, device=dev).float(), indexing='xy')
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)
    rays_d = points @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(world_view_transform, image_width, image_height, FoVx, FoVy, depth):
    """Central-difference normal map [H,W,3] of a depth map [1,H,W], border = 0 (:898-909). Uses the fused kernel."""
    H, W = depth.shape[-2], depth.shape[-1]
    raster = torch.zeros((1, 9, H, W), dtype=torch.float32, device=depth.device)
    raster[0, 6] = depth.reshape(H, W)
    _, dn = _epilogue(raster, world_view_transform.reshape(1, 4, 4), image_width, image_height, FoVx, FoVy,
                      want_normal=False)
    return dn[0].permute(1, 2, 0)


def _cat_sh(dc, rest):
    """cat(features_dc, features_rest) along the coefficient axis, on every call as the reference does (gr.py:1008). (Round 3 kept
    the result while both inputs looked unmodified; the package's own in-place writers -- ``splat_head(out=...)`` fills the buffers
    through raw pointers -- never touch the version counter that test relied on, so a refilled buffer rendered stale colours.)"""
    return torch.cat([dc, rest], dim=1).contiguous()


def _render_one(get, bs, world_view_transform, full_proj_transform, camera_center, bg_color, cfg, kernel_size,
                scaling_modifier, override_color, points3D=None):
    xyz = get("xyz")
    device = xyz.device
    # zero tensor whose gradient receives the screen-space mean gradients (reference :932-936)
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:       # inference: nothing will ever flow into it
        screenspace_points = torch.zeros_like(xyz)

    tanfovx = math.tan(cfg['model']['fov'] * np.pi / 360)
    tanfovy = math.tan(cfg['model']['fov'] * np.pi / 360)
    FovX = cfg['model']['fov'] * np.pi / 180
    FovY = cfg['model']['fov'] * np.pi / 180
    image_height = int(cfg['model']['training_resolution'])
    image_width = int(cfg['model']['training_resolution'])
    # the reference allocates a zero [H,W,2] subpixel_offset per call that no kernel reads (forward.cu:473 list of
    # unused arguments); an empty tensor keeps the 14-field settings tuple without the allocation + memset
    subpixel_offset = torch.empty((0,), dtype=torch.float32, device=device)

    raster_settings = GaussianRasterizationSettings_GOF(
        image_height=image_height, image_width=image_width, tanfovx=tanfovx, tanfovy=tanfovy,
        kernel_size=kernel_size, subpixel_offset=subpixel_offset, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=world_view_transform, projmatrix=full_proj_transform, sh_degree=cfg['model']['max_sh_degree'],
        campos=camera_center, prefiltered=False, debug=False)
    # cfg['model']['raster_exact'] (this build's key; absent = the process default): True composites THIS call in the reference's
    # float32 / float64 order -- what a consumer of `distortion_map` asks for --, False in the fast arithmetic
    exact = _raster_exact(cfg)
    rasterizer = GaussianRasterizer_GOF(raster_settings=raster_settings, exact=exact) if (torch.is_grad_enabled() or points3D is not None) else None

    means3D = xyz
    means2D = screenspace_points
    opacity = get("opacity")
    scales = get("scaling")
    rotations = get("rotation")

    extra = {}
    if points3D is not None:        # render_predicted_more_v2_gof_in (:1070-1228): integrate instead of render
        shs = torch.cat([get("features_dc"), get("features_rest")], dim=1).contiguous() if override_color is None else None
        colors_precomp = None if override_color is None else get("rgbs")
        rendered_image, alpha_integrated, color_integrated, radii = rasterizer.integrate(
            points3D=points3D, means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
            opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None, view2gaussian_precomp=None)
        extra = {"alpha_integrated": alpha_integrated, "color_integrated": color_integrated}
    elif not torch.is_grad_enabled():
        # inference: the same call without the nn.Module / autograd.Function wrapping (host time; the kernels are the same)
        shs = _cat_sh(get("features_dc"), get("features_rest")) if override_color is None else None
        rendered_image, radii = rasterize_nograd(means3D, shs, None if override_color is None else get("rgbs"), opacity, scales,
                                                 rotations, raster_settings, exact=exact, scan=_raster_scan(cfg))
    elif override_color is None:
        shs = _cat_sh(get("features_dc"), get("features_rest"))
        rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None,
                                           opacities=opacity, scales=scales, rotations=rotations,
                                           cov3D_precomp=None, view2gaussian_precomp=None)
    else:
        colors_precomp = get("rgbs")
        rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=None, colors_precomp=colors_precomp,
                                           opacities=opacity, scales=scales, rotations=rotations,
                                           cov3D_precomp=None, view2gaussian_precomp=None)

    wv = world_view_transform.reshape(1, 4, 4)
    if torch.is_grad_enabled() and rendered_image.requires_grad:
        # training: the two derived maps carry gradients in the reference (plain torch ops on the raster, gr.py:1043-1053), so a
        # normal-consistency or depth-normal loss must reach the rasterizer's backward -- same ops here instead of the fused kernel
        nw0, dn0 = _epilogue_autograd(rendered_image, wv[0], image_width, image_height, FovX, FovY)
    else:
        raster1 = rendered_image.detach().unsqueeze(0)
        nw, dn = _epilogue(raster1, wv, image_width, image_height, FovX, FovY)
        if deferred_status():         # (derived from the raster: repeated if the deferred status check finds that the call overflowed)
            add_redo(lambda: _epilogue(raster1, wv, image_width, image_height, FovX, FovY, out=(nw, dn)))
        nw0, dn0 = nw[0], dn[0]
    res = {"render": rendered_image[:3, :, :],
           "rendered_normal": nw0,
           "rendered_depth": rendered_image[6:7, :, :],
           "depth_normal": dn0,
           "rendered_alpha": rendered_image[7:8, :, :],
           "distortion_map": rendered_image[8:9, :, :],
           "viewspace_points": screenspace_points,
           "visibility_filter": radii > 0,
           "radii": radii}
    res.update(extra)
    return res


def render_predicted_more_v2_gof(pc: dict, bs, world_view_transform, full_proj_transform, camera_center,
                                 bg_color: torch.Tensor, cfg, kernel_size=0.0, scaling_modifier=1.0,
                                 override_color=None, subpixel_offset=None):
    """Render image ``bs`` of the batched Gaussian dict ``pc`` (every value [B,N,...]) from one camera.
    Matrices may carry leading singleton dims ([1,1,4,4], [1,1,3], [1,3]) exactly as visualize.py passes them."""
    return _render_one(lambda k: pc[k][bs], bs, world_view_transform, full_proj_transform, camera_center, bg_color,
                       cfg, kernel_size, scaling_modifier, override_color)


def render_predicted_more_v2_gof_in(points3D, pc: dict, bs, world_view_transform, full_proj_transform, camera_center,
                                    bg_color: torch.Tensor, cfg, kernel_size=0.0, scaling_modifier=1.0,
                                    override_color=None, subpixel_offset=None):
    """``render_predicted_more_v2_gof_in`` (:1070-1228): ``GaussianRasterizer_GOF.integrate`` of ``points3D`` [PN,3]
    against image ``bs`` of ``pc``; the returned dict additionally holds ``alpha_integrated`` [PN] and
    ``color_integrated`` [PN,3]. (``rendered_normal`` is all zero and ``distortion_map`` counts the points per pixel:
    that is what the reference's integrate kernel leaves in those channels.)"""
    return _render_one(lambda k: pc[k][bs], bs, world_view_transform, full_proj_transform, camera_center, bg_color,
                       cfg, kernel_size, scaling_modifier, override_color, points3D=points3D)


class AlphaSweep:
    """The opacity evaluation of the mesh extraction (visualize.py:449-464 and its 8 binary-search repeats :495-507):
    ``final_alpha = min over cameras of integrate(points).alpha_integrated``. The reference re-runs the whole rasterizer
    for every (camera, point set) pair; here every camera is prepared ONCE (``integrate_prepare``: projection, binning and
    the per-pixel pass stay resident in HBM) and each call only runs the per-point stage of every camera, with the minimum
    accumulated inside that kernel. Same numbers as looping ``render_predicted_more_v2_gof_in``.

        sweep = AlphaSweep(pc, bs, world_views, full_projs, camera_centers, bg, cfg, max_points=points.shape[0])
        final_alpha = sweep(points)          # [PN]; call again for the refined point sets
    """

    def __init__(self, pc: dict, bs, world_view_transforms, full_proj_transforms, camera_centers, bg_color, cfg, max_points,
                 kernel_size=0.0, scaling_modifier=1.0, override_color=None, streams=4, cameras_per_call=16):
        get = lambda k: pc[k][bs]
        xyz = get("xyz")
        tanfov = math.tan(cfg['model']['fov'] * np.pi / 360)
        res = int(cfg['model']['training_resolution'])
        shs = torch.cat([get("features_dc"), get("features_rest")], dim=1).contiguous() if override_color is None else None
        colors = None if override_color is None else get("rgbs")
        wv = world_view_transforms.reshape(-1, 4, 4)
        fp = full_proj_transforms.reshape(-1, 4, 4)
        cc = camera_centers.reshape(-1, 3)
        opacity, scaling, rotation = get("opacity"), get("scaling"), get("rotation")
        # The cameras are prepared `cameras_per_call` at a time, each group in ONE launch sequence (f3dg_integrate_prepare_batched):
        # the per-pixel pass of a single 256^2 camera is 256 workgroups, one wave per SIMD of an MI355X, so a camera alone is a chain
        # of latencies; sixteen of them fill the chip. (Round 2 kept four cameras in flight from four host threads on four streams;
        # `streams` is accepted and ignored.) A group shares one workspace: ~0.37 GB per camera at 589,824 Gaussians.
        V = wv.shape[0]
        self.views = []
        step = max(1, int(cameras_per_call))
        for v0 in range(0, V, step):
            self.views += integrate_prepare_batched(
                xyz, shs, colors, opacity, scaling, rotation, wv[v0:v0 + step], fp[v0:v0 + step], cc[v0:v0 + step], bg_color,
                image_height=res, image_width=res, tanfovx=tanfov, tanfovy=tanfov, sh_degree=cfg['model']['max_sh_degree'],
                max_points=max_points, scale_modifier=scaling_modifier, kernel_size=kernel_size)

    @property
    def nbytes(self):
        return sum(v.buffer.numel() for v in self.views if v.view == 0)

    def __call__(self, points3D):
        final_alpha = torch.ones((points3D.shape[0],), dtype=torch.float32, device=points3D.device)
        for v in self.views:
            integrate_points(v, points3D, alpha_min=final_alpha, want_outputs=False)
        return final_alpha


def render_predicted_more_v3_gof(pc, bs, world_view_transform, full_proj_transform, camera_center,
                                 bg_color: torch.Tensor, cfg, kernel_size=0.0, scaling_modifier=1.0,
                                 override_color=None, subpixel_offset=None):
    """Same as v2 but ``pc`` is a list of per-image dicts: ``pc[bs][key]`` (:1232-1380)."""
    return _render_one(lambda k: pc[bs][k], bs, world_view_transform, full_proj_transform, camera_center, bg_color,
                       cfg, kernel_size, scaling_modifier, override_color)


def render_views(pc: dict, bs, world_view_transforms, full_proj_transforms, camera_centers, bg_color, cfg,
                 kernel_size=0.0, scaling_modifier=1.0, override_color=None, workspace=None, epilogue=True,
                 check=True, channels="all"):
    """All V cameras of image ``bs`` in one launch sequence (no per-view Python loop, no per-view host sync).
    Returns a dict with the same keys as ``render_predicted_more_v2_gof`` but a leading view axis:
    render [V,3,H,W], rendered_normal [V,3,H,W], rendered_depth [V,1,H,W], depth_normal [V,3,H,W],
    rendered_alpha [V,1,H,W], distortion_map [V,1,H,W], radii [V,P], visibility_filter [V,P], plus 'workspace'.

    ``bs=None``: ALL B images of the batch through the same V cameras in one launch sequence (f3dg_forward_sets); the leading
    axis is then B * V, image-major (frame b * V + v).

    ``channels="rgb_depth_alpha"``: what visualize.py:304-306, 400-402 consume. The compositing kernel skips the normal and distortion
    accumulators; ``rendered_normal`` and ``distortion_map`` are None, the other maps bit-identical to the 9-channel call."""
    fov = cfg['model']['fov']
    tanfov = math.tan(fov * np.pi / 360)
    res = int(cfg['model']['training_resolution'])
    V = world_view_transforms.reshape(-1, 16).shape[0]
    n_sets = 1
    take = (lambda t: t[bs])
    wv, fp, cc = world_view_transforms, full_proj_transforms, camera_centers
    if bs is None:
        n_sets = pc["xyz"].shape[0]
        take = (lambda t: t.reshape((-1,) + tuple(t.shape[2:])))
        wv, fp, cc = (t.reshape(V, -1).repeat(n_sets, 1) for t in (wv, fp, cc))       # set-major: b * V + v
    if override_color is None:
        shs = torch.cat([take(pc["features_dc"]), take(pc["features_rest"])], dim=1).contiguous()
        colors = None
    else:
        shs, colors = None, take(pc["rgbs"])
    with torch.no_grad():
        raster, radii, ws = rasterize_views(
            take(pc["xyz"]), take(pc["opacity"]), wv, fp, cc, bg_color,
            image_height=res, image_width=res, tanfovx=tanfov, tanfovy=tanfov, sh=shs, colors_precomp=colors,
            scales=take(pc["scaling"]), rotations=take(pc["rotation"]), sh_degree=cfg['model']['max_sh_degree'],
            scale_modifier=scaling_modifier, kernel_size=kernel_size, workspace=workspace, check=check, n_sets=n_sets,
            channels=channels, exact=_raster_exact(cfg), scan=_raster_scan(cfg))
        lean = channels != "all"
        nw = dn = None
        if epilogue:
            nw, dn = _epilogue(raster, wv.reshape(V * n_sets, 4, 4).to(raster.device), res, res,
                               fov * np.pi / 180, fov * np.pi / 180, want_normal=not lean)
    return {"render": raster[:, :3], "rendered_normal": nw, "rendered_depth": raster[:, 6:7], "depth_normal": dn,
            "rendered_alpha": raster[:, 7:8], "distortion_map": None if lean else raster[:, 8:9], "visibility_filter": radii > 0,
            "radii": radii, "raster": raster, "workspace": ws}


def cycle_inputs(raster, B, V):
    """rasters [B * V, 9, H, W] (image-major) -> (xin [V, B, 4, H, W] = cat(clamp(rgb, 0, 1), alpha), depth [V, B, 1, H, W]):
    the next predictor inputs of the cycle aggregation (visualize.py:311, 331-333), view-major, in one kernel."""
    if raster.device.type != "cuda":
        raise RuntimeError("f3dgaus_amd.cycle_inputs needs a HIP tensor (no CPU fallback)")
    raster = raster.contiguous()
    H, W = raster.shape[-2:]
    xin = torch.empty((V, B, 4, H, W), dtype=torch.float32, device=raster.device)
    depth = torch.empty((V, B, 1, H, W), dtype=torch.float32, device=raster.device)
    rc = _lib.lib().f3dg_cycle_inputs(_stream(), int(B), int(V), int(H), int(W),
                                      _lib.ptr(raster), _lib.ptr(xin), _lib.ptr(depth))
    _lib.check(rc, "f3dg_cycle_inputs")
    return xin, depth
