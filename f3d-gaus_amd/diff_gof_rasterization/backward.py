"""Backward of ``_RasterizeGaussians`` through ``f3dg_backward`` (reference rast_py:106-165 -> rasterize_points.cu:124-211).
Gradient order returned to autograd follows rast_py:152-165:
(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, None)."""
import ctypes as C

import torch

from .. import _lib


def rasterize_backward_raw(ws, means3D, sh, colors_precomp, scales, rotations, radii, grad_out_color, sh_degree,
                           viewmatrices, projmatrices, camposs, bg, tanfovx, tanfovy, kernel_size, scale_modifier):
    """n_views-batched backward on a workspace produced with save_aux=True. Returns a dict of gradient tensors:
    per-view [V,P,..] for means2D / colors / view2gaussian, summed over the views for the Gaussian parameters."""
    from . import _dev_f32, _stream
    if not getattr(ws, "save_aux", False):
        # (the library checks the same on the device -- the gradients would all be zero and f3dg_backward_pairs reports ERR_STATE)
        raise RuntimeError("backward on a workspace whose last forward was an inference call (save_aux=False): the auxiliary "
                           "planes it reads were not written")
    device = means3D.device
    P = means3D.size(0)
    V = ws.n_views
    M = 0 if sh is None or sh.numel() == 0 else sh.size(1)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)
    # f3dg_backward writes every element of the per-view outputs (include/f3dg.h): no 64-float zero-fill per (view, Gaussian) -- 2 GB at
    # BASELINE C5; dL_dconic is identically zero in the reference (never written, never returned to autograd): one zero, broadcast
    e = (lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)) if P else z
    g = dict(dL_dmeans2D=e(V, P, 3), dL_dconic=z(1, 1, 1, 1).expand(V, P, 2, 2), dL_dopacity=z(P, 1), dL_dcolors=e(V, P, 3),
             dL_dmeans3D=z(P, 3), dL_dcov3D=z(P, 6), dL_dsh=z(P, M, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4),
             dL_dview2gaussian=e(V, P, 10))
    if P == 0:
        return g
    f = lambda t: _dev_f32(t, device)
    dpix = f(grad_out_color).reshape(V, 9, ws.H, ws.W)
    vm, pm, cp, bgt = f(viewmatrices).reshape(V, 16), f(projmatrices).reshape(V, 16), f(camposs).reshape(V, 3), f(bg).reshape(-1, 3)
    flags = _lib.FLAG_BG_PER_VIEW if (bgt.size(0) == V and V > 1) else 0
    means3D, sh, colors_precomp, scales, rotations = f(means3D), f(sh), f(colors_precomp), f(scales), f(rotations)
    radii = radii.contiguous()
    rc = _lib.lib().f3dg_backward(
        _stream(), C.c_void_p(ws.buffer.data_ptr()), ws.nbytes, ws.max_rendered, V, P, int(sh_degree), int(M),
        _lib.ptr(bgt), ws.W, ws.H, _lib.ptr(means3D), _lib.ptr(sh), _lib.ptr(colors_precomp), _lib.ptr(scales),
        float(scale_modifier), _lib.ptr(rotations), None, None, _lib.ptr(vm), _lib.ptr(pm), _lib.ptr(cp),
        float(tanfovx), float(tanfovy), float(kernel_size), _lib.ptr(radii), _lib.ptr(dpix),
        _lib.ptr(g["dL_dmeans2D"]), _lib.ptr(g["dL_dconic"]), _lib.ptr(g["dL_dopacity"]), _lib.ptr(g["dL_dcolors"]),
        _lib.ptr(g["dL_dmeans3D"]), _lib.ptr(g["dL_dcov3D"]), _lib.ptr(g["dL_dsh"]), _lib.ptr(g["dL_dscales"]),
        _lib.ptr(g["dL_drotations"]), _lib.ptr(g["dL_dview2gaussian"]), flags)
    _lib.check(rc, "f3dg_backward")
    return g


def rasterize_backward(ctx, grad_out_color):
    rs = ctx.raster_settings
    colors_precomp, means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, radii, sh = ctx.saved_tensors
    ws = ctx.workspace
    if ws is None:
        raise RuntimeError("backward called on a forward that did not keep its auxiliary buffers")
    if cov3Ds_precomp.numel() or view2gaussian_precomp.numel():
        raise NotImplementedError("gradients for cov3D_precomp / view2gaussian_precomp inputs are not produced by the "
                                  "reference either (their backward kernels are dead code, backward.cu:992-1007)")
    g = rasterize_backward_raw(ws, means3D, sh, colors_precomp, scales, rotations, radii.reshape(1, -1),
                               grad_out_color.reshape(1, 9, ws.H, ws.W), rs.sh_degree, rs.viewmatrix, rs.projmatrix,
                               rs.campos, rs.bg.reshape(-1)[:3], rs.tanfovx, rs.tanfovy, rs.kernel_size,
                               rs.scale_modifier)
    none_if_empty = lambda inp, grad: grad if inp.numel() else None
    return (g["dL_dmeans3D"], g["dL_dmeans2D"][0], none_if_empty(sh, g["dL_dsh"]),
            none_if_empty(colors_precomp, g["dL_dcolors"][0]), g["dL_dopacity"],
            none_if_empty(scales, g["dL_dscales"]), none_if_empty(rotations, g["dL_drotations"]),
            None, None, None, None)       # cov3Ds_precomp, view2gaussian_precomp, raster_settings, exact
