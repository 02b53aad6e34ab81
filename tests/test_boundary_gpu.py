"""The drop-in boundary, end to end: the reference-shaped blocking entry f3dg_forward, the `_C` shim of
INTEGRATION.md section 2 executed verbatim, debug=True behaviour, argument validation, and the small exports no other test calls."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import f3dgaus_amd as f3d
from f3dgaus_amd import _lib
from helpers import make_scene

pytestmark = pytest.mark.gpu


def _dev_scene(device, **kw):
    sc = make_scene(**kw)
    return sc, {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}


def _settings(d, debug=False):
    return f3d.GaussianRasterizationSettings_GOF(d["H"], d["W"], d["tanfovx"], d["tanfovy"], d["kernel_size"], torch.zeros(0), d["bg"],
                                                 d["scale_modifier"], d["viewmatrix"][0], d["projmatrix"][0], d["sh_degree"],
                                                 d["campos"][0], False, debug)


def _batched(d, save_aux=True):
    return f3d.rasterize_views(d["means3D"], d["opacities"], d["viewmatrix"], d["projmatrix"], d["campos"], d["bg"],
                               image_height=d["H"], image_width=d["W"], tanfovx=d["tanfovx"], tanfovy=d["tanfovy"], sh=d["shs"],
                               scales=d["scales"], rotations=d["rotations"], sh_degree=d["sh_degree"], save_aux=save_aux)


def test_f3dg_forward_reference_shaped_entry(gpu_device):
    """f3dg_forward = one view, blocking, returns num_rendered (rasterizer.h:35-55 shape); a too small capacity returns
    F3DG_ERR_OVERFLOW with the needed instance count in *h_needed."""
    sc, d = _dev_scene(gpu_device, P=3000, res=(96, 80), s0=0.05, view="oblique", bg=(0.1, 0.2, 0.3))
    L = _lib.lib()
    P, W, H = d["P"], d["W"], d["H"]
    ref, ref_radii, ws_ref = _batched(d)
    out = torch.full((9, H, W), -1.0, device=gpu_device)
    radii = torch.full((P,), -1, dtype=torch.int32, device=gpu_device)
    p = lambda t: C.c_void_p(t.data_ptr())

    def call(cap):
        ws = torch.empty(L.f3dg_workspace_bytes(P, W, H, 1, cap), dtype=torch.uint8, device=gpu_device)
        need = C.c_longlong(-1)
        n = L.f3dg_forward(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(ws), ws.numel(), cap, P, d["sh_degree"],
                           d["shs"].shape[1], p(d["bg"]), W, H, p(d["means3D"]), p(d["shs"]), None, p(d["opacities"]), p(d["scales"]),
                           1.0, p(d["rotations"]), None, None, p(d["viewmatrix"]), p(d["projmatrix"]), p(d["campos"]),
                           d["tanfovx"], d["tanfovy"], 0.0, 0, p(out), p(radii), _lib.FLAG_SAVE_AUX, C.byref(need))
        return n, need.value

    n, need = call(64)
    assert n == _lib.ERR_OVERFLOW and need == ws_ref.num_rendered
    n, need = call(need)
    assert n == ws_ref.num_rendered == need
    assert torch.equal(out, ref[0]) and torch.equal(radii, ref_radii[0])
    assert L.f3dg_forward(None, None, 0, 0, P, 1, 4, None, W, H, *([None] * 5), 1.0, *([None] * 6), 0.1, 0.1, 0.0, 0, None, None, 0,
                          None) == _lib.ERR_BAD_ARG
    assert L.f3dg_version().decode().startswith("f3dg-hip gfx950") and isinstance(L.f3dg_last_error(), bytes)


def test_integration_md_C_shim_verbatim(gpu_device):
    """Executes INTEGRATION.md section 2 as is (tests/shim_C.py) with the reference's argument tuples."""
    import shim_C
    assert "def rasterize_gaussians(" in shim_C.SOURCE and "def integrate_gaussians_to_points(" in shim_C.SOURCE
    sc, d = _dev_scene(gpu_device, P=4000, res=(100, 72), s0=0.04, view="oblique")
    rs = _settings(d)
    empty = torch.Tensor([])
    # RAST/diff_gof_rasterization/__init__.py:61-84
    args = (rs.bg, d["means3D"], empty, d["opacities"], d["scales"], d["rotations"], rs.scale_modifier, empty, empty, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, rs.image_height, rs.image_width, d["shs"],
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
    n, color, radii, geom, binning, img = shim_C.rasterize_gaussians(*args)
    ref, ref_radii, ws = _batched(d)
    assert n == ws.num_rendered and torch.equal(color, ref[0]) and torch.equal(radii, ref_radii[0])
    assert geom.dtype == torch.uint8 and binning.numel() == 1 and int(binning[0]) >= n and img.numel() == 0
    # RAST/diff_gof_rasterization/__init__.py:115-138: the backward half of the shim against the package's own backward
    assert "def rasterize_gaussians_backward(" in shim_C.SOURCE
    gen = torch.Generator().manual_seed(4)
    dpix = torch.randn(9, rs.image_height, rs.image_width, generator=gen).to(gpu_device)
    bargs = (rs.bg, d["means3D"], radii, empty, d["scales"], d["rotations"], rs.scale_modifier, empty, empty, rs.viewmatrix, rs.projmatrix,
             rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, dpix, d["shs"], rs.sh_degree, rs.campos, geom, n, binning, img,
             rs.debug)
    grads = shim_C.rasterize_gaussians_backward(*bargs)
    assert len(grads) == 9
    from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
    _, ref_radii_aux, ws_aux = _batched(d, save_aux=True)
    g = rasterize_backward_raw(ws_aux, d["means3D"], d["shs"], None, d["scales"], d["rotations"], ref_radii_aux, dpix[None], rs.sh_degree,
                               rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.scale_modifier)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    for got, key in zip(grads, ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                                "dL_drotations", "dL_dview2gaussian")):
        want = g[key][0] if g[key].shape[0] == 1 and g[key].ndim == got.ndim + 1 else g[key]
        assert got.shape == want.shape, key
        if key == "dL_dcov3D":
            assert not got.any()
        else:
            assert rel(got, want) <= (1e-3 if key in ("dL_dmeans3D", "dL_dscales", "dL_drotations") else 1e-5), key    # float atomics order
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        shim_C.rasterize_gaussians(args[0], d["means3D"].reshape(-1), *args[2:])
    # RAST/diff_gof_rasterization/__init__.py:269-293
    pts = (d["means3D"][:500] + 0.01).contiguous()
    iargs = (rs.bg, pts, d["means3D"], empty, d["opacities"], d["scales"], d["rotations"], rs.scale_modifier, empty, empty,
             rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, rs.image_height, rs.image_width,
             d["shs"], rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
    n2, out2, alpha_i, color_i, radii2, *_ = shim_C.integrate_gaussians_to_points(*iargs)
    c, a, ci, r = f3d.GaussianRasterizer_GOF(rs).integrate(pts, d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"],
                                                            rotations=d["rotations"])
    assert torch.equal(out2, c) and torch.equal(alpha_i, a) and torch.equal(color_i, ci) and torch.equal(radii2, r)
    # integrate always walks the reference's tile lists; the forward's default lists are tile-culled (f3dg.h "tile_cull")
    assert n2 >= n
    L = _lib.lib()
    L.f3dg_set_option(b"tile_cull", 0)
    try:
        n0, color0, *_ = shim_C.rasterize_gaussians(*args)
    finally:
        L.f3dg_set_option(b"tile_cull", 1)
    assert n0 == n2 and torch.equal(color0, color)


def test_debug_true_dumps_arguments_on_failure(gpu_device, tmp_path, monkeypatch):
    """rast_py:88-98 / :138-150: with debug=True a failing call leaves snapshot_fw.dump (the argument tuple, on the host)."""
    monkeypatch.chdir(tmp_path)
    sc, d = _dev_scene(gpu_device, P=500, res=(64, 64), s0=0.05)
    ok = f3d.GaussianRasterizer_GOF(_settings(d, debug=True))(d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"],
                                                              rotations=d["rotations"])
    assert ok[0].shape == (9, 64, 64) and not os.path.exists("snapshot_fw.dump")
    ref = f3d.GaussianRasterizer_GOF(_settings(d, debug=False))(d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"],
                                                                rotations=d["rotations"])
    assert torch.equal(ok[0], ref[0])
    with pytest.raises(RuntimeError, match="opacities must have"):
        f3d.GaussianRasterizer_GOF(_settings(d, debug=True))(d["means3D"], None, d["opacities"][:-3], shs=d["shs"], scales=d["scales"],
                                                             rotations=d["rotations"])
    dump = torch.load("snapshot_fw.dump")
    assert len(dump) == 22 and dump[1].shape == (500, 3) and dump[1].device.type == "cpu" and dump[21] is True
    # without debug the same failure raises and writes nothing
    os.remove("snapshot_fw.dump")
    with pytest.raises(RuntimeError):
        f3d.GaussianRasterizer_GOF(_settings(d))(d["means3D"], None, d["opacities"][:-3], shs=d["shs"], scales=d["scales"],
                                                 rotations=d["rotations"])
    assert not os.path.exists("snapshot_fw.dump")


def test_argument_validation(gpu_device):
    sc, d = _dev_scene(gpu_device, P=200, res=(32, 32), s0=0.05)
    kw = dict(image_height=32, image_width=32, tanfovx=d["tanfovx"], tanfovy=d["tanfovy"], sh=d["shs"], scales=d["scales"],
              rotations=d["rotations"], sh_degree=1)
    base = (d["means3D"], d["opacities"], d["viewmatrix"], d["projmatrix"], d["campos"], d["bg"])
    for bad in (dict(scales=d["scales"][:, :2]), dict(rotations=d["rotations"][:-1]), dict(sh=d["shs"][:, :, :2]),
                dict(out=torch.empty((1, 9, 32, 32), dtype=torch.float64, device=gpu_device)),
                dict(radii=torch.empty((1, 200), dtype=torch.int64, device=gpu_device))):
        with pytest.raises(RuntimeError):
            f3d.rasterize_views(*base, **{**kw, **bad})
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        f3d.rasterize_views(d["means3D"][:, :2], *base[1:], **kw)
    with pytest.raises(RuntimeError, match="HIP device"):
        f3d.rasterize_views(d["means3D"].cpu(), *base[1:], **kw)
    r = f3d.GaussianRasterizer_GOF(_settings(d))
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(d["means3D"], None, d["opacities"], scales=d["scales"], rotations=d["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        r(d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"])


def test_workspace_is_reused_for_smaller_calls_and_per_stream(gpu_device):
    sc, d = _dev_scene(gpu_device, P=3000, res=(64, 64), s0=0.05)
    rs = _settings(d)
    rast = f3d.GaussianRasterizer_GOF(rs)
    cache = f3d.diff_gof_rasterization._RasterizeGaussians.forward.__globals__["_WS_CACHE"]
    cache.clear()
    with torch.no_grad():
        a, _ = rast(d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
        assert len(cache) == 1, list(cache)
        (key, ws), = cache.items()
        buf = ws.buffer.data_ptr()
        b, _ = rast(d["means3D"][:1000], None, d["opacities"][:1000], shs=d["shs"][:1000], scales=d["scales"][:1000],
                    rotations=d["rotations"][:1000])            # fewer Gaussians: same buffer, no reallocation
        assert cache[key].buffer.data_ptr() == buf and cache[key].P == 1000
        side = torch.cuda.Stream(device=gpu_device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            c, _ = rast(d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
        side.synchronize()
        assert len(cache) == 2 and torch.equal(a, c)      # another stream gets its own workspace


def test_profile_and_timing_hooks(gpu_device):
    L = _lib.lib()
    sc, d = _dev_scene(gpu_device, P=2000, res=(64, 64), s0=0.05)
    L.f3dg_profile_enable(1)
    _batched(d, save_aux=False); _batched(d, save_aux=False)
    L.f3dg_profile_enable(0)
    ms = (C.c_double * 5)(); n = C.c_int(0)
    assert L.f3dg_profile_collect(ms, C.byref(n)) == 0 and n.value == 2 and all(0 < m < 100 for m in ms[:3]) and ms[3] == ms[4] == 0
    t = (C.c_ulonglong * 8)(*([7] * 8))
    assert L.f3dg_debug_timing(t, 1) == 0 and list(t) == [0] * 8      # product build: the counters are compiled out
