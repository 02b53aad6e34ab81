"""Per-call settings of the rasterizer (F3DG_FLAG_EXACT / _FAST / _NO_TILE_CULL / _NO_SMALL_PATH, include/f3dg.h): the reference hands
every knob of a call through GaussianRasterizationSettings_GOF (RAST/diff_gof_rasterization/__init__.py:168-182); the knobs this build
adds travel in the call's flags word the same way, so two streams can render with different settings at the same time."""
import numpy as np
import pytest
import torch

from helpers import make_scene, run_oracle

pytestmark = pytest.mark.gpu


def _call(scene, device, **kw):
    import f3dgaus_amd as f3d
    dev = lambda t: None if t is None else t.to(device)
    return f3d.rasterize_views(
        dev(scene["means3D"]), dev(scene["opacities"]), dev(scene["viewmatrix"]), dev(scene["projmatrix"]), dev(scene["campos"]),
        dev(scene["bg"]), image_height=scene["H"], image_width=scene["W"], tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        sh=dev(scene["shs"]), colors_precomp=dev(scene["colors_precomp"]), scales=dev(scene["scales"]), rotations=dev(scene["rotations"]),
        sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"], kernel_size=scene["kernel_size"], **kw)


def test_two_streams_render_exact_and_fast_concurrently(gpu_device):
    """Exact and fast arithmetic on two streams at once, five rounds: every frame bit-identical to the serial run of its mode (a
    process-wide option would make the two calls race for it)."""
    scene = make_scene(P=60000, res=(256, 256), s0=0.012, view="oblique", n_views=6, seed=5)
    ex, _, ws_x = _call(scene, gpu_device, exact=True)
    fa, _, ws_f = _call(scene, gpu_device, exact=False)
    ex, fa = ex.clone(), fa.clone()
    assert not torch.equal(ex[:, 8], fa[:, 8])            # the modes do differ (distortion channel)
    assert float(((ex[:, :3] - fa[:, :3]).abs() <= 1e-5).float().mean()) >= 0.999      # (isolated pixels flip an alpha >= 1/255 decision)
    cap = max(ws_x.max_rendered, ws_f.max_rendered)
    s1, s2 = torch.cuda.Stream(gpu_device), torch.cuda.Stream(gpu_device)
    torch.cuda.synchronize(gpu_device)
    outs = []
    for _ in range(5):
        with torch.cuda.stream(s1):
            a = _call(scene, gpu_device, exact=True, check=False, max_rendered=cap)
        with torch.cuda.stream(s2):
            b = _call(scene, gpu_device, exact=False, check=False, max_rendered=cap)
        outs.append((a, b))
    torch.cuda.synchronize(gpu_device)
    for a, b in outs:
        assert torch.equal(a[0], ex)
        assert torch.equal(b[0], fa)


def test_flags_override_the_process_defaults(gpu_device):
    """exact=True / False against the process-wide render_fast option set the other way; tile_cull=False gives the reference's
    instance count whatever the option says; small_path=False keeps a one-view call on the general launch sequence."""
    from f3dgaus_amd import _lib
    L = _lib.lib()
    scene = make_scene(P=20000, res=(128, 128), s0=0.02, view="oblique")
    o = run_oracle(scene)
    try:
        L.f3dg_set_option(b"render_fast", 0)
        ref_exact = _call(scene, gpu_device)[0].clone()
        got_fast = _call(scene, gpu_device, exact=False)[0].clone()
        L.f3dg_set_option(b"render_fast", 1)
        ref_fast = _call(scene, gpu_device)[0].clone()
        got_exact = _call(scene, gpu_device, exact=True)[0].clone()
    finally:
        L.f3dg_set_option(b"render_fast", 1)
    assert torch.equal(ref_exact, got_exact) and torch.equal(ref_fast, got_fast)
    assert not torch.equal(ref_exact, ref_fast)
    culled = _call(scene, gpu_device)[2].num_rendered
    full = _call(scene, gpu_device, tile_cull=False)[2].num_rendered
    assert full == o["num_rendered"] and culled <= full
    _call(scene, gpu_device)
    L.f3dg_debug_launch_count(1)
    b = _call(scene, gpu_device)[0].clone()
    n_small = L.f3dg_debug_launch_count(1)
    a = _call(scene, gpu_device, small_path=False)[0].clone()
    n_general = L.f3dg_debug_launch_count(1)
    assert torch.equal(a, b)
    assert n_small == 3 and n_general > n_small       # (the one-view shape takes the three-launch path by default)


def test_exact_flag_reaches_the_backward(gpu_device):
    """A SAVE_AUX forward with exact=False (F3DG_FLAG_FAST) records its arithmetic in the workspace header; the backward repeats it:
    gradients finite and within 1e-4 of the exact pair's."""
    from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
    scene = make_scene(P=8000, res=(96, 96), s0=0.03, view="oblique")
    dev = lambda t: None if t is None else t.to(gpu_device)
    gen = torch.Generator().manual_seed(3)
    dpix = torch.randn(1, 9, scene["H"], scene["W"], generator=gen).to(gpu_device)
    dpix[:, 7] = 0
    grads = {}
    for mode in (True, False):
        out, radii, ws = _call(scene, gpu_device, save_aux=True, exact=mode)
        g = rasterize_backward_raw(ws, dev(scene["means3D"]), dev(scene["shs"]), None, dev(scene["scales"]), dev(scene["rotations"]), radii, dpix,
                                   scene["sh_degree"], dev(scene["viewmatrix"]), dev(scene["projmatrix"]), dev(scene["campos"]), dev(scene["bg"]),
                                   scene["tanfovx"], scene["tanfovy"], scene["kernel_size"], scene["scale_modifier"])
        grads[mode] = {k: v.clone() for k, v in g.items()}
    for k in ("dL_dopacity", "dL_dcolors"):
        a, b = grads[True][k], grads[False][k]
        assert torch.isfinite(b).all()
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale < 1e-4, k


@pytest.mark.parametrize("name", ["depth_spread", "small_splats"])
def test_dropin_distortion_map_meets_survey_8d_with_raster_exact(name, gpu_device):
    """SURVEY 8d asks rel 1e-3 of the distortion term (forward.cu:552-557). The drop-in renderer composites inference calls in the fast
    arithmetic by default, which moves `distortion_map` by 3-17 % relative (<= 2.5e-6 absolute: INTEGRATION.md, first bullet); a caller
    that consumes the key sets cfg['model']['raster_exact'] = True, and THAT path -- render_predicted_more_v2_gof as visualize.py calls
    it, under torch.no_grad() -- is held to 8d here: |d| <= 1e-6 + 1e-3 |ref| on >= 99.9 % of the pixels, rel 1e-3 wherever the
    reference's value exceeds 1e-4."""
    import f3dgaus_amd as f3d
    from f3dgaus_amd import cameras
    from helpers import frac_within
    kw = {"depth_spread": dict(P=800, res=(64, 64), s0=0.3, view="canonical", depth_range=(1.0, 30.0)),
          "small_splats": dict(P=30000, res=(128, 128), s0=0.01, view="oblique")}[name]
    scene = make_scene(**kw)
    o = run_oracle(scene)["out_color"]
    res = scene["W"]
    cfg = cameras.default_cfg(res)
    import math
    assert math.tan(cfg['model']['fov'] * np.pi / 360) == scene["tanfovx"]       # the scene's cameras are the default configuration's
    cfg['model']['max_sh_degree'] = scene["sh_degree"]
    pc = {"xyz": scene["means3D"], "opacity": scene["opacities"], "scaling": scene["scales"], "rotation": scene["rotations"],
          "features_dc": scene["shs"][:, :1], "features_rest": scene["shs"][:, 1:]}
    pc = {k: v.unsqueeze(0).to(gpu_device) for k, v in pc.items()}
    args = (scene["viewmatrix"][:1].unsqueeze(0).to(gpu_device), scene["projmatrix"][:1].unsqueeze(0).to(gpu_device),
            scene["campos"][:1].unsqueeze(0).to(gpu_device), scene["bg"].reshape(1, 3).to(gpu_device))
    with torch.no_grad():
        cfg['model']['raster_exact'] = True
        exact = f3d.render_predicted_more_v2_gof(pc, 0, *args, cfg)["distortion_map"][0].cpu().numpy()
        cfg['model']['raster_exact'] = False
        fast = f3d.render_predicted_more_v2_gof(pc, 0, *args, cfg)["distortion_map"][0].cpu().numpy()
    assert frac_within(exact, o[8], 1e-6, 1e-3) >= 0.999, frac_within(exact, o[8], 1e-6, 1e-3)
    big = np.abs(o[8]) > 1e-4
    if big.any():
        assert frac_within(exact[big], o[8][big], 0.0, 1e-3) >= 0.999
    # the default (fast) key stays inside the north_star's absolute 1e-4 -- and is NOT the reference's to 1e-3 everywhere
    assert np.abs(fast - o[8]).max() <= 1e-4
