"""SURVEY 8c fixture F6, "a real pipeline slice": images/1/n01644373_4548.jpg + its depth map through the reference's OWN dataset class,
predictor and renderer wrapper at 64 x 64, the plain-C oracle as the rasterizer (tests/tools/gen_real_golden.py, build container).
tests/golden/real_F6.npz holds the 4,096 pixel-aligned Gaussians on the real depth map and, for orbit views 1 and 5, the raster,
radii, instance count and the wrapper's post-processed maps.
  * CPU: the oracle on this host reproduces the stored rasters (host libm / compiler drift guard);
  * GPU: the HIP path -- the batched call AND the drop-in `render_predicted_more_v2_gof` -- reproduces them to the usual bars, in
    both arithmetic modes; the build's own predictor reproduces the stored Gaussians from the stored image + depth."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import assert_render_parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "golden", "real_F6.npz")


def _scene(g, v):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    tanfov = math.tan(float(g["fov"]) * math.pi / 360)
    return dict(P=g["g_xyz"].shape[0], W=64, H=64, sh_degree=int(g["sh_degree"]), kernel_size=0.0, scale_modifier=1.0,
                tanfovx=tanfov, tanfovy=tanfov, bg=torch.zeros(3), viewmatrix=t(g[f"v{v}_wv"]).reshape(1, 4, 4),
                projmatrix=t(g[f"v{v}_fp"]).reshape(1, 4, 4), campos=t(g[f"v{v}_cc"]).reshape(1, 3), means3D=t(g["g_xyz"]),
                opacities=t(g["g_opacity"]), scales=t(g["g_scaling"]), rotations=t(g["g_rotation"]),
                shs=torch.cat([t(g["g_features_dc"]), t(g["g_features_rest"])], 1).contiguous(), colors_precomp=None)


def test_fixture_is_a_real_image():
    g = np.load(PATH)
    assert str(g["name"]) == "n01644373_4548.jpg" and g["images"].shape == (1, 3, 64, 64) and g["g_xyz"].shape == (4096, 3)
    # pixel-aligned on the real depth map: Gaussian y * 64 + x sits on the ray of pixel (x, y) of the canonical camera at the
    # depth the dataset class read from the LeReS PNG (visualize.py:283 -> gaussian_predictor.py:961-1007)
    d = g["depth"].reshape(-1)
    assert 6.6 < d.min() < d.max() < 8.7 and np.unique(np.round(d, 3)).size > 500


@pytest.mark.parametrize("v", [1, 5])
def test_oracle_reproduces_real_fixture(v):
    from helpers import run_oracle
    g = np.load(PATH)
    o = run_oracle(_scene(g, v))
    assert o["num_rendered"] == int(g[f"v{v}_num_rendered"]) and np.array_equal(o["radii"], g[f"v{v}_radii"])
    assert_render_parity(o["out_color"], g[f"v{v}_raster"], "oracle-vs-real-fixture")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("v", [1, 5])
def test_hip_reproduces_real_fixture(v, mode, gpu_device):
    import helpers
    from f3dgaus_amd import _lib
    from helpers import run_hip
    g = np.load(PATH)
    L = _lib.lib()
    assert L.f3dg_set_option(b"render_fast", 2 if mode == "fast" else 0) == 0
    helpers.RENDER_MODE = mode
    try:
        h = run_hip(_scene(g, v), gpu_device)
    finally:
        helpers.RENDER_MODE = None
        L.f3dg_set_option(b"render_fast", 1)
    assert h["num_rendered"] == int(g[f"v{v}_num_rendered"]) and np.array_equal(h["radii"][0], g[f"v{v}_radii"])
    assert_render_parity(h["out_color"][0], g[f"v{v}_raster"], f"hip-vs-real-fixture view {v} ({mode})")


@pytest.mark.gpu
@pytest.mark.parametrize("v", [1, 5])
def test_dropin_wrapper_on_real_fixture(v, gpu_device):
    """`render_predicted_more_v2_gof` exactly as visualize.py:293-300 calls it, on the stored Gaussians: the dict the reference's
    wrapper returned around the oracle (incl. its world normals and depth normals) is reproduced."""
    import f3dgaus_amd as f3d
    from f3dgaus_amd import cameras
    g = np.load(PATH)
    cfg = cameras.default_cfg(64)
    dev = gpu_device
    pc = {k[2:]: torch.from_numpy(g[k]).unsqueeze(0).to(dev) for k in g.files if k.startswith("g_") and g[k].ndim >= 2}
    wv, fp, cc = (torch.from_numpy(g[f"v{v}_{n}"]).to(dev) for n in ("wv", "fp", "cc"))
    with torch.no_grad():
        r = f3d.render_predicted_more_v2_gof(pc, 0, wv, fp, cc, torch.zeros(1, 3, device=dev), cfg)
    ref = g[f"v{v}_raster"]
    got = torch.cat([r["render"], r["rendered_normal"], r["rendered_depth"], r["rendered_alpha"], r["distortion_map"]], 0).cpu().numpy()
    for name, a, b in (("render", got[:3], ref[:3]), ("alpha", got[7], ref[7])):
        assert np.mean(np.abs(a - b) <= 1e-4) >= 0.999, name
    assert np.mean(np.abs(got[6] - ref[6]) <= 1e-4 * np.abs(ref[6])) >= 0.999
    assert np.array_equal(r["radii"].cpu().numpy(), g[f"v{v}_radii"])
    # post-processed maps of the reference's wrapper (gr.py:1043-1053) on the oracle's raster
    for name in ("rendered_normal", "depth_normal"):
        a, b = r[name].cpu().numpy(), g[f"v{v}_{name}"]
        assert np.mean(np.abs(a - b) <= 2e-3) >= 0.995, (name, float(np.mean(np.abs(a - b) <= 2e-3)))


@pytest.mark.gpu
def test_build_predictor_reproduces_real_gaussians(gpu_device):
    """The build's own predictor (SongUNet through MIOpen + the fused splat-head kernel) on the stored image + depth, with the weights
    the generator gave the reference's module: the stored 4,096 Gaussians."""
    import f3dgaus_amd as f3d
    from f3dgaus_amd import cameras
    from helpers_weights import formula_state_dict
    g = np.load(PATH)
    cfg = cameras.default_cfg(64)
    cfg['model']['opacity_bias'] = 0.0
    model = f3d.Unet_GS_gtunet(cfg, renderer=None).eval()
    sd = model.state_dict()
    keep = {k: v for k, v in sd.items() if k.split(".")[-1] in ("ray_dirs", "sh_to_v_transform", "v_to_sh_transform") or k.endswith("resample_filter")}
    new = formula_state_dict({k: tuple(v.shape) for k, v in sd.items()}, keep=keep)
    for k in new:
        if k.endswith("network_with_offset.out.weight"):
            new[k] = torch.from_numpy(g["out_weight"])
        if k.endswith("network_with_offset.out.bias"):
            new[k] = torch.from_numpy(g["out_bias"])
    model.load_state_dict(new)
    model = model.to(gpu_device)
    images = torch.from_numpy(g["images"]).to(gpu_device)
    x = torch.cat([images.unsqueeze(1), torch.ones_like(images.unsqueeze(1)[:, :, 0:1])], 2)
    with torch.no_grad():
        _, _, gsb = model(x, torch.zeros(1, 3, device=gpu_device), torch.from_numpy(g["cano_v2w"]).to(gpu_device),
                          torch.from_numpy(g["cano_quat"]).to(gpu_device), return_3d_features=True, render=False,
                          squre_clip=cfg['opt']['squre_clip'], unet_depth=torch.from_numpy(g["depth"]).to(gpu_device))
    for k in ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest"):
        a, b = gsb[k][0].cpu().numpy(), g["g_" + k]
        assert a.shape == b.shape, k
        assert np.abs(a - b).max() <= 5e-4 * max(1.0, np.abs(b).max()), (k, float(np.abs(a - b).max()))
