"""GPU parity of GaussianRasterizer_GOF.integrate (f3dg_integrate through the C ABI) against the CPU oracle's literal
restatement of Rasterizer::integrate / integrateCUDA.

The HIP path is organised differently from the reference kernel (contributor lists in a global table, one lane per
point, no point sort, arithmetic emulation of the 256-point sweeps), so these tests are what shows the reorganisation
changes nothing: integer outputs (radii, points per pixel incl. the >256-points-per-pixel sweep quirk) are exact,
float outputs within the north-star 1e-4 (only expf may differ by an ulp between device and glibc)."""
import numpy as np
import pytest
import torch

import f3dgaus_amd as f3d
from f3dgaus_amd import _lib
from helpers import make_scene

pytestmark = pytest.mark.gpu

from helpers_integrate import assert_integrate_parity, hip_integrate, make_points, npy, oracle_integrate


def run_both(scene, pts, device):
    return oracle_integrate(scene, pts), hip_integrate(scene, pts, device)


SCENES = {
    "I1_tiny": (dict(P=2000, res=(64, 64), s0=0.05, view="canonical"), 20000, 0),
    "I2_oblique_bg": (dict(P=5000, res=(128, 128), s0=0.02, view="oblique", behind_fraction=0.05, bg=(0.2, 0.5, 0.7)), 60000, 0),
    "I3_colors_precomp": (dict(P=3000, res=(64, 64), s0=0.05, view="oblique", colors_precomp=True), 20000, 0),
    "I4_odd_size_filter": (dict(P=4000, res=(100, 72), s0=0.04, view="oblique", kernel_size=0.1), 30000, 0),
    "I5_sweeps": (dict(P=3000, res=(64, 64), s0=0.05, view="oblique"), 5000, 700),       # > 256 points in one pixel
    "I6_dense_lists": (dict(P=20000, res=(64, 64), s0=0.05, view="oblique"), 20000, 0),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_integrate_matches_oracle(name):
    kw, n_pts, cluster = SCENES[name]
    scene = make_scene(**kw)
    pts = make_points(scene, n_pts, cluster=cluster)
    o, h = run_both(scene, pts, torch.device("cuda:0"))
    if cluster:
        assert o["out"][8].max() > 256, "the cluster must overflow one pixel's 256-point sweep"
    assert_integrate_parity(o, h, name)


@pytest.mark.lab
@pytest.mark.parametrize("kw", [dict(P=20000, res=(128, 128), s0=0.03, view="oblique"),
                                dict(P=3000, res=(96, 80), s0=0.3, view="oblique", aniso=True),        # large, flat splats
                                dict(P=60000, res=(64, 64), s0=0.004, view="canonical")],               # sub-pixel splats
                         ids=["mid", "large_aniso", "tiny"])
def test_integrate_filters_are_bit_identical(kw):
    """The culled lists + K pre-test of pass 1 only remove (ray, Gaussian) pairs the reference `continue`s on."""
    scene = make_scene(**kw)
    pts = make_points(scene, 50000)
    L = _lib.lib()
    res = []
    try:
        # (2, 1): the default, per-pixel ellipse test with Gaussians across the lanes; (1, 1): round 1's per-ray pre-test + box masks;
        # (2, 0): the plain transcription every variant must match bit for bit
        # (3, 1): the default, 545 shared rays per tile (integrate_pass1_rays_kernel); (2, 1): round 2's per-pixel pass
        for kernel, on in ((3, 1), (2, 1), (1, 1), (2, 0)):
            L.f3dg_set_option(b"render_kernel", kernel)
            L.f3dg_set_option(b"render_pretest", on)
            L.f3dg_set_option(b"render_cull", on)
            res.append(run_both(scene, pts, torch.device("cuda:0"))[1])
    finally:
        L.f3dg_set_option(b"render_kernel", 3)
        L.f3dg_set_option(b"render_pretest", 1)
        L.f3dg_set_option(b"render_cull", 1)
    for r in res[:3]:
        for k in ("out", "ai", "ci", "radii"):
            assert np.array_equal(r[k].view(np.uint32), res[3][k].view(np.uint32)), k


def test_contributor_limit_tiles_are_redone_per_pixel():
    """Pixels that reach the reference's 1,024 contributors (forward.cu:972-976) stop there while the rays they share with their
    neighbours go on: the shared-ray kernel hands such tiles to the per-pixel kernel. Thousands of nearly transparent, large
    splats per pixel; the result must be the plain transcription's bit for bit and the oracle's within tolerance."""
    import ctypes as C
    from f3dgaus_amd.diff_gof_rasterization import GaussianRasterizationSettings_GOF, integrate_prepare
    dev = torch.device("cuda:0")
    scene = make_scene(P=30000, res=(40, 36), s0=0.2, view="canonical")
    scene["opacities"] = torch.full_like(scene["opacities"], 0.011)
    pts = make_points(scene, 8000)
    L = _lib.lib()
    d = lambda t: None if t is None else t.to(dev)
    rs = GaussianRasterizationSettings_GOF(36, 40, scene["tanfovx"], scene["tanfovy"], 0.0, torch.zeros(0), d(scene["bg"]), 1.0,
                                           d(scene["viewmatrix"][0]), d(scene["projmatrix"][0]), scene["sh_degree"],
                                           d(scene["campos"][0]), False, False)
    prep = integrate_prepare(d(scene["means3D"]), d(scene["shs"]), None, d(scene["opacities"]), d(scene["scales"]),
                             d(scene["rotations"]), None, None, rs, max_points=8000)
    n = C.c_int(-1)
    assert L.f3dg_debug_integrate_redo(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(prep.buffer.data_ptr()),
                                       prep.P, prep.max_points, prep.W, prep.H, 1, prep.capacity, C.byref(n)) == 0
    assert 0 < n.value <= 9, n.value                       # some of the 3 x 3 tiles, i.e. the limit was reached
    o, h = run_both(scene, pts, dev)
    assert_integrate_parity(o, h, "contributor limit")
    if not L.f3dg_version().endswith(b"lab"):
        return                                             # (the plain transcription is compiled in lab builds only)
    try:
        L.f3dg_set_option(b"render_pretest", 0); L.f3dg_set_option(b"render_cull", 0)
        plain = hip_integrate(scene, pts, dev)
    finally:
        L.f3dg_set_option(b"render_pretest", 1); L.f3dg_set_option(b"render_cull", 1)
    for k in ("out", "ai", "ci", "radii"):
        assert np.array_equal(h[k].view(np.uint32), plain[k].view(np.uint32)), k


def test_integrate_empty_inputs():
    """P == 0 or PN == 0: nothing runs and the binding's fills stay (rasterize_points.cu:273-276, 300)."""
    dev = torch.device("cuda:0")
    scene = make_scene(P=500, res=(64, 64), s0=0.05, view="canonical")
    o, h = run_both(scene, np.zeros((0, 3), np.float32), dev)
    assert not h["out"].any() and h["ai"].shape == (0,) and h["ci"].shape == (0, 3)
    assert np.array_equal(o["out"], h["out"])


def test_renderer_wrapper_in():
    """render_predicted_more_v2_gof_in returns the reference's dict keys (gaussian_renderer/__init__.py:1212-1228)."""
    dev = torch.device("cuda:0")
    from f3dgaus_amd import cameras
    cfg = cameras.default_cfg(resolution=64)
    scene = make_scene(P=2000, res=(64, 64), s0=0.05, view="oblique")
    P = scene["P"]
    pc = {"xyz": scene["means3D"][None].to(dev), "opacity": scene["opacities"][None].to(dev),
          "scaling": scene["scales"][None].to(dev), "rotation": scene["rotations"][None].to(dev),
          "features_dc": scene["shs"][None, :, :1].to(dev), "features_rest": scene["shs"][None, :, 1:].to(dev)}
    cfg["model"]["max_sh_degree"] = scene["sh_degree"]
    pts = torch.from_numpy(make_points(scene, 5000)).to(dev)
    out = f3d.render_predicted_more_v2_gof_in(pts, pc, 0, scene["viewmatrix"][0].to(dev), scene["projmatrix"][0].to(dev),
                                              scene["campos"][0].to(dev), scene["bg"].to(dev), cfg)
    for k in ("render", "rendered_normal", "rendered_depth", "depth_normal", "rendered_alpha", "distortion_map",
              "viewspace_points", "visibility_filter", "alpha_integrated", "color_integrated", "radii"):
        assert k in out, k
    assert out["alpha_integrated"].shape == (5000,) and out["color_integrated"].shape == (5000, 3)
    assert out["render"].shape == (3, 64, 64) and not out["rendered_normal"].any()
    assert int(out["distortion_map"].sum().item()) > 0


@pytest.mark.parametrize("seed", range(6))
def test_integrate_random_configurations(seed):
    rng = np.random.default_rng(2000 + seed)
    W, H = int(rng.integers(17, 150)), int(rng.integers(17, 150))
    kw = dict(P=int(rng.integers(200, 5000)), res=(W, H), s0=float(np.exp(rng.uniform(np.log(0.01), np.log(0.1)))),
              seed=int(rng.integers(0, 1000)), view="oblique" if seed % 2 else "canonical",
              kernel_size=float(rng.choice([0.0, 0.1])), bg=tuple(float(x) for x in rng.uniform(0, 1, 3)),
              colors_precomp=bool(seed % 3 == 2))
    scene = make_scene(**kw)
    pts = make_points(scene, int(rng.integers(100, 20000)), seed=seed, spread=float(rng.uniform(0.01, 0.2)))
    o, h = run_both(scene, pts, torch.device("cuda:0"))
    assert_integrate_parity(o, h, str(kw))


def test_prepared_integration_and_alpha_sweep():
    """integrate_prepare + integrate_points give the bits of the one-shot integrate, for several point sets against the same
    preparation; AlphaSweep equals the reference-shaped loop min_v integrate_v(points).alpha_integrated."""
    from f3dgaus_amd import cameras
    from f3dgaus_amd.diff_gof_rasterization import (GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF,
                                                    integrate_points, integrate_prepare)
    dev = torch.device("cuda:0")
    scene = make_scene(P=6000, res=(64, 64), s0=0.04, view=[1, 3, 6])
    d = lambda t: None if t is None else t.to(dev)
    sets = [torch.from_numpy(make_points(scene, n, seed=s)).to(dev) for n, s in ((9000, 0), (4000, 1), (9000, 2))]
    loops = [torch.ones(len(p), device=dev) for p in sets]
    for v in range(3):
        rs = GaussianRasterizationSettings_GOF(64, 64, scene["tanfovx"], scene["tanfovy"], 0.0, torch.zeros(0), d(scene["bg"]), 1.0,
                                               d(scene["viewmatrix"][v]), d(scene["projmatrix"][v]), scene["sh_degree"],
                                               d(scene["campos"][v]), False, False)
        prep = integrate_prepare(d(scene["means3D"]), d(scene["shs"]), None, d(scene["opacities"]), d(scene["scales"]),
                                 d(scene["rotations"]), None, None, rs, max_points=9000)
        for k, pts in enumerate(sets):
            color, ai, ci, radii = GaussianRasterizer_GOF(rs).integrate(
                points3D=pts, means3D=d(scene["means3D"]), means2D=None, opacities=d(scene["opacities"]), shs=d(scene["shs"]),
                scales=d(scene["scales"]), rotations=d(scene["rotations"]))
            ai2, ci2 = integrate_points(prep, pts)
            assert torch.equal(ai, ai2) and torch.equal(ci, ci2) and torch.equal(color, prep.color), (v, k)
            assert torch.equal(radii, prep.radii)
            loops[k] = torch.min(loops[k], ai)
    cfg = cameras.default_cfg(resolution=64)
    cfg["model"]["max_sh_degree"] = scene["sh_degree"]
    pc = {"xyz": d(scene["means3D"])[None], "opacity": d(scene["opacities"])[None], "scaling": d(scene["scales"])[None],
          "rotation": d(scene["rotations"])[None], "features_dc": d(scene["shs"])[None, :, :1],
          "features_rest": d(scene["shs"])[None, :, 1:]}
    sweep = f3d.AlphaSweep(pc, 0, d(scene["viewmatrix"]), d(scene["projmatrix"]), d(scene["campos"]), d(scene["bg"]), cfg,
                           max_points=9000)
    # the sweep uses cfg's field of view; the scene was built with the same one
    assert abs(np.tan(cfg["model"]["fov"] * np.pi / 360) - scene["tanfovx"]) < 1e-12
    for k, pts in enumerate(sets):
        assert torch.equal(sweep(pts), loops[k]), k
    # cameras prepared one per call and all in one call are the same bits
    one = f3d.AlphaSweep(pc, 0, d(scene["viewmatrix"]), d(scene["projmatrix"]), d(scene["campos"]), d(scene["bg"]), cfg,
                         max_points=9000, cameras_per_call=1)
    assert len(one.views) == len(sweep.views) == 3 and sweep.views[0].buffer.data_ptr() == sweep.views[2].buffer.data_ptr()
    assert torch.equal(one(sets[0]), sweep(sets[0]))
    for a, b in zip(one.views, sweep.views):      # (after the same point set: channel 8 holds its points per pixel)
        assert torch.equal(a.color, b.color) and torch.equal(a.radii, b.radii)


def test_batched_prepare_many_cameras_full_size():
    """f3dg_integrate_prepare_batched at the mesh extraction's size (589,824 Gaussians @256^2): 12 cameras in one launch sequence give
    the per-pixel images, radii and point integrals of 12 single-camera preparations, bit for bit."""
    from f3dgaus_amd import cameras, synthetic
    from f3dgaus_amd.diff_gof_rasterization import integrate_points, integrate_prepare_batched
    dev = torch.device("cuda:0")
    P, V, RES, PN = 589824, 12, 256, 200000
    g = synthetic.make_gaussians(P, s0=0.01, seed=3, device=dev)
    oc = synthetic.orbit_cameras(V, resolution=RES, device=dev)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    gen = torch.Generator().manual_seed(5)
    pts = (g["xyz"][torch.randint(0, P, (PN,), generator=gen).to(dev)] + 0.03 * torch.randn(PN, 3, generator=gen).to(dev)).contiguous()
    kw = dict(image_height=RES, image_width=RES, tanfovx=oc["tanfovx"], tanfovy=oc["tanfovy"], sh_degree=1, max_points=PN)
    bg = torch.zeros(3, device=dev)
    many = integrate_prepare_batched(g["xyz"], shs, None, g["opacity"], g["scaling"], g["rotation"], oc["viewmatrix"], oc["projmatrix"],
                                     oc["campos"], bg, **kw)
    assert len(many) == V
    for v in (0, 5, 11):
        single = integrate_prepare_batched(g["xyz"], shs, None, g["opacity"], g["scaling"], g["rotation"], oc["viewmatrix"][v:v + 1],
                                           oc["projmatrix"][v:v + 1], oc["campos"][v:v + 1], bg, **kw)[0]
        # (channel 8 is written by the point stage only: the points per pixel, forward.cu:1216)
        assert torch.equal(single.color[:8], many[v].color[:8]) and torch.equal(single.radii, many[v].radii)
        a1, c1 = integrate_points(single, pts)
        a2, c2 = integrate_points(many[v], pts)
        assert torch.equal(a1, a2) and torch.equal(c1, c2) and torch.equal(single.color, many[v].color)
