"""GPU parity of the forward rasterizer (through the C ABI) against the CPU oracle, stage by stage.

Per-Gaussian intermediates, instance counts, the sorted instance list and tile ranges are required to be
BIT-EXACT (integer/index work, and float work whose only operations are IEEE +,-,*,/,sqrt in a fixed order);
rendered channels are held to the north-star tolerance of 1e-4 (only expf may differ by an ulp)."""
import numpy as np
import pytest
import torch

from helpers import assert_render_parity, make_scene, run_hip, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fast", "exact"])
def render_mode(request):
    """Every test of this module runs in both compositing modes: "fast" (the default: error-free float32 pairs instead of the
    float64 island, csrc/f3dg_render.hip) and "exact" (the reference's float32/float64 operation order). Both are held to
    the same tolerances."""
    from f3dgaus_amd import _lib
    L = _lib.lib()
    import helpers
    assert L.f3dg_set_option(b"render_fast", 2 if request.param == "fast" else 0) == 0      # 2: fast also in SAVE_AUX calls
    helpers.RENDER_MODE = request.param
    yield request.param
    helpers.RENDER_MODE = None
    L.f3dg_set_option(b"render_fast", 1)

SCENES = {
    "F1_tiny_identity": dict(P=2000, res=(64, 64), s0=0.05, view="canonical"),
    "F2_oblique_aniso": dict(P=5000, res=(128, 128), s0=0.02, view="oblique", aniso=True, behind_fraction=0.05),
    "F3_colors_precomp": dict(P=3000, res=(64, 64), s0=0.05, view="oblique", colors_precomp=True, bg=(0.2, 0.5, 0.7)),
    "F4_filter_scalemod": dict(P=3000, res=(96, 96), s0=0.05, view="oblique", kernel_size=0.1, scale_modifier=0.5),
    "F5_odd_size": dict(P=4000, res=(100, 72), s0=0.04, view="oblique"),
    "F6_small_splats": dict(P=30000, res=(128, 128), s0=0.01, view="oblique"),
    "F8_sh0": dict(P=1500, res=(64, 64), s0=0.05, view="oblique", sh_degree=0),
    "F9_long_tile_lists": dict(P=30000, res=(128, 128), s0=0.05, view="oblique"),      # tile lists of 4k..16k entries
    "F11_wide_radix": dict(P=20000, res=(320, 272), s0=0.02, view="oblique"),         # 340 tiles: two 8-bit tile passes
    "F12_depth_spread": dict(P=800, res=(64, 64), s0=0.3, view="canonical", depth_range=(1.0, 30.0)),   # distortion values > 1e-4: rel 1e-3 applies
    "F10_huge_tile_lists": dict(P=50000, res=(32, 32), s0=0.05, view="canonical"),     # 4 tiles with lists > 16320 entries
    "F13_single_tile": dict(P=300, res=(16, 12), s0=0.05, view="canonical"),          # one tile: no tile pass at all
    "F14_sh_degree2": dict(P=3000, res=(96, 64), s0=0.05, view="oblique", sh_degree=2),    # M = 9  (forward.cu:40-51)
    "F15_sh_degree3": dict(P=3000, res=(64, 96), s0=0.05, view="oblique", sh_degree=3),    # M = 16 (forward.cu:53-66)
}


def _check_view(h, o, v, label, colors_precomp=None):
    vis = o["radii"] > 0
    if colors_precomp is not None:      # the oracle leaves its rgb buffer untouched when colours are given
        o = dict(o, rgb=colors_precomp)
    assert np.array_equal(h["radii"][v], o["radii"]), f"{label} radii"
    assert np.array_equal(h["tiles_touched"][v], o["tiles_touched"]), f"{label} tiles_touched"
    for name, hk, ok in (("view2gaussian", h["view2gaussian"][v], o["view2gaussian"]),
                         ("depths", h["depths"][v], o["depths"]), ("means2D", h["means2D"][v], o["means2D"]),
                         ("opacity*coef", h["opac"][v], o["conic_opacity"][:, 3]),
                         ("conic", h["conic_opacity"][v][:, :3], o["conic_opacity"][:, :3]),
                         ("rgb", h["rgb"][v], o["rgb"])):
        a, b = hk[vis], ok[vis]
        same = a.view(np.uint32) == b.view(np.uint32)
        assert same.all(), f"{label} {name}: {(~same).sum()} of {same.size} words differ, max abs {np.abs(a - b).max()}"
    if o["clamped"].size:
        bits = o["clamped"][:, 0] | (o["clamped"][:, 1] << 1) | (o["clamped"][:, 2] << 2)
        assert np.array_equal(h["clamped"][v][vis], bits[vis]), f"{label} clamped"


@pytest.mark.parametrize("name", list(SCENES))
def test_forward_stagewise(name, gpu_device):
    scene = make_scene(**SCENES[name])
    h = run_hip(scene, gpu_device)
    o = run_oracle(scene)
    assert h["num_rendered"] == o["num_rendered"], name
    cp = scene["colors_precomp"]
    _check_view(h, o, 0, name, None if cp is None else cp.numpy())
    assert np.array_equal(h["point_offsets"][0], o["point_offsets"])
    assert np.array_equal(h["keys_sorted"], o["keys_sorted"]), f"{name} sorted keys"
    assert np.array_equal(h["point_list"], o["point_list"]), f"{name} point list"
    assert np.array_equal(h["ranges"][0], o["ranges"]), f"{name} ranges"
    assert_render_parity(h["out_color"][0], o["out_color"], name)
    nc_same = (h["n_contrib"][0] == o["n_contrib"]).mean()
    assert nc_same >= 0.999, f"{name} n_contrib {nc_same}"
    d = np.abs(h["final_T"][0] - o["final_T"])
    assert (d[0] <= 1e-5).mean() >= 0.999


def test_forward_without_aux_matches_with_aux(gpu_device):
    scene = make_scene(**SCENES["F2_oblique_aniso"])
    a = run_hip(scene, gpu_device, save_aux=True)
    b = run_hip(scene, gpu_device, save_aux=False)
    assert np.array_equal(a["out_color"], b["out_color"])


@pytest.mark.parametrize("kw", [
    dict(P=8000, res=(128, 128), s0=0.03, view=[0, 1, 3, 5, 6, 2, 4, 7, 8]),       # 9 views: one XCD group of 8 + one spread view
    # odd P (a view's key segment starts at any address), depths over 1..30 (key ranges beyond 2^24: the four-pass depth sort)
    dict(P=4001, res=(96, 80), s0=0.2, view=[0, 3, 5], depth_range=(1.0, 30.0)),
    dict(P=12289, res=(320, 272), s0=0.02, view=[1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 2]),  # 340 tiles: two tile passes, 11 views, odd P
], ids=["orbit9", "odd_P_depth_spread", "two_tile_passes_11_views"])
def test_forward_batched_views_equal_single_views(gpu_device, kw):
    scene = make_scene(**kw)
    h = run_hip(scene, gpu_device)
    T = h["ranges"].shape[1]
    total = 0
    for v in range(scene["viewmatrix"].shape[0]):
        o = run_oracle(scene, view=v)
        _check_view(h, o, v, f"view{v}")
        R = o["num_rendered"]
        seg = h["point_list"][total:total + R]
        assert np.array_equal(seg, o["point_list"]), f"view {v} point list"
        assert np.array_equal(h["ranges"][v] - np.uint32(total) * (h["ranges"][v].sum(1, keepdims=True) > 0), o["ranges"])
        tb = int(T - 1).bit_length()                      # key = ((view << tile_bits) | tile) << 32 | depth bits
        hi = (h["keys_sorted"][total:total + R] >> np.uint64(32)).astype(np.int64)
        assert (hi >> tb == v).all() and np.array_equal(hi & ((1 << tb) - 1), (o["keys_sorted"] >> np.uint64(32)).astype(np.int64))
        assert np.array_equal(h["keys_sorted"][total:total + R] & np.uint64(0xFFFFFFFF), o["keys_sorted"] & np.uint64(0xFFFFFFFF))
        assert_render_parity(h["out_color"][v], o["out_color"], f"view{v}")
        total += R
    assert total == h["num_rendered"]


@pytest.mark.lab
@pytest.mark.parametrize("kw", [
    dict(P=8000, res=(128, 128), s0=0.03, view=[0, 1, 3, 5, 6, 2, 4, 7, 8]),
    dict(P=4001, res=(96, 80), s0=0.2, view=[0, 3, 5], depth_range=(1.0, 30.0)),       # four-pass views: the last pass is pass 3
], ids=["compact", "four_pass"])
def test_fused_rectangle_gather_is_invisible(gpu_device, kw):
    """Option sort_fused_rects (a view's last depth pass also delivers its tile rectangles in sorted order; measured equal, off by
    default) against the separate gather launch: lists, ranges and images identical."""
    from f3dgaus_amd import _lib
    scene = make_scene(**kw)
    L = _lib.lib()
    a = run_hip(scene, gpu_device)
    try:
        assert L.f3dg_set_option(b"sort_fused_rects", 1) == 0
        b = run_hip(scene, gpu_device)
    finally:
        L.f3dg_set_option(b"sort_fused_rects", 0)
    assert a["num_rendered"] == b["num_rendered"]
    for k in ("point_list", "ranges", "keys_sorted"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["out_color"].view(np.uint32), b["out_color"].view(np.uint32))


@pytest.mark.lab
@pytest.mark.parametrize("save_aux", [True, False])
def test_projection_hoist_is_bit_identical(gpu_device, save_aux):
    """Option pre_hoist (round 5): the view-independent part of the projection -- 3D covariance, rotation matrix, float64 scale
    reciprocals -- computed once per Gaussian by preprocess_hoist_kernel and read by the per-(view, Gaussian) threads instead of being
    recomputed per view. Same operations in the same order: records, radii, lists and images identical to the bit."""
    from f3dgaus_amd import _lib
    scene = make_scene(P=9000, res=(128, 96), s0=0.03, view=[0, 1, 3, 5, 6, 2], aniso=True, scale_modifier=0.8)
    L = _lib.lib()
    a = run_hip(scene, gpu_device, save_aux=save_aux)
    try:
        assert L.f3dg_set_option(b"pre_hoist", 1) == 0
        L.f3dg_debug_launch_count(1)
        b = run_hip(scene, gpu_device, save_aux=save_aux)
    finally:
        L.f3dg_set_option(b"pre_hoist", 0)
    assert a["num_rendered"] == b["num_rendered"]
    vis = a["radii"] > 0
    for k in ("radii", "point_list", "ranges"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("view2gaussian", "opac", "rgb"):
        assert np.array_equal(a[k][vis].view(np.uint32), b[k][vis].view(np.uint32)), k
    assert np.array_equal(a["out_color"].view(np.uint32), b["out_color"].view(np.uint32))


@pytest.mark.parametrize("which", ["cov3D+view2gaussian", "view2gaussian"])
def test_precomputed_covariance_and_view2gaussian(which, gpu_device):
    """cov3D_precomp replaces scales / rotations in the 2D footprint (forward.cu:338-348), view2gaussian_precomp replaces the
    computed view2gaussian in the compositing stage (forward.cu:396-403); a caller that precomputes the library's own values gets the
    scale/rotation render back, and both paths match the oracle called the same way."""
    import f3dgaus_amd as f3d
    scene = make_scene(P=3000, res=(96, 80), s0=0.05, view="oblique", bg=(0.1, 0.3, 0.2))
    base = run_oracle(scene)
    cov3D, v2g = base["cov3D"], base["view2gaussian"]
    npy = lambda t: t.detach().cpu().numpy()
    kw = dict(means3D=npy(scene["means3D"]), opacities=npy(scene["opacities"]), viewmatrix=npy(scene["viewmatrix"][0]),
              projmatrix=npy(scene["projmatrix"][0]), campos=npy(scene["campos"][0]), tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
              W=scene["W"], H=scene["H"], bg=npy(scene["bg"]), shs=npy(scene["shs"]), sh_degree=1, view2gaussian_precomp=v2g)
    from oracle import gof
    o = gof.Oracle()
    if which == "view2gaussian":
        out_o, radii_o, R_o = o.forward(scales=npy(scene["scales"]), rotations=npy(scene["rotations"]), **kw)
    else:
        out_o, radii_o, R_o = o.forward(cov3D_precomp=cov3D, **kw)
    dev = lambda t: None if t is None else t.to(gpu_device)
    pre = dict(view2gaussian_precomp=torch.from_numpy(v2g).to(gpu_device).unsqueeze(0))
    if which == "view2gaussian":
        pre.update(scales=dev(scene["scales"]), rotations=dev(scene["rotations"]))
    else:
        pre.update(cov3Ds_precomp=torch.from_numpy(cov3D).to(gpu_device))
    out, radii, ws = f3d.rasterize_views(
        dev(scene["means3D"]), dev(scene["opacities"]), dev(scene["viewmatrix"]), dev(scene["projmatrix"]), dev(scene["campos"]),
        dev(scene["bg"]), image_height=scene["H"], image_width=scene["W"], tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        sh=dev(scene["shs"]), sh_degree=1, save_aux=True, **pre)
    assert np.array_equal(radii[0].cpu().numpy(), radii_o) and np.array_equal(radii_o, base["radii"])
    assert_render_parity(out[0].cpu().numpy(), out_o, which)
    assert_render_parity(out[0].cpu().numpy(), base["out_color"], which + " vs scale/rotation render")


def test_c2_view_with_large_splats_full_size(gpu_device):
    """SURVEY 8d's second sweep: one full-size C2 view at sigma0 = 0.05 (R / P ~ 14, tile lists of ~10 k entries: the
    compositing-heavy, numerically benign regime), in both arithmetic modes (this module's fixture)."""
    scene = make_scene(P=196608, res=(256, 256), s0=0.05, view="oblique")
    h = run_hip(scene, gpu_device)
    o = run_oracle(scene)
    assert h["num_rendered"] == o["num_rendered"] and h["num_rendered"] > 8 * 196608
    assert np.array_equal(h["point_list"], o["point_list"])
    assert np.array_equal(h["ranges"][0], o["ranges"])
    assert_render_parity(h["out_color"][0], o["out_color"], "C2 sigma0=0.05")
    assert (h["n_contrib"][0] == o["n_contrib"]).mean() >= 0.999


@pytest.mark.lab
def test_wide_group_stream_is_identical(gpu_device):
    """The binning stage carries (view << tile_bits | tile) as u16 when it fits, else u32: both must give the same lists
    (the u32 path is otherwise only reached with more than 65,536 (view, tile) groups)."""
    from f3dgaus_amd import _lib
    L = _lib.lib()
    res = []
    try:
        for wide in (0, 1):
            L.f3dg_set_option(b"sort_wide_groups", wide)
            for name in ("F5_odd_size", "F9_long_tile_lists", "F10_huge_tile_lists", "F11_wide_radix"):
                res.append((wide, name, run_hip(make_scene(**SCENES[name]), gpu_device)))
            res.append((wide, "multi", run_hip(make_scene(P=8000, res=(128, 128), s0=0.03, view=[0, 1, 3, 5, 6, 2, 4, 7, 8]), gpu_device)))
    finally:
        L.f3dg_set_option(b"sort_wide_groups", 0)
    half = len(res) // 2
    for (w0, n0, a), (w1, n1, b) in zip(res[:half], res[half:]):
        assert n0 == n1 and w0 != w1
        for k in ("point_list", "keys_sorted", "ranges", "out_color"):
            assert np.array_equal(a[k], b[k]), (n0, k)


def test_empty_and_all_culled(gpu_device):
    import f3dgaus_amd as f3d
    scene = make_scene(P=100, res=(64, 64))
    # P == 0: zeros (rasterize_points.cu:72,85)
    out, radii, ws = f3d.rasterize_views(
        torch.zeros(0, 3, device=gpu_device), torch.zeros(0, 1, device=gpu_device), scene["viewmatrix"].to(gpu_device),
        scene["projmatrix"].to(gpu_device), scene["campos"].to(gpu_device), torch.tensor([0.3, 0.2, 0.1]),
        image_height=64, image_width=64, tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        sh=torch.zeros(0, 4, 3, device=gpu_device), scales=torch.zeros(0, 3, device=gpu_device),
        rotations=torch.zeros(0, 4, device=gpu_device), sh_degree=1)
    assert out.shape == (1, 9, 64, 64) and float(out.abs().max()) == 0.0 and ws.num_rendered == 0
    # every Gaussian behind the near plane: background only, radii 0
    scene["means3D"] = scene["means3D"].clone()
    scene["means3D"][:, 2] = -1.0
    scene["bg"] = torch.tensor([0.3, 0.2, 0.1])
    h = run_hip(scene, gpu_device)
    o = run_oracle(scene)
    assert h["num_rendered"] == 0 == o["num_rendered"]
    assert (h["radii"] == 0).all()
    assert np.allclose(h["out_color"][0], o["out_color"], atol=0)
    assert np.allclose(h["out_color"][0, :3].reshape(3, -1).T, [0.3, 0.2, 0.1])


def test_overflow_grows_and_retries(gpu_device):
    scene = make_scene(**SCENES["F1_tiny_identity"])
    h = run_hip(scene, gpu_device, max_rendered=64)     # far too small -> F3DG_ERR_OVERFLOW -> grow -> same result
    o = run_oracle(scene)
    assert h["num_rendered"] == o["num_rendered"]
    assert h["workspace"].max_rendered >= o["num_rendered"]
    assert_render_parity(h["out_color"][0], o["out_color"], "overflow-retry")


def test_c1_full_size_single_view(gpu_device):
    """BASELINE config C1 shape: 65,536 Gaussians, one 256x256 view (sigma0 = 0.01, the numerically fragile regime)."""
    scene = make_scene(P=65536, res=(256, 256), s0=0.01, view="oblique")
    h = run_hip(scene, gpu_device)
    o = run_oracle(scene)
    assert h["num_rendered"] == o["num_rendered"]
    assert np.array_equal(h["point_list"], o["point_list"])
    _check_view(h, o, 0, "C1")
    assert_render_parity(h["out_color"][0], o["out_color"], "C1")


@pytest.mark.lab
@pytest.mark.parametrize("name", ["F1_tiny_identity", "F2_oblique_aniso", "F5_odd_size", "F6_small_splats", "F4_filter_scalemod", "C1", "tiny_sigma", "huge_sigma"])
def test_pretest_is_conservative_bit_identical_outputs(name, gpu_device):
    """The float32 pre-test of the compositing kernel may only skip pairs whose alpha is certainly < 1/255:
    every output and every auxiliary plane must be bit-identical with it on and off."""
    from f3dgaus_amd import _lib
    extra = {"C1": dict(P=65536, res=(256, 256), s0=0.01, view="oblique"),
             "tiny_sigma": dict(P=20000, res=(256, 256), s0=0.003, view="oblique"),
             "huge_sigma": dict(P=1500, res=(64, 64), s0=0.3, view="canonical")}
    scene = make_scene(**(SCENES[name] if name in SCENES else extra[name]))
    L = _lib.lib()
    try:
        assert L.f3dg_set_option(b"render_kernel", 1) == 0
        for o in (b"render_pretest", b"render_cull", b"render_queue"):
            assert L.f3dg_set_option(o, 0) == 0
        a = run_hip(scene, gpu_device)              # plain transcription-order kernel
        variants = []
        for pre, cull, que in ((1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)):
            L.f3dg_set_option(b"render_pretest", pre); L.f3dg_set_option(b"render_cull", cull); L.f3dg_set_option(b"render_queue", que)
            variants.append(run_hip(scene, gpu_device))
        L.f3dg_set_option(b"render_kernel", 2)      # render2: four waves per tile, Gaussians across the lanes + conservative ellipse in phase 1
        for rnd in (192, 256):                      # list entries staged per round (fast arithmetic: 192 by default)
            L.f3dg_set_option(b"render_round", rnd)
            variants.append(run_hip(scene, gpu_device))
        L.f3dg_set_option(b"render_kernel", 3)      # render3 (default): one wave64 per 8x8 quadrant, quadrant masks from the binning stage
        variants.append(run_hip(scene, gpu_device))   # the default for launches this small: render3l_fwd_kernel (prefetching windows)
        L.f3dg_set_option(b"render_lowocc", 0)
        variants.append(run_hip(scene, gpu_device))   # the default for everything larger: sliding half-windows (render3s_fwd_kernel)
        for tail in (64, 24, 5, 0):                   # ... with the tail schedule from at most `tail` unsaturated pixels per quadrant on
            L.f3dg_set_option(b"render_tail", tail)
            variants.append(run_hip(scene, gpu_device))
        L.f3dg_set_option(b"render_slide", 0)         # fixed 64-entry windows,
        for dma in (1, 0):                          # records staged by global_load_lds / through registers
            L.f3dg_set_option(b"render_dma", dma)
            variants.append(run_hip(scene, gpu_device))
    finally:
        L.f3dg_set_option(b"render_kernel", 3)
        L.f3dg_set_option(b"render_slide", 1)
        L.f3dg_set_option(b"render_lowocc", 1)
        L.f3dg_set_option(b"render_dma", 1)
        L.f3dg_set_option(b"render_round", 192)
        L.f3dg_set_option(b"render_tail", -1)
        for o in (b"render_pretest", b"render_cull", b"render_queue"):
            L.f3dg_set_option(o, 1)
    for b in variants:
        for k in ("out_color", "final_T", "n_contrib"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert L.f3dg_set_option(b"no_such_option", 1) == _lib.ERR_BAD_ARG


@pytest.mark.parametrize("name", ["F2_oblique_aniso", "F5_odd_size", "F6_small_splats", "F9_long_tile_lists", "C1", "pixel_aligned", "thin"])
def test_packed_schedule_bit_identical(name, gpu_device):
    """The rank-packed compositing kernel (option render_pack, csrc/f3dg_render4.hip) evaluates the stateless part of sparse
    trips for pairs of different pixels in one wave trip and hands the results to the lanes that own the pixels. Per pixel the
    sequence of blended entries and every operation on them is render3s's: the inference outputs must be bit-identical for every
    packing threshold (64: every trip packed, 0: none), and meet the oracle."""
    from f3dgaus_amd import _lib
    extra = {"C1": dict(P=65536, res=(256, 256), s0=0.01, view="oblique"),
             "pixel_aligned": dict(P=65536, res=(256, 256), s0=0.01, view="oblique", n_views=3, seed=3),
             "thin": dict(P=40000, res=(120, 88), s0=0.03, view="oblique", n_views=3, seed=7)}
    scene = make_scene(**(SCENES[name] if name in SCENES else extra[name]))
    if name == "thin":
        scene["opacities"] = scene["opacities"] * 0.04        # nothing saturates: every quadrant walks its whole list
    L = _lib.lib()
    try:
        L.f3dg_set_option(b"render_lowocc", 0)
        assert L.f3dg_set_option(b"render_pack", 0) == 0
        a = run_hip(scene, gpu_device, save_aux=False)          # render3s
        assert b"render3s" in L.f3dg_debug_last_render_kernel()
        assert L.f3dg_set_option(b"render_pack", 1) == 0
        variants = []
        for th in (64, 32, 12, 3, 0):
            assert L.f3dg_set_option(b"render_pack_th", th) == 0
            variants.append(run_hip(scene, gpu_device, save_aux=False))
        assert b"render4" in L.f3dg_debug_last_render_kernel()
        # the default: packed for inference launches in the reference's arithmetic, render3s in fast arithmetic
        L.f3dg_set_option(b"render_pack", -1)
        L.f3dg_set_option(b"render_pack_th", 32)
        variants.append(run_hip(scene, gpu_device, save_aux=False))
        import helpers
        assert (b"render4" if helpers.RENDER_MODE == "exact" else b"render3s") in L.f3dg_debug_last_render_kernel()
    finally:
        L.f3dg_set_option(b"render_pack", -1)
        L.f3dg_set_option(b"render_pack_th", 32)
        L.f3dg_set_option(b"render_lowocc", 1)
    for b in variants:
        assert np.array_equal(a["out_color"].view(np.uint32), b["out_color"].view(np.uint32))
    for v in range(scene["viewmatrix"].shape[0]):
        o = run_oracle(scene, view=v)
        assert_render_parity(variants[1]["out_color"][v], o["out_color"], "packed %s view %d" % (name, v))
    # the SAVE_AUX variant (a forward that a backward follows): also the auxiliary planes -- final transmittance, the distortion sums,
    # the last and the median contributor as 1-based list positions -- identical to render3s's
    try:
        L.f3dg_set_option(b"render_lowocc", 0)
        L.f3dg_set_option(b"render_pack", 0)
        a = run_hip(scene, gpu_device, save_aux=True)
        assert b"render3s" in L.f3dg_debug_last_render_kernel()
        aux = []
        for th in (64, 32, 5):
            L.f3dg_set_option(b"render_pack", 1)
            L.f3dg_set_option(b"render_pack_th", th)
            aux.append(run_hip(scene, gpu_device, save_aux=True))
            assert b"render4" in L.f3dg_debug_last_render_kernel() and b"SAVE_AUX=true" in L.f3dg_debug_last_render_kernel()
    finally:
        L.f3dg_set_option(b"render_pack", -1)
        L.f3dg_set_option(b"render_pack_th", 32)
        L.f3dg_set_option(b"render_lowocc", 1)
    for b in aux:
        for k in ("out_color", "final_T", "n_contrib"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


@pytest.mark.parametrize("name", ["F2_oblique_aniso", "F5_odd_size", "F6_small_splats", "F9_long_tile_lists", "C1", "thin", "two_views"])
def test_small_launch_kernels_bit_identical(name, gpu_device):
    """Launches of at most 2,048 quadrant waves (one or two 256^2 views: the reference's per-view loop) take the latency-chain
    kernels of csrc/f3dg_render4.hip: render3p_fwd_kernel (option render_split = 1: a producer wave prepares the next window while the
    consumer wave composites, render_unroll = 2: two entries per phase-2 trip evaluated as independent instruction streams) and
    render3q_fwd_kernel (render_split = 2 / 3: consumer + evaluator waves + producer). Images and auxiliary planes must be the general kernel's
    (render_lowocc 0) to the bit, for every variant, and meet the oracle."""
    from f3dgaus_amd import _lib
    extra = {"C1": dict(P=65536, res=(256, 256), s0=0.01, view="oblique"),
             "thin": dict(P=40000, res=(120, 88), s0=0.03, view="oblique", n_views=2, seed=7),
             "two_views": dict(P=65536, res=(256, 256), s0=0.01, view="oblique", n_views=2, seed=5)}      # 2,048 quadrants
    scene = make_scene(**(SCENES[name] if name in SCENES else extra[name]))
    if name == "thin":
        scene["opacities"] = scene["opacities"] * 0.04        # nothing saturates: every quadrant walks its whole list
    V = scene["viewmatrix"].shape[0]
    assert V * ((scene["W"] + 15) // 16) * ((scene["H"] + 15) // 16) * 4 <= 2048
    L = _lib.lib()
    res = {}
    try:
        L.f3dg_set_option(b"render_lowocc", 0)           # the general kernel (render3s; render4 in the reference's arithmetic)
        for aux in (False, True):
            res[1, aux] = run_hip(scene, gpu_device, save_aux=aux)
            assert b"render3s" in L.f3dg_debug_last_render_kernel() or b"render4" in L.f3dg_debug_last_render_kernel(), L.f3dg_debug_last_render_kernel()
        L.f3dg_set_option(b"render_lowocc", 1)
        # two waves per quadrant (option render_split): the producer wave prepares the next window while the consumer composites
        L.f3dg_set_option(b"render_split", 1)
        for U in (1, 2):
            L.f3dg_set_option(b"render_unroll", U)
            for aux in (False, True):
                res["split", U, aux] = run_hip(scene, gpu_device, save_aux=aux)
                assert b"render3p" in L.f3dg_debug_last_render_kernel(), L.f3dg_debug_last_render_kernel()
        # the pipeline of waves (render_split = 2 / 3: a consumer wave, 2 / 3 evaluator waves, the producer wave)
        for sp in (2, 3):
            L.f3dg_set_option(b"render_split", sp)
            for aux in (False, True):
                res["split", 10 * sp, aux] = run_hip(scene, gpu_device, save_aux=aux)
                assert b"render3q" in L.f3dg_debug_last_render_kernel(), L.f3dg_debug_last_render_kernel()
        # the defaults: by launch size and arithmetic (one view: render3p in fast arithmetic, render3q in the reference's)
        L.f3dg_set_option(b"render_unroll", -1)
        L.f3dg_set_option(b"render_split", -1)
        res["split", 0, False] = run_hip(scene, gpu_device, save_aux=False)
        import helpers
        one_view = V * ((scene["W"] + 15) // 16) * ((scene["H"] + 15) // 16) * 4 <= 1024
        want = b"render3p" if helpers.RENDER_MODE == "fast" else (b"render3q" if one_view else b"render3p") if helpers.RENDER_MODE == "exact" else b"render3"
        assert want in L.f3dg_debug_last_render_kernel(), L.f3dg_debug_last_render_kernel()
        res["split", 0, True] = run_hip(scene, gpu_device, save_aux=True)
    finally:
        L.f3dg_set_option(b"render_unroll", -1)
        L.f3dg_set_option(b"render_split", -1)
        L.f3dg_set_option(b"render_lowocc", 1)
    for U in (0, 1, 2, 20, 30):
        assert np.array_equal(res[1, False]["out_color"].view(np.uint32), res["split", U, False]["out_color"].view(np.uint32)), ("split", U)
        for k in ("out_color", "final_T", "n_contrib"):
            assert np.array_equal(res[1, True][k].view(np.uint32), res["split", U, True][k].view(np.uint32)), ("split", U, k)
    for v in range(V):
        o = run_oracle(scene, view=v)
        assert_render_parity(res["split", 2, False]["out_color"][v], o["out_color"], "small launch %s view %d" % (name, v))


@pytest.mark.lab
@pytest.mark.parametrize("tail", [64, 16, 3])
def test_tail_schedule_thin_coverage(tail, gpu_device):
    """The tail schedule of the one-wave kernel (option render_tail) on the case it exists for: long tile lists of faint
    Gaussians, so that pixels never saturate and walk the whole list (several views, odd image size, inference and SAVE_AUX
    calls). Held against the oracle and bit-identical to the sliding-window schedule."""
    from f3dgaus_amd import _lib
    scene = make_scene(P=40000, res=(120, 88), s0=0.03, view="oblique", n_views=3, seed=7)
    scene["opacities"] = scene["opacities"] * 0.04        # ~40 entries per pixel, alpha <= 0.04 each: T stays far above 1e-4
    L = _lib.lib()
    try:
        L.f3dg_set_option(b"render_lowocc", 0)
        L.f3dg_set_option(b"render_tail", 0)
        a = run_hip(scene, gpu_device)
        a_inf = run_hip(scene, gpu_device, save_aux=False)
        L.f3dg_set_option(b"render_tail", tail)
        b = run_hip(scene, gpu_device)
        b_inf = run_hip(scene, gpu_device, save_aux=False)
    finally:
        L.f3dg_set_option(b"render_lowocc", 1)
        L.f3dg_set_option(b"render_tail", -1)
    for k in ("out_color", "final_T", "n_contrib"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(a_inf["out_color"].view(np.uint32), b_inf["out_color"].view(np.uint32))
    assert float(b["final_T"][:, 0].min()) > 1e-3         # nothing saturated: every quadrant went through its whole list
    for v in range(3):
        o = run_oracle(scene, view=v)
        assert_render_parity(b["out_color"][v], o["out_color"], "tail %d view %d" % (tail, v))


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations(seed, gpu_device):
    """Randomised sweep over image sizes (incl. non-multiples of 16 and single-tile images), splat sizes, filters and
    background: integer state bit-exact, renders within the north-star tolerance."""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(9, 200)), int(rng.integers(9, 200))
    kw = dict(P=int(rng.integers(50, 6000)), res=(W, H), s0=float(np.exp(rng.uniform(np.log(0.008), np.log(0.12)))),
              seed=int(rng.integers(0, 1000)), view="oblique" if seed % 2 else "canonical", sh_degree=int(rng.integers(0, 2)),
              kernel_size=float(rng.choice([0.0, 0.05, 0.3])), scale_modifier=float(rng.choice([1.0, 0.7, 1.5])),
              behind_fraction=float(rng.choice([0.0, 0.2])), bg=tuple(float(x) for x in rng.uniform(0, 1, 3)),
              aniso=bool(seed % 3 == 0), colors_precomp=bool(seed % 5 == 4))
    scene = make_scene(**kw)
    h = run_hip(scene, gpu_device)
    o = run_oracle(scene)
    assert h["num_rendered"] == o["num_rendered"], kw
    assert np.array_equal(h["point_list"], o["point_list"]), kw
    assert np.array_equal(h["ranges"][0], o["ranges"]), kw
    _check_view(h, o, 0, str(kw), colors_precomp=None if scene["colors_precomp"] is None else scene["colors_precomp"].numpy())
    assert_render_parity(h["out_color"][0], o["out_color"], str(kw))
