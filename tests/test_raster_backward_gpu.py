"""GPU parity of the backward pass (f3dg_backward through the C ABI, and through torch autograd) against the CPU oracle.

Tolerances follow SURVEY section 8c: compositing-stage gradients (view2gaussian, opacity, colours, sh, mean2D) within
1e-5 of the maximum; dmean3D / drot / dscale are cancellation-dominated in float32 -- the reference's own run-to-run
spread is 3e-4 / 2e-2 / 0.4 -- so they are asserted as "error against the fp64 chain-rule truth no worse than the
oracle's own error" (this build accumulates dL/dview2gaussian in float64, so it is typically equal or better)."""
import numpy as np
import pytest
import torch

import f3dgaus_amd as f3d
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
from grad_truth import per_gaussian_truth
from helpers import make_scene, run_oracle

pytestmark = pytest.mark.gpu


def _hip_fwd_bwd(scene, dpix, device):
    dev = lambda t: None if t is None else t.to(device)
    out, radii, ws = f3d.rasterize_views(
        dev(scene["means3D"]), dev(scene["opacities"]), dev(scene["viewmatrix"]), dev(scene["projmatrix"]),
        dev(scene["campos"]), dev(scene["bg"]), image_height=scene["H"], image_width=scene["W"],
        tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], sh=dev(scene["shs"]),
        colors_precomp=dev(scene["colors_precomp"]), scales=dev(scene["scales"]), rotations=dev(scene["rotations"]),
        sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"], kernel_size=scene["kernel_size"], save_aux=True)
    g = rasterize_backward_raw(ws, dev(scene["means3D"]), dev(scene["shs"]), dev(scene["colors_precomp"]),
                               dev(scene["scales"]), dev(scene["rotations"]), radii, torch.from_numpy(dpix).to(device),
                               scene["sh_degree"], dev(scene["viewmatrix"]), dev(scene["projmatrix"]), dev(scene["campos"]),
                               dev(scene["bg"]), scene["tanfovx"], scene["tanfovy"], scene["kernel_size"], scene["scale_modifier"])
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in g.items()}, radii.cpu().numpy()


def _rel(a, b):
    m = np.abs(b).max()
    return 0.0 if m == 0 else float(np.abs(a.astype(np.float64) - b).max() / m)


CASES = {
    "B1_identity_rgb_ones": (dict(P=2000, res=(64, 64), s0=0.05, view="canonical"), "rgb_ones"),
    "B1_identity_random": (dict(P=2000, res=(64, 64), s0=0.05, view="canonical"), "random"),
    "B2_oblique_aniso_random": (dict(P=5000, res=(128, 128), s0=0.02, view="oblique", aniso=True, behind_fraction=0.05, bg=(0.3, 0.1, 0.6)), "random"),
    "B3_filter_scalemod_random": (dict(P=3000, res=(96, 96), s0=0.05, view="oblique", kernel_size=0.1, scale_modifier=0.5), "random"),
    "B4_colors_precomp": (dict(P=2500, res=(100, 72), s0=0.05, view="oblique", colors_precomp=True), "random"),
    "B5_sh_degree2": (dict(P=2500, res=(80, 64), s0=0.05, view="oblique", sh_degree=2, bg=(0.2, 0.4, 0.1)), "random"),      # M = 9  (backward.cu:62-104)
    "B6_sh_degree3": (dict(P=2500, res=(64, 80), s0=0.05, view="oblique", sh_degree=3), "random"),                          # M = 16 (backward.cu:106-138)
}


@pytest.mark.parametrize("name", list(CASES))
def test_backward_vs_oracle(name, gpu_device):
    kw, mode = CASES[name]
    scene = make_scene(**kw)
    H, W = scene["H"], scene["W"]
    if mode == "rgb_ones":
        dpix = np.zeros((9, H, W), np.float32)
        dpix[:3] = 1.0
    else:
        dpix = np.random.default_rng(5).standard_normal((9, H, W)).astype(np.float32)
    o = run_oracle(scene)
    go = o["oracle"].backward(dpix)
    gh, radii = _hip_fwd_bwd(scene, dpix[None], gpu_device)
    assert np.array_equal(radii[0], o["radii"])

    assert not gh["dL_dconic"].any() and not gh["dL_dcov3D"].any()
    culled = o["radii"] == 0
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity"):
        assert not gh[k][culled].any(), k
    # compositing stage + sh: tight
    assert _rel(gh["dL_dview2gaussian"][0], go["dL_dview2gaussian"]) <= 1e-5, name
    assert _rel(gh["dL_dopacity"], go["dL_dopacity"]) <= 1e-5
    assert _rel(gh["dL_dcolors"][0], go["dL_dcolor"]) <= 1e-5
    assert _rel(gh["dL_dmeans2D"][0], go["dL_dmean2D"]) <= 2e-5
    if scene["shs"] is not None:
        assert _rel(gh["dL_dsh"], go["dL_dsh"]) <= 1e-5
    # per-Gaussian stage: no worse than the oracle against the fp64 truth
    truth = per_gaussian_truth(scene, 0, o["radii"], go["dL_dview2gaussian"], go["dL_dcolor"])
    if scene["shs"] is not None:        # dL/dsh is linear in dL/dcolour: well conditioned, checked against the float64 chain rule too
        assert _rel(gh["dL_dsh"], truth["dL_dsh"]) <= 1e-5, name
    # floor = the reference's own run-to-run spread for that gradient (3e-4 / 2e-2 / 0.4 of the maximum)
    for hk, ok, floor in (("dL_dmeans3D", "dL_dmean3D", 3e-4), ("dL_drotations", "dL_drot", 2e-2), ("dL_dscales", "dL_dscale", 0.4)):
        e_h, e_o = _rel(gh[hk], truth[ok]), _rel(go[ok], truth[ok])
        print(f"{name} {hk}: hip-vs-fp64 {e_h:.2e}  oracle-vs-fp64 {e_o:.2e}")
        assert e_h <= max(2.0 * e_o, floor), (name, hk, e_h, e_o)


def test_autograd_function_matches_raw_backward(gpu_device):
    scene = make_scene(P=3000, res=(64, 64), s0=0.05, view="oblique")
    dev = lambda t: t.to(gpu_device)
    leaf = {k: dev(scene[k]).clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    means2D = torch.zeros_like(leaf["means3D"], requires_grad=True)
    rs = f3d.GaussianRasterizationSettings_GOF(64, 64, scene["tanfovx"], scene["tanfovy"], 0.0, torch.zeros(0), dev(scene["bg"]), 1.0,
                                               dev(scene["viewmatrix"][0]), dev(scene["projmatrix"][0]), 1, dev(scene["campos"][0]), False, False)
    color, radii = f3d.GaussianRasterizer_GOF(rs)(means3D=leaf["means3D"], means2D=means2D, shs=leaf["shs"],
                                                 opacities=leaf["opacities"], scales=leaf["scales"], rotations=leaf["rotations"])
    assert color.shape == (9, 64, 64) and radii.dtype == torch.int32 and not radii.requires_grad
    w = torch.from_numpy(np.random.default_rng(2).standard_normal((9, 64, 64)).astype(np.float32)).to(gpu_device)
    (color * w).sum().backward()
    gh, _ = _hip_fwd_bwd(scene, w.cpu().numpy()[None], gpu_device)
    assert _rel(leaf["opacities"].grad.cpu().numpy(), gh["dL_dopacity"]) <= 1e-5
    assert _rel(means2D.grad.cpu().numpy(), gh["dL_dmeans2D"][0]) <= 1e-5
    assert _rel(leaf["shs"].grad.cpu().numpy(), gh["dL_dsh"]) <= 1e-5
    for k, hk in (("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations")):
        assert leaf[k].grad is not None and leaf[k].grad.shape == leaf[k].shape
        assert _rel(leaf[k].grad.cpu().numpy(), gh[hk]) <= 1e-3, k      # float32 atomics order on dcolor feeds the SH->mean path


def test_multi_view_backward_sums_single_view_backwards(gpu_device):
    scene = make_scene(P=2500, res=(64, 64), s0=0.05, view=[1, 4, 7])
    V = 3
    dpix = np.random.default_rng(9).standard_normal((V, 9, 64, 64)).astype(np.float32)
    gb, _ = _hip_fwd_bwd(scene, dpix, gpu_device)
    acc = None
    for v in range(V):
        sv = dict(scene)
        for k in ("viewmatrix", "projmatrix", "campos"):
            sv[k] = scene[k][v:v + 1]
        g1, _ = _hip_fwd_bwd(sv, dpix[v:v + 1], gpu_device)
        assert _rel(gb["dL_dview2gaussian"][v], g1["dL_dview2gaussian"][0]) <= 1e-6
        assert _rel(gb["dL_dcolors"][v], g1["dL_dcolors"][0]) <= 1e-5
        assert _rel(gb["dL_dmeans2D"][v], g1["dL_dmeans2D"][0]) <= 1e-5
        acc = {k: g1[k].astype(np.float64) for k in g1} if acc is None else {k: acc[k] + g1[k] for k in g1}
    for k in ("dL_dopacity", "dL_dsh", "dL_dmeans3D"):
        assert _rel(gb[k], acc[k]) <= 1e-4, k


def test_known_answers_and_kernel_variants(gpu_device):
    """SURVEY 8c known answers on the HIP path: the alpha channel's gradient has no effect (backward.cu never reads
    dL_dpixels[ALPHA_OFFSET]); the culled / DPP-reduced kernel and the lock-step kernel (option render_cull = 0) agree to
    float32 summation order."""
    from f3dgaus_amd import _lib
    scene = make_scene(P=4000, res=(96, 80), s0=0.04, view="oblique", bg=(0.3, 0.1, 0.6))
    dpix = np.random.default_rng(3).standard_normal((1, 9, 80, 96)).astype(np.float32)
    g0, _ = _hip_fwd_bwd(scene, dpix, gpu_device)
    dpix2 = dpix.copy()
    dpix2[:, 7] += 5.0
    g1, _ = _hip_fwd_bwd(scene, dpix2, gpu_device)
    for k in ("dL_dview2gaussian", "dL_dopacity", "dL_dcolors", "dL_dmeans2D"):
        assert _rel(g1[k], g0[k]) <= 1e-6, k
    L = _lib.lib()
    if not L.f3dg_version().endswith(b"lab"):
        return                                             # (the lock-step kernel is compiled in lab builds only)
    try:
        L.f3dg_set_option(b"render_cull", 0)
        g2, _ = _hip_fwd_bwd(scene, dpix, gpu_device)
    finally:
        L.f3dg_set_option(b"render_cull", 1)
    for k in ("dL_dview2gaussian", "dL_dopacity", "dL_dcolors", "dL_dmeans2D", "dL_dsh"):
        assert _rel(g2[k], g0[k]) <= 2e-6, k


@pytest.mark.parametrize("seed", range(5))
def test_backward_random_configurations(seed, gpu_device):
    """Compositing-stage gradients on random image sizes / splat sizes / filters against the oracle."""
    rng = np.random.default_rng(3000 + seed)
    W, H = int(rng.integers(17, 120)), int(rng.integers(17, 120))
    kw = dict(P=int(rng.integers(200, 4000)), res=(W, H), s0=float(np.exp(rng.uniform(np.log(0.02), np.log(0.1)))),
              seed=int(rng.integers(0, 1000)), view="oblique" if seed % 2 else "canonical",
              kernel_size=float(rng.choice([0.0, 0.1])), scale_modifier=float(rng.choice([1.0, 0.7])),
              bg=tuple(float(x) for x in rng.uniform(0, 1, 3)), colors_precomp=bool(seed % 3 == 2))
    scene = make_scene(**kw)
    dpix = rng.standard_normal((9, H, W)).astype(np.float32)
    o = run_oracle(scene)
    go = o["oracle"].backward(dpix)
    gh, radii = _hip_fwd_bwd(scene, dpix[None], gpu_device)
    assert np.array_equal(radii[0], o["radii"])
    assert _rel(gh["dL_dview2gaussian"][0], go["dL_dview2gaussian"]) <= 1e-5, kw
    assert _rel(gh["dL_dopacity"], go["dL_dopacity"]) <= 1e-5, kw
    assert _rel(gh["dL_dcolors"][0], go["dL_dcolor"]) <= 1e-5, kw
    assert _rel(gh["dL_dmeans2D"][0], go["dL_dmean2D"]) <= 2e-5, kw


def test_backward_refuses_a_stale_workspace(gpu_device):
    """f3dg_backward on a workspace whose last forward was an inference call (no auxiliary planes, possibly the small-call path):
    the Python wrapper raises; through the raw C ABI every gradient is zero and f3dg_backward_pairs reports F3DG_ERR_STATE."""
    import ctypes as C
    import f3dgaus_amd as f3d
    from f3dgaus_amd import _lib
    from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
    sc = make_scene(P=3000, res=(64, 64), s0=0.05, view="oblique")
    dev = lambda t: None if t is None else t.to(gpu_device)
    L = _lib.lib()
    for small in (1, 0):
        L.f3dg_set_option(b"small_path", 2 if small else 0)
        try:
            out, radii, ws = f3d.rasterize_views(
                dev(sc["means3D"]), dev(sc["opacities"]), dev(sc["viewmatrix"]), dev(sc["projmatrix"]), dev(sc["campos"]), dev(sc["bg"]),
                image_height=64, image_width=64, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], sh=dev(sc["shs"]), scales=dev(sc["scales"]),
                rotations=dev(sc["rotations"]), sh_degree=1, save_aux=False)
        finally:
            L.f3dg_set_option(b"small_path", 2)
        args = (dev(sc["means3D"]), dev(sc["shs"]), None, dev(sc["scales"]), dev(sc["rotations"]), radii, torch.ones_like(out), 1,
                dev(sc["viewmatrix"]), dev(sc["projmatrix"]), dev(sc["campos"]), dev(sc["bg"]), sc["tanfovx"], sc["tanfovy"], 0.0, 1.0)
        with pytest.raises(RuntimeError, match="inference call"):
            rasterize_backward_raw(ws, *args)
        ws.save_aux = True                      # bypass the host check: the device-side guard
        g = rasterize_backward_raw(ws, *args)
        assert all(float(v.abs().max()) == 0.0 for v in g.values())
        n = C.c_longlong(0)
        assert L.f3dg_backward_pairs(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), C.byref(n)) == _lib.ERR_STATE
        ws.save_aux = False


@pytest.mark.parametrize("kw", [
    dict(P=30000, res=(128, 128), s0=0.01, view="oblique", n_views=3, seed=2),                                     # small splats, several views
    dict(P=4 * 64 * 64, res=(64, 64), s0=0.01, view="oblique", n_views=2, seed=4, pixel_ordered=True),             # pixel-aligned, merged sets
    dict(P=6000, res=(100, 72), s0=0.06, view="oblique", seed=9, bg=(0.4, 0.2, 0.7)),                              # long runs: many pixels per entry
])
def test_dense_backward_agrees_with_the_lock_step_walk(kw, gpu_device):
    """The default compositing backward (render5_bwd_kernel: entry-major batches of (pixel, entry) pairs, the per-pixel recurrence folded
    to one dot product, segmented scans, 128-byte accumulation records) against render3_bwd_kernel (option bwd_dense 0: the lock-step walk
    with the transposed wave reduction): the same arithmetic per pair, sums in a different order -- every gradient array within 2e-6 of
    its maximum -- and the same count of contributing pairs."""
    import ctypes as C
    from f3dgaus_amd import _lib
    L = _lib.lib()
    scene = make_scene(**kw)
    V, H, W = scene["viewmatrix"].shape[0], scene["H"], scene["W"]
    dpix = np.random.default_rng(17).standard_normal((V, 9, H, W)).astype(np.float32)
    try:
        assert L.f3dg_set_option(b"bwd_dense", 0) == 0
        g0, _ = _hip_fwd_bwd(scene, dpix, gpu_device)
        assert L.f3dg_set_option(b"bwd_dense", 1) == 0
        g1, _ = _hip_fwd_bwd(scene, dpix, gpu_device)
    finally:
        L.f3dg_set_option(b"bwd_dense", 1)
    # the compositing stage (what the two kernels compute) and the colour chain, which is linear in it. The per-Gaussian chain rule
    # amplifies a last-bit difference of dL/dview2gaussian by the conditioning of the scene -- at sigma0 = 0.01 by 1e2 (mean3D) to 1e4
    # (rotation, scale): SURVEY 0.9 measures the same spread between two runs of the reference -- and is held to the float64 truth by
    # test_backward_vs_oracle with either kernel, not compared here
    for k in ("dL_dview2gaussian", "dL_dopacity", "dL_dcolors", "dL_dmeans2D", "dL_dsh"):
        assert _rel(g1[k], g0[k]) <= 5e-6, (k, _rel(g1[k], g0[k]))


@pytest.mark.parametrize("dense", [1, 0])
def test_per_view_outputs_are_written_in_full(dense, gpu_device):
    """include/f3dg.h: dL_dmean2D, dL_dcolor and dL_dview2gaussian need no zero-fill by the caller -- the wrapper hands them over
    uninitialised (2 GB less to fill at BASELINE C5). The allocator's free blocks are poisoned with NaN first; Gaussians behind the
    camera and off screen (no list entry, no gradient) must come back as exact zeros, with either compositing backward."""
    from f3dgaus_amd import _lib
    L = _lib.lib()
    scene = make_scene(P=6000, res=(96, 80), s0=0.03, view="oblique", n_views=3, seed=5, behind_fraction=0.2, bg=(0.1, 0.5, 0.3))
    V, H, W = scene["viewmatrix"].shape[0], scene["H"], scene["W"]
    dpix = np.random.default_rng(3).standard_normal((V, 9, H, W)).astype(np.float32)
    try:
        assert L.f3dg_set_option(b"bwd_dense", dense) == 0
        ref, radii = _hip_fwd_bwd(scene, dpix, gpu_device)
        poison = [torch.full((V * 6000 * n,), float("nan"), device=gpu_device) for n in (3, 3, 10, 3, 10)]
        del poison
        got, _ = _hip_fwd_bwd(scene, dpix, gpu_device)
    finally:
        L.f3dg_set_option(b"bwd_dense", 1)
    hidden = radii.reshape(V, -1) <= 0
    assert hidden.any()
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dview2gaussian"):
        assert np.isfinite(got[k]).all(), k
        assert not got[k][hidden].any(), k
    for k in got:           # (both kernels add with float atomics in a run-dependent order: equal to the first run up to that)
        assert _rel(got[k], ref[k]) <= 1e-5 or k in ("dL_dmeans3D", "dL_dscales", "dL_drotations"), k
