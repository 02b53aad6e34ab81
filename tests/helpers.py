"""Shared test plumbing: scene construction, the HIP path through the C ABI, the oracle, comparison metrics."""
import ctypes as C
import math

import numpy as np
import torch

import f3dgaus_amd as f3d
from f3dgaus_amd import _lib, synthetic
from oracle import gof as oracle_gof


RENDER_MODE = None      # "fast" / "exact" when a module's fixture forces the compositing arithmetic; None: the library default per call


def make_scene(P, res=(64, 64), s0=0.05, seed=0, view="canonical", n_views=1, sh_degree=1, colors_precomp=False,
               kernel_size=0.0, scale_modifier=1.0, behind_fraction=0.0, bg=(0.0, 0.0, 0.0), aniso=False, depth_range=None, pixel_ordered=False):
    W, H = res
    if pixel_ordered:       # one Gaussian per pixel of a res x res input image, id = y * res + x (what the predictor hands the rasterizer)
        assert P % (W * H) == 0 and W == H          # (a multiple: that many blocks one after the other, as a merged set of several images)
        blocks = [synthetic.make_pixel_gaussians(W, s0=s0, seed=seed + 17 * b, sh_rest=max(3, (sh_degree + 1) ** 2 - 1)) for b in range(P // (W * H))]
        g = {k: torch.cat([b[k] for b in blocks], 0).contiguous() for k in blocks[0]}
    else:
        g = synthetic.make_gaussians(P, s0=s0, seed=seed, behind_fraction=behind_fraction, sh_rest=max(3, (sh_degree + 1) ** 2 - 1))
    if depth_range is not None:   # spread the Gaussians over view-space depths z0..z1 (log-uniform), keeping their image positions:
        gen = torch.Generator().manual_seed(seed + 29)      # the NDC depth map then spans ~0.2, so the distortion channel reaches 1e-3..1e-2
        z0, z1 = depth_range
        znew = torch.exp(torch.rand(P, generator=gen) * (math.log(z1) - math.log(z0)) + math.log(z0))
        ratio = (znew / g["xyz"][:, 2]).unsqueeze(1)
        g["xyz"] = g["xyz"] * ratio
        g["scaling"] = g["scaling"] * ratio
    if aniso:   # anisotropic scales spanning 1e-3 .. 0.2 (fixture F2)
        gen = torch.Generator().manual_seed(seed + 17)
        g["scaling"] = torch.exp(torch.rand(P, 3, generator=gen) * (math.log(0.2) - math.log(1e-3)) + math.log(1e-3))
    cams = synthetic.orbit_cameras(8, resolution=max(W, H), include_canonical=True)
    if view == "canonical":
        idx = [0]
    elif view == "oblique":
        idx = [3]                     # orbit view 2 (index 0 is the canonical camera)
    else:
        idx = list(view)
    if n_views > 1 and len(idx) == 1:
        idx = list(range(n_views))
    scene = dict(
        P=P, W=W, H=H, sh_degree=sh_degree, kernel_size=kernel_size, scale_modifier=scale_modifier,
        tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], bg=torch.tensor(bg, dtype=torch.float32),
        viewmatrix=cams["viewmatrix"][idx].contiguous(), projmatrix=cams["projmatrix"][idx].contiguous(),
        campos=cams["campos"][idx].contiguous(), means3D=g["xyz"], opacities=g["opacity"], scales=g["scaling"],
        rotations=g["rotation"], shs=torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous(),
        colors_precomp=None)
    if colors_precomp:
        gen = torch.Generator().manual_seed(seed + 5)
        scene["colors_precomp"] = torch.rand(P, 3, generator=gen)
        scene["shs"] = None
    return scene


def run_oracle(scene, view=0):
    o = oracle_gof.Oracle()
    npy = lambda t: None if t is None else t.detach().cpu().numpy()
    out, radii, R = o.forward(
        means3D=npy(scene["means3D"]), opacities=npy(scene["opacities"]), viewmatrix=npy(scene["viewmatrix"][view]),
        projmatrix=npy(scene["projmatrix"][view]), campos=npy(scene["campos"][view]), tanfovx=scene["tanfovx"],
        tanfovy=scene["tanfovy"], W=scene["W"], H=scene["H"], bg=npy(scene["bg"]), shs=npy(scene["shs"]),
        colors_precomp=npy(scene["colors_precomp"]), scales=npy(scene["scales"]), rotations=npy(scene["rotations"]),
        sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"], kernel_size=scene["kernel_size"])
    res = o.intermediates()
    res.update(out_color=out, radii=radii, oracle=o)
    return res


def run_hip(scene, device, save_aux=True, max_rendered=None):
    """All views of the scene through f3dg_forward_batched; returns outputs + exported internal state (numpy)."""
    dev = lambda t: None if t is None else t.to(device)
    # The exported lists are compared with the reference's, so they are built without tile culling (option "tile_cull" = 0). The
    # default (culled lists: a Gaussian is instantiated only in the tiles its conservative ellipse reaches) must give the same
    # images to the bit with no more instances: checked here on every scene that goes through this helper.
    culled = None
    if max_rendered is None:
        culled = f3d.rasterize_views(
            dev(scene["means3D"]), dev(scene["opacities"]), dev(scene["viewmatrix"]), dev(scene["projmatrix"]),
            dev(scene["campos"]), dev(scene["bg"]), image_height=scene["H"], image_width=scene["W"],
            tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], sh=dev(scene["shs"]),
            colors_precomp=dev(scene["colors_precomp"]), scales=dev(scene["scales"]), rotations=dev(scene["rotations"]),
            sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"], kernel_size=scene["kernel_size"],
            save_aux=save_aux, exact=_exact_flag())
        culled = (culled[0].clone(), culled[1].clone(), culled[2].num_rendered)
    # (per call -- F3DG_FLAG_NO_TILE_CULL --, not through the process-wide option)
    return _run_hip_reference_lists(scene, device, save_aux, max_rendered, culled)


def _exact_flag():
    """The per-call arithmetic selection of the module fixture's RENDER_MODE (None: the process default)."""
    return None if RENDER_MODE is None else RENDER_MODE == "exact"


def _run_hip_reference_lists(scene, device, save_aux, max_rendered, culled):
    dev = lambda t: None if t is None else t.to(device)
    out, radii, ws = f3d.rasterize_views(
        dev(scene["means3D"]), dev(scene["opacities"]), dev(scene["viewmatrix"]), dev(scene["projmatrix"]),
        dev(scene["campos"]), dev(scene["bg"]), image_height=scene["H"], image_width=scene["W"],
        tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], sh=dev(scene["shs"]),
        colors_precomp=dev(scene["colors_precomp"]), scales=dev(scene["scales"]), rotations=dev(scene["rotations"]),
        sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"], kernel_size=scene["kernel_size"],
        save_aux=save_aux, max_rendered=max_rendered, tile_cull=False, small_path=False, exact=_exact_flag())      # (the general path's internals are exported below)
    V, P, W, H = scene["viewmatrix"].shape[0], scene["P"], scene["W"], scene["H"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    cap = ws.max_rendered
    R = ws.num_rendered
    e = dict(
        rec=torch.zeros(V * max(P, 1) * 16, dtype=torch.float32, device=device),
        means2D=torch.zeros(V * max(P, 1) * 2, dtype=torch.float32, device=device),
        conic=torch.zeros(V * max(P, 1) * 4, dtype=torch.float32, device=device),
        tiles=torch.zeros(V * max(P, 1), dtype=torch.int32, device=device),
        offsets=torch.zeros(V * max(P, 1), dtype=torch.int32, device=device),
        clamped=torch.zeros(V * max(P, 1), dtype=torch.uint8, device=device),
        keys=torch.zeros(max(cap, 1), dtype=torch.int64, device=device),
        point_list=torch.zeros(max(cap, 1), dtype=torch.int32, device=device),
        ranges=torch.zeros(V * T * 2, dtype=torch.int32, device=device),
        final_T=torch.zeros(V * 4 * H * W, dtype=torch.float32, device=device),
        n_contrib=torch.zeros(V * 2 * H * W, dtype=torch.int32, device=device),
        depths=torch.zeros(V * max(P, 1), dtype=torch.float32, device=device))
    if P > 0:
        # an inference call (no SAVE_AUX) keeps the records, the final list and the ranges only: the other planes stay zero here
        always = ("rec", "point_list", "ranges")
        rc = _lib.lib().f3dg_debug_export(
            C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), P, W, H, V, cap,
            *[C.c_void_p(e[k].data_ptr()) if (save_aux or k in always) else None
              for k in ("rec", "means2D", "conic", "tiles", "offsets", "clamped", "keys", "point_list", "ranges", "final_T", "n_contrib", "depths")])
        assert rc == 0
    torch.cuda.synchronize()
    if culled is not None:
        assert torch.equal(culled[0], out) and torch.equal(culled[1], radii) and culled[2] <= R, "tile culling changed the result"
    rec = e["rec"].cpu().numpy().reshape(V, max(P, 1), 16)
    res = dict(
        out_color=out.cpu().numpy(), radii=radii.cpu().numpy(), num_rendered=R,
        view2gaussian=rec[:, :, 0:10], opac=rec[:, :, 10], rgb=rec[:, :, 12:15], depths=e["depths"].cpu().numpy().reshape(V, max(P, 1)),
        means2D=e["means2D"].cpu().numpy().reshape(V, max(P, 1), 2),
        conic_opacity=e["conic"].cpu().numpy().reshape(V, max(P, 1), 4),
        tiles_touched=e["tiles"].cpu().numpy().view(np.uint32).reshape(V, max(P, 1)),
        point_offsets=e["offsets"].cpu().numpy().view(np.uint32).reshape(V, max(P, 1)),
        clamped=e["clamped"].cpu().numpy().reshape(V, max(P, 1)),
        keys_sorted=e["keys"].cpu().numpy().view(np.uint64)[:R],
        point_list=e["point_list"].cpu().numpy().view(np.uint32)[:R],
        ranges=e["ranges"].cpu().numpy().view(np.uint32).reshape(V, T, 2),
        final_T=e["final_T"].cpu().numpy().reshape(V, 4, H, W),
        n_contrib=e["n_contrib"].cpu().numpy().view(np.uint32).reshape(V, 2, H, W), workspace=ws)
    return res


def psnr(a, b, peak=1.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 200.0 if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def frac_within(a, b, atol, rtol=0.0):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return float(np.mean(d <= atol + rtol * np.abs(b)))


def assert_render_parity(hip_out, ora_out, label="", dist_big_rtol=1e-3):
    """SURVEY section 8d 'Parity check used with timing' -- tolerance 1e-4 (north_star), robust to the isolated
    pixels a 1-ulp expf difference can flip at the alpha<1/255, T<1e-4 and T>0.5 discontinuities."""
    rgb_h, rgb_o = hip_out[0:3], ora_out[0:3]
    assert frac_within(rgb_h, rgb_o, 1e-4) >= 0.999, f"{label} rgb: {frac_within(rgb_h, rgb_o, 1e-4)}"
    assert psnr(rgb_h, rgb_o) >= 80.0, f"{label} rgb psnr {psnr(rgb_h, rgb_o)}"
    assert frac_within(hip_out[7], ora_out[7], 1e-4) >= 0.999, f"{label} alpha"
    assert frac_within(hip_out[3:6], ora_out[3:6], 1e-4) >= 0.999, f"{label} normal"
    assert frac_within(hip_out[6], ora_out[6], 0.0, 1e-4) >= 0.999, f"{label} depth"
    # distortion: values ~1e-7..1e-5 built from float32 accumulations (dist1, dist2, distortion) that cancel strongly; a 1-ulp
    # difference of exp() is amplified to a few percent of the value (SURVEY appendix A.2 measured ~3 % relative / 3.5e-7 absolute
    # between faithful implementations; the oracle against itself with tan_fov moved by one ulp: 4-17 % median relative). Stated per
    # arithmetic mode (RENDER_MODE is set by the fixture of the modules that run both):
    #   exact (the reference's operation order; only expf differs):  |d| <= 1e-6 + 1e-3 |ref|
    #   fast  (hardware exp / rcp, FMA-contracted accumulations):    |d| <= 1e-6 + 5e-2 |ref|  -- 3-17 % median relative on the
    #         sigma0 = 0.01 scenes (profiles/*/parity_report.md), absolute <= 2.5e-6; putting the reference's operations back into
    #         the fast path costs 27 % of the kernel and still leaves 3-7 % (tools/ab_exact_dist.sh), so the deviation is declared
    rtol_small = 1e-3 if RENDER_MODE == "exact" else 5e-2
    assert frac_within(hip_out[8], ora_out[8], 1e-6, rtol_small) >= 0.999, f"{label} distortion ({RENDER_MODE})"
    # ... and where the channel is well conditioned (values above 1e-4: scenes with a real depth spread), SURVEY 8d's rel 1e-3
    big = np.abs(ora_out[8]) > 1e-4
    if big.any():
        assert frac_within(hip_out[8][big], ora_out[8][big], 0.0, dist_big_rtol) >= 0.999, \
            f"{label} distortion (rel {dist_big_rtol:g} on {int(big.sum())} px): {frac_within(hip_out[8][big], ora_out[8][big], 0.0, dist_big_rtol)}"
