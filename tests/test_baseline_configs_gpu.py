"""The BASELINE.json configurations at their FULL sizes, each as one call of the product path (SURVEY 8d / VERDICT round 1 #1):

  C2  196,608 Gaussians, the 120-view orbit @256x256 in ONE f3dg_forward_batched call (30,720 (view, tile) groups);
  C5  1,000,000 Gaussians, 32 views @512x512, SAVE_AUX forward + backward in one call each;
  C3  B = 8 images @256x256 through the cycle aggregation (predictor + 8 renders + 8 re-predictions, merged sets of 589,824).

At these sizes the oracle cannot check every view, so each configuration combines (a) size-independent properties of the whole
batch, evaluated on the device (sortedness and stability of the instance list, tile ranges = segment sizes, alpha + T = 1, a view
rendered alone == the same view inside the batch, additivity of the backward over views) with (b) the oracle on a few complete
views at full size: forward parity at the north-star tolerance and, for C5, compositing-stage gradients within 1e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

import f3dgaus_amd as f3d
from f3dgaus_amd import _lib, cameras, synthetic
from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
from helpers import assert_render_parity, run_oracle

pytestmark = pytest.mark.gpu


def _scene(P, res, V, device, seed=0):
    g = synthetic.make_gaussians(P, s0=0.01, seed=seed, device=device)
    cams = synthetic.orbit_cameras(V, resolution=res, device=device)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    return g, cams, shs


def _render(g, cams, shs, res, views, device, save_aux, workspace=None):
    return f3d.rasterize_views(
        g["xyz"], g["opacity"], cams["viewmatrix"][views], cams["projmatrix"][views], cams["campos"][views],
        torch.zeros(3, device=device), image_height=res, image_width=res, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"],
        sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1, save_aux=save_aux, workspace=workspace)


def _oracle_scene(g, cams, shs, res, view):
    cpu = lambda t: t.detach().cpu()
    return dict(P=g["xyz"].shape[0], W=res, H=res, sh_degree=1, kernel_size=0.0, scale_modifier=1.0, tanfovx=cams["tanfovx"],
                tanfovy=cams["tanfovy"], bg=torch.zeros(3), viewmatrix=cpu(cams["viewmatrix"][view:view + 1]),
                projmatrix=cpu(cams["projmatrix"][view:view + 1]), campos=cpu(cams["campos"][view:view + 1]),
                means3D=cpu(g["xyz"]), opacities=cpu(g["opacity"]), scales=cpu(g["scaling"]), rotations=cpu(g["rotation"]),
                shs=cpu(shs), colors_precomp=None)


def _binning_properties(ws, P, res, V, device, out):
    """Sortedness / stability / range consistency of the whole batch's instance list, on the device."""
    T = ((res + 15) // 16) ** 2
    R, cap = ws.num_rendered, ws.max_rendered
    keys = torch.zeros(cap, dtype=torch.int64, device=device)
    pl = torch.zeros(cap, dtype=torch.int32, device=device)
    ranges = torch.zeros(V * T * 2, dtype=torch.int32, device=device)
    tiles = torch.zeros(V * P, dtype=torch.int32, device=device)
    fT = torch.zeros(V * 4 * res * res, dtype=torch.float32, device=device)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = _lib.lib().f3dg_debug_export(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(ws.buffer), P, res, res, V, cap,
                                      None, None, None, p(tiles), None, None, p(keys), p(pl), p(ranges), p(fT), None, None)
    assert rc == 0
    keys, pl = keys[:R], pl[:R].long()
    assert R == int(tiles.long().sum().item()) and R > 2 * P * V
    d = keys[1:] - keys[:-1]                              # keys < 2^63: signed compare is the unsigned one
    assert bool((d >= 0).all()), "instance list not sorted by (view, tile, depth)"
    ties = d == 0
    assert bool(((pl[1:] - pl[:-1])[ties] > 0).all()), "equal keys not in ascending Gaussian order (stability)"
    tb = int(T - 1).bit_length()
    hi = keys >> 32
    seg = (hi >> tb) * T + (hi & ((1 << tb) - 1))
    counts = torch.bincount(seg, minlength=V * T)
    rg = ranges.view(V * T, 2).long()
    assert torch.equal(rg[:, 1] - rg[:, 0], counts), "tile ranges != segment sizes"
    nz = counts > 0
    assert torch.equal(rg[nz, 0], (torch.cumsum(counts, 0) - counts)[nz])
    assert bool(torch.isfinite(out).all()) and float(out[:, 7].min()) >= 0 and float(out[:, 7].max()) <= 1 + 1e-5
    assert float((out[:, 7] + fT.view(V, 4, res, res)[:, 0] - 1).abs().max()) < 2e-5, "alpha + T != 1"
    return R


def test_c2_120_views_in_one_call(gpu_device):
    P, res, V = 196608, 256, 120
    g, cams, shs = _scene(P, res, V, gpu_device)
    out, radii, ws = _render(g, cams, shs, res, slice(None), gpu_device, save_aux=True)
    assert out.shape == (V, 9, res, res) and radii.shape == (V, P)
    _binning_properties(ws, P, res, V, gpu_device, out)
    # a view rendered alone == the same view inside the 120-view launch, bit for bit
    for v in (0, 77):
        single, _, _ = _render(g, cams, shs, res, slice(v, v + 1), gpu_device, save_aux=True)      # (same arithmetic mode as `out`)
        assert torch.equal(single[0], out[v]), v
    # the oracle on three spread views at full size
    # ... in both arithmetic modes: `out` is a SAVE_AUX call (the reference's operation order), `inf` the inference call (fast mode)
    inf, _, _ = _render(g, cams, shs, res, slice(None), gpu_device, save_aux=False)
    for v in (3, 59, 118):
        o = run_oracle(_oracle_scene(g, cams, shs, res, v))
        assert np.array_equal(radii[v].cpu().numpy(), o["radii"])
        assert_render_parity(out[v].cpu().numpy(), o["out_color"], f"C2 view {v}")
        assert_render_parity(inf[v].cpu().numpy(), o["out_color"], f"C2 view {v} (fast mode)")


def test_c5_one_million_gaussians_32_views_forward_backward(gpu_device):
    P, res, V = 1000000, 512, 32
    g, cams, shs = _scene(P, res, V, gpu_device)
    out, radii, ws = _render(g, cams, shs, res, slice(None), gpu_device, save_aux=True)
    R = _binning_properties(ws, P, res, V, gpu_device, out)
    single, _, _ = _render(g, cams, shs, res, slice(17, 18), gpu_device, save_aux=True)
    assert torch.equal(single[0], out[17])

    # backward of the whole batch: random dL/dpix on channels 0-6 and 8
    gen = torch.Generator(device="cpu").manual_seed(11)
    dpix = torch.randn(V, 9, res, res, generator=gen).to(gpu_device)
    dpix[:, 7] = 0
    bg = torch.zeros(3, device=gpu_device)
    bwd = lambda w, r, views, dp: rasterize_backward_raw(
        w, g["xyz"], shs, None, g["scaling"], g["rotation"], r, dp, 1, cams["viewmatrix"][views], cams["projmatrix"][views],
        cams["campos"][views], bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)
    gr = bwd(ws, radii, slice(None), dpix)
    for k, v in gr.items():
        assert bool(torch.isfinite(v).all()), k
    assert float(gr["dL_dconic"].abs().max()) == 0 and float(gr["dL_dcov3D"].abs().max()) == 0      # known answers
    never = (radii == 0).all(0)
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"):
        assert not bool(gr[k][never].any()), k
    del ws

    # additivity over views at full size: backward of the 2-view batch {5, 20} == sum of the two single-view backwards
    pair = torch.tensor([5, 20], device=gpu_device)
    o2, r2, w2 = _render(g, cams, shs, res, pair, gpu_device, save_aux=True)
    g2 = bwd(w2, r2, pair, dpix[pair])
    acc = None
    singles = {}
    for v in (5, 20):
        o1, r1, w1 = _render(g, cams, shs, res, slice(v, v + 1), gpu_device, save_aux=True)
        g1 = bwd(w1, r1, slice(v, v + 1), dpix[v:v + 1])
        singles[v] = (o1, r1, g1)
        acc = {k: g1[k].clone() for k in g1} if acc is None else {k: acc[k] + g1[k] for k in acc}
    for k in ("dL_dopacity", "dL_dsh", "dL_dmeans3D"):
        m = float(acc[k].abs().max())
        assert float((g2[k] - acc[k]).abs().max()) <= 2e-5 * m, k

    # the oracle on ONE complete view at full size: forward parity + compositing-stage gradients
    v = 20
    o = run_oracle(_oracle_scene(g, cams, shs, res, v))
    o1, r1, g1 = singles[v]
    assert np.array_equal(r1[0].cpu().numpy(), o["radii"])
    assert_render_parity(o1[0].cpu().numpy(), o["out_color"], "C5 view 20")
    go = o["oracle"].backward(dpix[v].cpu().numpy())
    rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
    assert rel(g1["dL_dview2gaussian"][0].cpu().numpy(), go["dL_dview2gaussian"]) <= 1e-5
    assert rel(g1["dL_dopacity"].cpu().numpy(), go["dL_dopacity"]) <= 1e-5
    assert rel(g1["dL_dcolors"][0].cpu().numpy(), go["dL_dcolor"]) <= 1e-5
    assert rel(g1["dL_dmeans2D"][0].cpu().numpy(), go["dL_dmean2D"]) <= 2e-5
    assert rel(g1["dL_dsh"].cpu().numpy(), go["dL_dsh"]) <= 1e-5
    print(f"C5: {R} instances ({R / P / V:.2f} per Gaussian and view)")


def cycle_loop_check(device, B, res, V=8, seed=0):
    """cycle_aggregate (batched renders, in-place merge) against the reference-shaped loop (visualize.py:283-340: per-view
    renderer calls, per-view predictor calls, torch.cat merge) run with the same operators. Random weights."""
    return _cycle_loop_check(device, B, res, V, seed)


def _cycle_loop_check(device, B, res, V, seed):
    torch.manual_seed(seed)
    cfg = cameras.default_cfg(res)
    model = f3d.Unet_GS_gtunet(cfg, renderer=f3d.render_predicted_more_v2_gof).to(device).eval()
    gen = torch.Generator().manual_seed(3)
    images = torch.rand(B, 3, res, res, generator=gen).to(device)
    depth = (torch.rand(B, 1, res, res, generator=gen) * 2 + 6.667).to(device)
    rig = cameras.OrbitRig(cfg)
    merged, renders = f3d.cycle.cycle_aggregate(model, images, depth, cfg, rig=rig, num_views=V, return_renders=True)
    HW = res * res
    assert merged["xyz"].shape == (B, (1 + V) * HW, 3) and merged["features_rest"].shape == (B, (1 + V) * HW, 3, 3)
    assert merged["opacity"].shape == (B, (1 + V) * HW, 1) and merged["rotation"].shape == (B, (1 + V) * HW, 4)
    with torch.no_grad():
        bg = torch.zeros(B, 3, device=device)
        cano, ob = rig.canonical, rig.orbit(V)
        x0 = torch.cat([images, torch.ones_like(images[:, :1])], 1).unsqueeze(1)
        _, _, gsb = model(x0, bg, cano.view_to_world_transforms.expand(B, 1, 4, 4).to(device),
                          cano.source_cv2wT_quat.expand(B, 1, 4).to(device), unet_depth=depth)
        # two U-Net passes over the same input. The build's own kernels are run-to-run bit-reproducible (the channels-last GroupNorm sums
        # its partial moments in a fixed order since round 5) and so is the attention; what is not are three of MIOpen's 36 convolution
        # configurations -- the 3x3 / 1x1 convolutions with 256 / 512 input channels at 32x32, split-K kernels that add with atomics --
        # measured by tools/unet_determinism.py: 2e-6 absolute on a backbone output of 4.6, which the splat head turns into 5.5e-6 ..
        # 8.1e-6 relative on the Gaussians (four runs, profiles/r05_final/unet_determinism.md). MIOpen's deterministic attribute is no
        # way out (it falls back to kernels 200x slower: 15 s per 8-image pass). Bars: 3e-5 here (round 4: 1e-4), 1e-4 after the
        # re-predictions (round 4: 1e-3; measured 1.6e-6 .. 2.2e-6)
        first = worst = 0.0
        for k in gsb:
            d = (gsb[k] - merged[k][:, :HW]).abs().max().item()
            first = max(first, d / max(1.0, gsb[k].abs().max().item()))
            assert d <= 3e-5 * max(1.0, gsb[k].abs().max().item()), (k, d)
        # from here on use the SAME first-pass Gaussians for both loops (sigma ~ 0.01 scenes amplify 1-ulp input differences to
        # 1e-2 in the render), so renders must agree bit for bit
        gsb = {k: merged[k][:, :HW].contiguous() for k in gsb}
        wv, fp, cc = (t.to(device) for t in (ob.world_view_transforms, ob.full_proj_transforms, ob.camera_centers))
        ref = {k: [v] for k, v in gsb.items()}
        for th in range(V):
            rgb, dep, alp = [], [], []
            for bb in range(B):
                od = f3d.render_predicted_more_v2_gof(gsb, bb, wv[th:th + 1], fp[th:th + 1], cc[th:th + 1], bg[0:1], cfg)
                rgb.append(od["render"].reshape(1, 3, res, res)); dep.append(od["rendered_depth"].reshape(1, 1, res, res))
                alp.append(od["rendered_alpha"].reshape(1, 1, res, res))
            rgb, dep, alp = torch.cat(rgb).clamp(0, 1), torch.cat(dep), torch.cat(alp)
            assert torch.equal(rgb, renders["rgb"][:, th]) and torch.equal(dep, renders["depth"][:, th])
            assert torch.equal(alp, renders["alpha"][:, th])
            xin = torch.cat([rgb, alp], 1).unsqueeze(1)
            _, _, gi = model(xin, bg, ob.view_to_world_transforms[th:th + 1].expand(B, 1, 4, 4).to(device),
                             ob.source_cv2wT_quat[th:th + 1].expand(B, 1, 4).to(device), unet_depth=dep)
            for k in ref:
                ref[k].append(gi[k])
        ref = {k: torch.cat(v, 1) for k, v in ref.items()}
    for k in ref:
        assert ref[k].shape == merged[k].shape, k
        d = (ref[k] - merged[k]).abs().max().item()
        worst = max(worst, d / max(1.0, ref[k].abs().max().item()))
        assert d <= 1e-4 * max(1.0, ref[k].abs().max().item()), (k, d)
    print(f"cycle_loop_check B={B} V={V}: first pass max rel {first:.2e} (bar 3e-5), merged sets max rel {worst:.2e} (bar 1e-4)")
    return merged, cfg, rig


def test_c3_cycle_aggregation_batch_8_at_256(gpu_device):
    B, res = 8, 256
    merged, cfg, rig = cycle_loop_check(gpu_device, B, res)
    assert merged["xyz"].shape == (B, 589824, 3)
    orbit = f3d.cycle.render_orbit(merged, cfg, rig=rig, num_views=4, views_per_call=4)
    assert orbit["render"].shape == (B, 4, 3, res, res) and bool(torch.isfinite(orbit["render"]).all())


@pytest.mark.parametrize("V", [2, 8])
def test_c3_cycle_aggregation_full_batch_64_at_256(V, gpu_device):
    """BASELINE C3 at its full batch: 64 images @256^2 through predict -> render V views -> re-predict -> merge (the
    loop was held against the reference-shaped loop at B = 8 only). The same check at B = 64 -- every cycle render bit-identical to the
    per-image, per-view renderer calls, the merged sets equal to the reference-shaped concatenation -- and two views of two images' merged
    sets through the oracle at full size. V = 8 is C3 itself (589,824 Gaussians per merged set), V = 2 the same loop with two cycle views.
    Until round 4 the first 64-image backbone pass spent 6-7 minutes in MIOpen on a fresh box; the predictor now runs fp32 passes of more
    than 8 images as chunks of 8 (cfg['model']['backbone_chunk']: the per-image cost is flat from 8 on, and the 8-image shapes are the
    ones test_c3_cycle_aggregation_batch_8_at_256 has prepared already): V = 8 -- C3 itself -- runs in the default suite; the two-view variant stays behind F3DG_SLOW_TESTS."""
    import os
    if V != 8 and not os.environ.get("F3DG_SLOW_TESTS"):
        pytest.skip("the two-view variant of C3's cycle loop at B = 64: set F3DG_SLOW_TESTS=1")
    B, res = 64, 256
    merged, cfg, rig = cycle_loop_check(gpu_device, B, res, V=V)
    assert merged["xyz"].shape == (B, (1 + V) * 65536, 3)
    cams = synthetic.orbit_cameras(16, resolution=res, device=gpu_device)
    for b, c in ((3, 5), (41, 12)):
        g = {k: merged[k][b].contiguous() for k in ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest")}
        shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
        out, radii, _ = _render(g, cams, shs, res, slice(c, c + 1), gpu_device, save_aux=False)
        o = run_oracle(_oracle_scene(g, cams, shs, res, c))
        assert np.array_equal(radii[0].cpu().numpy(), o["radii"])
        assert_render_parity(out[0].cpu().numpy(), o["out_color"], f"C3 merged set of image {b}, orbit camera {c}")


def test_c3_batch_64_cycle_views_in_one_sets_call(gpu_device):
    """BASELINE C3's cycle render at its full batch: B = 64 images x 65,536 Gaussians x 8 views @256x256 as ONE f3dg_forward_sets
    call of 512 views = 131,072 (view, tile) groups, which does not fit the 16-bit group stream: the natural u32 path of the
    binning stage. An inference call keeps no auxiliary planes, so the list properties are evaluated against the depth of every
    (view, Gaussian) recomputed here in the kernel's float32 operation order; two views go through the oracle at full size."""
    B, Pset, res, Vs = 64, 65536, 256, 8
    V, T = B * Vs, (res // 16) ** 2
    sets = [synthetic.make_gaussians(Pset, s0=0.01, seed=100 + b, device=gpu_device) for b in range(B)]
    g = {k: torch.cat([s[k] for s in sets], 0).contiguous() for k in sets[0]}
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    cams = synthetic.orbit_cameras(Vs, resolution=res, device=gpu_device)
    vm, pm, cp = (cams[k].repeat(B, *([1] * (cams[k].ndim - 1))) for k in ("viewmatrix", "projmatrix", "campos"))
    out, radii, ws = f3d.rasterize_views(
        g["xyz"], g["opacity"], vm, pm, cp, torch.zeros(3, device=gpu_device), image_height=res, image_width=res,
        tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"], rotations=g["rotation"], sh_degree=1,
        n_sets=B)
    assert out.shape == (V, 9, res, res) and radii.shape == (V, Pset)
    assert (V - 1) << 8 | 255 > 0xFFFF            # (view << tile_bits | tile) needs more than 16 bits
    R, cap = ws.num_rendered, ws.max_rendered
    pl = torch.zeros(cap, dtype=torch.int32, device=gpu_device)
    ranges = torch.zeros(V * T * 2, dtype=torch.int32, device=gpu_device)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = _lib.lib().f3dg_debug_export(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(ws.buffer), Pset, res, res, V, cap,
                                      None, None, None, None, None, None, None, p(pl), p(ranges), None, None, None)
    assert rc == 0
    pl = pl[:R].long()
    rg = ranges.view(V * T, 2).long()
    counts = rg[:, 1] - rg[:, 0]
    nz = counts > 0
    assert int(counts.sum()) == R and bool((counts >= 0).all()) and R > 2 * Pset * V
    assert torch.equal(rg[nz, 0], (torch.cumsum(counts, 0) - counts)[nz]), "ranges do not partition the list in (view, tile) order"
    assert int(pl.min()) >= 0 and int(pl.max()) < Pset
    # depth of every (view, Gaussian) of its own set in the kernel's order: ((m2 x + m6 y) + m10 z) + m14 (auxiliary.h:86-94)
    seg_of = torch.repeat_interleave(torch.arange(V * T, device=gpu_device), counts)
    view_of = seg_of // T
    xyz = g["xyz"].view(B, Pset, 3)[view_of // Vs, pl]
    m = vm.reshape(V, 16)[view_of]
    depth = ((m[:, 2] * xyz[:, 0] + m[:, 6] * xyz[:, 1]) + m[:, 10] * xyz[:, 2]) + m[:, 14]
    key = (seg_of << 32) | depth.view(torch.int32).long()           # depths are positive: their bit patterns order like the values
    d = key[1:] - key[:-1]
    assert bool((d >= 0).all()), "list not sorted by (view, tile, depth)"
    assert bool(((pl[1:] - pl[:-1])[d == 0] > 0).all()), "equal keys not in ascending Gaussian order"
    # every visible Gaussian is in the list of the tile that holds its projected centre... at least: radii > 0 <=> instantiated somewhere
    inst = torch.zeros(V * Pset, dtype=torch.bool, device=gpu_device)
    inst[view_of * Pset + pl] = True
    assert bool((inst.view(V, Pset) <= (radii > 0)).all())
    assert bool(torch.isfinite(out).all()) and float(out[:, 7].min()) >= 0 and float(out[:, 7].max()) <= 1 + 1e-5
    # the same view rendered alone (one set, one camera) is bit-identical; two views against the oracle
    for v in (5, 8 * 37 + 2):
        b, c = divmod(v, Vs)
        single, _, _ = _render(sets[b], cams, torch.cat([sets[b]["features_dc"], sets[b]["features_rest"]], 1).contiguous(), res,
                               slice(c, c + 1), gpu_device, save_aux=False)
        assert torch.equal(single[0], out[v]), v
        o = run_oracle(_oracle_scene(sets[b], cams, torch.cat([sets[b]["features_dc"], sets[b]["features_rest"]], 1), res, c))
        assert np.array_equal(radii[v].cpu().numpy(), o["radii"])
        assert_render_parity(out[v].cpu().numpy(), o["out_color"], f"C3 set {b} camera {c}")
