"""Shared plumbing of the integrate tests: point clouds, the oracle call, the HIP call through the drop-in API."""
import numpy as np
import torch

from oracle.gof import Oracle

npy = lambda t: None if t is None else t.detach().cpu().numpy()


def make_points(scene, n, seed=0, spread=0.05, cluster=0):
    """Points around the Gaussians (plus some far outside the image / behind the camera); `cluster` extra points
    packed inside ONE pixel footprint (exercises the 256-point sweeps of integrateCUDA, forward.cu:1011-1189)."""
    rng = np.random.default_rng(seed)
    m = npy(scene["means3D"])
    pts = m[rng.integers(0, len(m), n)] + rng.normal(0, spread, (n, 3))
    pts[: n // 50] *= 5.0                      # outside the image
    pts[n // 50: n // 25, 2] = -1.0            # behind the camera
    if cluster:
        base = m[len(m) // 2]
        ray = base / np.linalg.norm(base)
        pts = np.concatenate([pts, base[None] + rng.normal(0, 2e-4, (cluster, 3)) + np.linspace(0, 0.3, cluster)[:, None] * ray[None]])
    return np.ascontiguousarray(pts, dtype=np.float32)


def oracle_integrate(scene, pts, view=0):
    o = Oracle()
    out, ai, ci, radii, R, NI = o.integrate(
        points3D=pts, means3D=npy(scene["means3D"]), opacities=npy(scene["opacities"]),
        viewmatrix=npy(scene["viewmatrix"][view]), projmatrix=npy(scene["projmatrix"][view]),
        campos=npy(scene["campos"][view]), tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], W=scene["W"], H=scene["H"],
        bg=npy(scene["bg"]), shs=npy(scene["shs"]), colors_precomp=npy(scene["colors_precomp"]),
        scales=npy(scene["scales"]), rotations=npy(scene["rotations"]), sh_degree=scene["sh_degree"],
        scale_modifier=scene["scale_modifier"], kernel_size=scene["kernel_size"])
    return dict(out=out, ai=ai, ci=ci, radii=radii, R=R, NI=NI, oracle=o)


def hip_integrate(scene, pts, device, view=0):
    from f3dgaus_amd.diff_gof_rasterization import GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF
    dev = lambda t: None if t is None else t.to(device)
    rs = GaussianRasterizationSettings_GOF(
        image_height=scene["H"], image_width=scene["W"], tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        kernel_size=scene["kernel_size"], subpixel_offset=torch.zeros((scene["H"], scene["W"], 2), device=device),
        bg=dev(scene["bg"]), scale_modifier=scene["scale_modifier"], viewmatrix=dev(scene["viewmatrix"][view]),
        projmatrix=dev(scene["projmatrix"][view]), sh_degree=scene["sh_degree"], campos=dev(scene["campos"][view]),
        prefiltered=False, debug=False)
    color, ai, ci, radii = GaussianRasterizer_GOF(rs).integrate(
        points3D=torch.from_numpy(pts).to(device), means3D=dev(scene["means3D"]), means2D=None,
        opacities=dev(scene["opacities"]), shs=dev(scene["shs"]), colors_precomp=dev(scene["colors_precomp"]),
        scales=dev(scene["scales"]), rotations=dev(scene["rotations"]))
    return dict(out=npy(color), ai=npy(ai), ci=npy(ci), radii=npy(radii))


def assert_integrate_parity(o, h, label, tol=1e-4):
    assert np.array_equal(o["radii"], h["radii"]), f"{label} radii"
    assert np.array_equal(o["out"][8], h["out"][8]), f"{label} points per pixel (distortion channel)"
    assert np.array_equal(o["out"][3:6], h["out"][3:6]) and not o["out"][3:6].any(), f"{label} channels 3..5 must stay 0"
    for name, a, b in (("colour", o["out"][:3], h["out"][:3]), ("alpha", o["out"][7], h["out"][7]),
                       ("alpha_integrated", o["ai"], h["ai"]), ("color_integrated", o["ci"], h["ci"])):
        d = np.abs(a - b)
        assert (d <= tol).mean() >= 0.999 and d.max() <= 100 * tol, f"{label} {name}: max {d.max()}, frac ok {(d <= tol).mean()}"
    d = np.abs(o["out"][6] - h["out"][6])
    assert (d <= 1e-4 * np.maximum(o["out"][6], 1.0)).mean() >= 0.999, f"{label} max depth"
    # the same points are left at the binding's fills (alpha exactly 1, colour 0: outside the frustum / image)
    untouched_o = (o["ai"] == 1.0) & ~o["ci"].any(axis=1)
    untouched_h = (h["ai"] == 1.0) & ~h["ci"].any(axis=1)
    assert (untouched_o == untouched_h).mean() >= 0.999, f"{label} untouched points"
