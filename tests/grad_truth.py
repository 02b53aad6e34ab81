"""float64 autograd ground truth for the per-Gaussian stage of the backward pass (test infrastructure).

The reference's computeView2Gaussian_backward (backward.cu:381-587) and SH backward (:20-139) are the plain chain
rule of  view2gaussian(mean, scale, rot; view)  and  colour(mean, sh; campos).  In float32 they are dominated by
cancellation noise (the reference's own run-to-run spread on dL/dscale is 0.35-0.45 of the maximum), so
parity of dmean3D / drot / dscale is asserted as "error against this fp64 truth <= the oracle's own error"."""
import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def view2gaussian64(means, scales, rots, view):
    """[P,10] in float64; view = viewmatrix tensor [4,4] (row-vector convention)."""
    r, x, y, z = rots.unbind(-1)
    R = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)     # gaussian -> world
    Wr = view[:3, :3].T
    Rgv = Wr @ R                                              # gaussian -> view rotation
    tg = means @ view[:3, :3] + view[3, :3]                   # view-space mean
    t2 = -(Rgv.transpose(1, 2) @ tg.unsqueeze(-1)).squeeze(-1)
    S = 1.0 / (scales * scales + 1e-7)
    C = (t2 * t2 * S).sum(-1)
    Sigma = Rgv @ torch.diag_embed(S) @ Rgv.transpose(1, 2)
    B = (Rgv @ (S * t2).unsqueeze(-1)).squeeze(-1)
    return torch.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2],
                        B[:, 0], B[:, 1], B[:, 2], C], -1)


def colour64(means, shs, campos, deg):
    d = means - campos
    d = d / d.norm(dim=-1, keepdim=True)
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:       # real spherical harmonics of degree 2 and 3 in the 3DGS basis (forward.cu:40-66)
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6] \
            + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8]
        if deg > 2:
            res = res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10] \
                + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] \
                + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14] \
                + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15]
    return torch.clamp_min(res + 0.5, 0.0)


def per_gaussian_truth(scene, view_idx, radii, dL_dv2g, dL_dcolor):
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    means = t64(scene["means3D"]).requires_grad_(True)
    scales = t64(scene["scales"]).requires_grad_(True)
    rots = t64(scene["rotations"]).requires_grad_(True)
    view = t64(scene["viewmatrix"][view_idx])
    vis = torch.tensor(np.asarray(radii) > 0)
    L = (view2gaussian64(means, scales, rots, view) * t64(dL_dv2g))[vis].sum()
    shs = None
    if scene["shs"] is not None:
        shs = t64(scene["shs"]).requires_grad_(True)
        col = colour64(means, shs, t64(scene["campos"][view_idx]), scene["sh_degree"])
        L = L + (col * t64(dL_dcolor))[vis].sum()
    L.backward()
    return dict(dL_dmean3D=means.grad.numpy(), dL_dscale=scales.grad.numpy(), dL_drot=rots.grad.numpy(),
                dL_dsh=None if shs is None else shs.grad.numpy())


def compositing_truth(scene, o, weights, view_idx=0):
    """float64 autograd of the compositing stage for a loss  sum(weights[0:6] * out[0:6])  (RGB + normal channels:
    the channels whose gradient the reference differentiates completely). The per-tile lists, the per-pixel number of
    blended entries and the skip decisions are taken as constants, exactly as the analytic backward assumes.
    Returns gradients w.r.t. the per-Gaussian compositing inputs: opacity*coef [P], colour [P,3], view2gaussian [P,10]."""
    W, H = scene["W"], scene["H"]
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(scene["tanfovx"]))
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(scene["tanfovy"]))
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    v2g = t64(o["view2gaussian"]).requires_grad_(True)
    opac = t64(o["conic_opacity"][:, 3]).requires_grad_(True)
    col = t64(o["rgb"] if scene["colors_precomp"] is None else scene["colors_precomp"]).requires_grad_(True)
    wts = t64(weights)
    last = torch.tensor(o["n_contrib"][0].astype(np.int64))
    gx = (W + 15) // 16
    loss = torch.zeros((), dtype=torch.float64)
    for tile in range(o["ranges"].shape[0]):
        r0, r1 = int(o["ranges"][tile, 0]), int(o["ranges"][tile, 1])
        if r1 <= r0:
            continue
        ids = torch.tensor(o["point_list"][r0:r1].astype(np.int64))
        ty, tx = divmod(tile, gx)
        ys, xs = torch.meshgrid(torch.arange(ty * 16, min(ty * 16 + 16, H)), torch.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        # the ray is rounded to float32 in the reference (forward.cu:448); keep that rounding, the exponent amplifies it
        rx = ((xs.double() + 0.5 - W / 2.) / float(focal_x)).float().double().unsqueeze(1)
        ry = ((ys.double() + 0.5 - H / 2.) / float(focal_y)).float().double().unsqueeze(1)
        v = v2g[ids].unsqueeze(0)                                   # [1,n,10]
        n0 = v[..., 0] * rx + v[..., 1] * ry + v[..., 2]
        n1 = v[..., 1] * rx + v[..., 3] * ry + v[..., 4]
        n2 = v[..., 2] * rx + v[..., 4] * ry + v[..., 5]
        AA = rx * n0 + ry * n1 + n2
        BB = 2 * (v[..., 6] * rx + v[..., 7] * ry + v[..., 8])
        t = -BB / (2 * AA)
        power = torch.clamp_max(-0.5 * (-(BB / AA) * (BB / 4.) + v[..., 9]), 0.0)
        alpha = torch.clamp_max(opac[ids].unsqueeze(0) * torch.exp(power), 0.99)
        with torch.no_grad():
            mask = (t > 0.2) & (alpha >= 1.0 / 255.0) & (torch.arange(len(ids)).unsqueeze(0) < last[ys, xs].unsqueeze(1))
        a = alpha * mask
        T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a[:, :-1]], 1), 1)
        wgt = a * T
        length = torch.sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7)
        chans = [wgt @ col[ids][:, 0], wgt @ col[ids][:, 1], wgt @ col[ids][:, 2],
                 (wgt * (-n0 / length)).sum(1), (wgt * (-n1 / length)).sum(1), (wgt * (-n2 / length)).sum(1)]
        for ch in range(6):
            loss = loss + (chans[ch] * wts[ch, ys, xs]).sum()
    loss.backward()
    return dict(dL_dopacity=opac.grad.numpy(), dL_dcolor=col.grad.numpy(), dL_dview2gaussian=v2g.grad.numpy())
