"""Independent float64 evaluation of the GOF forward render, written from the formulas, not from the
oracle's code: plain linear algebra in numpy float64 (Sigma' = Rgv diag(S^-2) Rgv^T, B = -Sigma' tg, C = tg^T Sigma' tg-form),
one pass over the depth-sorted Gaussians with [H, W] state arrays, no tiles except the reference's tile-rectangle membership.
It cross-examines the plain-C oracle (which cannot be pinned against the CUDA reference in this image): on well-conditioned
scenes (sigma0 >= 0.05) the two must agree on all 9 channels up to float32 rounding. Test infrastructure only."""
import math

import numpy as np

SH_C0, SH_C1 = 0.28209479177387814, 0.4886025119029199


def _rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]              # NOT normalised (forward.cu:138-149)
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def render_fp64(scene, view=0):
    f = lambda t: None if t is None else np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64)
    W, H = scene["W"], scene["H"]
    p, op, sc, q = f(scene["means3D"]), f(scene["opacities"]).reshape(-1), f(scene["scales"]), f(scene["rotations"])
    vm, pm, cam = f(scene["viewmatrix"][view]), f(scene["projmatrix"][view]), f(scene["campos"][view])
    bg = f(scene["bg"])
    mod, ks = float(scene["scale_modifier"]), float(scene["kernel_size"])
    # the camera of the rasterizer: focal in float32 (rasterizer_impl.cu:274-275), tan_fov as a C float
    tfx, tfy = np.float32(scene["tanfovx"]), np.float32(scene["tanfovy"])
    fx, fy = float(np.float32(W) / (np.float32(2.0) * tfx)), float(np.float32(H) / (np.float32(2.0) * tfy))
    P = len(p)
    ph = np.concatenate([p, np.ones((P, 1))], 1)
    pv = ph @ vm                                        # row-vector convention
    hom = ph @ pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    vis = pv[:, 2] > 0.2
    R = _rot(q)
    Sig = R @ (np.eye(3)[None] * ((mod * sc) ** 2)[:, None, :]) @ R.transpose(0, 2, 1)
    Wv = vm[:3, :3].T                                   # x_view = Wv x_world + t
    tz = pv[:, 2]
    tx = np.clip(pv[:, 0] / tz, -1.3 * float(tfx), 1.3 * float(tfx)) * tz
    ty = np.clip(pv[:, 1] / tz, -1.3 * float(tfy), 1.3 * float(tfy)) * tz
    J = np.zeros((P, 2, 3))
    J[:, 0, 0] = fx / tz; J[:, 0, 2] = -fx * tx / (tz * tz); J[:, 1, 1] = fy / tz; J[:, 1, 2] = -fy * ty / (tz * tz)
    cov = J @ Wv[None] @ Sig @ Wv.T[None] @ J.transpose(0, 2, 1)
    a, b, c = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = np.maximum(1e-6, a * c - b * b)
    det1 = np.maximum(1e-6, (a + ks) * (c + ks) - b * b)
    coef = np.sqrt(det0 / (det1 + 1e-6) + 1e-6)
    coef[(det0 <= 1e-6) | (det1 <= 1e-6)] = 0.0
    a, c = a + ks, c + ks
    det = a * c - b * b
    mid = 0.5 * (a + c)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    radius = np.ceil(3.0 * np.sqrt(lam))
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tr = lambda v: np.trunc(v).astype(np.int64)
    rx0, rx1 = np.clip(tr((px - radius) / 16), 0, gx), np.clip(tr((px + radius + 15) / 16), 0, gx)
    ry0, ry1 = np.clip(tr((py - radius) / 16), 0, gy), np.clip(tr((py + radius + 15) / 16), 0, gy)
    vis &= (det != 0) & ((rx1 - rx0) * (ry1 - ry0) > 0)
    # colour
    if scene.get("colors_precomp") is not None:
        rgb = f(scene["colors_precomp"])
    else:
        sh = f(scene["shs"])
        d = p - cam[None]
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        rgb = SH_C0 * sh[:, 0]
        if scene["sh_degree"] > 0:
            rgb = rgb - SH_C1 * d[:, 1:2] * sh[:, 1] + SH_C1 * d[:, 2:3] * sh[:, 2] - SH_C1 * d[:, 0:1] * sh[:, 3]
        rgb = np.maximum(rgb + 0.5, 0.0)
    # view2gaussian
    Rgv = Wv[None] @ R
    tg = pv[:, :3]
    Sinv = 1.0 / (sc * sc + 1e-7)                       # raw scales: the reference ignores scale_modifier here (forward.cu:255)
    Sigp = Rgv @ (np.eye(3)[None] * Sinv[:, None, :]) @ Rgv.transpose(0, 2, 1)
    Bv = -np.einsum("pij,pj->pi", Sigp, tg)
    Cc = np.einsum("pi,pij,pj->p", tg, Sigp, tg)
    opc = op * coef

    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    rxp = (xs + 0.5 - W / 2.0) / fx
    ryp = (ys + 0.5 - H / 2.0) / fy
    tile_x, tile_y = xs // 16, ys // 16
    T = np.ones((H, W)); Cacc = np.zeros((8, H, W)); d1 = np.zeros((H, W)); d2 = np.zeros((H, W)); dist = np.zeros((H, W))
    done = np.zeros((H, W), bool)
    order = np.argsort(pv[:, 2].astype(np.float32), kind="stable")       # keys are the float32 depth bits, ties by index
    for g in order:
        if not vis[g]:
            continue
        m = (~done) & (tile_x >= rx0[g]) & (tile_x < rx1[g]) & (tile_y >= ry0[g]) & (tile_y < ry1[g])
        if not m.any():
            continue
        S = Sigp[g]
        n0 = S[0, 0] * rxp + S[0, 1] * ryp + S[0, 2]
        n1 = S[1, 0] * rxp + S[1, 1] * ryp + S[1, 2]
        n2 = S[2, 0] * rxp + S[2, 1] * ryp + S[2, 2]
        AA = rxp * n0 + ryp * n1 + n2
        BB = 2.0 * (Bv[g, 0] * rxp + Bv[g, 1] * ryp + Bv[g, 2])
        t = -BB / (2.0 * AA)
        m &= t > 0.2
        power = np.minimum(-0.5 * (Cc[g] - BB * BB / (4.0 * AA)), 0.0)
        alpha = np.minimum(0.99, opc[g] * np.exp(power))
        m &= alpha >= 1.0 / 255.0
        testT = T * (1.0 - alpha)
        stop = m & (testT < 1e-4)
        done |= stop
        m &= ~stop
        if not m.any():
            continue
        mp = (100.0 * t - 20.0) / (99.8 * t)
        ln = np.sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7)
        w = np.where(m, alpha * T, 0.0)
        A = 1.0 - T
        err = mp * mp * A + d2 - 2.0 * mp * d1
        dist += err * w; d1 += mp * w; d2 += mp * mp * w
        for ch in range(3):
            Cacc[ch] += rgb[g, ch] * w
        Cacc[3] += -n0 / ln * w; Cacc[4] += -n1 / ln * w; Cacc[5] += -n2 / ln * w
        Cacc[6] = np.where(m & (T > 0.5), t, Cacc[6])
        Cacc[7] += w
        T = np.where(m, testT, T)
    out = np.empty((9, H, W))
    out[:3] = Cacc[:3] + T[None] * bg[:, None, None]
    out[3:8] = Cacc[3:8]
    out[8] = dist / ((1.0 - T) * (1.0 - T) + 1e-7)
    return out
