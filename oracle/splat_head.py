"""numpy restatement of the cycle-aggregative projection ("splat head") -- TEST INFRASTRUCTURE ONLY.

Follows reference src/gaussian_predictor.py: get_pos_from_network_output :857-881, forward :961-1002,
transform_rotations :839-855 with quaternion_raw_multiply :45-64, transform_SHs :821-837 with the constant
matrices of init_sh_transform_matrices :649-655, flatten_vector :788-791.
Pinned against the imported reference by tests/golden/splat_head.npz (tests/test_python_half_pins.py).
"""
import numpy as np

V_TO_SH = np.array([[0, 0, -1], [-1, 0, 0], [0, 1, 0]], dtype=np.float32)
SH_TO_V = V_TO_SH.T.copy()


def flatten_vector(x):
    """[B,C,H,W] -> [B,HW,C]"""
    B, C = x.shape[:2]
    return x.reshape(B, C, -1).transpose(0, 2, 1)


def splat_head(net_out, depth, ray_dirs, view_to_world, cam_quat, squre_clip=10000.0):
    f32 = np.float32
    net_out, depth, ray_dirs = net_out.astype(f32), depth.astype(f32), ray_dirs.astype(f32).reshape(1, 3, *net_out.shape[2:])
    v2w, quat = view_to_world.astype(f32).reshape(-1, 4, 4), cam_quat.astype(f32).reshape(-1, 4)
    offset, opacity, scaling, rotation, fdc, frest = np.split(net_out, np.cumsum([3, 1, 3, 4, 3]), axis=1)
    pos = ray_dirs * depth + offset
    pos = flatten_vector(pos)
    pos = np.concatenate([pos, np.ones_like(pos[:, :, :1])], 2)
    pos = np.einsum("bnk,bkj->bnj", pos, v2w).astype(f32)
    xyz = pos[:, :, :3] / (pos[:, :, 3:] + f32(1e-10))
    if squre_clip < 10.0:
        xyz[:, :, 0] = np.clip(xyz[:, :, 0], -squre_clip, squre_clip)
        xyz[:, :, 1] = np.clip(xyz[:, :, 1], -squre_clip, squre_clip)
    out = {"xyz": xyz.astype(f32)}
    out["opacity"] = flatten_vector(f32(1) / (f32(1) + np.exp(-opacity)))
    out["scaling"] = flatten_vector(np.exp(scaling))
    nrm = np.maximum(np.sqrt((rotation * rotation).sum(1, keepdims=True)), f32(1e-12))
    q = flatten_vector(rotation / nrm)
    a = quat[:, None, :]
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    out["rotation"] = np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                                aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1).astype(f32)
    out["features_dc"] = flatten_vector(fdc)[:, :, None, :]
    rest = flatten_vector(frest)
    rest = rest.reshape(rest.shape[0], rest.shape[1], 3, 3)                 # [B,N,sh,rgb]
    T = np.einsum("ij,bjk,kl->bil", SH_TO_V, v2w[:, :3, :3], V_TO_SH)        # [B,3,3]
    out["features_rest"] = np.einsum("bnsc,bst->bntc", rest, T).astype(f32)
    out["unet_depth"] = flatten_vector(depth)
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def init_ray_dirs(res, fov_deg, inverted_x=False, inverted_y=True):
    """gaussian_predictor.py:657-681"""
    x = np.linspace(-res // 2 + 0.5, res // 2 - 0.5, res, dtype=np.float32)
    y = np.linspace(res // 2 - 0.5, -res // 2 + 0.5, res, dtype=np.float32)
    if inverted_x:
        x = -x
    if inverted_y:
        y = -y
    gx, gy = np.meshgrid(x, y, indexing="xy")
    rd = np.stack([gx, gy, np.ones_like(gx)])[None].astype(np.float32)
    focal = res / (2 * np.tan(fov_deg * np.pi / 180 / 2))
    rd[:, :2] /= np.float32(focal)
    return rd
