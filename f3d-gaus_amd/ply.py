"""Gaussian PLY export of the inference path: the on-disk format on the far side of the hot path.

Mirrors reference visualize.py:146-179 (``load_ply``: the per-image re-layout of the Gaussian dict, with 45 zero ``f_rest``
columns -- F3D-Gaus exports SH degree 0 only -- and the ACTIVATED opacity / scale values the predictor returns) and the file
layout of the vendored 3DGS writer src/gaussian-splatting/scene/gaussian_model.py:177-208 (``construct_list_of_attributes`` +
``save_ply``): binary little-endian PLY, one ``vertex`` element, float32 properties
    x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3
with the SH planes channel-major (``features.transpose(1, 2).flatten(1)``). The reference's own write branch is broken (it
references an undefined element, visualize.py:176) and every caller passes ``path=None``; here both branches work. ``plyfile`` is
not needed: the header is written by hand, the body is the packed float32 rows. Host-side file IO, no device code."""
import os

import numpy as np
import torch


def construct_list_of_attributes(n_dc, n_rest, n_scale=3, n_rot=4):
    names = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    names += ['f_dc_{}'.format(i) for i in range(n_dc)]
    names += ['f_rest_{}'.format(i) for i in range(n_rest)]
    names.append('opacity')
    names += ['scale_{}'.format(i) for i in range(n_scale)]
    names += ['rot_{}'.format(i) for i in range(n_rot)]
    return names


def write_ply(path, columns, names):
    """columns: list of [N, k] float arrays in file order; names: one property name per column of their concatenation."""
    rows = np.concatenate([np.asarray(c, dtype=np.float32).reshape(len(columns[0]), -1) for c in columns], axis=1)
    assert rows.shape[1] == len(names), (rows.shape, len(names))
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % rows.shape[0]
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(rows.astype("<f4")).tobytes())


def read_ply(path):
    """(names, rows [N, len(names)] float32) of a PLY written by ``write_ply`` (binary little-endian, float properties only)."""
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode("ascii").splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0", lines[:2]
    n = int([ln for ln in lines if ln.startswith("element vertex")][0].split()[-1])
    names = [ln.split()[2] for ln in lines if ln.startswith("property float ")]
    rows = np.frombuffer(raw[end:], dtype="<f4").reshape(n, len(names))
    return names, rows


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """The vendored writer's layout (gaussian_model.py:190-208): features_dc [N,1,3], features_rest [N,K,3] (any K), everything
    else [N,k]; tensors or arrays, on any device."""
    t = lambda a: a.detach().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
    xyz, features_dc, features_rest, opacity, scaling, rotation = map(t, (xyz, features_dc, features_rest, opacity, scaling, rotation))
    f_dc = features_dc.transpose(1, 2).flatten(start_dim=1).contiguous()
    f_rest = features_rest.transpose(1, 2).flatten(start_dim=1).contiguous()
    names = construct_list_of_attributes(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    write_ply(path, [xyz.numpy(), np.zeros_like(xyz.numpy()), f_dc.numpy(), f_rest.numpy(), opacity.reshape(len(xyz), -1).numpy(),
                     scaling.numpy(), rotation.numpy()], names)


def load_ply(gs_dic, bb, path):
    """visualize.py:146-179. ``path is None``: returns (xyz [N,3], f_dc [N,3], f_rest [N,45] zeros, opacities [N,1], scale [N,3],
    rotation [N,4]) of image ``bb`` (tensors on the dict's device); otherwise writes them (plus zero normals) as a PLY."""
    xyz = gs_dic['xyz'][bb].detach()
    f_dc = gs_dic['features_dc'][bb].detach().transpose(1, 2).flatten(start_dim=1).contiguous()
    f_rest = torch.zeros_like(gs_dic['features_dc'][bb])
    f_rest = f_rest.expand([-1, (3 + 1) ** 2 - 1, -1]).detach().transpose(1, 2).flatten(start_dim=1).contiguous()
    opacities = gs_dic['opacity'][bb].detach()
    scale = gs_dic['scaling'][bb].detach()
    rotation = gs_dic['rotation'][bb].detach()
    if path is None:
        return xyz, f_dc, f_rest, opacities, scale, rotation
    names = construct_list_of_attributes(f_dc.shape[1], f_rest.shape[1], scale.shape[1], rotation.shape[1])
    c = lambda a: a.cpu().numpy()
    write_ply(path, [c(xyz), np.zeros_like(c(xyz)), c(f_dc), c(f_rest), c(opacities), c(scale), c(rotation)], names)
