"""PLY export against tests/golden/ply_16.npz, which tests/tools/gen_ply_golden.py produced by executing the
reference's own ``load_ply`` (visualize.py:146-179) and the vendored 3DGS ``save_ply`` (gaussian_model.py:177-208) on a
16-Gaussian set. CPU only: this is host-side file IO."""
import os

import numpy as np
import torch

from f3dgaus_amd import ply

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ply_16.npz"))


def _gs():
    return {k[3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("in_")}


def test_load_ply_path_none_matches_reference():
    out = ply.load_ply(_gs(), int(GOLD["bb"]), None)
    for got, key in zip(out, ("lp_xyz", "lp_f_dc", "lp_f_rest", "lp_opacities", "lp_scale", "lp_rotation")):
        assert tuple(got.shape) == GOLD[key].shape and np.array_equal(got.numpy(), GOLD[key]), key


def test_written_file_is_the_reference_vertex_array(tmp_path):
    path = str(tmp_path / "sub" / "point_cloud.ply")
    ply.load_ply(_gs(), int(GOLD["bb"]), path)
    raw = open(path, "rb").read()
    names = [str(n) for n in GOLD["names"]]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 16\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n").encode()
    assert raw.startswith(header)
    assert raw[len(header):] == GOLD["rows"].astype("<f4").tobytes()            # the packed rows the reference hands to plyfile
    rnames, rows = ply.read_ply(path)
    assert rnames == names and len(names) == 62 and np.array_equal(rows, GOLD["rows"])
    # the general writer (any number of f_rest coefficients) agrees on the same data
    gs, bb = _gs(), int(GOLD["bb"])
    path2 = str(tmp_path / "b.ply")
    ply.save_ply(path2, gs["xyz"][bb], gs["features_dc"][bb], torch.zeros(16, 15, 3), gs["opacity"][bb], gs["scaling"][bb], gs["rotation"][bb])
    assert open(path2, "rb").read() == raw
    path3 = str(tmp_path / "c.ply")
    ply.save_ply(path3, gs["xyz"][bb], gs["features_dc"][bb], gs["features_rest"][bb], gs["opacity"][bb], gs["scaling"][bb], gs["rotation"][bb])
    n3, r3 = ply.read_ply(path3)
    assert len(n3) == 6 + 3 + 9 + 1 + 3 + 4 and np.array_equal(r3[:, 9:18], gs["features_rest"][bb].transpose(1, 2).flatten(1).numpy())
