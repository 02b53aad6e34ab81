"""Generates tests/golden/ply_16.npz from the reference's OWN PLY code, in the build container (the reference does not travel):
  * visualize.py:146-179 (``load_ply``) is read from where it lies and executed as is (path=None branch) on a 16-Gaussian batch;
  * src/gaussian-splatting/scene/gaussian_model.py:177-208 (``construct_list_of_attributes`` + ``save_ply``) is read from where it
    lies and executed as is on the same Gaussians (with the 15 x 3 zero ``f_rest`` the reference exports), with stand-ins for the
    absent ``plyfile`` classes that only CAPTURE the structured vertex array the reference hands to PlyElement.describe.
The fixture holds the input dict, the six arrays ``load_ply`` returns, the property names and the packed rows of that array."""
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
npy = lambda t: t.detach().cpu().numpy()


def main():
    g = torch.Generator().manual_seed(7)
    B, N = 2, 16
    gs = {"xyz": torch.randn(B, N, 3, generator=g), "opacity": torch.rand(B, N, 1, generator=g), "scaling": torch.rand(B, N, 3, generator=g) * 0.1,
          "rotation": torch.nn.functional.normalize(torch.randn(B, N, 4, generator=g), dim=-1), "features_dc": torch.randn(B, N, 1, 3, generator=g),
          "features_rest": torch.randn(B, N, 3, 3, generator=g) * 0.1}
    bb = 1
    lines = open(os.path.join(REF, "visualize.py")).read().splitlines()
    assert lines[145].startswith("def load_ply(") and lines[178].strip().startswith("return xyz, f_dc, f_rest")
    ns = {"torch": torch, "os": os}
    exec(compile("\n".join(lines[145:179]), "/root/reference/visualize.py:146-179", "exec"), ns)
    xyz, f_dc, f_rest, opac, scale, rot = ns["load_ply"](gs, bb, None)

    captured = {}

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            captured["elements"], captured["name"] = elements, name
            return "el"

    class PlyData:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = path

    gm = open(os.path.join(REF, "src/gaussian-splatting/scene/gaussian_model.py")).read().splitlines()
    assert gm[176].strip().startswith("def construct_list_of_attributes(self)") and gm[207].strip() == "PlyData([el]).write(path)"
    ns2 = {"np": np, "torch": torch, "os": os, "PlyData": PlyData, "PlyElement": PlyElement, "mkdir_p": lambda p: None}
    exec(compile(textwrap.dedent("\n".join(gm[176:208])), "gaussian_model.py:177-208", "exec"), ns2)
    me = types.SimpleNamespace(_xyz=gs["xyz"][bb], _features_dc=gs["features_dc"][bb],
                               _features_rest=torch.zeros_like(gs["features_dc"][bb]).expand(-1, 15, -1),
                               _opacity=gs["opacity"][bb], _scaling=gs["scaling"][bb], _rotation=gs["rotation"][bb])
    me.construct_list_of_attributes = lambda: ns2["construct_list_of_attributes"](me)
    ns2["save_ply"](me, "unused/point_cloud.ply")
    el = captured["elements"]
    assert captured["name"] == "vertex" and all(el.dtype[n] == np.dtype("f4") for n in el.dtype.names)
    out = os.path.join(ROOT, "tests", "golden", "ply_16.npz")
    np.savez_compressed(out, bb=np.int64(bb), names=np.array(el.dtype.names), rows=np.frombuffer(el.tobytes(), dtype="<f4").reshape(N, -1),
                        lp_xyz=npy(xyz), lp_f_dc=npy(f_dc), lp_f_rest=npy(f_rest), lp_opacities=npy(opac), lp_scale=npy(scale),
                        lp_rotation=npy(rot), **{"in_" + k: npy(v) for k, v in gs.items()})
    print(out, os.path.getsize(out), len(el.dtype.names), "properties")


if __name__ == "__main__":
    main()
