"""Writes a small fixture of the integrate path (inputs + the CPU oracle's outputs) to tests/golden/integrate_*.npz.
Like tests/tools/gen_raster_golden.py it freezes the ORACLE's results of this container (the reference's integrate is CUDA-only
and cannot run here); the oracle's integrate is cross-checked against an independent float64 numpy evaluation in
tests/test_oracle_integrate.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_scene  # noqa: E402
from helpers_integrate import make_points, oracle_integrate  # noqa: E402

CASES = {
    "I1_oblique": (dict(P=1200, res=(64, 64), s0=0.05, view="oblique", bg=(0.2, 0.5, 0.7)), 4000, 0),
    "I5_sweeps": (dict(P=800, res=(48, 48), s0=0.05, view="oblique"), 1500, 600),
}
for name, (kw, n, cluster) in CASES.items():
    sc = make_scene(**kw)
    pts = make_points(sc, n, cluster=cluster)
    o = oracle_integrate(sc, pts)
    npy = lambda t: None if t is None else t.numpy()
    data = dict(W=sc["W"], H=sc["H"], sh_degree=sc["sh_degree"], kernel_size=sc["kernel_size"],
                scale_modifier=sc["scale_modifier"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=npy(sc["bg"]),
                viewmatrix=npy(sc["viewmatrix"]), projmatrix=npy(sc["projmatrix"]), campos=npy(sc["campos"]),
                means3D=npy(sc["means3D"]), opacities=npy(sc["opacities"]), scales=npy(sc["scales"]),
                rotations=npy(sc["rotations"]), shs=npy(sc["shs"]), points3D=pts, out_color=o["out"],
                alpha_integrated=o["ai"], color_integrated=o["ci"], radii=o["radii"], num_integrated=np.int64(o["NI"]),
                num_rendered=np.int64(o["R"]))
    path = os.path.join(ROOT, "tests", "golden", f"integrate_{name}.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path))
