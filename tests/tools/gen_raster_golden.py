"""Writes small rasterizer fixtures (inputs + the CPU oracle's outputs and key intermediates) to tests/golden/raster_*.npz.
They freeze the oracle's results of THIS container (glibc expf etc.), so the GPU tests can also compare against committed
vectors and the CPU tests can detect an oracle that behaves differently on another host. The oracle itself is pinned as
described in DESIGN.md section 4 (it is not a run of the reference rasterizer, which cannot be built here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_scene, run_oracle  # noqa: E402

CASES = {
    "F1_tiny_identity": dict(P=1200, res=(64, 64), s0=0.05, view="canonical"),
    "F2_oblique_aniso": dict(P=1500, res=(64, 64), s0=0.03, view="oblique", aniso=True, behind_fraction=0.05),
    "F3_colors_precomp": dict(P=1000, res=(48, 48), s0=0.05, view="oblique", colors_precomp=True, bg=(0.2, 0.5, 0.7)),
    "F4_filter_scalemod": dict(P=1000, res=(48, 48), s0=0.05, view="oblique", kernel_size=0.1, scale_modifier=0.5),
    "F5_odd_size": dict(P=1500, res=(100, 72), s0=0.04, view="oblique"),
}
out_dir = os.path.join(ROOT, "tests", "golden")
for name, kw in CASES.items():
    sc = make_scene(**kw)
    o = run_oracle(sc)
    dpix = np.random.default_rng(7).standard_normal((9, sc["H"], sc["W"])).astype(np.float32)
    g = o["oracle"].backward(dpix)
    npy = lambda t: None if t is None else t.numpy()
    data = dict(
        W=sc["W"], H=sc["H"], sh_degree=sc["sh_degree"], kernel_size=sc["kernel_size"], scale_modifier=sc["scale_modifier"],
        tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=npy(sc["bg"]), viewmatrix=npy(sc["viewmatrix"]),
        projmatrix=npy(sc["projmatrix"]), campos=npy(sc["campos"]), means3D=npy(sc["means3D"]), opacities=npy(sc["opacities"]),
        scales=npy(sc["scales"]), rotations=npy(sc["rotations"]),
        out_color=o["out_color"], radii=o["radii"], num_rendered=np.int64(o["num_rendered"]), point_list=o["point_list"],
        ranges=o["ranges"], view2gaussian=o["view2gaussian"], depths=o["depths"], means2D=o["means2D"],
        conic_opacity=o["conic_opacity"], n_contrib=o["n_contrib"], final_T=o["final_T"], dL_dpix=dpix,
        **{"g_" + k: v for k, v in g.items() if k not in ("dL_dconic", "dL_dcov3D")})
    if sc["shs"] is not None:
        data["shs"] = npy(sc["shs"]); data["rgb"] = o["rgb"]
    else:
        data["colors_precomp"] = npy(sc["colors_precomp"])
    path = os.path.join(out_dir, f"raster_{name}.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path))
