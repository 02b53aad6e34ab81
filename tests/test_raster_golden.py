"""Committed rasterizer fixtures (tests/golden/raster_*.npz, written by tests/tools/gen_raster_golden.py from the CPU oracle):
  * CPU: the oracle on this host reproduces them (guards against host libm / compiler differences);
  * GPU: the HIP path reproduces them (forward bit-exact intermediates, 1e-4 renders, 1e-5 compositing gradients)."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import assert_render_parity, run_hip, run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "raster_*.npz")))


def _scene(g):
    t = lambda k: torch.from_numpy(g[k]) if k in g.files else None
    return dict(P=g["means3D"].shape[0], W=int(g["W"]), H=int(g["H"]), sh_degree=int(g["sh_degree"]),
                kernel_size=float(g["kernel_size"]), scale_modifier=float(g["scale_modifier"]), tanfovx=float(g["tanfovx"]),
                tanfovy=float(g["tanfovy"]), bg=t("bg"), viewmatrix=t("viewmatrix"), projmatrix=t("projmatrix"),
                campos=t("campos"), means3D=t("means3D"), opacities=t("opacities"), scales=t("scales"),
                rotations=t("rotations"), shs=t("shs"), colors_precomp=t("colors_precomp"))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[7:-4] for f in FILES])
def test_oracle_reproduces_fixture(path):
    g = np.load(path)
    o = run_oracle(_scene(g))
    assert o["num_rendered"] == int(g["num_rendered"])
    assert np.array_equal(o["radii"], g["radii"]) and np.array_equal(o["point_list"], g["point_list"])
    assert np.array_equal(o["ranges"], g["ranges"])
    for k in ("view2gaussian", "depths", "means2D", "conic_opacity"):
        assert np.array_equal(o[k].view(np.uint32), g[k].view(np.uint32)), k
    assert_render_parity(o["out_color"], g["out_color"], "oracle-vs-fixture")
    go = o["oracle"].backward(g["dL_dpix"])
    for k in ("dL_dview2gaussian", "dL_dopacity", "dL_dcolor", "dL_dmean2D"):
        assert np.abs(go[k] - g["g_" + k]).max() <= 1e-5 * np.abs(g["g_" + k]).max(), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[7:-4] for f in FILES])
def test_hip_reproduces_fixture(path, gpu_device):
    from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
    g = np.load(path)
    sc = _scene(g)
    h = run_hip(sc, gpu_device)
    assert h["num_rendered"] == int(g["num_rendered"])
    assert np.array_equal(h["radii"][0], g["radii"]) and np.array_equal(h["point_list"], g["point_list"])
    assert np.array_equal(h["ranges"][0], g["ranges"])
    vis = g["radii"] > 0
    for hk, gk in ((h["view2gaussian"][0], g["view2gaussian"]), (h["depths"][0], g["depths"]), (h["means2D"][0], g["means2D"]),
                   (h["conic_opacity"][0], g["conic_opacity"])):
        assert np.array_equal(hk[vis].view(np.uint32), gk[vis].view(np.uint32))
    assert_render_parity(h["out_color"][0], g["out_color"], "hip-vs-fixture")
    assert (h["n_contrib"][0] == g["n_contrib"]).mean() >= 0.999
    dev = lambda t: None if t is None else t.to(gpu_device)
    gr = rasterize_backward_raw(h["workspace"], dev(sc["means3D"]), dev(sc["shs"]), dev(sc["colors_precomp"]), dev(sc["scales"]),
                                dev(sc["rotations"]), torch.from_numpy(h["radii"]).to(gpu_device),
                                torch.from_numpy(g["dL_dpix"][None]).to(gpu_device), sc["sh_degree"], dev(sc["viewmatrix"]),
                                dev(sc["projmatrix"]), dev(sc["campos"]), dev(sc["bg"]), sc["tanfovx"], sc["tanfovy"],
                                sc["kernel_size"], sc["scale_modifier"])
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(gr["dL_dview2gaussian"][0].cpu().numpy(), g["g_dL_dview2gaussian"]) <= 1e-5
    assert rel(gr["dL_dopacity"].cpu().numpy(), g["g_dL_dopacity"]) <= 1e-5
    assert rel(gr["dL_dcolors"][0].cpu().numpy(), g["g_dL_dcolor"]) <= 1e-5
    assert rel(gr["dL_dmeans2D"][0].cpu().numpy(), g["g_dL_dmean2D"]) <= 2e-5
    if "g_dL_dsh" in g.files and g["g_dL_dsh"].size:
        assert rel(gr["dL_dsh"].cpu().numpy(), g["g_dL_dsh"]) <= 1e-5
