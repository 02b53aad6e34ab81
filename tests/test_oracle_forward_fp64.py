"""Cross-examination of the (unpinnable) C oracle's forward render: an independent float64 numpy evaluation of the GOF formulas
(tests/fwd_truth.py) must agree with it on all 9 channels. The float32 storage of view2gaussian alone moves the exponent by
~6e-8 * C (C = t^2/sigma^2), so the agreement is quoted per conditioning class: large splats (C ~ 1e3) to float32 rounding,
sigma0 <= 0.08 (C ~ 1e4) at the level of the reference's own conditioning. CPU only."""
import numpy as np
import pytest

from fwd_truth import render_fp64
from helpers import frac_within, make_scene, psnr, run_oracle

# name -> (scene, abs tolerance on rgb / normal / alpha, minimum PSNR)
SCENES = {
    "canonical_s0.4": (dict(P=150, res=(64, 64), s0=0.4, view="canonical", bg=(0.2, 0.4, 0.1)), 2e-4, 90.0),
    "oblique_s0.3": (dict(P=200, res=(64, 48), s0=0.3, view="oblique"), 2e-4, 90.0),
    "colors_precomp_s0.25": (dict(P=200, res=(48, 64), s0=0.25, view="oblique", colors_precomp=True), 5e-4, 90.0),
    "filter_scalemod_sh0_s0.3": (dict(P=200, res=(48, 48), s0=0.3, view="oblique", kernel_size=0.1, scale_modifier=0.7, sh_degree=0), 2e-3, 85.0),
    "canonical_s0.08": (dict(P=500, res=(64, 64), s0=0.08, view="canonical"), 5e-3, 75.0),
    "oblique_s0.05": (dict(P=800, res=(64, 48), s0=0.05, view="oblique"), 1e-2, 62.0),
    "deep_s0.05": (dict(P=600, res=(64, 64), s0=0.05, view="oblique", depth_range=(1.5, 20.0)), 1e-2, 62.0),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_oracle_forward_matches_independent_fp64(name):
    kw, atol, min_psnr = SCENES[name]
    scene = make_scene(**kw)
    o = run_oracle(scene)["out_color"].astype(np.float64)
    t = render_fp64(scene)
    assert o[7].max() > 0.5                                        # the scene is not empty
    for ch, label in ((slice(0, 3), "rgb"), (slice(3, 6), "normal"), (slice(7, 8), "alpha")):
        assert frac_within(o[ch], t[ch], atol) >= 0.995, (label, frac_within(o[ch], t[ch], atol))
        assert psnr(o[ch], t[ch]) >= min_psnr, (label, psnr(o[ch], t[ch]))
    assert frac_within(o[6], t[6], 0.0, max(atol, 1e-5)) >= 0.99, "median depth"
    assert frac_within(o[8], t[8], 2e-6, 5e-2) >= 0.99, "distortion"


def test_oracle_distortion_on_a_depth_spread_scene():
    """Where the distortion channel is well conditioned (depths 1 .. 30: NDC depths spread over ~0.2, values up to 6e-4) the
    oracle's float32 accumulations agree with the float64 evaluation to 1e-3 relative -- the tolerance SURVEY 8d asks for."""
    scene = make_scene(P=800, res=(64, 64), s0=0.3, view="canonical", depth_range=(1.0, 30.0))
    o = run_oracle(scene)["out_color"].astype(np.float64)
    t = render_fp64(scene)
    big = t[8] > 1e-4
    assert big.sum() > 1000 and t[8].max() > 5e-4, (big.sum(), t[8].max())
    assert frac_within(o[8][big], t[8][big], 0.0, 1e-3) >= 0.99, frac_within(o[8][big], t[8][big], 0.0, 1e-3)
