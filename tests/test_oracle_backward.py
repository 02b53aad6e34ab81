"""CPU checks of the oracle's backward: finite differences of the oracle's own forward for the compositing-stage
gradients, and the fp64 chain-rule truth for the per-Gaussian stage."""
import numpy as np

from grad_truth import compositing_truth, per_gaussian_truth
from helpers import make_scene, run_oracle


def _loss_weights(H, W, seed=0):
    rng = np.random.default_rng(seed)
    w = np.zeros((9, H, W), np.float32)
    w[0:6] = rng.standard_normal((6, H, W)).astype(np.float32)      # RGB + normal channels (smooth, fully differentiated)
    return w


def test_compositing_gradients_match_fp64_autograd():
    """The oracle's analytic compositing backward (restated from backward.cu:634-955) against float64 autograd of the
    compositing recurrence itself, for both colour paths."""
    # (sigma0, tolerances): the float32 forward evaluates the exponent -(C - B^2/4A)/2 with an absolute error of about
    # ulp(C) ~ ulp(t^2/sigma^2); large splats (sigma 0.3 -> C ~ 5e2) make that negligible so the MATH is checked tightly,
    # small splats (sigma 0.06 -> C ~ 1.4e4) show the float32 conditioning the reference itself has
    for s0, tol_op, tol_col, tol_v in ((0.3, 2e-4, 2e-5, 2e-4), (0.06, 2e-2, 2e-2, 0.15)):
        for kw in (dict(colors_precomp=True), dict(colors_precomp=False)):
            scene = make_scene(P=300, res=(48, 48), s0=s0, view="oblique", **kw)
            o = run_oracle(scene)
            w = _loss_weights(48, 48)
            g = o["oracle"].backward(w)
            truth = compositing_truth(scene, o, w)
            rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
            assert rel(g["dL_dopacity"][:, 0], truth["dL_dopacity"]) < tol_op, (s0, rel(g["dL_dopacity"][:, 0], truth["dL_dopacity"]))
            assert rel(g["dL_dcolor"], truth["dL_dcolor"]) < tol_col, (s0, rel(g["dL_dcolor"], truth["dL_dcolor"]))
            assert rel(g["dL_dview2gaussian"], truth["dL_dview2gaussian"]) < tol_v, (s0, rel(g["dL_dview2gaussian"], truth["dL_dview2gaussian"]))


def test_known_answers_and_per_gaussian_stage_vs_fp64_truth():
    scene = make_scene(P=1500, res=(64, 64), s0=0.05, view="oblique", behind_fraction=0.05)
    o = run_oracle(scene)
    rng = np.random.default_rng(1)
    dpix = rng.standard_normal((9, 64, 64)).astype(np.float32)
    g = o["oracle"].backward(dpix)
    assert not g["dL_dconic"].any() and not g["dL_dcov3D"].any()
    culled = o["radii"] == 0
    assert culled.any()
    for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dsh", "dL_dopacity", "dL_dcolor", "dL_dview2gaussian"):
        assert not g[k][culled].any(), k
    # the alpha channel has no gradient path
    dpix2 = dpix.copy()
    dpix2[7] += 5.0
    g2 = o["oracle"].backward(dpix2)
    for k in g:
        assert np.array_equal(g[k], g2[k]), k
    truth = per_gaussian_truth(scene, 0, o["radii"], g["dL_dview2gaussian"], g["dL_dcolor"])
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(g["dL_dsh"], truth["dL_dsh"]) < 1e-5
    assert rel(g["dL_dmean3D"], truth["dL_dmean3D"]) < 5e-3          # float32 cancellation (6e-5 .. 7e-4)
    assert rel(g["dL_drot"], truth["dL_drot"]) < 5e-2                # (3e-3 .. 8e-3)
    assert rel(g["dL_dscale"], truth["dL_dscale"]) < 0.5             # (0.15 .. 0.20)
