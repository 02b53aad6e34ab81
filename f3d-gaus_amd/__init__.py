"""f3dgaus_amd -- MI355X-native GOF rasterization + cycle-aggregative projection for F3D-Gaus.

The directory is named ``f3d-gaus_amd`` (not an identifier); import it as ``f3dgaus_amd`` through the shim module
at the repo root, or with ``importlib.import_module("f3d-gaus_amd")``. Importing the package also registers the
drop-in ``diff_gof_rasterization`` module name so the reference's own import line resolves to this build.
"""
import sys as _sys

from . import _lib, build, cameras, synthetic  # noqa: F401
from . import diff_gof_rasterization  # noqa: F401

# `from diff_gof_rasterization import GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF`
# (reference src/gaussian_renderer/__init__.py:10) resolves to this build unless another one is already loaded.
_sys.modules.setdefault("diff_gof_rasterization", diff_gof_rasterization)

from .diff_gof_rasterization import (GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF,  # noqa: E402,F401
                                     rasterize_views, set_deferred_status, deferred_status, flush)

from . import gaussian_renderer, gaussian_predictor, unet_gs, cycle, dist, ply  # noqa: E402,F401
from .gaussian_renderer import (render_predicted_more_v2_gof, render_predicted_more_v3_gof,  # noqa: E402,F401
                                render_predicted_more_v2_gof_in, AlphaSweep,
                                render_views, depth_to_normal, depths_to_points)
from .gaussian_predictor import GaussianSplatPredictor_gtunet, splat_head  # noqa: E402,F401
from .unet_gs import Unet_GS_gtunet  # noqa: E402,F401



def set_option(name, value):
    """Process-wide runtime switch of the HIP library (include/f3dg.h, f3dg_set_option): e.g. ``set_option("tile_cull", 0)`` for the
    reference's tile lists bit for bit, ``set_option("render_fast", 0)`` for the reference's float32/float64 compositing order."""
    _lib.check(_lib.lib().f3dg_set_option(name.encode() if isinstance(name, str) else name, int(value)), "f3dg_set_option")


__version__ = "0.1.0"
