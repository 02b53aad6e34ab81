"""Differential fuzz of every filter of the compositing path (needs a GPU; not part of the test suite): random scenes, exact
arithmetic, the plain transcription (pixel-lane kernel without pre-test / culling / queues, the reference's tile lists) against the
default configuration (render2: ellipse test across the lanes, tile-culled lists). The filters only remove work whose result is a
bare `continue`, so the nine channels must agree bit for bit. Also integrate: plain pass 1 against the culled one.
    python tests/tools/fuzz_culling.py [n_scenes] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import f3dgaus_amd as f3d  # noqa: E402
from helpers import make_scene  # noqa: E402
from helpers_integrate import hip_integrate, make_points  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")


def render(sc):
    d = lambda t: None if t is None else t.to(dev)
    out, radii, ws = f3d.rasterize_views(
        d(sc["means3D"]), d(sc["opacities"]), d(sc["viewmatrix"]), d(sc["projmatrix"]), d(sc["campos"]), d(sc["bg"]),
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], sh=d(sc["shs"]),
        colors_precomp=d(sc["colors_precomp"]), scales=d(sc["scales"]), rotations=d(sc["rotations"]), sh_degree=sc["sh_degree"],
        scale_modifier=sc["scale_modifier"], kernel_size=sc["kernel_size"], save_aux=False)
    return out.clone(), radii.clone(), ws.num_rendered


def plain(on):
    f3d.set_option("render_kernel", 1 if on else 2)
    for o in ("render_pretest", "render_cull", "render_queue"):
        f3d.set_option(o, 0 if on else 1)
    f3d.set_option("tile_cull", 0 if on else 1)


f3d.set_option("render_fast", 0)
bad = 0
for i in range(n_scenes):
    big = os.environ.get("BIG") and i % 4 == 0      # BIG=1: every fourth scene up to 520 px (two tile passes) and 200 k Gaussians
    hi = 66 if big else 20
    W, H = int(rng.integers(2, hi)) * 8 + int(rng.integers(0, 8)), int(rng.integers(2, hi)) * 8 + int(rng.integers(0, 8))
    kw = dict(P=int(rng.integers(200, 200000 if big else 30000)), res=(W, H), s0=float(10 ** rng.uniform(-2.6, -0.4)), seed=int(rng.integers(1 << 30)),
              view=[int(v) for v in rng.choice(9, size=int(rng.integers(1, 4)), replace=False)], aniso=bool(rng.integers(0, 2)),
              kernel_size=float(rng.choice([0.0, 0.0, 0.1, 0.3])), scale_modifier=float(rng.choice([1.0, 1.0, 0.5, 2.0])),
              behind_fraction=float(rng.choice([0.0, 0.05])), sh_degree=int(rng.integers(0, 2)))
    if rng.integers(0, 4) == 0:
        kw["depth_range"] = (1.0, float(rng.uniform(4, 40)))
    sc = make_scene(**kw)
    try:
        plain(True); a = render(sc)
        plain(False); b = render(sc)
        same = torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1], b[1])
        msg = ""
        if i % 3 == 0:            # integrate on one view of the scene: plain pass 1 against the culled one
            one = make_scene(**dict(kw, view=kw["view"][:1]))
            pts = make_points(one, 20000)
            plain(True); ia = hip_integrate(one, pts, dev)
            plain(False); ib = hip_integrate(one, pts, dev)
            isame = all(np.array_equal(ia[k].view(np.uint32), ib[k].view(np.uint32)) for k in ("out", "ai", "ci", "radii"))
            same, msg = same and isame, " integrate " + ("ok" if isame else "DIFFERS")
    finally:
        plain(False)
    if not same:
        bad += 1
    print(f"{i:3d} {'ok ' if same else 'DIFF'} instances {a[2]} -> {b[2]}{msg} {kw}", flush=True)
f3d.set_option("render_fast", 1)
print("scenes:", n_scenes, "mismatches:", bad)
sys.exit(1 if bad else 0)
