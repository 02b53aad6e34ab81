import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import f3dgaus_amd as f3d
from helpers import make_scene
dev = torch.device("cuda:0")
sc = make_scene(P=5000, res=(96, 96), s0=0.05, view="oblique")
d = lambda t: t.to(dev)
for name, mod in (("nan_means", lambda s: s["means3D"].__setitem__(slice(0, 50), float("nan"))),
                  ("inf_scales", lambda s: s["scales"].__setitem__(slice(0, 50), float("inf"))),
                  ("zero_scales", lambda s: s["scales"].__setitem__(slice(0, 50), 0.0)),
                  ("neg_opacity", lambda s: s["opacities"].__setitem__(slice(0, 50), -1.0)),
                  ("nan_opacity", lambda s: s["opacities"].__setitem__(slice(0, 50), float("nan"))),
                  ("zero_rot", lambda s: s["rotations"].__setitem__(slice(0, 50), 0.0)),
                  ("huge_scales", lambda s: s["scales"].__setitem__(slice(0, 20), 50.0))):
    s = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sc.items()}
    mod(s)
    out, radii, ws = f3d.rasterize_views(d(s["means3D"]), d(s["opacities"]), d(s["viewmatrix"]), d(s["projmatrix"]), d(s["campos"]), d(s["bg"]),
        image_height=96, image_width=96, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], sh=d(s["shs"]), scales=d(s["scales"]), rotations=d(s["rotations"]), sh_degree=1, save_aux=True)
    torch.cuda.synchronize()
    print(name, "instances", ws.num_rendered, "finite frac", float(torch.isfinite(out).float().mean()))
