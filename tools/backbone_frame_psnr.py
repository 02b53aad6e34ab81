"""What the 16-bit options of the backbone (bf16: 0.05-0.10 relative error on the 23 output channels of a ~100-layer network with
formula-defined weights; fp16: three more mantissa bits) do to a rendered frame: the real image of fixture F6 through predictor + cycle
aggregation with the fp32 and with the 16-bit backbone, the same 32-view orbit of the merged sets, PSNR of the 8-bit RGB frames (and of
depth / alpha) between them.

  python tools/backbone_frame_psnr.py [bf16 fp16] [--checkpoint PATH]

Without --checkpoint the weights are the fixture generator's FORMULA weights (no checkpoint travels with this repository): what the
numbers then say is how far a perturbation of that size is amplified by an untrained network of this architecture -- nothing follows
from them for the released checkpoint. --checkpoint PATH loads a `state_dict` (a .pth / .pt file as the reference's visualize.py loads
it: either the dict itself or {"model_state_dict": ...}) into the predictor, so a maintainer who has the checkpoint can measure the
16-bit options on the trained network.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import f3dgaus_amd as f3d  # noqa: E402
from real_data import load_real_image, real_predictor  # noqa: E402

dev = torch.device("cuda:0")
images, depth, _ = load_real_image(dev)
frames = {}
OPTIONS = [a for a in sys.argv[1:] if a in ("bf16", "fp16")] or ["bf16", "fp16"]
CKPT = sys.argv[sys.argv.index("--checkpoint") + 1] if "--checkpoint" in sys.argv else None
print("weights: %s" % (CKPT or "formula weights of the fixture generator (NOT a trained network: see the module docstring)"))
for backbone in ["fp32"] + OPTIONS:
    model, cfg = real_predictor(dev, 256, backbone)
    if CKPT:
        sd = torch.load(CKPT, map_location="cpu")
        sd = sd.get("model_state_dict", sd) if isinstance(sd, dict) else sd
        missing, unexpected = model.load_state_dict(sd, strict=False)
        print("  %s: loaded %s (%d missing, %d unexpected keys)" % (backbone, CKPT, len(missing), len(unexpected)))
        model = model.to(dev)
    with torch.no_grad():
        merged = f3d.cycle.cycle_aggregate(model, images, depth, cfg)
        frames[backbone] = f3d.cycle.render_orbit(merged, cfg, num_views=32, views_per_call=32)
    del model


def psnr(a, b, peak):
    mse = ((a.double() - b.double()) ** 2).mean(dim=(-3, -2, -1))
    return 10.0 * torch.log10(peak * peak / mse.clamp_min(1e-30))


q = lambda t: (t.clamp(0, 1) * 255.0).round()
fmt = lambda p: "min %.1f / median %.1f / max %.1f dB" % (float(p.min()), float(p.median()), float(p.max()))
a = frames["fp32"]
dmax = float(a["rendered_depth"][0].max())
for opt in OPTIONS:
    b = frames[opt]
    finite = bool(torch.isfinite(b["render"]).all())
    p_rgb = psnr(q(a["render"][0]), q(b["render"][0]), 255.0)
    p_alpha = psnr(a["rendered_alpha"][0], b["rendered_alpha"][0], 1.0)
    p_depth = psnr(a["rendered_depth"][0], b["rendered_depth"][0], dmax)
    print("%s backbone against fp32 backbone, real image, 32 orbit views of the merged sets (589,824 Gaussians each)%s:" % (opt, "" if finite else " -- NON-FINITE FRAMES"))
    print("  8-bit RGB frames   PSNR " + fmt(p_rgb))
    print("  alpha              PSNR " + fmt(p_alpha))
    print("  median depth       PSNR " + fmt(p_depth) + " (peak = %.2f)" % dmax)
