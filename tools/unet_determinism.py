"""Is a backbone pass run-to-run bit-reproducible? Runs the same 8-image pass several times per layout and reports the max difference;
then each distinct convolution shape of the pass twice (F.conv2d alone) and the attention (SDPA) twice."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand(B, 4, 256, 256, device=dev)
for layout in ("nhwc", "nchw"):
    cfg = cameras.default_cfg(256)
    cfg["model"]["backbone_layout"] = layout
    torch.manual_seed(1)
    pred = f3d.GaussianSplatPredictor_gtunet(cfg).to(dev).eval()
    xin = x.contiguous(memory_format=torch.channels_last) if layout == "nhwc" else x
    with torch.no_grad():
        ys = [pred.network_with_offset(xin, N_views_xa=1).float().contiguous() for _ in range(4)]
    d = [float((ys[0] - y).abs().max()) for y in ys[1:]]
    print(layout, "max |diff| between passes:", d, "of max", float(ys[0].abs().max()), flush=True)
    # which convolution shapes are not reproducible?
    shapes = {}
    hooks = []
    orig = F.conv2d

    def spy(inp, w, b=None, stride=1, padding=0, *a, **k):
        key = (tuple(inp.shape), tuple(w.shape), padding, inp.is_contiguous(memory_format=torch.channels_last) and inp.shape[1] > 1)
        if key not in shapes:
            shapes[key] = (inp.detach().clone(), w.detach().clone())
        return orig(inp, w, b, stride, padding, *a, **k)
    F.conv2d = spy
    with torch.no_grad():
        pred.network_with_offset(xin, N_views_xa=1)
    F.conv2d = orig
    bad = 0
    for key, (inp, w) in shapes.items():
        with torch.no_grad():
            a = orig(inp, w, None, 1, key[2])
            outs = [orig(inp, w, None, 1, key[2]) for _ in range(3)]
        dd = max(float((a - o).abs().max()) for o in outs)
        if dd != 0.0:
            bad += 1
            print("   conv not reproducible:", key, "max |diff|", dd, "of", float(a.abs().max()), flush=True)
    print(layout, "distinct conv shapes", len(shapes), "not reproducible", bad, flush=True)
q = torch.randn(B, 1, 1024, 256, device=dev)
with torch.no_grad():
    a = F.scaled_dot_product_attention(q, q, q)
    print("SDPA reproducible:", all(torch.equal(a, F.scaled_dot_product_attention(q, q, q)) for _ in range(3)))
