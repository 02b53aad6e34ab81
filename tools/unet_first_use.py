"""First-use and steady-state time of backbone passes on a fresh box: python tools/unet_first_use.py <dtype> <deterministic 0/1> B [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras  # noqa: E402

dtype, det = sys.argv[1], int(sys.argv[2])
torch.backends.cudnn.deterministic = bool(det)
dev = torch.device("cuda:0")
cfg = cameras.default_cfg(256)
cfg["model"]["backbone_dtype"] = dtype
cfg["model"]["backbone_chunk"] = int(os.environ.get("CHUNK", "0"))
torch.manual_seed(0)
pred = f3d.GaussianSplatPredictor_gtunet(cfg).to(dev).eval()
rig = cameras.OrbitRig(cfg).canonical
for B in [int(a) for a in sys.argv[3:]]:
    x = torch.rand(B, 1, 4, 256, 256, device=dev)
    v2w = rig.view_to_world_transforms.expand(B, 1, 4, 4).to(dev)
    quat = rig.source_cv2wT_quat.expand(B, 1, 4).to(dev)
    depth = torch.rand(B, 1, 256, 256, device=dev) * 2 + 6.667
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            pred(x, v2w, quat, unet_depth=depth)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{dtype} deterministic={det} chunk={cfg['model']['backbone_chunk']} B={B}: first {ts[0]:.2f} s, then {ts[1] * 1e3:.1f} / {ts[2] * 1e3:.1f} ms", flush=True)
