"""Times the SongUNet backbone of GaussianSplatPredictor_gtunet (random weights) on the GPU box: B images of 4x256x256.
Run under `rocprofv3 --kernel-trace --stats` to see which kernels the time goes to."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))
cfg = cameras.default_cfg(256)
torch.manual_seed(0)
pred = f3d.GaussianSplatPredictor_gtunet(cfg).to(dev).eval()
x = torch.rand(B, 4, 256, 256, device=dev)
mode = os.environ.get("MODE", "fp32")
if os.environ.get("BENCHMARK"):
    torch.backends.cudnn.benchmark = True
if mode in ("channels_last", "bf16_cl"):
    pred = pred.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)


def run():
    with torch.no_grad():
        if mode in ("bf16", "bf16_cl"):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return pred.network_with_offset(x)
        return pred.network_with_offset(x)


for _ in range(2):
    y = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    y = run()
torch.cuda.synchronize()
print(f"SongUNet forward mode={mode} benchmark={bool(os.environ.get('BENCHMARK'))} B={B}: {(time.perf_counter() - t0) / n * 1e3:.1f} ms, out {tuple(y.shape)} {y.dtype}")
