"""The backbone's convolutions in the two memory layouts (NCHW = torch default, NHWC = channels_last), fp32 and bf16, with MIOpen's
algorithm search on: does the layout remove the batched_transpose kernels MIOpen wraps around its NHWC kernels, and what does a
whole pass of the backbone cost in channels_last (native GroupNorm: the fused kernel is NCHW)?

  python tools/ubench_conv_layout.py
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
shapes = [((8, 128, 256, 256), (128, 128, 3, 3)), ((8, 256, 128, 128), (256, 256, 3, 3)), ((8, 256, 256, 256), (128, 256, 3, 3)),
          ((8, 256, 64, 64), (256, 256, 3, 3)), ((8, 128, 256, 256), (128, 128, 1, 1))]


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("| input | weight | dtype | NCHW ms | TFLOP/s | NHWC ms | TFLOP/s |")
print("|---|---|---|---:|---:|---:|---:|")
for ish, wsh in shapes:
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(ish, device=dev, dtype=dt)
        w = torch.randn(wsh, device=dev, dtype=dt) * 0.05
        b = torch.zeros(wsh[0], device=dev, dtype=dt)
        pad = wsh[2] // 2
        fl = 2.0 * ish[0] * wsh[0] * ish[2] * ish[3] * wsh[1] * wsh[2] * wsh[3]
        t0 = bench(lambda: F.conv2d(x, w, b, padding=pad))
        xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        t1 = bench(lambda: F.conv2d(xc, wc, b, padding=pad))
        print(f"| {list(ish)} | {list(wsh)} | {str(dt)[6:]} | {t0:.3f} | {fl / t0 / 1e9:.1f} | {t1:.3f} | {fl / t1 / 1e9:.1f} |", flush=True)

# the whole backbone, bf16 autocast: the build's kernels in both layouts, and torch's GroupNorm in both (which is what made channels_last
# lose in round 2: it converts the tensor back)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras, gaussian_predictor as gp  # noqa: E402
cfg = cameras.default_cfg(256)
torch.manual_seed(0)
pred = f3d.GaussianSplatPredictor_gtunet(cfg).to(dev).eval()
x = torch.rand(8, 4, 256, 256, device=dev)


def run(inp):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return pred.network_with_offset(inp)


t_nchw = bench(lambda: run(x), 3)
pred_cl = pred.to(memory_format=torch.channels_last)
xcl = x.contiguous(memory_format=torch.channels_last)
t_cl_fused = bench(lambda: run(xcl), 3)
_gn = gp.GroupNorm.forward


def gn_native(self, x, N_views_xa=1, silu=False, pre_bias=None):
    if pre_bias is not None:
        x = x + pre_bias.to(x.dtype).reshape(1, -1, 1, 1)
    y = F.group_norm(x, self.num_groups, self.weight.to(x.dtype), self.bias.to(x.dtype), self.eps)
    return F.silu(y) if silu else y


gp.GroupNorm.forward = gn_native
t_cl = bench(lambda: run(xcl), 3)
t_nchw_native = bench(lambda: run(x), 3)
gp.GroupNorm.forward = _gn
print(f"\nbackbone, 8 images, bf16 autocast: NCHW + the build's kernels {t_nchw:.1f} ms; channels_last + the build's kernels {t_cl_fused:.1f} ms; "
      f"NCHW + torch GroupNorm {t_nchw_native:.1f} ms; channels_last + torch GroupNorm {t_cl:.1f} ms")
