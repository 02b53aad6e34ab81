"""One pass of the SongUNet backbone (reference src/gaussian_predictor.py:137-511; this build's f3d-gaus_amd/gaussian_predictor.py) under
torch.profiler: device time per (operator, input shapes) with the FLOPs torch derives from the shapes, i.e. achieved TFLOP/s of the
convolutions and matrix products that make up the pass, plus the GPU kernels behind the five most expensive ones.

  python tools/prof_unet.py B mode       B = images per pass; mode = fp32 | bf16 (the opt-in autocast option) | bf16_resident
                                         (the same with the convolution weights held in bfloat16: no per-call cast kernels)
Prints a markdown section (profiles/r04_final/unet.md is assembled from these runs)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras, gaussian_predictor as gp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
graph = mode.endswith("_graph")      # ..._graph: the pass replayed from the HIP graph the predictor captures for small batches (wall time only)
mode = mode.replace("_graph", "")
nhwc = mode.endswith("_nhwc")        # fp32_nhwc / bf16_nhwc: the channels-last layout option (activations and filters)
mode_label, mode = mode + ("_graph" if graph else ""), mode.replace("_nhwc", "")
dev = torch.device("cuda:0")
cfg = cameras.default_cfg(256)
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
pred = f3d.GaussianSplatPredictor_gtunet(cfg).to(dev).eval()
x = torch.rand(B, 4, 256, 256, device=dev)
if nhwc:
    pred.network_with_offset.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)

if mode == "bf16_resident":
    # experiment: every convolution weight / bias cast ONCE (Conv2d.forward casts `self.weight.to(x.dtype)` per call; under autocast
    # the cast is cached per autocast region, i.e. repeated every pass)
    for m in pred.modules():
        if isinstance(m, gp.Conv2d) and m.weight is not None:
            m.weight.data = m.weight.data.to(torch.bfloat16)
            m.bias.data = m.bias.data.to(torch.bfloat16)


def run():
    with torch.no_grad():
        if mode.startswith("bf16"):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return pred.network_with_offset(x)
        return pred.network_with_offset(x)


if graph:
    # the same pass captured into a HIP graph (torch.cuda.CUDAGraph, static input) and replayed: is a small pass bound by host launches?
    eager = run
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = eager()

    def run():
        g.replay()
        return static_out
for _ in range(3):
    y = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10 if graph else 3):
    y = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / (10 if graph else 3) * 1e3
if graph:
    ye = eager()
    print(f"### {B} images, {mode_label}: {ms:.2f} ms per pass (wall, 10 replays); max |graph - eager| = {(y.float() - ye.float()).abs().max().item():.3g} "
          f"of {ye.float().abs().max().item():.3g}\n")
    sys.exit(0)

# ---- every convolution of the pass between two HIP events: milliseconds, FLOPs (2 N Cout Hout Wout Cin k^2) and TFLOP/s per shape.
# (torch's with_flops knows aten::conv2d but the device time sits on aten::miopen_convolution and the kernels below it, so the
# profiler table further down has no FLOP column worth reading.)
import torch.nn.functional as F  # noqa: E402
_conv2d = F.conv2d
conv_log = []


def timed_conv2d(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = _conv2d(inp, weight, bias, stride, padding, dilation, groups)
    e1.record()
    conv_log.append((tuple(inp.shape), tuple(weight.shape), tuple(out.shape), str(inp.dtype).replace("torch.", ""), e0, e1))
    return out


F.conv2d = timed_conv2d
run()
torch.cuda.synchronize()
F.conv2d = _conv2d
groups = {}
for ishape, wshape, oshape, dt_, e0, e1 in conv_log:
    fl = 2.0 * oshape[0] * oshape[1] * oshape[2] * oshape[3] * wshape[1] * wshape[2] * wshape[3]
    g = groups.setdefault((ishape, wshape, dt_), [0, 0.0, 0.0])
    g[0] += 1; g[1] += e0.elapsed_time(e1); g[2] += fl
tot_ms = sum(g[1] for g in groups.values()); tot_fl = sum(g[2] for g in groups.values())
peak = 157.3 if mode == "fp32" else 2500.0
print(f"### {B} images, {mode_label}: convolutions between HIP events: {len(conv_log)} calls, {tot_ms:.1f} ms, {tot_fl / 1e12:.2f} TFLOP -> "
      f"**{tot_fl / tot_ms / 1e9:.1f} TFLOP/s** ({100 * tot_fl / tot_ms / 1e9 / peak:.1f} % of the {peak:.0f} TFLOP/s dense {'fp32 vector / matrix' if mode == 'fp32' else 'bf16 MFMA'} peak)\n")
print("| input | weight | dtype | calls | ms | GFLOP | TFLOP/s |")
print("|---|---|---|---:|---:|---:|---:|")
for (ishape, wshape, dt_), g in sorted(groups.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"| {list(ishape)} | {list(wshape)} | {dt_} | {g[0]} | {g[1]:.3f} | {g[2] / 1e9:.1f} | {g[2] / g[1] / 1e9:.1f} |")
print()

from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_flops=True) as prof:
    run()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
dt = lambda e: getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
rows = sorted((e for e in ka if dt(e) > 0), key=dt, reverse=True)
total_us = sum(dt(e) for e in rows)
print(f"### {B} images, {mode_label}: {ms:.1f} ms per pass (wall, 3 passes), {total_us / 1e3:.1f} ms of device time in {len(rows)} (operator, shape) groups\n")
print("| operator | input shapes | calls | device ms | % | GFLOP | TFLOP/s |")
print("|---|---|---:|---:|---:|---:|---:|")
for e in rows[:14]:
    fl = e.flops or 0
    shapes = str(e.input_shapes)[:90]
    print(f"| `{e.key}` | {shapes} | {e.count} | {dt(e) / 1e3:.3f} | {100 * dt(e) / total_us:.1f} | {fl / 1e9:.1f} | "
          f"{(fl / (dt(e) * 1e-6) / 1e12) if fl else 0:.1f} |")
conv_us = sum(dt(e) for e in rows if "conv" in e.key)
conv_fl = sum((e.flops or 0) for e in rows if "conv" in e.key)
mm_us = sum(dt(e) for e in rows if e.key in ("aten::bmm", "aten::mm", "aten::addmm", "aten::matmul", "aten::_scaled_dot_product_flash_attention", "aten::_scaled_dot_product_efficient_attention"))
cast_us = sum(dt(e) for e in rows if e.key in ("aten::_to_copy", "aten::copy_", "aten::to"))
gn_us = total_us - conv_us - mm_us - cast_us
print(f"\nconvolutions: {conv_us / 1e3:.1f} ms, {conv_fl / 1e12:.2f} TFLOP -> {conv_fl / max(conv_us, 1) / 1e6:.1f} TFLOP/s overall; matrix products / attention "
      f"{mm_us / 1e3:.2f} ms; dtype casts / copies {cast_us / 1e3:.2f} ms; everything else (GroupNorm+SiLU kernel, resampling, adds) {gn_us / 1e3:.2f} ms\n")
# kernels behind the device time
kern = {}
for ev in prof.events():
    if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower() or str(getattr(ev, "device_type", "")).endswith("CUDA"):
        kern.setdefault(ev.name, [0, 0.0])
        kern[ev.name][0] += 1
        kern[ev.name][1] += getattr(ev, "device_time", 0) or getattr(ev, "cuda_time", 0) or 0
top = sorted(kern.items(), key=lambda kv: kv[1][1], reverse=True)[:8]
if top and top[0][1][1] > 0:
    print("| GPU kernel | launches | ms |")
    print("|---|---:|---:|")
    for name, (n, us) in top:
        print(f"| `{name[:110]}` | {n} | {us / 1e3:.3f} |")
    print()
