"""Which engine moves a device -> pinned-host copy issued by torch (hipMemcpyAsync)? Run under `rocprofv3 --kernel-trace --stats`:
a shader copy shows up as the kernel __amd_rocclr_copyBuffer, an SDMA copy does not. Prints the copy's time alone and while a
long VALU kernel runs on another stream (what the frame copy of bench.py does behind the next step's rendering)."""
import os
import time
import torch
dev = torch.device("cuda:0")
n = 120 * 256 * 256 * 3
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()
def copy_alone(k=20):
    t0 = time.perf_counter()
    for _ in range(k):
        with torch.cuda.stream(side):
            dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
def busy(k=20, copy=True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        if copy:
            with torch.cuda.stream(side):
                dst.copy_(src, non_blocking=True)
        b = torch.sin(a) * 1.0001          # a few ms of VALU work on the main stream
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
copy_alone(3); busy(3)
print("env", {k: v for k, v in os.environ.items() if k.startswith(("GPU_", "HSA_ENABLE_SDMA", "HSA_FORCE", "DEBUG_CLR"))})
print("copy alone %.3f ms (%.1f GB/s)" % (copy_alone(), n / copy_alone() / 1e6))
print("main-stream work alone %.3f ms / iteration, with the copy on a side stream %.3f ms" % (busy(copy=False), busy(copy=True)))
