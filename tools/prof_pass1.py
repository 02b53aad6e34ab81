"""AlphaSweep preparation (f3dg_integrate_prepare_batched) of 16 cameras of 589,824 Gaussians in a loop, for rocprofv3 runs on
integrate_pass1_cull_kernel (tools/pmc_pass1.sh). Prints the time per camera."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras, synthetic  # noqa: E402
dev = torch.device("cuda:0")
RES, P, PN = 256, int(os.environ.get("P", 589824)), 1_000_000
V = int(os.environ.get("V", 16))
cfg = cameras.default_cfg(RES)
g = synthetic.make_gaussians(P, s0=float(os.environ.get("S0", 0.01)), seed=0, device=dev)
oc = synthetic.orbit_cameras(V, resolution=RES, device=dev)
pc = {k: g[k][None] for k in ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest")}
bg = torch.zeros(3, device=dev)
for rep in range(int(os.environ.get("REPS", 3))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sweep = f3d.AlphaSweep(pc, 0, oc["viewmatrix"], oc["projmatrix"], oc["campos"], bg, cfg, max_points=PN, cameras_per_call=V)
    torch.cuda.synchronize()
    print("prepare: %.3f ms per camera" % ((time.perf_counter() - t0) / V * 1e3), flush=True)
