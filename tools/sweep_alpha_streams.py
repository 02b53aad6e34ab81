"""AlphaSweep camera preparation with 1..16 cameras in flight (589,824 Gaussians, 64 cameras @256^2); the first lines include the
one-time device allocation of the pooled workspaces."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f3dgaus_amd as f3d
from f3dgaus_amd import cameras as _cams, synthetic
dev = torch.device("cuda:0")
RES, P, PN, V = 256, 589824, 1_000_000, 64
cfg = _cams.default_cfg(RES)
g = synthetic.make_gaussians(P, s0=0.01, seed=0, device=dev)
oc = synthetic.orbit_cameras(V, resolution=RES, device=dev)
pc = {"xyz": g["xyz"][None], "opacity": g["opacity"][None], "scaling": g["scaling"][None], "rotation": g["rotation"][None],
      "features_dc": g["features_dc"][None], "features_rest": g["features_rest"][None]}
bg = torch.zeros(3, device=dev)
for ns in (1, 2, 4, 8, 16, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sweep = f3d.AlphaSweep(pc, 0, oc["viewmatrix"], oc["projmatrix"], oc["campos"], bg, cfg, max_points=PN, streams=ns)
    torch.cuda.synchronize(); print(ns, "streams:", round((time.perf_counter() - t0) / V * 1e3, 2), "ms per camera", flush=True)
    del sweep
