"""Where the host time of the drop-in loop goes: cProfile over N calls of render_predicted_more_v2_gof (one view per call, deferred
status on), top functions by own time, and the wall time per call.   python tools/prof_dropin_host.py [N]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f3dgaus_amd as f3d  # noqa: E402
from f3dgaus_amd import cameras, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
RES = 256
cfg = cameras.default_cfg(RES)
g = synthetic.make_pixel_gaussians(RES, s0=0.01, seed=0, device=dev)
pc = {k: g[k][None] for k in ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest")}
V = 60
cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
wv, fp, cc = (cams[k].unsqueeze(1) for k in ("viewmatrix", "projmatrix", "campos"))
wvs, fps, ccs = ([t[i:i + 1].contiguous() for i in range(V)] for t in (wv, fp, cc))
bg = torch.zeros(1, 3, device=dev)
frames = torch.empty((V, 3, RES, RES), dtype=torch.float32, device=dev)
f3d.set_deferred_status(True, depth=2)


def loop(n):
    with torch.no_grad():
        for i in range(n):
            th = i % V
            o = f3d.render_predicted_more_v2_gof(pc, 0, wvs[th], fps[th], ccs[th], bg, cfg)
            frames[th] = o["render"]
    f3d.flush()
    torch.cuda.synchronize()


loop(120)
t0 = time.perf_counter()
loop(N)
dt = time.perf_counter() - t0
print("wall %.1f us per call (%d calls), %.0f views/s" % (1e6 * dt / N, N, N / dt))
pr = cProfile.Profile()
pr.enable()
loop(N)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
