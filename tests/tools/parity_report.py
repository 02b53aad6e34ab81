"""Prints per-channel parity statistics of the HIP path vs the CPU oracle on the test scenes (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_scene, psnr, run_hip, run_oracle  # noqa: E402
from test_raster_forward_gpu import SCENES  # noqa: E402

dev = torch.device("cuda:0")
scenes = dict(SCENES)
scenes["C1_65536_256"] = dict(P=65536, res=(256, 256), s0=0.01, view="oblique")
names = ["rgb", "rgb", "rgb", "nrm", "nrm", "nrm", "depth", "alpha", "dist"]
for name, kw in scenes.items():
    sc = make_scene(**kw)
    h, o = run_hip(sc, dev), run_oracle(sc)
    a, b = h["out_color"][0], o["out_color"]
    d = np.abs(a.astype(np.float64) - b)
    print(f"{name}: R={h['num_rendered']} rgb max {d[:3].max():.2e} psnr {psnr(a[:3], b[:3]):.1f} dB | normal max {d[3:6].max():.2e} | "
          f"depth max {d[6].max():.2e} (#>1e-4*z: {(d[6] > 1e-4 * np.abs(b[6])).sum()}) | alpha max {d[7].max():.2e} | "
          f"dist max abs {d[8].max():.2e} max rel {(d[8] / np.maximum(np.abs(b[8]), 1e-12))[np.abs(b[8]) > 1e-7].max() if (np.abs(b[8]) > 1e-7).any() else 0:.2e} | "
          f"n_contrib equal {(h['n_contrib'][0] == o['n_contrib']).mean():.6f}")
