"""Writes the merged Gaussian set of the real image (tests/real_data.py; needs a HIP device for the predictor) to
gpurun_out/real_set.npz (float16 except positions; SH rest dropped) so that schedule models can run on it on a CPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from real_data import real_merged_set  # noqa: E402

g = {k: v.cpu().numpy() for k, v in real_merged_set(torch.device("cuda:0")).items()}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "real_set.npz"), xyz=g["xyz"].astype(np.float32), opacity=g["opacity"].astype(np.float16),
                    scaling=g["scaling"].astype(np.float32), rotation=g["rotation"].astype(np.float16), features_dc=g["features_dc"].astype(np.float16))
print({k: v.shape for k, v in g.items()}, os.path.getsize(os.path.join(ROOT, "gpurun_out", "real_set.npz")) / 1e6, "MB")
