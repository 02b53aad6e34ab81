"""Per-channel parity statistics of the HIP path vs the CPU oracle on the test scenes, in BOTH arithmetic modes (exact = the
reference's float32/float64 order, fast = error-free float32 pairs), next to the NOISE FLOOR of the reference's own arithmetic
(SURVEY 8d): the oracle against itself with tan_fov moved by one float32 ulp -- the GOF exponent amplifies that 6e-8 relative
change by C = t^2/sigma^2. Run on the GPU box; the output is committed under profiles/ (markdown)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from f3dgaus_amd import _lib  # noqa: E402
from helpers import frac_within, make_scene, psnr, run_hip, run_oracle  # noqa: E402
from test_raster_forward_gpu import SCENES  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
scenes = {k: v for k, v in SCENES.items() if k not in ("F10_huge_tile_lists",)}
scenes["C1_65536_256"] = dict(P=65536, res=(256, 256), s0=0.01, view="oblique")
scenes["C2_view_196608_256"] = dict(P=196608, res=(256, 256), s0=0.01, view="oblique")


def stats(a, b):
    d = np.abs(a.astype(np.float64) - b)
    rel = (d[8] / np.maximum(np.abs(b[8]), 1e-12))[np.abs(b[8]) > 1e-7]
    return (f"{d[:3].max():.1e} | {psnr(a[:3], b[:3]):.1f} | {frac_within(a[:3], b[:3], 1e-4):.5f} | {d[3:6].max():.1e} | "
            f"{int((d[6] > 1e-4 * np.abs(b[6])).sum())} | {d[7].max():.1e} | {d[8].max():.1e} | {np.median(rel) if rel.size else 0:.1e}")


print("| scene | instances | what | rgb max abs | rgb PSNR dB | rgb frac <= 1e-4 | normal max | depth px off | alpha max | dist max abs | dist median rel |")
print("|---|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for name, kw in scenes.items():
    sc = make_scene(**kw)
    o = run_oracle(sc)
    res = {}
    for mode, val in (("exact", 0), ("fast", 2)):
        L.f3dg_set_option(b"render_fast", val)
        res[mode] = run_hip(sc, dev)
    L.f3dg_set_option(b"render_fast", 1)
    sc2 = dict(sc)
    sc2["tanfovx"] = float(np.nextafter(np.float32(sc["tanfovx"]), np.float32(1)))      # +1 ulp in float32
    sc2["tanfovy"] = float(np.nextafter(np.float32(sc["tanfovy"]), np.float32(1)))
    o2 = run_oracle(sc2)
    for what, a in (("HIP exact vs oracle", res["exact"]["out_color"][0]), ("HIP fast vs oracle", res["fast"]["out_color"][0]),
                    ("noise floor: oracle vs oracle(tan_fov + 1 ulp)", o2["out_color"])):
        print(f"| {name} | {o['num_rendered']} | {what} | {stats(a, o['out_color'])} |")
    same = (res["exact"]["n_contrib"][0] == o["n_contrib"]).mean(), (res["fast"]["n_contrib"][0] == o["n_contrib"]).mean()
    print(f"| {name} | | n_contrib identical: exact {same[0]:.6f}, fast {same[1]:.6f} | | | | | | | | |")
