"""CPU model of the compositing backward's lock-step walk (csrc/f3dg_backward.hip: render3_bwd_kernel) on a C5-shaped view: how many
(wave, entry) steps does a quadrant's wave take when its 64 pixels walk every entry that reaches ANY of them (the kernel), and how many
when each 16-lane row owns a 4x4 pixel block and walks only the entries that reach ITS block (the four blocks advance
together, a window of 64 kept entries lasts as long as its busiest block)? Also the active pixels per step and the atomic operations.
Needs the oracle; test infrastructure only.

  python tests/tools/bwd_block_model.py [P] [res] [n_tiles] [s0]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_scene, run_oracle

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 32
S0 = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
sc = make_scene(P=P, res=(RES, RES), s0=S0, view="oblique")
o = run_oracle(sc)
f32 = np.float32
fx = float(f32(RES) / (f32(2.0) * f32(sc["tanfovx"])))
v64 = o["view2gaussian"].astype(np.float64)
opac = o["conic_opacity"][:, 3].astype(np.float64)
ranges, pl = o["ranges"], o["point_list"]
nc = o["n_contrib"][0]
tx_n = RES // 16
rng = np.random.default_rng(0)
tiles = rng.choice(tx_n * tx_n, NT, replace=False)
tot = dict(steps_q=0, act_q=0, steps_b=0, act_b=0, units_b=0, kept=0)
for tile in tiles:
    r0, r1 = ranges[tile]
    ids = pl[r0:r1]
    n = len(ids)
    if n == 0:
        continue
    ty, tx = divmod(tile, tx_n)
    ys, xs = np.meshgrid(np.arange(ty * 16, ty * 16 + 16), np.arange(tx * 16, tx * 16 + 16), indexing="ij")
    rx = ((xs + 0.5 - RES / 2) / fx).reshape(-1, 1)
    ry = ((ys + 0.5 - RES / 2) / fx).reshape(-1, 1)
    v = v64[ids][None]
    n0 = v[..., 0] * rx + v[..., 1] * ry + v[..., 2]
    n1 = v[..., 1] * rx + v[..., 3] * ry + v[..., 4]
    n2 = v[..., 2] * rx + v[..., 4] * ry + v[..., 5]
    a = rx * n0 + ry * n1 + n2
    b = v[..., 6] * rx + v[..., 7] * ry + v[..., 8]
    p = np.minimum(-0.5 * (v[..., 9] - b * b / a), 0)
    thr = np.log(1.0 / (255.0 * np.maximum(opac[ids][None], 1e-12)))
    t = -b / a
    last = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].astype(np.int64).reshape(-1, 1)
    pos = np.arange(n)[None, :]
    act = (p >= thr) & (t > 0.2) & (pos < last)          # contributing (pixel, entry) pairs          [256, n]
    ell = (p >= 1.11 * np.minimum(thr, 0) - 1e-3) & (pos < last)   # what the conservative ellipse lets into the walk
    py, px = np.divmod(np.arange(256), 16)
    for q in range(4):
        lanes = np.nonzero(((py // 8) * 2 + px // 8) == q)[0]
        eq = ell[lanes]                                  # [64, n]
        anyq = eq.any(0)
        kept = np.nonzero(anyq)[0]                       # entries the wave walks (any pixel of the quadrant)
        tot["kept"] += len(kept)
        tot["steps_q"] += len(kept)
        tot["act_q"] += int(act[lanes][:, kept].sum())
        blk = ((py[lanes] % 8) // 4) * 2 + (px[lanes] % 8) // 4
        anyb = np.stack([eq[blk == bb].any(0) for bb in range(4)])      # [4, n]
        tot["units_b"] += int(anyb.sum())
        tot["act_b"] += int(act[lanes][:, kept].sum())
        for s in range(0, len(kept), 64):                # a window of 64 kept entries lasts as long as its busiest block
            tot["steps_b"] += int(anyb[:, kept[s:s + 64]].sum(1).max())
print("C5-shaped view: P %d, %d^2, sigma0 %.3g, %d tiles sampled" % (P, RES, S0, NT))
print("kernel (quadrant walks):   steps %d, contributing pixels per step %.1f of 64, atomic lane-operations %d (17 per step)" %
      (tot["steps_q"], tot["act_q"] / max(tot["steps_q"], 1), 17 * tot["steps_q"]))
print("4x4 blocks (row walks):    steps %d (x%.2f), contributing pixels per step %.1f of 64, atomic lane-operations %d (17 per (block, entry): x%.2f)" %
      (tot["steps_b"], tot["steps_b"] / max(tot["steps_q"], 1), tot["act_b"] / max(tot["steps_b"], 1), 17 * tot["units_b"],
       tot["units_b"] / max(tot["steps_q"], 1)))
