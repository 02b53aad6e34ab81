"""Lab tool (needs a -DF3DG_LAB build: `F3DG_LAB=1 python f3d-gaus_amd/build.py --force`): what bounds the compositing launch?

Renders the workload's views once with the counting kernel (option render_count = 1, render_replay = 1: every quadrant wave logs its
number of slides), then times, with HIP events of the library's profile marks, (a) the normal launch, (b) render3s_stage_only_kernel
replaying exactly those slides -- list scan, 64-byte record gathers by global_load_lds, the ellipse ballots of phase 1, no phase 2 --
and (c) the same without the record gathers. (b) is what the launch costs as a stream of memory requests + its fixed per-slide work.

usage: python tools/replay_staging.py [--data real|synthetic] [--views 128] [--reps 6]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="real")
    ap.add_argument("--views", type=int, default=128)
    ap.add_argument("--gaussians", type=int, default=196608)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--channels", default="all")
    args = ap.parse_args()
    import f3dgaus_amd as f3d
    from f3dgaus_amd import _lib, synthetic
    L = _lib.lib()
    dev = torch.device("cuda:0")
    if args.data == "real":
        from real_data import real_merged_set
        g = real_merged_set(dev)
    else:
        g = synthetic.make_gaussians(args.gaussians, s0=0.01, seed=0, device=dev)
    P, V, RES = g["xyz"].shape[0], args.views, 256
    cams = synthetic.orbit_cameras(V, resolution=RES, device=dev)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    bg = torch.zeros(3, device=dev)
    out = torch.empty((V, 9, RES, RES), dtype=torch.float32, device=dev)
    state = {"ws": None}

    def call(check):
        _, _, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, image_height=RES,
                                       image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"],
                                       rotations=g["rotation"], sh_degree=1, workspace=state["ws"], out=out, check=check, channels=args.channels)
        state["ws"] = ws

    def opt(name, v):
        _lib.check(L.f3dg_set_option(name.encode(), v), "f3dg_set_option " + name)

    def timed(label):
        call(False)
        torch.cuda.synchronize()
        L.f3dg_profile_enable(1)
        for _ in range(args.reps):
            call(False)
        torch.cuda.synchronize()
        L.f3dg_profile_enable(0)
        st = (C.c_double * 5)()
        nc = C.c_int(0)
        per = (C.c_double * (3 * 64))()
        _lib.check(L.f3dg_profile_collect_calls(st, C.byref(nc), per, 64), "collect")
        rows = sorted(per[3 * k + 2] for k in range(min(nc.value, 64)))
        return {"what": label, "kernel": L.f3dg_debug_last_render_kernel().decode(), "compositing_ms_median": rows[len(rows) // 2], "min": rows[0], "max": rows[-1]}

    call(True)
    res = [timed("normal launch")]
    opt("render_count", 1); opt("render_replay", 1)
    cb = (C.c_ulonglong * 16)()
    L.f3dg_debug_render_counts(cb, 1)
    call(False)
    torch.cuda.synchronize()
    L.f3dg_debug_render_counts(cb, 1)
    opt("render_count", 0)
    counts = {"staged": int(cb[0]), "scanned": int(cb[1]), "wave_trips": int(cb[2]), "slides": int(cb[3]), "lane_trips": int(cb[4]), "waves": int(cb[5])}
    opt("render_replay", 2)
    res.append(timed("replay: scan + record gathers + phase 1 of the logged slides, no phase 2"))
    opt("render_replay", 3)
    res.append(timed("replay without the record gathers (list scan + ellipse loads + phase 1)"))
    opt("render_replay", 0)
    res.append(timed("normal launch again"))
    print(json.dumps({"P": P, "V": V, "counts": counts, "bytes_staged_80B": 80.0 * counts["staged"] + 4.0 * counts["scanned"], "runs": res}, indent=1))


if __name__ == "__main__":
    main()
