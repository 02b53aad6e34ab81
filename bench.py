#!/usr/bin/env python
"""bench.py -- headline benchmark of the GOF rasterization hot path on MI355X.

Metric (BASELINE.json): rendered views/s at 256x256 for (N Gaussians, K cameras), plus the achieved bytes/s of the
per-tile compositing kernel against the gfx950 HBM roofline. Workload at N=1 = BASELINE config C2: one image's
196,608 Gaussians ("~200k" = 3 cycle views x 65,536) rendered along a 120-view orbit at 256x256, forward only.
A "step" = one pass of the hot path over that batch: 120 views. With --gpus N every rank renders its own image
(weak scaling: the batch of input images shards embarrassingly) and the RGB frames are gathered to
rank 0 over RCCL inside the timed region.

Prints ONE JSON line (rank 0). `roofline` is measured live with HIP events recorded by the library on the launch
stream; `cpu_baseline` times the CPU oracle (the build's plain-C restatement: kind "port") on a bounded sample of
the same workload on this box's host cores.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=196608)
    ap.add_argument("--views", type=int, default=120)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--sigma0", type=float, default=0.01)
    ap.add_argument("--views-per-call", type=int, default=int(os.environ.get("F3DG_VIEWS_PER_CALL", "120")))
    ap.add_argument("--streams", type=int, default=int(os.environ.get("F3DG_STREAMS", "1")),
                    help="HIP streams the view chunks are distributed over (chunk i -> stream i %% streams)")
    ap.add_argument("--render-mode", choices=["fast", "exact"], default=os.environ.get("F3DG_RENDER_MODE", "fast"),
                    help="compositing arithmetic: fast = error-free float32 pairs for the float64 island (default, parity-gated "
                         "at 1e-4 / 99.9 %% / 80 dB), exact = the reference's float32/float64 operation order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-views", type=int, default=12)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the product path)"
    # F3DG_DIST_BACKEND=gloo is a functional check of the N>1 logic on a box with fewer GPUs than ranks (ranks share
    # devices, frames are gathered through host memory); the measured configuration is always nccl = RCCL.
    backend = os.environ.get("F3DG_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")

    import f3dgaus_amd as f3d
    from f3dgaus_amd import _lib, synthetic
    L = _lib.lib()
    _lib.check(L.f3dg_set_option(b"render_fast", 1 if args.render_mode == "fast" else 0), "f3dg_set_option")
    if os.environ.get("F3DG_DEBUG_SKIP_ALL"):    # experiment: no Gaussian ever passes -> the compositing kernel only stages
        _lib.check(L.f3dg_set_option(b"debug_skip_all", 1), "f3dg_set_option")
    if os.environ.get("F3DG_RENDER_KERNEL"):      # A/B of the compositing kernel generations (default: the library's)
        _lib.check(L.f3dg_set_option(b"render_kernel", int(os.environ["F3DG_RENDER_KERNEL"])), "f3dg_set_option")

    P, V, RES = args.gaussians, args.views, args.res
    g = synthetic.make_gaussians(P, s0=args.sigma0, seed=rank, device=device)     # every rank = a different image
    cams = synthetic.orbit_cameras(V, resolution=RES, device=device)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    bg = torch.zeros(3, device=device)
    out = torch.empty((V, 9, RES, RES), dtype=torch.float32, device=device)
    radii = torch.empty((V, P), dtype=torch.int32, device=device)
    chunks = [(a, min(a + args.views_per_call, V)) for a in range(0, V, args.views_per_call)]
    workspaces = {}
    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else None

    def render_chunk(a, b, check):
        ws = workspaces.get((a, b) if streams else b - a)
        o, r, ws = f3d.rasterize_views(
            g["xyz"], g["opacity"], cams["viewmatrix"][a:b], cams["projmatrix"][a:b], cams["campos"][a:b], bg,
            image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs,
            scales=g["scaling"], rotations=g["rotation"], sh_degree=1, workspace=ws, out=out[a:b], radii=radii[a:b],
            save_aux=False, check=check)
        return ws

    # calibration pass: sizes every chunk's workspace (capacity = max over chunks, +25 %), counts instances
    counts = []
    for a, b in chunks:
        ws = render_chunk(a, b, check=True)
        counts.append((b - a, ws.num_rendered))
    for n in set(c[0] for c in counts):
        cap = int(max(c[1] for c in counts if c[0] == n) * 1.25) + 4096
        if streams:      # concurrent chunks need their own workspace
            for (a, b) in chunks:
                if b - a == n:
                    workspaces[(a, b)] = f3d.diff_gof_rasterization.Workspace(P, RES, RES, n, cap, device)
        else:
            workspaces[n] = f3d.diff_gof_rasterization.Workspace(P, RES, RES, n, cap, device)
    R_total = sum(c[1] for c in counts)

    # final exchange of the path: the rendered frames go to rank 0 as 8-bit RGB, which is what the reference turns every
    # frame into before writing the video (visualize.py:416); 120 x 3 x 256 x 256 B = 23.6 MB per rank and step. The
    # gather is asynchronous on RCCL's stream and overlaps the next step's rendering; all of them are waited for inside
    # the timed region.
    gather_buf = None
    pending = []
    if world > 1:
        gather_buf = [torch.empty((V, RES, RES, 3), dtype=torch.uint8, device=comm_device) for _ in range(world)] if rank == 0 else None

    def step():
        if streams:
            main = torch.cuda.current_stream()
            for i, (a, b) in enumerate(chunks):
                st = streams[i % len(streams)]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    render_chunk(a, b, check=False)
            for st in streams:
                main.wait_stream(st)
        else:
            for a, b in chunks:
                render_chunk(a, b, check=False)
        if world > 1:     # final gather of the RGB frames (the only exchange of the path)
            frames = f3d.gaussian_renderer.pack_frames(out).to(comm_device)          # uint8 [V,H,W,3], one kernel
            work = dist.gather(frames, gather_buf, dst=0, async_op=True)
            pending.append((work, frames))
            while len(pending) > 1:           # at most one gather in flight behind the current step
                pending.pop(0)[0].wait()

    def barrier():
        while pending:
            pending.pop(0)[0].wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    L.f3dg_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    L.f3dg_profile_enable(0)
    stage_ms = (C.c_double * 3)()
    ncalls = C.c_int(0)
    _lib.check(L.f3dg_profile_collect(stage_ms, C.byref(ncalls)), "f3dg_profile_collect")
    for n, ws in workspaces.items():      # no overflow happened in the timed region
        f3d.diff_gof_rasterization.read_status(ws)

    t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device if world > 1 else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    copy_gbs = measured_copy_bandwidth(device) if rank == 0 else None

    if rank == 0:
        T = ((RES + 15) // 16) ** 2
        launches = max(int(ncalls.value), 1)
        # algorithmic bytes of the compositing kernel per launch: 72*R + 36*W*H (inference mode: the
        # aux planes final_T / n_contrib are not written) + 8*T, summed over the views of the launch
        bytes_per_step = 72.0 * R_total + (36.0 * RES * RES + 8.0 * T) * V
        render_ms_per_launch = stage_ms[2] / launches
        bytes_per_launch = bytes_per_step / len(chunks)
        achieved = bytes_per_launch / (render_ms_per_launch * 1e-3) / 1e9 if render_ms_per_launch > 0 else 0.0
        result = {
            "metric": "rendered views/sec at 256x256 (N Gaussians, K cams)",
            "value": world * V * args.steps / elapsed,
            "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (+f64 islands, as the reference)", "data": "synthetic",
            "config": {"workload": "C2: 1 image/GPU, %d Gaussians (sigma0=%g), %d-view orbit @%dx%d, GOF forward raster only"
                       % (P, args.sigma0, V, RES, RES), "gaussians": P, "views": V, "resolution": RES,
                       "instances_per_step": R_total, "views_per_call": args.views_per_call,
                       "parallelism": "image-sharded x%d + RCCL gather" % world if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "render_fwd_kernel<SAVE_AUX=false, PRETEST, CULL, QUEUE>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(P, V, RES, args.views_per_call),
                         "peak_measured_copy": copy_gbs,     # device-to-device copy on THIS box, read + write bytes
                         "valu": measured_valu(P, V, RES, args.views_per_call),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "ms_per_launch": render_ms_per_launch,
                         "stage_ms_per_step": {"preprocess": stage_ms[0] / args.steps, "binning": stage_ms[1] / args.steps,
                                               "compositing": stage_ms[2] / args.steps}},
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(g, cams, shs, P, RES, args.cpu_sample_views)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_copy_bandwidth(device, nbytes=1 << 30, reps=5):
    """GB/s (read + write) of a plain device-to-device copy of 1 GiB on this box: the practical HBM ceiling next to the
    8 TB/s nominal peak the roofline fraction is quoted against."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    a.zero_()
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measured_valu(P, V, RES, views_per_call):
    """Vector-ALU figures of the compositing kernel from the committed SQ counter passes (profiles/r01_final/traffic.json,
    same configuration only): busy = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x kernel cycles), lanes = active lanes per VALU
    instruction / 64. The kernel is VALU-bound (SURVEY 8d asks for this next to the HBM fraction)."""
    path = os.path.join(ROOT, "profiles", "r01_final", "traffic.json")
    try:
        t = json.load(open(path))
        c = t["config"]
        if (c["gaussians"], c["views"], c["resolution"], c["views_per_call"]) == (P, V, RES, views_per_call):
            return t.get("valu")
    except Exception:
        pass
    return None


def measured_traffic(P, V, RES, views_per_call):
    """HBM bytes per compositing launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    runs of this same command; gfx950 FETCH_SIZE correction applied) -- only when they were taken on this exact
    configuration, else null. PMC counters cannot be collected from inside the process."""
    path = os.path.join(ROOT, "profiles", "r01_final", "traffic.json")
    try:
        t = json.load(open(path))
        c = t["config"]
        if (c["gaussians"], c["views"], c["resolution"], c["views_per_call"]) == (P, V, RES, views_per_call):
            return t["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(g, cams, shs, P, RES, n_sample):
    """The CPU oracle (plain-C restatement, OpenMP tile-parallel; kind 'port') timed on a bounded sample: the first
    n_sample views of the same orbit over the same Gaussians, on this box's host cores."""
    from oracle import gof
    npy = lambda t: t.detach().cpu().numpy()
    a = dict(means3D=npy(g["xyz"]), opacities=npy(g["opacity"]), scales=npy(g["scaling"]),
             rotations=npy(g["rotation"]), shs=npy(shs))
    vm, pm, cp = npy(cams["viewmatrix"]), npy(cams["projmatrix"]), npy(cams["campos"])
    o = gof.Oracle()
    cores = os.cpu_count() or 1
    V = vm.shape[0]
    idx = [int(round(i * (V - 1) / max(n_sample - 1, 1))) for i in range(n_sample)]
    t0 = time.perf_counter()
    done = 0
    for v in idx:
        o.forward(viewmatrix=vm[v], projmatrix=pm[v], campos=cp[v], tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"],
                  W=RES, H=RES, bg=[0, 0, 0], sh_degree=1, **a)
        done += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "views/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d orbit views of the same %d Gaussians @%dx%d, oracle/gof_oracle.c with OpenMP on %d threads"
                      % (done, V, P, RES, RES, cores)}


if __name__ == "__main__":
    main()
