#!/usr/bin/env python
"""bench.py -- headline benchmark of the GOF rasterization hot path on MI355X.

Metric (BASELINE.json): rendered views/s at 256x256 for (N Gaussians, K cameras), plus the achieved bytes/s of the
per-tile compositing kernel against the gfx950 HBM roofline.

Workloads (`--workload`):
  c2 (default; the configuration the metric is quoted on): one image's 196,608 Gaussians ("~200k" = 3 cycle views x 65,536)
      rendered along a 120-view orbit at 256x256, forward only, all 120 views in one launch sequence. A "step" = those 120 views.
      With --gpus N every rank renders its own image (weak scaling: the batch of input images shards embarrassingly)
      and the RGB frames are gathered to rank 0 over RCCL inside the timed region.
  c5 (BASELINE config C5): 1,000,000 Gaussians, 32 views @512x512, SAVE_AUX forward + backward (random dL/dpix on channels 0-6
      and 8) per step; adds the roofline record of the compositing backward (80 R + 60 W H + 68 C bytes, C counted by the kernel).
  dropin: the reference's OWN call pattern (visualize.py:293-314, 387-416): `render_predicted_more_v2_gof(pc, bb, wv[th:th+1], ...)`, one
      view per rasterizer call, through this build's drop-in wrapper (the small-call path of the library: three launches per call).
      --gaussians defaults to 65,536 here (one predicted view's set). A "step" = --views such calls.
  c4 (BASELINE config C4's shape per rank; C3 at N = 1): --images B input images per rank @256x256 through predictor (random
      weights) + cycle aggregation (8 novel views of all B images in one launch sequence, 8 re-predictions, in-place merge)
      + the 8 orbit views of every merged set + frame packing + the gather. A "step" = B x 8 final views.

Every number of the JSON line is measured in THIS run except the `*_from_profiles` objects, which are read from the committed
rocprofv3 PMC passes of the same command (profiles/<round>/traffic.json) and say so. Hardware counters cannot be read from inside
the process: at N = 1 the C2 line's `roofline.traffic` comes from two child runs of the same workload under `rocprofv3 --pmc`
(FETCH_SIZE, WRITE_SIZE; `roofline.traffic_live`; --no-pmc or F3DG_BENCH_PMC=0 skips them, a failure leaves null and says why). C2 at N = 1 runs three timed loops of K steps each: frames left in HBM (`value_in_hbm`), frames packed to 8-bit RGB
and copied to pinned host memory behind the next step's rendering (`value`: SURVEY 8d "views/s ... including the final D2H of RGB"),
and the in-HBM loop again in the reference's float32/float64 arithmetic (`value_exact`, `roofline_exact`). `roofline` is the
compositing kernel, timed with HIP events the library records on the launch stream, on the list entries the launch is handed
(`frac_on_reference_instances`: the same formula on the reference's instance count); `rooflines_other` prices the projection and
binning stages with this build's own byte streams; `cpu_baseline` times the CPU oracle (the build's plain-C restatement: kind
"port") on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with hipIpcGetMemHandle otherwise); the variable is
# exported on the boxes already -- kept here for launches whose environment was rebuilt
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c2", "c4", "c5", "dropin"], default="c2")
    ap.add_argument("--gaussians", type=int, default=196608)
    ap.add_argument("--views", type=int, default=120)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--sigma0", type=float, default=0.01)
    ap.add_argument("--images", type=int, default=16, help="c4: input images per rank (BASELINE C4 = 64 per rank at 8 GPUs)")
    ap.add_argument("--views-per-call", type=int, default=int(os.environ.get("F3DG_VIEWS_PER_CALL", "0")),
                    help="views per rasterizer call; 0 (default): equal parts of at most 128 views (C2's 120 views are one call)")
    ap.add_argument("--render-mode", choices=["fast", "exact"], default=os.environ.get("F3DG_RENDER_MODE", "fast"),
                    help="compositing arithmetic: fast = error-free float32 pairs for the float64 island (default, parity-gated "
                         "at 1e-4 / 99.9 %% / 80 dB), exact = the reference's float32/float64 operation order")
    ap.add_argument("--backbone", choices=["fp32", "bf16", "fp16"], default="fp32",
                    help="c4: precision of the SongUNet backbone (bf16 = the opt-in autocast option, SURVEY 8f-3)")
    ap.add_argument("--backbone-chunk", type=int, default=-1,
                    help="c4: images per backbone pass (cfg['model']['backbone_chunk']; -1 = the predictor's default: 8 for fp32 -- bounds MIOpen's "
                         "first-use find at 43 s --, 0 = whole batches for the 16-bit options)")
    ap.add_argument("--backbone-layout", choices=["auto", "nchw", "nhwc"], default="auto",
                    help="c4: memory layout of the SongUNet backbone (auto, the default: channels-last for passes of two images or more; nhwc: "
                         "channels-last activations and filters -- MIOpen's NHWC kernels + the channels-last GroupNorm+SiLU / residual-join "
                         "kernels; nchw: torch's layout)")
    ap.add_argument("--tile-cull", type=int, choices=[0, 1], default=1,
                    help="1 (library default): a Gaussian is instantiated only in the tiles its alpha >= 1/255 ellipse reaches; "
                         "0: the reference's tile lists (every tile of the 3-sigma square)")
    ap.add_argument("--data", choices=["synthetic", "real"], default="synthetic",
                    help="c2: synthetic = the recipe of SURVEY 8d; real = the image of fixture F6 (tests/golden/real_image_256.npz: "
                         "images/1/n01644373_4548.jpg + its LeReS depth) through the build's own predictor and cycle aggregation "
                         "(formula weights, no checkpoint travels): the 589,824 merged Gaussians along the 128-view orbit")
    ap.add_argument("--channels", choices=["all", "rgb_depth_alpha"], default="all",
                    help="c2: output channels of the rasterizer calls; rgb_depth_alpha = what the reference's loops consume (visualize.py:304-306, "
                         "400-402) and the build's cycle / orbit loops ask for (F3DG_FLAG_SKIP_NORMAL | _SKIP_DISTORTION); the headline stays 9-channel")
    ap.add_argument("--scan", type=int, choices=[0, 1], default=int(os.environ.get("F3DG_SCAN", "0")),
                    help="1: the calls carry F3DG_FLAG_SCAN -- the split-pixel compositing schedule (render5_fwd_kernel), its own 1e-4-gated mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) that fill roofline.traffic at N = 1 (also: F3DG_BENCH_PMC=0)")
    ap.add_argument("--no-d2h", action="store_true", help="skip the timed loops with frame packing + device-to-host copy: `value` is then the in-HBM rate")
    ap.add_argument("--d2h-issue", choices=["thread", "main"], default=os.environ.get("F3DG_D2H_ISSUE", "main"),
                    help="who issues the pack + device-to-host copy of a finished step on the side stream: the step's own host thread "
                         "(three asynchronous calls behind an event) or a second host thread")
    ap.add_argument("--d2h-path", choices=["direct", "copy"], default=os.environ.get("F3DG_D2H_PATH", "copy"),
                    help="how a finished step's 8-bit frames reach pinned host memory: copy (default) = f3dg_pack_frames into HBM + a "
                         "device-to-host copy (HIP runs it as a whole-chip shader copy, __amd_rocclr_copyBuffer); direct = "
                         "f3dg_pack_frames_host writes them there (one kernel of a few workgroups; measured 1 % slower, notes/r06.md section 3)")
    ap.add_argument("--d2h-workgroups", type=int, default=int(os.environ.get("F3DG_D2H_WG", "0")), help="direct path: workgroups of the pack kernel (0: 64)")
    ap.add_argument("--no-exact", action="store_true", help="skip the extra timed loop in the reference's arithmetic (value_exact / roofline_exact)")
    ap.add_argument("--cpu-sample-views", type=int, default=12)
    return ap.parse_args()


def setup_dist():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the product path)"
    # F3DG_DIST_BACKEND=gloo is a functional check of the N>1 logic on a box with fewer GPUs than ranks (ranks share
    # devices, frames are gathered through host memory); the measured configuration is always nccl = RCCL.
    backend = os.environ.get("F3DG_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    return rank, world, dist, device, (device if backend == "nccl" else torch.device("cpu"))


class Gatherer:
    """The only exchange of the path: the rendered frames go to rank 0 as 8-bit RGB (what the reference turns every frame into
    before writing its video, visualize.py:416). Asynchronous on RCCL's stream, at most one gather in flight behind the current
    step; all of them are waited for inside the timed region."""

    def __init__(self, dist, world, rank, shape, comm_device):
        self.dist, self.world, self.comm_device, self.pending = dist, world, comm_device, []
        self.buf = [torch.empty(shape, dtype=torch.uint8, device=comm_device) for _ in range(world)] if (world > 1 and rank == 0) else None

    def submit(self, frames_u8):
        if self.world == 1:
            return
        frames = frames_u8.to(self.comm_device)
        # the previous step's gather has had this whole step's rendering to finish; it is waited for BEFORE the next one is issued, because
        # both write rank 0's buffers: RCCL runs a communicator's collectives in issue order, gloo's worker threads need not (seen once as
        # a stale frame in tests/test_dist_cpu.py::test_c4_step_gather_gloo_world2)
        while self.pending:
            self.pending.pop(0)[0].wait()
        self.pending.append((self.dist.gather(frames, self.buf, dst=0, async_op=True), frames))

    def barrier(self):
        while self.pending:
            self.pending.pop(0)[0].wait()
        if self.world > 1:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


def timed(step, barrier, warmup, steps):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(elapsed, dist, world, device):
    if world == 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    args = parse()
    rank, world, dist, device, comm_device = setup_dist()
    import f3dgaus_amd as f3d
    from f3dgaus_amd import _lib
    L = _lib.lib()
    _lib.check(L.f3dg_set_option(b"render_fast", 1 if args.render_mode == "fast" else 0), "f3dg_set_option")
    if os.environ.get("F3DG_DEBUG_SKIP_ALL"):    # experiment: no Gaussian ever passes -> the compositing kernel only stages
        _lib.check(L.f3dg_set_option(b"debug_skip_all", 1), "f3dg_set_option")
    if os.environ.get("F3DG_RENDER_ROUND"):       # A/B: list entries staged per round by render2 (256 / 192 / 128)
        _lib.check(L.f3dg_set_option(b"render_round", int(os.environ["F3DG_RENDER_ROUND"])), "f3dg_set_option")
    if os.environ.get("F3DG_RENDER_KERNEL"):      # A/B of the compositing kernel generations (default: the library's)
        _lib.check(L.f3dg_set_option(b"render_kernel", int(os.environ["F3DG_RENDER_KERNEL"])), "f3dg_set_option")
    for env, opt in (("F3DG_RENDER_DMA", b"render_dma"), ("F3DG_RENDER_LDS_PAD", b"render_lds_pad"), ("F3DG_BWD_OCC", b"bwd_occ"), ("F3DG_RENDER_SLIDE", b"render_slide"), ("F3DG_RENDER_LOWOCC", b"render_lowocc"), ("F3DG_RENDER_TAIL", b"render_tail"), ("F3DG_SMALL_DEBUG", b"small_debug"), ("F3DG_RENDER_PACK_TH", b"render_pack_th"), ("F3DG_RENDER_PACK", b"render_pack"), ("F3DG_PRE_HOIST", b"pre_hoist"), ("F3DG_RENDER_WPB", b"render_wpb")):     # A/B switches of render3
        if os.environ.get(env):
            _lib.check(L.f3dg_set_option(opt, int(os.environ[env])), "f3dg_set_option")
    for item in filter(None, os.environ.get("F3DG_OPTIONS", "").split(",")):     # "name=value,...": any library option, for A/B runs
        _lib.check(L.f3dg_set_option(item.split("=")[0].strip().encode(), int(item.split("=")[1])), "f3dg_set_option")
    _lib.check(L.f3dg_set_option(b"tile_cull", args.tile_cull), "f3dg_set_option")
    result = {"c2": run_c2, "c4": run_c4, "c5": run_c5, "dropin": run_dropin}[args.workload](args, rank, world, dist, device, comm_device, f3d, L)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------- C2
def run_c2(args, rank, world, dist, device, comm_device, f3d, L):
    from f3dgaus_amd import _lib, synthetic
    P, V, RES = args.gaussians, args.views, args.res
    if args.data == "real":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from real_data import real_merged_set
        assert RES == 256, "--data real is the 256 x 256 image"
        g = real_merged_set(device)                # predict -> 8 novel views -> 8 re-predictions -> merge (untimed set-up)
        P = g["xyz"].shape[0]
        if V == 120:
            V = args.views = 128                   # the reference's final orbit (visualize.py:343-416)
        torch.cuda.synchronize()
    else:
        g = synthetic.make_gaussians(P, s0=args.sigma0, seed=rank, device=device)     # every rank = a different image
    cams = synthetic.orbit_cameras(V, resolution=RES, device=device)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    bg = torch.zeros(3, device=device)
    out = torch.empty((V, 9, RES, RES), dtype=torch.float32, device=device)
    radii = torch.empty((V, P), dtype=torch.int32, device=device)
    if args.views_per_call <= 0:        # equal parts of at most 128 views (a 120 + 8 split of 128 views costs 4 %: measured)
        parts = (V + 127) // 128
        args.views_per_call = (V + parts - 1) // parts
    chunks = [(a, min(a + args.views_per_call, V)) for a in range(0, V, args.views_per_call)]
    workspaces = {}

    call_opts = {"exact": None, "tile_cull": None, "channels": args.channels, "scan": bool(args.scan)}     # per-call settings (F3DG_FLAG_EXACT / _NO_TILE_CULL): None = the process default

    def render_chunk(a, b, check, out=out):
        o, r, ws = f3d.rasterize_views(
            g["xyz"], g["opacity"], cams["viewmatrix"][a:b], cams["projmatrix"][a:b], cams["campos"][a:b], bg,
            image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs,
            scales=g["scaling"], rotations=g["rotation"], sh_degree=1, workspace=workspaces.get(b - a), out=out[a:b],
            radii=radii[a:b], save_aux=False, check=check, **call_opts)
        return ws

    # calibration passes (untimed). First with the reference's tile lists: R_total = the reference's num_rendered, the unit count of
    # SURVEY 8d's byte formulas (the workload's size, whatever the implementation then skips). Then in the benched configuration:
    # sizes every chunk's workspace (capacity = max over chunks, +25 %) and counts the instances the launches really process.
    call_opts["tile_cull"] = False
    R_total = sum(render_chunk(a, b, check=True).num_rendered for a, b in chunks)
    call_opts["tile_cull"] = None
    counts = []
    for a, b in chunks:
        ws = render_chunk(a, b, check=True)
        counts.append((b - a, ws.num_rendered))
    for n in set(c[0] for c in counts):
        cap = int(max(c[1] for c in counts if c[0] == n) * 1.25) + 4096
        workspaces[n] = f3d.diff_gof_rasterization.Workspace(P, RES, RES, n, cap, device)
    R_proc = sum(c[1] for c in counts)
    # length of the (view, tile) lists the compositing kernel is handed (last chunk's call: the ranges are still in its workspace)
    a_, b_ = chunks[-1]
    T_ = ((RES + 15) // 16) ** 2
    rng_ = torch.zeros((b_ - a_) * T_ * 2, dtype=torch.int32, device=device)
    _lib.check(L.f3dg_debug_export(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), P, RES, RES, b_ - a_,
                                   ws.max_rendered, None, None, None, None, None, None, None, None, C.c_void_p(rng_.data_ptr()), None, None, None),
               "f3dg_debug_export")
    rr_ = rng_.reshape(-1, 2).cpu()
    lens_ = (rr_[:, 1] - rr_[:, 0]).float()
    list_stats = {"tile_list_mean": float(lens_.mean()), "tile_list_max": int(lens_.max()), "instances_per_view_and_gaussian": R_proc / float(V * P),
                  "reference_instances_per_view_and_gaussian (R/P per view)": R_total / float(V * P)}

    gat = Gatherer(dist, world, rank, (V, RES, RES, 3), comm_device)

    def step():
        for a, b in chunks:
            render_chunk(a, b, check=False)
        if world > 1:
            gat.submit(f3d.gaussian_renderer.pack_frames(out))          # uint8 [V,H,W,3], one kernel

    def collect(steps):
        """per-stage HIP-event milliseconds of the calls recorded since profiling was enabled: (sums, calls, per-call rows [calls][3])"""
        cap = steps * len(chunks) + 8
        st = (C.c_double * 5)()
        nc = C.c_int(0)
        per = (C.c_double * (3 * cap))()
        _lib.check(L.f3dg_profile_collect_calls(st, C.byref(nc), per, cap), "f3dg_profile_collect_calls")
        n = min(int(nc.value), cap)
        return [st[i] for i in range(5)], max(int(nc.value), 1), [[per[3 * k + i] for i in range(3)] for k in range(n)]

    def measure(fn, warmup, steps):
        """(seconds for `steps` steps, per-stage HIP-event milliseconds summed over them, forward calls, kernel launches, per-call rows)"""
        timed(fn, gat.barrier, warmup, 0)
        L.f3dg_profile_enable(1)
        L.f3dg_debug_launch_count(1)
        t = timed(fn, gat.barrier, 0, steps)
        launches = int(L.f3dg_debug_launch_count(1))
        L.f3dg_profile_enable(0)
        st, nc, rows = collect(steps)
        for ws in workspaces.values():      # no overflow happened in the timed region
            f3d.diff_gof_rasterization.read_status(ws)
        return t, st, nc, launches, rows

    # (1) frames left in HBM (N > 1: + the RCCL gather of the packed frames)
    elapsed_hbm, stage_ms, ncalls, nlaunch, rows = measure(step, args.warmup, args.steps)
    kernel_name = L.f3dg_debug_last_render_kernel().decode()           # what the library launched, with its template arguments
    elapsed_hbm = max_over_ranks(elapsed_hbm, dist, world, comm_device if world > 1 else device)

    # (2) N = 1, SURVEY 8d "views/s = views / wall time including the final D2H of RGB only": the frames leave the GPU as 8-bit RGB
    # (what the reference turns every frame into before it writes the video, visualize.py:407,416). Two output buffers: while
    # step i + 1 renders, a side stream packs step i's frames (f3dg_pack_frames) and copies them to pinned host memory; everything
    # is waited for inside the timed region. Also timed: the float32 RGB planes, un-pipelined (the PCIe-heavy variant).
    elapsed, d2h = elapsed_hbm, None
    if world == 1 and not args.no_d2h:
        import queue
        import threading
        side = torch.cuda.Stream(device=device)       # (stream priorities -1 / 0 on either side: no difference, notes/r06.md section 3)
        outs = [out, torch.empty_like(out)]
        packed = [torch.empty((V, RES, RES, 3), dtype=torch.uint8, device=device) for _ in range(2)]
        host = [torch.empty((V, RES, RES, 3), dtype=torch.uint8) for _ in range(2)]
        if not os.environ.get("F3DG_BENCH_PAGEABLE"):          # (test switch: a host buffer that makes every copy block its caller)
            host = [h.pin_memory() for h in host]
        rendered = [torch.cuda.Event() for _ in range(2)]
        copied = [torch.cuda.Event() for _ in range(2)]
        issued = [threading.Event() for _ in range(2)]          # the copy thread has recorded copied[k]
        state = {"i": 0}
        jobs = queue.Queue()

        # The pack + copy of a finished step is issued by a second host thread: where a device-to-host copy blocks its caller (a box
        # whose pinned allocation or copy engine misbehaves: seen once, 20 k instead of 29 k views/s) it blocks that thread, not the one
        # that issues the next step's kernels.
        copier_error = []

        def pack_and_copy(k):           # (on the side stream) frames of buffer k -> host[k]
            if args.d2h_path == "direct" and host[k].is_pinned():
                f3d.gaussian_renderer.pack_frames(outs[k], out=host[k], max_workgroups=args.d2h_workgroups)
            else:
                f3d.gaussian_renderer.pack_frames(outs[k], out=packed[k])
                host[k].copy_(packed[k], non_blocking=True)

        def copier():
            torch.cuda.set_device(device)
            while True:
                k = jobs.get()
                if k is None:
                    return
                try:
                    with torch.cuda.stream(side):
                        side.wait_event(rendered[k])
                        pack_and_copy(k)
                        copied[k].record()
                except Exception as ex:          # the main thread must not wait for ever on issued[k]
                    copier_error.append(ex)
                finally:
                    issued[k].set()

        def wait_issued(ev):
            if not ev.wait(timeout=120.0):
                raise RuntimeError("bench: the frame-copy thread did not answer within 120 s")
            if copier_error:
                raise RuntimeError("bench: the frame-copy thread failed") from copier_error[0]

        worker = threading.Thread(target=copier, daemon=True)
        worker.start()

        def step_d2h():
            k = state["i"] & 1
            state["i"] += 1
            if args.d2h_issue == "main":
                torch.cuda.current_stream().wait_event(copied[k])    # the side stream has read buffer k (two steps ago)
                for a, b in chunks:
                    render_chunk(a, b, check=False, out=outs[k])
                rendered[k].record()
                with torch.cuda.stream(side):
                    side.wait_event(rendered[k])
                    pack_and_copy(k)
                    copied[k].record()
                return
            wait_issued(issued[k])
            issued[k].clear()
            torch.cuda.current_stream().wait_event(copied[k])        # the side stream has read buffer k (two steps ago)
            for a, b in chunks:
                render_chunk(a, b, check=False, out=outs[k])
            rendered[k].record()
            jobs.put(k)

        def d2h_barrier():
            for ev in issued:                                        # every copy has been issued ...
                wait_issued(ev)
            gat.barrier()                                            # ... and (device synchronisation) has arrived

        for k in range(2):
            copied[k].record()
            issued[k].set()
        timed(step_d2h, d2h_barrier, 2, 0)
        L.f3dg_profile_enable(1)
        L.f3dg_debug_launch_count(1)
        elapsed = timed(step_d2h, d2h_barrier, 0, args.steps)
        nlaunch = int(L.f3dg_debug_launch_count(1)) - args.steps       # (the pack kernel of every step is not the call's)
        L.f3dg_profile_enable(0)
        stage_ms, ncalls, rows = collect(args.steps)
        for ws in workspaces.values():      # no overflow happened in the timed region
            f3d.diff_gof_rasterization.read_status(ws)
        jobs.put(None)
        worker.join(timeout=10)

        # the pack + copy leg alone, nothing else on the device (what has to hide behind the next step)
        leg = []
        for _ in range(3):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(side):
                e0.record()
                f3d.gaussian_renderer.pack_frames(outs[0], out=packed[0])
                e1.record()
                host[0].copy_(packed[0], non_blocking=True)
                e2.record()
            e2.synchronize()
            leg.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        leg_pack_ms, leg_copy_ms = min(x[0] for x in leg), min(x[1] for x in leg)
        leg_direct_ms = None
        if host[0].is_pinned():
            legd = []
            for _ in range(3):
                e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                with torch.cuda.stream(side):
                    e0.record()
                    f3d.gaussian_renderer.pack_frames(outs[0], out=host[0], max_workgroups=args.d2h_workgroups)
                    e1.record()
                e1.synchronize()
                legd.append(e0.elapsed_time(e1))
            leg_direct_ms = min(legd)

        host_f32 = torch.empty((V, 3, RES, RES), dtype=torch.float32).pin_memory()

        def step_f32():
            step()
            host_f32.copy_(out[:, :3], non_blocking=True)

        e32 = timed(step_f32, gat.barrier, 1, args.steps)
        d2h = {"uint8_rgb": {"value": V * args.steps / elapsed, "unit": "views/s", "ms_per_step": 1e3 * elapsed / args.steps,
                             "bytes_per_step": V * RES * RES * 3, "pipelined": True, "issued_by": args.d2h_issue, "path": args.d2h_path,
                             "leg_alone_ms": {"pack_frames_host (direct path: one kernel writes pinned host memory)": leg_direct_ms,
                                              "direct_GBps": V * RES * RES * 3 / (leg_direct_ms * 1e-3) / 1e9 if leg_direct_ms else None,
                                              "pack_frames": leg_pack_ms, "copy_to_pinned": leg_copy_ms,
                                              "copy_GBps": V * RES * RES * 3 / (leg_copy_ms * 1e-3) / 1e9 if leg_copy_ms > 0 else None}},
               "float32_rgb": {"value": V * args.steps / e32, "unit": "views/s", "ms_per_step": 1e3 * e32 / args.steps,
                               "bytes_per_step": V * RES * RES * 12, "pipelined": False},
               "note": "uint8: the frames of a finished step go to pinned host memory on a side stream, double-buffered behind the next "
                       "step's rendering, all waited for inside the timed region (= `value`) -- path direct: f3dg_pack_frames_host writes "
                       "them there (one kernel, a few workgroups); path copy: f3dg_pack_frames + a device-to-host copy, which HIP runs as a "
                       "whole-chip shader copy; float32: the three RGB planes copied after every step on the same stream"}

    # (3) the same build in the reference's own arithmetic (float32 with the float64 island, forward.cu:511-579), frames left in HBM
    exact = None
    if args.render_mode == "fast" and not args.no_exact:
        call_opts["exact"] = True
        e_x, st_x, nc_x, _, rows_x = measure(step, 1, args.steps)
        kernel_name_x = L.f3dg_debug_last_render_kernel().decode()
        call_opts["exact"] = None
        e_x = max_over_ranks(e_x, dist, world, comm_device if world > 1 else device)
        exact = (e_x, st_x, nc_x, rows_x, kernel_name_x)

    # (4) what the compositing kernel did, counted by its counting variant in ONE untimed call sequence (option render_count): the
    # list entries it staged (a quadrant stops reading its tile's list once its 64 pixels are saturated), its phase-2 trips and lanes
    counts = None
    if True:        # (every rank: at N > 1 a step ends in the collective gather)
        _lib.check(L.f3dg_set_option(b"render_count", 1), "f3dg_set_option")
        cbuf = (C.c_ulonglong * 16)()
        L.f3dg_debug_render_counts(cbuf, 1)
        L.f3dg_debug_render4_counts(None, 1)
        L.f3dg_debug_render5_counts(None, 1)
        step()
        gat.barrier()
        _lib.check(L.f3dg_debug_render_counts(cbuf, 1), "f3dg_debug_render_counts")
        _lib.check(L.f3dg_set_option(b"render_count", 0), "f3dg_set_option")
        c4 = (C.c_ulonglong * 16)()
        _lib.check(L.f3dg_debug_render4_counts(c4, 1), "f3dg_debug_render4_counts")
        c5 = (C.c_ulonglong * 16)()
        _lib.check(L.f3dg_debug_render5_counts(c5, 1), "f3dg_debug_render5_counts")
        if c5[5]:       # the split-pixel kernel (F3DG_FLAG_SCAN) ran: its own counters
            pairs = int(c5[4]) + int(c5[7])
            counts = {"list_entries_staged": int(c5[0]), "list_entries_scanned": int(c5[1]), "fused_trips": int(c5[2]), "slides": int(c5[3]),
                      "fused_lane_trips": int(c5[4]), "waves": int(c5[5]), "dense_batches": int(c5[6]), "pairs_in_dense_batches": int(c5[7]),
                      "pixels_compacted": int(c5[8]), "slides_with_a_compaction": int(c5[9]), "phase2_lane_trips": pairs,
                      "fused_trip_lane_utilisation": int(c5[4]) / (64.0 * int(c5[2])) if c5[2] else None,
                      "dense_batch_lane_utilisation": int(c5[7]) / (64.0 * int(c5[6])) if c5[6] else None,
                      "note": "one untimed step with option render_count = 1 (render5_fwd_kernel with work counters); a fused trip runs a pair's whole "
                              "arithmetic in the pixel's lane, a dense batch = 64 (pixel, entry) pairs, one per lane, combined by segmented scans"}
        elif c4[5]:       # the rank-packed kernel (render_kernel = 4) ran: its own counters
            pairs = int(c4[4]) + int(c4[9])
            counts = {"list_entries_staged": int(c4[0]), "list_entries_scanned": int(c4[1]), "fused_trips": int(c4[2]), "slides": int(c4[3]),
                      "fused_lane_trips": int(c4[4]), "waves": int(c4[5]), "packed_batches": int(c4[6]), "blend_trips": int(c4[7]),
                      "pairs_evaluated_in_dense_trips": int(c4[8]), "pairs_reaching_a_blend_trip": int(c4[9]),
                      "phase2_lane_trips": pairs,
                      # lanes doing useful work in the trips that carry the stateless two thirds of a pair's arithmetic
                      "stateless_lane_utilisation": (int(c4[4]) + int(c4[8])) / (64.0 * (int(c4[2]) + int(c4[6]))) if c4[2] + c4[6] else None,
                      "blend_trip_lane_utilisation": int(c4[9]) / (64.0 * int(c4[7])) if c4[7] else None,
                      "note": "one untimed step with option render_count = 1 (render4_fwd_kernel with work counters); a fused trip runs a pair's "
                              "whole arithmetic in the pixel's lane, a packed batch = one dense trip (stateless part, one pair per lane) + its blend trips"}
        elif cbuf[5]:
            counts = {"list_entries_staged": int(cbuf[0]), "list_entries_scanned": int(cbuf[1]), "phase2_wave_trips": int(cbuf[2]),
                      "slides": int(cbuf[3]), "phase2_lane_trips": int(cbuf[4]), "waves": int(cbuf[5]),
                      "phase2_lane_utilisation": cbuf[4] / (64.0 * cbuf[2]) if cbuf[2] else None,
                      "phase2_trips_of_slides_with_at_most_8_live_pixels": int(cbuf[6]), "..._at_most_24": int(cbuf[7]),
                      "slides_with_at_most_8_live_pixels": int(cbuf[8]), "slides_with_at_most_24": int(cbuf[9]),
                      "tail_steps": int(cbuf[10]), "tail_wave_trips": int(cbuf[11]), "tail_entries_tested": int(cbuf[12]),
                      "staged_entries_reaching_a_live_pixel": int(cbuf[13]),
                      "two_pixels_per_lane_emulation": {"trips_two_32_lane_halves_walked_separately": int(cbuf[14]),
                                                        "trips_one_32_lane_walk_with_pixels_i_and_i_plus_32_per_lane": int(cbuf[15]),
                                                        "ratio": (cbuf[15] / cbuf[14]) if cbuf[14] else None},
                      "note": "one untimed step with option render_count = 1 (the same kernel with work counters); staged entries count "
                              "a list entry once per quadrant wave that gathers its record"}

    if rank != 0:
        return None
    copy_gbs = measured_copy_bandwidth(device)
    T = ((RES + 15) // 16) ** 2
    per = lambda ms, n=None: ms / (n or ncalls)
    nl = len(chunks)
    # ALGORITHMIC bytes per launch, summed over the views of a launch. The unit count is what the launch is handed
    # (R_proc: the culled lists); the same formula on the reference's num_rendered (R_total) is the secondary figure.
    b_render = (72.0 * R_proc + (36.0 * RES * RES + 8.0 * T) * V) / nl      # inference mode: the aux planes are not written
    b_render_ref = (72.0 * R_total + (36.0 * RES * RES + 8.0 * T) * V) / nl
    # this build's own streams: projection reads the 92 B of a Gaussian once per call and writes 96 B per
    # (view, Gaussian) (record 64, ellipse 16, radius 4, rectangle 8, sort key 4); binning moves 108 B per (view, Gaussian) (three
    # depth-sort passes 56, rectangle gather 24, prefix sum 12, instance generation 16) + 18 B per instance (generation 6, tile pass
    # histogram 2 + scatter 10) + the ranges
    b_pre = (92.0 * P + 96.0 * P * V) / nl
    b_bin = (108.0 * P * V + 18.0 * R_proc + 8.0 * T * V) / nl
    gbs = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    prof = profiles_record(P, V, RES, args.views_per_call, args.render_mode, args.tile_cull, args.sigma0)
    import statistics

    def spread(rws, col):
        v = sorted(r[col] for r in rws) if col is not None else sorted(sum(r) for r in rws)
        return {"min": v[0], "median": statistics.median(v), "max": v[-1], "n": len(v)} if v else None

    def roofline(stage, kernel, n, rws):
        ms = per(stage[2], n)
        return {"bound": "hbm", "kernel": kernel,
                "achieved": gbs(b_render, ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(b_render, ms) / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": b_render, "ms_per_launch": ms, "ms_per_launch_spread": spread(rws, 2),
                "units": "72 B x instances_processed_per_step (the list entries the launch is handed) + 36 B x pixels + 8 B x tiles",
                "frac_on_reference_instances": gbs(b_render_ref, ms) / HBM_PEAK_GBS,
                "units_reference": "the same formula on instances_per_step = the reference's num_rendered for this input (tile_cull 0)"}

    rf = roofline(stage_ms, kernel_name, ncalls, rows)
    if counts:
        # the same formula on what the kernel READ: 4 B per scanned list entry + 80 B (record + ellipse) per staged one + the
        # pixels and ranges. Where saturated quadrants stop long before their list ends (large splats) this, not the algorithmic
        # figure, is what the launch moved -- and it cannot exceed 1
        b_staged = (4.0 * counts["list_entries_scanned"] + 80.0 * counts["list_entries_staged"] + (36.0 * RES * RES + 8.0 * T) * V) / nl
        rf["frac_on_staged_entries"] = gbs(b_staged, rf["ms_per_launch"]) / HBM_PEAK_GBS
        rf["bytes_read_per_launch_counted"] = b_staged
        rf["kernel_counters"] = counts
        # `frac` never exceeds what the kernel read: where the formula on the handed list entries is more than 1.3 x the counted
        # figure (saturated quadrants stop long before their tile's list ends) the counted one is the primary number
        if rf["frac"] > 1.3 * rf["frac_on_staged_entries"]:
            rf["frac_formula_on_all_list_entries"] = rf["frac"]
            rf["frac"] = rf["frac_on_staged_entries"]
            rf["achieved"] = gbs(b_staged, rf["ms_per_launch"])
            rf["units"] = ("COUNTED by the kernel: 4 B x list entries scanned + 80 B x list entries staged + 36 B x pixels + 8 B x tiles -- the "
                           "72 B x list-entry formula gives more than 1.3 x that here (frac_formula_on_all_list_entries) because saturated "
                           "quadrants never read most of their tile's list")
    # HBM bytes per launch need PMC counters (separate rocprofv3 --pmc passes, which cannot run inside this process): at N = 1 `traffic` is
    # filled at the end by two child runs of this workload under rocprofv3 (live_pmc_traffic); the figure of the newest committed
    # profile of THIS configuration stays beside it under traffic_from_profiles, with its source
    rf.update({"traffic": None,
               "traffic_from_profiles": prof.get("traffic"), "valu_from_profiles": prof.get("valu"),
               "valu_issue_frac_from_profiles": (prof.get("valu") or {}).get("valu_issue_frac"),
               "peak_measured_copy": copy_gbs,     # device-to-device copy on THIS box, read + write bytes
               "stage_ms_per_step": {"preprocess": stage_ms[0] / args.steps, "binning": stage_ms[1] / args.steps,
                                     "compositing": stage_ms[2] / args.steps}})
    caveat = ("fast = error-free float32 pairs for the float64 island + hardware exp/rcp/rsq; every channel within 1e-4 of the oracle "
              "(RGB <= 3e-7), EXCEPT that the distortion channel (values 1e-7..1e-5 from cancelling float32 sums) moves by 3-17 % "
              "relative (median) at sigma0 = 0.01 -- absolute <= 1e-6, the reference's own 1-ulp noise on that channel") \
        if args.render_mode == "fast" else "exact = the reference's float32/float64 operation order"
    result = {
        "metric": "rendered views/sec at 256x256 (N Gaussians, K cams)",
        "value": world * V * args.steps / elapsed,
        "unit": "views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.render_mode == "fast" else "f32 (+f64 islands, as the reference)",
        "data": "synthetic" if args.data == "synthetic" else
                "real image n01644373_4548.jpg + LeReS depth (tests/golden/real_image_256.npz) through this build's predictor + cycle "
                "aggregation with formula weights (no checkpoint travels); cameras as the reference's orbit",
        "config": {"workload": "%s: 1 image/GPU, %d Gaussians (%s), %d-view orbit @%dx%d, GOF forward raster only; timed region = "
                               "%s; compositing arithmetic: %s" % (
                                   "C2" if args.data == "synthetic" else "C2-shaped on the real merged set",
                                   P, ("sigma0=%g" % args.sigma0) if args.data == "synthetic" else "9 x 65,536 predicted, sigma ~ 0.01", V, RES, RES,
                                   "render + RCCL gather of the 8-bit frames to rank 0" if world > 1 else
                                   ("render, frames left in HBM" if d2h is None else "render + 8-bit RGB frames copied to pinned host memory"),
                                   caveat),
                   "gaussians": P, "views": V, "resolution": RES, "sigma0": args.sigma0, "instances_per_step": R_total, "lists": list_stats,
                   "instances_processed_per_step": R_proc, "tile_cull": args.tile_cull,
                   "views_per_call": args.views_per_call, "render_mode": args.render_mode, "channels": args.channels,
                   "kernel_launches_per_call": nlaunch / float(ncalls),
                   "parallelism": "image-sharded x%d + RCCL gather" % world if world > 1 else "single GPU"},
        "value_in_hbm": world * V * args.steps / elapsed_hbm, "ms_per_step_in_hbm": 1e3 * elapsed_hbm / args.steps,
        "call_ms_spread": {"note": "device time of one call (projection + binning + compositing, HIP events) over the calls of the timed "
                                   "loop `value` comes from: a box-to-box or run-to-run effect shows here, not only in the mean",
                           "slowest_call_index": (max(range(len(rows)), key=lambda i: sum(rows[i])) if rows else None),
                           "all_stages": spread(rows, None), "preprocess": spread(rows, 0), "binning": spread(rows, 1),
                           "compositing": spread(rows, 2)},
        "dist_backend": (os.environ.get("F3DG_DIST_BACKEND", "nccl") if world > 1 else None),
        "ranks_seen": (dist.get_world_size() if world > 1 else 1),
        "roofline": rf,
        "rooflines_other": {
            "preprocess_kernel": {"bound": "hbm", "algorithmic_bytes_per_launch": b_pre, "ms_per_launch": per(stage_ms[0]),
                                  "achieved": gbs(b_pre, per(stage_ms[0])), "unit": "GB/s", "frac": gbs(b_pre, per(stage_ms[0])) / HBM_PEAK_GBS,
                                  "formula": "this build's streams: 92 B x P (inputs, once per call) + 96 B x P x views written",
                                  "traffic_from_profiles": (prof.get("stages") or {}).get("preprocess")},
            "binning (depth sort per Gaussian, instance generation, tile pass, ranges)": {
                "bound": "hbm", "algorithmic_bytes_per_launch": b_bin, "ms_per_launch": per(stage_ms[1]),
                "achieved": gbs(b_bin, per(stage_ms[1])), "unit": "GB/s", "frac": gbs(b_bin, per(stage_ms[1])) / HBM_PEAK_GBS,
                "formula": "this build's streams: 108 B x P x views + 18 B x instances_processed + 8 B x tiles x views",
                "traffic_from_profiles": (prof.get("stages") or {}).get("binning")}},
    }
    if d2h:
        result["with_d2h"] = d2h
    if exact:
        e_x, st_x, nc_x, rows_x, kernel_name_x = exact
        result["value_exact"] = world * V * args.steps / e_x
        result["ms_per_step_exact"] = 1e3 * e_x / args.steps
        result["roofline_exact"] = roofline(st_x, kernel_name_x, nc_x, rows_x)
        result["roofline_exact"]["note"] = "same run, option render_fast = 0: the reference's float32/float64 operation order; frames left in HBM"
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(g, cams, shs, P, RES, args.cpu_sample_views if args.data == "synthetic" else max(2, args.cpu_sample_views // 3))
    if world == 1 and not args.no_pmc and os.environ.get("F3DG_BENCH_PMC", "1") != "0":
        live = live_pmc_traffic(kernel_name, ["--workload", "c2", "--steps", "3", "--warmup", "1", "--no-d2h", "--no-exact",
                                              "--gaussians", str(args.gaussians), "--views", str(args.views), "--res", str(args.res),
                                              "--sigma0", repr(args.sigma0), "--views-per-call", str(args.views_per_call),
                                              "--render-mode", args.render_mode, "--tile-cull", str(args.tile_cull), "--data", args.data,
                                              "--channels", args.channels, "--scan", str(args.scan)])
        rf["traffic"] = live.get("bytes_per_launch")
        rf["traffic_live"] = live
        if rf["traffic"]:
            rf["frac_on_counter_traffic"] = rf["traffic"] / (rf["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return result


def pmc_counter_mean(directory, kernel_substring, counter):
    """(mean value, dispatches) of `counter` over the dispatches with the LARGEST grid among the kernels whose name contains
    `kernel_substring`, from the *counter_collection.csv files rocprofv3 --pmc left under `directory` (set-up launches of the same kernel
    on fewer views have smaller grids and are left out); None when no row matches."""
    import csv
    rows = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if kernel_substring in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
    if not rows:
        return None
    big = max(int(r["Grid_Size"]) for r in rows)
    v = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == big]
    return sum(v) / len(v), len(v)


def live_pmc_traffic(kernel_name, workload_args):
    """HBM bytes per launch of the compositing kernel from the PMC counters, measured NOW: this same workload run again in two child
    processes under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one counter per pass, nothing else traced -- the recipe of
    MI355X_MICROARCH.md's HBM section), 3 steps each, frames left in HBM. Per pass: the mean over the dispatches of that kernel with the
    largest grid (the timed launches). traffic = 2 x FETCH_SIZE + WRITE_SIZE, both in KB in the CSV (the gfx950 correction: FETCH_SIZE
    tallies a 128-byte line as 64 -- calibrated for this kernel's 64-byte gathers in notes/r06.md section 1). Any failure (no rocprofv3,
    time-out, no matching dispatch) leaves bytes_per_launch None and says why; the bench line is printed either way."""
    import shutil
    import signal
    import subprocess
    import tempfile
    out = {"bytes_per_launch": None, "kernel": kernel_name, "method": "2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of this workload "
           "in child processes of this run (a few steps each, mean over the largest-grid dispatches of the kernel)"}
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        out["error"] = "rocprofv3 not found"
        return out
    sub = (kernel_name or "").split("<")[0].strip()
    if not sub:
        out["error"] = "no kernel name"
        return out
    cmd_tail = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-pmc"] + list(workload_args)
    env = dict(os.environ, TMPDIR="/tmp", F3DG_BENCH_PMC="0")
    vals = {}
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="f3dg_pmc_", dir="/tmp")
        try:
            p = subprocess.Popen([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "b", "--"] + cmd_tail,
                                 cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=240)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                out["error"] = "%s pass timed out" % counter
                return out
            got = pmc_counter_mean(d, sub, counter)
            if got is None:
                out["error"] = "%s pass: no dispatch of %s in the counter file (exit code %s)" % (counter, sub, p.returncode)
                return out
            vals[counter] = got
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0] * 1024.0, vals["WRITE_SIZE"][0] * 1024.0
    out.update({"bytes_per_launch": 2.0 * fetch + write, "fetch_size_bytes_x2": 2.0 * fetch, "write_size_bytes": write,
                "dispatches_averaged": vals["FETCH_SIZE"][1], "seconds": round(time.time() - t0, 1)})
    return out


# ---------------------------------------------------------------------------------------------------------------- C4
def run_c4(args, rank, world, dist, device, comm_device, f3d, L):
    """BASELINE C4 per rank (C3 at N = 1): B images -> predictor -> 8 cycle views -> 8 re-predictions -> merged sets of 589,824
    Gaussians -> their 8 orbit views -> 8-bit frames -> gather. Random weights (no checkpoint travels), synthetic images."""
    from f3dgaus_amd import _lib, cameras
    B, RES, V = args.images, args.res, 8
    cfg = cameras.default_cfg(RES)
    cfg['model']['backbone_dtype'] = args.backbone
    cfg['model']['backbone_layout'] = args.backbone_layout
    if args.backbone_chunk >= 0:
        cfg['model']['backbone_chunk'] = args.backbone_chunk
    torch.backends.cudnn.benchmark = True               # MIOpen picks its convolution algorithms once per shape
    torch.manual_seed(0)
    model = f3d.Unet_GS_gtunet(cfg, renderer=f3d.render_predicted_more_v2_gof).to(device).eval()
    gen = torch.Generator().manual_seed(100 + rank)
    images = torch.rand(B, 3, RES, RES, generator=gen).to(device)
    depth = (torch.rand(B, 1, RES, RES, generator=gen) * 2 + 6.667).to(device)
    rig = cameras.OrbitRig(cfg)
    per_call = max(1, min(B, 8))
    gat = Gatherer(dist, world, rank, (B * V, RES, RES, 3), comm_device)
    t_cycle = [0.0, 0.0]

    def step():
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        merged = f3d.cycle.cycle_aggregate(model, images, depth, cfg, rig=rig, num_views=V)
        e1.record()
        orbit = f3d.cycle.render_orbit(merged, cfg, rig=rig, num_views=V, views_per_call=V, epilogue=False, images_per_call=per_call)
        frames = f3d.gaussian_renderer.pack_frames(orbit["render"].reshape(B * V, 3, RES, RES))
        e2.record()
        gat.submit(frames)
        step.events.append((e0, e1, e2))
    step.events = []

    timed(step, gat.barrier, args.warmup, 0)
    step.events.clear()
    L.f3dg_profile_enable(1)
    L.f3dg_debug_launch_count(1)
    elapsed = timed(step, gat.barrier, 0, args.steps)
    nlaunch = int(L.f3dg_debug_launch_count(1))      # kernels of libf3dg_hip.so: rasterizer, splat head, hand-off, GroupNorm, packing
    L.f3dg_profile_enable(0)
    stage_ms = (C.c_double * 5)()
    ncalls = C.c_int(0)
    _lib.check(L.f3dg_profile_collect(stage_ms, C.byref(ncalls)), "f3dg_profile_collect")
    for e0, e1, e2 in step.events:
        t_cycle[0] += e0.elapsed_time(e1)
        t_cycle[1] += e1.elapsed_time(e2)
    elapsed = max_over_ranks(elapsed, dist, world, comm_device if world > 1 else device)
    if rank != 0:
        return None
    calls_per_step = ncalls.value / max(args.steps, 1)
    return {
        "metric": "rendered views/sec at 256x256 (N Gaussians, K cams)",
        "value": world * B * V * args.steps / elapsed, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (rasterizer, splat head); backbone %s" % args.backbone, "data": "synthetic",
        "config": {"workload": "C4 shape per rank (C3 at 1 GPU): %d images/GPU @%dx%d, predictor (SongUNet, random weights) + cycle "
                               "aggregation (8 views, 8 re-predictions, merged sets of 589,824 Gaussians) + 8 orbit views of every merged set "
                               "+ frame packing + gather" % (B, RES, RES),
                   "images_per_gpu": B, "views_per_image": V, "resolution": RES, "backbone": args.backbone, "backbone_layout": args.backbone_layout, "backbone_chunk": args.backbone_chunk,
                   "parallelism": "image-sharded x%d + RCCL gather" % world if world > 1 else "single GPU"},
        "breakdown_ms_per_step": {"predictor + cycle aggregation": t_cycle[0] / args.steps, "orbit render + frame packing": t_cycle[1] / args.steps,
                                  "rasterizer stages (HIP events)": {"preprocess": stage_ms[0] / args.steps, "binning": stage_ms[1] / args.steps,
                                                                     "compositing": stage_ms[2] / args.steps}},
        "launches": {"rasterizer_calls_per_step": calls_per_step, "rasterizer_calls_per_image": calls_per_step / B,
                     "library_kernel_launches_per_image": nlaunch / float(max(args.steps, 1)) / B,
                     "reference_rasterizer_calls_per_image": 2 * V,
                     "note": "the reference issues one rasterizer call (>= 10 launches + a blocking D2H) per (image, view): "
                             "visualize.py:293-314 and :387-416"},
    }




# ------------------------------------------------------------------------------------------------------------ drop-in
def run_dropin(args, rank, world, dist, device, comm_device, f3d, L):
    """The reference's per-view loop around the drop-in operator: one `render_predicted_more_v2_gof` call per view, exactly the
    arguments visualize.py:394-399 passes ([th:th+1] slices of the camera stacks, a [1,3] background, the config dict)."""
    from f3dgaus_amd import _lib, cameras, synthetic
    P = args.gaussians if args.gaussians != 196608 else 65536
    V, RES = args.views, args.res
    cfg = cameras.default_cfg(RES)
    # the predictor's Gaussians are PIXEL-ORDERED (id = y * res + x on the input image's depth map): that is what these loops feed
    # the rasterizer, and the id order matters to the small-call path (a wave's share of the ids is a band of image rows)
    # (a multiple of res^2: a MERGED set -- visualize.py:387-416 renders the orbit from the nine predicted sets of the cycle, 589,824
    # Gaussians --: that many pixel-ordered blocks one after the other)
    pixel = P % (RES * RES) == 0 and not os.environ.get("F3DG_DROPIN_RANDOM_IDS")
    if pixel:
        blocks = [synthetic.make_pixel_gaussians(RES, s0=args.sigma0, seed=rank + 17 * b, device=device) for b in range(P // (RES * RES))]
        g = {k: torch.cat([b[k] for b in blocks], 0).contiguous() for k in blocks[0]}
    else:
        g = synthetic.make_gaussians(P, s0=args.sigma0, seed=rank, device=device)
    pc = {"xyz": g["xyz"][None], "opacity": g["opacity"][None], "scaling": g["scaling"][None], "rotation": g["rotation"][None],
          "features_dc": g["features_dc"][None], "features_rest": g["features_rest"][None]}
    cams = synthetic.orbit_cameras(V, resolution=RES, device=device)
    wv, fp, cc = cams["viewmatrix"].unsqueeze(1), cams["projmatrix"].unsqueeze(1), cams["campos"].unsqueeze(1)
    bg = torch.zeros(1, 3, device=device)
    frames = torch.empty((V, 3, RES, RES), dtype=torch.float32, device=device)
    host = torch.empty((V, 3, RES, RES), dtype=torch.float32).pin_memory()

    def step_device():          # frames stay on the device
        with torch.no_grad():
            for th in range(V):
                o = f3d.render_predicted_more_v2_gof(pc, 0, wv[th:th + 1].contiguous(), fp[th:th + 1].contiguous(), cc[th:th + 1].contiguous(), bg, cfg)
                frames[th] = o["render"]

    def step_reference():       # as the reference: every frame goes to the host before the next call (visualize.py:400)
        with torch.no_grad():
            for th in range(V):
                o = f3d.render_predicted_more_v2_gof(pc, 0, wv[th:th + 1].contiguous(), fp[th:th + 1].contiguous(), cc[th:th + 1].contiguous(), bg, cfg)
                host[th] = o["render"].reshape(3, RES, RES).cpu()

    sync = lambda: torch.cuda.synchronize()
    # the rasterizer alone, same arguments (f3dg_forward_batched with one view): what the wrapper adds is the difference
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    out1 = torch.empty((1, 9, RES, RES), dtype=torch.float32, device=device)
    rad1 = torch.empty((1, P), dtype=torch.int32, device=device)
    _, _, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"][:1], cams["projmatrix"][:1], cams["campos"][:1], bg[0],
                                   image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs,
                                   scales=g["scaling"], rotations=g["rotation"], sh_degree=1, out=out1, radii=rad1)

    def step_raster():
        for th in range(V):
            f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"][th:th + 1], cams["projmatrix"][th:th + 1], cams["campos"][th:th + 1],
                                bg[0], image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs,
                                scales=g["scaling"], rotations=g["rotation"], sh_degree=1, workspace=ws, out=out1, radii=rad1, check=False)

    e_ras = timed(step_raster, sync, args.warmup, args.steps)
    # opt-in: the status of a call is checked when the next call arrives on the stream instead of blocking (set_deferred_status)
    f3d.set_deferred_status(True)

    def sync_flush():
        f3d.flush()
        torch.cuda.synchronize()
    e_def = timed(step_device, sync_flush, args.warmup, args.steps)
    f3d.set_deferred_status(True, depth=2)      # call k checked when call k + 2 arrives: the host runs a whole call ahead of the device
    e_def2 = timed(step_device, sync_flush, args.warmup, args.steps)
    f3d.set_deferred_status(False)
    timed(step_device, sync, args.warmup, 0)
    L.f3dg_profile_enable(1)
    L.f3dg_debug_launch_count(1)
    e_dev = timed(step_device, sync, 0, args.steps)
    launches = int(L.f3dg_debug_launch_count(1))
    L.f3dg_profile_enable(0)
    st = (C.c_double * 5)()
    nc = C.c_int(0)
    _lib.check(L.f3dg_profile_collect(st, C.byref(nc)), "f3dg_profile_collect")
    e_ref = timed(step_reference, sync, 1, args.steps)      # (last: the blocking pageable copies leave the process in a slower state)
    if rank != 0:
        return None
    n = V * args.steps
    calls = max(int(nc.value), 1)
    return {
        "metric": "rendered views/sec at 256x256 (N Gaussians, K cams)", "value": n / e_dev, "unit": "views/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * e_dev / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_deferred_status": n / e_def,
        "value_deferred_status_depth2": n / e_def2,
        "gaussian_order": ("pixel-ordered (id = y * res + x, as the predictor emits them)" + (", %d blocks (a merged set)" % (P // (RES * RES)) if P > RES * RES else "")) if pixel else "random ids",
        "config": {"workload": "drop-in: render_predicted_more_v2_gof one view per call (the reference's loop, visualize.py:387-416), %d Gaussians "
                               "(sigma0=%g), %d calls per step @%dx%d, frames left on the device" % (P, args.sigma0, V, RES, RES),
                   "gaussians": P, "views": V, "resolution": RES, "kernel_launches_per_call": launches / float(calls)},
        "us_per_call": {"wrapper, frames on the device (= value)": 1e6 * e_dev / n,
                        "wrapper with set_deferred_status(True): the status of call k is checked when call k + 1 arrives": 1e6 * e_def / n,
                        "wrapper + .cpu() of every frame before the next call (the reference's loop)": 1e6 * e_ref / n,
                        "rasterizer alone (f3dg_forward_batched, one view, no host sync)": 1e6 * e_ras / n,
                        "rasterizer stages per call (HIP events)": {"preprocess": 1e3 * st[0] / calls, "binning": 1e3 * st[1] / calls,
                                                                     "compositing": 1e3 * st[2] / calls}},
    }


# ---------------------------------------------------------------------------------------------------------------- C5
def run_c5(args, rank, world, dist, device, comm_device, f3d, L):
    """BASELINE C5: 1 M Gaussians, 32 views @512x512, forward with the auxiliary planes + backward, one call each per step."""
    from f3dgaus_amd import _lib, synthetic
    from f3dgaus_amd.diff_gof_rasterization.backward import rasterize_backward_raw
    P, V, RES = 1000000, 32, 512
    g = synthetic.make_gaussians(P, s0=args.sigma0, seed=rank, device=device)
    cams = synthetic.orbit_cameras(V, resolution=RES, device=device)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).contiguous()
    bg = torch.zeros(3, device=device)
    gen = torch.Generator().manual_seed(11)
    dpix = torch.randn(V, 9, RES, RES, generator=gen).to(device)
    dpix[:, 7] = 0
    kw = dict(image_height=RES, image_width=RES, tanfovx=cams["tanfovx"], tanfovy=cams["tanfovy"], sh=shs, scales=g["scaling"],
              rotations=g["rotation"], sh_degree=1, save_aux=True)
    # the reference's num_rendered: unit count of the byte formulas (per call: F3DG_FLAG_NO_TILE_CULL)
    R = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, tile_cull=False, **kw)[2].num_rendered
    out, radii, ws = f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, **kw)
    R_proc = ws.num_rendered

    def step():
        f3d.rasterize_views(g["xyz"], g["opacity"], cams["viewmatrix"], cams["projmatrix"], cams["campos"], bg, workspace=ws,
                            out=out, radii=radii, check=False, **kw)
        rasterize_backward_raw(ws, g["xyz"], shs, None, g["scaling"], g["rotation"], radii, dpix, 1, cams["viewmatrix"],
                               cams["projmatrix"], cams["campos"], bg, cams["tanfovx"], cams["tanfovy"], 0.0, 1.0)

    sync = lambda: torch.cuda.synchronize()
    timed(step, sync, args.warmup, 0)
    L.f3dg_profile_enable(1)
    elapsed = timed(step, sync, 0, args.steps)
    L.f3dg_profile_enable(0)
    stage_ms = (C.c_double * 5)()
    ncalls = C.c_int(0)
    _lib.check(L.f3dg_profile_collect(stage_ms, C.byref(ncalls)), "f3dg_profile_collect")
    fwd_kernel = L.f3dg_debug_last_render_kernel().decode()        # what the library launched for the forward (the packed kernel, render4, by default)
    pairs = C.c_longlong(0)
    _lib.check(L.f3dg_backward_pairs(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.buffer.data_ptr()), C.byref(pairs)),
               "f3dg_backward_pairs")
    if rank != 0:
        return None
    T = ((RES + 15) // 16) ** 2
    n = max(args.steps, 1)
    gbs = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    b_fwd = 72.0 * R + (60.0 * RES * RES + 8.0 * T) * V                       # with the auxiliary planes
    b_bwd = 80.0 * R + 60.0 * RES * RES * V + 68.0 * pairs.value
    bwd_kernel = "render3_bwd_kernel" if "bwd_dense=0" in os.environ.get("F3DG_OPTIONS", "").replace(" ", "") else "render5_bwd_kernel"
    bwd_rf = {"bound": "hbm", "kernel": bwd_kernel, "algorithmic_bytes_per_launch": b_bwd, "ms_per_launch": stage_ms[3] / n,
              "achieved": gbs(b_bwd, stage_ms[3] / n), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(b_bwd, stage_ms[3] / n) / HBM_PEAK_GBS,
              "traffic": None, "formula": "80 R + 60 W H V + 68 C (R = instances_per_step, the reference's num_rendered; C = contributing "
                                          "pairs, counted by the kernel)",
              "frac_on_processed_instances": gbs(80.0 * R_proc + 60.0 * RES * RES * V + 68.0 * pairs.value, stage_ms[3] / n) / HBM_PEAK_GBS,
              **c5_profile_record(stage_ms[3] / n)}
    # `frac` never exceeds what the kernel moved: SURVEY 8d's formula prices every contributing pair at 68 bytes of atomics, which the
    # wave-level reduction of this kernel never issues; where a committed PMC profile of this configuration exists and the formula gives
    # more than 1.3 x its figure, the counter figure is the primary number and the formula's stays under its own key
    if world == 1 and not args.no_pmc and os.environ.get("F3DG_BENCH_PMC", "1") != "0":
        live = live_pmc_traffic(bwd_kernel, ["--workload", "c5", "--steps", "2", "--warmup", "1", "--sigma0", repr(args.sigma0)])
        bwd_rf["traffic"] = live.get("bytes_per_launch")
        bwd_rf["traffic_live"] = live
        if bwd_rf["traffic"]:      # this run's own counter passes take precedence over the committed profile's
            bwd_rf["frac_on_counter_traffic"] = bwd_rf["traffic"] / (bwd_rf["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    fc = bwd_rf.get("frac_on_counter_traffic")
    if fc and bwd_rf["frac"] > 1.3 * fc:
        bwd_rf["frac_formula_80R_60WHV_68C"] = bwd_rf["frac"]
        bwd_rf["frac"] = fc
        bwd_rf["achieved"] = fc * HBM_PEAK_GBS
        bwd_rf["units"] = ("HBM bytes per launch from PMC passes (2 x FETCH_SIZE + WRITE_SIZE: this run's child passes when `traffic` is set, "
                           "else the newest committed profile of this configuration) / this run's kernel time")
    return {
        "metric": "rendered views/sec at 256x256 (N Gaussians, K cams)", "value": world * V * args.steps / elapsed, "unit": "views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (+f64 islands, as the reference)", "data": "synthetic",
        "config": {"workload": "C5: %d Gaussians (sigma0=%g), %d views @%dx%d, forward with auxiliary planes + backward "
                               "(random dL/dpix on channels 0-6, 8); views/s counts a forward + backward as one view" % (P, args.sigma0, V, RES, RES),
                   "gaussians": P, "views": V, "resolution": RES, "instances_per_step": R, "instances_processed_per_step": R_proc,
                   "tile_cull": args.tile_cull, "contributing_pairs_per_step": pairs.value},
        "roofline": bwd_rf,
        "rooflines_other": {
            fwd_kernel: {"bound": "hbm", "algorithmic_bytes_per_launch": b_fwd, "ms_per_launch": stage_ms[2] / n,
                                                               "achieved": gbs(b_fwd, stage_ms[2] / n), "unit": "GB/s",
                                                               "frac": gbs(b_fwd, stage_ms[2] / n) / HBM_PEAK_GBS}},
        "stage_ms_per_step": {"preprocess": stage_ms[0] / n, "binning": stage_ms[1] / n, "compositing": stage_ms[2] / n,
                              "compositing backward": stage_ms[3] / n, "per-Gaussian backward": stage_ms[4] / n},
    }


# ---------------------------------------------------------------------------------------------------------------- helpers
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


def c5_profile_record(bwd_ms):
    """Counter figures of the compositing backward from the newest committed C5 profile (profiles/*/c5_traffic.json, written by
    tools/assemble_profile.py from separate rocprofv3 --pmc passes of `bench.py --workload c5`): the HBM bytes per launch and what
    fraction of the roofline they are at THIS run's kernel time -- next to the formula's 68 C atomic term, which the wave reduction never issues."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "c5_traffic.json"))):
        try:
            best = (f, json.load(open(f)))
        except Exception:
            pass
    if best is None:
        return {"traffic_from_profiles": None}
    f, t = best
    tb = t.get("traffic_bytes_per_launch")
    return {"traffic_from_profiles": {"bytes_per_launch": tb, "kernel": t.get("kernel"), "source": os.path.relpath(f, os.path.dirname(os.path.abspath(__file__))),
                                      "note": "builder-side PMC passes, not measured in this run"},
            "frac_on_counter_traffic": (tb / (bwd_ms * 1e-3) / 8e12) if tb else None,
            "valu_from_profiles": t.get("valu")}


def profiles_record(P, V, RES, views_per_call, mode, tile_cull=1, sigma0=0.01):
    """Counter-derived figures of the compositing kernel from the newest committed profile of THIS configuration
    (profiles/*/traffic.json, written by tools/make_profile.py from separate rocprofv3 --pmc passes of this same command):
    `traffic` = HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md),
    `valu` = SQ instruction counts / lane utilisation. Returned with their source path; {} when no profile matches."""
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json"))):
        try:
            t = json.load(open(path))
            c = t["config"]
            if (c["gaussians"], c["views"], c["resolution"], c["views_per_call"]) == (P, V, RES, views_per_call) and \
                    c.get("render_mode", "exact") == mode and c.get("tile_cull", 0) == tile_cull and abs(c.get("sigma0", 0.01) - sigma0) < 1e-9:
                src = os.path.relpath(path, ROOT)
                best = {"traffic": {"bytes_per_launch": t["traffic_bytes_per_launch"], "kernel": t.get("kernel"), "source": src,
                                    "note": "builder-side PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), not measured in this run"},
                        "valu": dict(t.get("valu") or {}, source=src),
                        "stages": {k: dict(v, source=src) for k, v in (t.get("stages") or {}).items()}}
        except Exception:
            pass
    return best


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
