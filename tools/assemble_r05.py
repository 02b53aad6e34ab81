"""Assembles profiles/r05_final/ from a tools/collect_r05.sh run and the round's A/B directories:
    python tools/assemble_r05.py gpurun_out/r05 profiles/r05_final
summary.md (bench lines), bench_kernel_stats.csv, traffic.json (the format bench.py's profiles_record reads), compositing_packed.md (render3s
against the rank-packed kernel: A/B lines + PMC of the timed launches), real_data.md, projection_hoist.md, backbone.md, pytest_gpu_final.log."""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
GO = os.path.dirname(os.path.abspath(src))


def lj(path):
    if not os.path.exists(path):
        return None
    l = [x for x in open(path) if x.startswith("{")]
    return json.loads(l[-1]) if l else None


def pmc(tag):
    p = os.path.join(src, tag, "summary.txt")
    if not os.path.exists(p):
        return None
    s = json.load(open(p))
    best = max(s.items(), key=lambda kv: kv[1].get("dispatches_averaged", 0))
    return best[0], best[1]


# ---- summary.md
rows = [("C2 default (196,608 Gaussians, 120 views, fast arithmetic)", "bench_default"), ("the same, packed kernel off (render_pack 0)", "bench_default_pack0"),
        ("C2, channels rgb + depth + alpha (the build's own loops)", "bench_lean"), ("sigma0 = 0.05", "bench_sigma005"),
        ("589,824 synthetic x 128 views", "bench_589k"), ("real image, merged 589,824 x 128 views (--data real)", "bench_real"),
        ("the same, packed kernel off", "bench_real_pack0"), ("real, channels rgb + depth + alpha", "bench_real_lean")]
out = ["# Round 5 evidence run (`tools/collect_r05.sh`, one MI355X box)", "",
       "| line | views/s (`value`; in HBM) | reference arithmetic (`value_exact`) | ms/step | projection | binning | compositing | `frac` | compositing, reference arithmetic (ms; kernel) |",
       "|---|---:|---:|---:|---:|---:|---:|---:|---|"]
for label, f in rows:
    j = lj(os.path.join(src, f + ".log"))
    if not j:
        continue
    rf, rx = j["roofline"], j.get("roofline_exact") or {}
    st = rf["stage_ms_per_step"]
    out.append("| %s | %.0f; %.0f | %s | %.2f | %.2f | %.2f | %.2f | %.3f | %s |" % (
        label, j["value"], j.get("value_in_hbm", 0), ("%.0f" % j["value_exact"]) if j.get("value_exact") else "", j["ms_per_step"], st["preprocess"], st["binning"],
        st["compositing"], rf["frac"], ("%.2f; `%s`" % (rx["ms_per_launch"], rx["kernel"].split("<")[0])) if rx else ""))
j = lj(os.path.join(src, "bench_dropin.log"))
if j:
    u = j.get("us_per_call", {})
    st = [v for k, v in u.items() if isinstance(v, dict)]
    out += ["", "Drop-in loop (one view per call, 65,536 pixel-ordered Gaussians): %.0f views/s, %.0f with `set_deferred_status(True)`, %.0f with `depth=2`; kernels per call %s us." % (
        j["value"], j.get("value_deferred_status", 0), j.get("value_deferred_status_depth2", 0), json.dumps({k: round(v, 1) for k, v in st[0].items()}) if st else "?")]
    jx = lj(os.path.join(src, "bench_dropin_exact.log"))
    if jx:
        stx = [v for k, v in jx.get("us_per_call", {}).items() if isinstance(v, dict)]
        out += ["The same in the reference's arithmetic: %.0f / %.0f / %.0f views/s; kernels per call %s us." % (
            jx["value"], jx.get("value_deferred_status", 0), jx.get("value_deferred_status_depth2", 0), json.dumps({k: round(v, 1) for k, v in stx[0].items()}) if stx else "?")]
j = lj(os.path.join(src, "bench_c5.log"))
if j:
    out += ["", "C5 (1 M Gaussians, 32 views @512^2, forward + backward): %.0f views/s, %.2f ms/step; stages %s." % (j["value"], j["ms_per_step"], json.dumps({k: round(v, 2) for k, v in j.get("stage_ms_per_step", {}).items()}))]
    rf = j.get("roofline", {})
    out += ["Backward `frac` %.3f (%s)." % (rf.get("frac", 0), rf.get("units", rf.get("formula", "")))]
for f, label in (("bench_c4_fp32", "C4 shape, 16 images, fp32 backbone, backbone_chunk 8 (default)"), ("bench_c4_fp32_chunk0", "the same, whole 16-image passes (backbone_chunk 0)")):
    j = lj(os.path.join(src, f + ".log"))
    if j:
        out += ["", "%s: %.1f views/s, %.0f ms/step." % (label, j["value"], j["ms_per_step"])]
log = os.path.join(src, "pytest_gpu.log")
if os.path.exists(log):
    tail = [l.strip() for l in open(log) if "passed" in l or l.startswith("real")]
    out += ["", "GPU suite on the same box: " + "; ".join(tail)]
    shutil.copy(log, os.path.join(dst, "pytest_gpu_final.log"))
open(os.path.join(dst, "summary.md"), "w").write("\n".join(out) + "\n")
for f in glob.glob(os.path.join(src, "stats", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "bench_kernel_stats.csv"))
for f in ("parity_report.md", "unet_determinism.log", "unet_first_use.log", "issue_rate.log", "r3q_clocks.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f.replace(".log", ".md")))

# ---- traffic.json for bench.py's profiles_record (C2 fast)
p = pmc("pmc_c2_fast")
pp = pmc("pmc_c2_preprocess")
if p:
    k, e = p
    c = e["counters"]
    t = {"kernel": k[:64], "config": {"gaussians": 196608, "views": 120, "resolution": 256, "views_per_call": 120, "render_mode": "fast", "tile_cull": 1, "sigma0": 0.01},
         "FETCH_SIZE_KB_per_launch": c.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": c.get("WRITE_SIZE"), "traffic_bytes_per_launch": e["hbm_bytes"]["total"],
         "valu": {"lane_utilisation": e.get("lane_utilisation"), "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"), "SQ_ACTIVE_INST_VALU": c.get("SQ_ACTIVE_INST_VALU"),
                  "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES"), "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES"), "kernel_us_rocprof": e["kernel_stats"]["avg_us_all_calls"],
                  "effective_clock_ghz": e.get("effective_clock_ghz"), "valu_issue_frac": e.get("valu_issue_frac"), "lds_busy_frac": e.get("lds_busy_frac"),
                  "note": "means over the dispatches with the largest grid of the kernel name (tools/pmc_kernel.py); valu_issue_frac = (SQ_INSTS_VALU x 2 + transcendentals x 6 cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"},
         "stages": {}, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncalibrated"}
    if pp:
        t["stages"]["preprocess"] = {"bytes_per_launch": pp[1]["hbm_bytes"]["total"], "note": "PMC, 2 x FETCH_SIZE + WRITE_SIZE"}
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

# ---- compositing_packed.md
L = ["# render3s against the rank-packed kernel (render4, `csrc/f3dg_render4.hip`)", "",
     "PMC of the TIMED launches only (`tools/pmc_kernel.sh`: dispatches with the largest grid of the kernel name), one box, `tools/collect_r05.sh`:", "",
     "| workload, arithmetic | kernel | avg us | SQ_INSTS_VALU | SQ_INSTS_SALU | SQ_INSTS_LDS | lane utilisation | VALU issue frac | LDS busy frac | 2 x FETCH + WRITE (GB) | clock GHz |",
     "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
for label, tag in (("C2, fast", "pmc_c2_fast"), ("C2, reference arithmetic, render3s", "pmc_c2_exact_pack0"), ("C2, reference arithmetic, packed (default)", "pmc_c2_exact"),
                   ("real set, fast, render3s (default)", "pmc_real_fast"), ("real set, fast, packed (th 16)", "pmc_real_fast_pack1"),
                   ("real set, reference arithmetic, render3s", "pmc_real_exact_pack0"), ("real set, reference arithmetic, packed (default)", "pmc_real_exact")):
    p = pmc(tag)
    if not p:
        continue
    k, e = p
    c = e["counters"]
    L.append("| %s | `%s` | %.1f | %.3e | %.3e | %.3e | %.3f | %.3f | %.3f | %.2f | %.2f |" % (
        label, k.replace("void ", "").split("(")[0][:44], e["kernel_stats"]["avg_us_all_calls"], c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0),
        e.get("lane_utilisation") or 0, e.get("valu_issue_frac") or 0, e.get("lds_busy_frac") or 0, e["hbm_bytes"]["total"] / 1e9, e.get("effective_clock_ghz") or 0))
L += ["", "(avg us is the mean over all calls of the name under `rocprofv3 --kernel-trace --stats`; the real-set launches are 128 views, C2's 120.)", ""]
for tag, title in (("r05h", "Packing threshold sweep (HIP-event compositing ms from bench.py; `k3` = render3s, `k4_thN` = render4 with render_pack_th N; `c2x` / `realx` = reference arithmetic)"),
                   ("r05_lean", "Channels rgb + depth + alpha (`p0` = render3s, `p1_thN` = packed)")):
    d = os.path.join(GO, tag)
    if not os.path.isdir(d):
        continue
    L += ["## " + title, "", "| run | views/s | step ms | compositing ms | frac | counters |", "|---|---:|---:|---:|---:|---|"]
    for f in sorted(glob.glob(os.path.join(d, "*.log"))):
        j = lj(f)
        if not j:
            continue
        rf = j["roofline"]
        kc = rf.get("kernel_counters") or {}
        keep = ("fused_trips", "packed_batches", "blend_trips", "phase2_wave_trips", "stateless_lane_utilisation", "phase2_lane_utilisation")
        L.append("| %s | %.0f | %.3f | %.3f | %.3f | %s |" % (os.path.basename(f)[:-4], j["value"], j["ms_per_step"], rf["ms_per_launch"], rf["frac"],
                                                            ", ".join("%s %s" % (k, ("%.3f" % v) if isinstance(v, float) else v) for k, v in kc.items() if k in keep)))
    L.append("")
open(os.path.join(dst, "compositing_packed.md"), "w").write("\n".join(L) + "\n")

# ---- projection_hoist.md
d = os.path.join(GO, "r05_hoist")
if os.path.isdir(d):
    L = ["# Projection with the per-Gaussian hoist through memory (option `pre_hoist`; `tools/ab_r05e.sh`, alternating runs on one box)", "",
         "| run | views/s | projection ms | binning ms | compositing ms |", "|---|---:|---:|---:|---:|"]
    for f in sorted(glob.glob(os.path.join(d, "*.log"))):
        j = lj(f)
        if j:
            st = j["roofline"]["stage_ms_per_step"]
            L.append("| %s | %.0f | %.3f | %.3f | %.3f |" % (os.path.basename(f)[:-4].replace("_h0", " hoist off").replace("_h1", " hoist ON"), j["value"], st["preprocess"], st["binning"], st["compositing"]))
    open(os.path.join(dst, "projection_hoist.md"), "w").write("\n".join(L) + "\n")
print(open(os.path.join(dst, "summary.md")).read())
