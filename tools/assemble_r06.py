"""Assembles profiles/r06_final/ from a tools/collect_r06.sh run:   python tools/assemble_r06.py gpurun_out/r06 profiles/r06_final
summary.md (bench lines), bench_kernel_stats.csv, traffic.json + c5_traffic.json (the formats bench.py reads), compositing_scan.md (the
split-pixel mode against render3s: bench lines, kernel counters, PMC of the timed launches), backward_dense.md (lock-step against dense
compositing backward), the pytest log. The staging replay and the FETCH_SIZE calibration of the lab build are copied by hand (notes/r06.md)."""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def lj(name):
    path = os.path.join(src, name + ".log")
    if not os.path.exists(path):
        return None
    l = [x for x in open(path) if x.startswith("{")]
    return json.loads(l[-1]) if l else None


def pmc(tag):
    p = os.path.join(src, tag, "summary.txt")
    if not os.path.exists(p):
        return None
    s = json.load(open(p))
    return max(s.items(), key=lambda kv: kv[1].get("dispatches_averaged", 0))


rows = [("C2 default (196,608 Gaussians, 120 views, fast arithmetic; frame copy issued by the step's own thread)", "bench_default"),
        ("the same, frame copy issued by a second host thread (rounds 2-5)", "bench_default_thread"),
        ("C2 with F3DG_FLAG_SCAN (split-pixel schedule)", "bench_default_scan"),
        ("C2, channels rgb + depth + alpha (the build's own loops)", "bench_lean"), ("sigma0 = 0.05", "bench_sigma005"),
        ("589,824 synthetic x 128 views", "bench_589k"), ("real image, merged 589,824 x 128 views (--data real)", "bench_real"),
        ("the same with F3DG_FLAG_SCAN", "bench_real_scan"), ("real, channels rgb + depth + alpha", "bench_real_lean"),
        ("the same with F3DG_FLAG_SCAN", "bench_real_lean_scan")]
out = ["# Round 6 evidence run (`tools/collect_r06.sh`, one MI355X box; library " + (open(os.path.join(src, "version.txt")).read().strip() if os.path.exists(os.path.join(src, "version.txt")) else "?") + ")", "",
       "| line | views/s (`value`; in HBM) | reference arithmetic (`value_exact`) | ms/step | projection | binning | compositing | `frac` | slowest / median call (ms) |",
       "|---|---:|---:|---:|---:|---:|---:|---:|---|"]
for label, f in rows:
    d = lj(f)
    if not d:
        continue
    r = d["roofline"]
    st = r["stage_ms_per_step"]
    sp = (d.get("call_ms_spread") or {}).get("all_stages") or {}
    out.append("| %s | %.0f; %.0f | %s | %.2f | %.2f | %.2f | %.2f | %.3f | %.2f / %.2f |" % (
        label, d["value"], d.get("value_in_hbm", 0), ("%.0f" % d["value_exact"]) if d.get("value_exact") else "", d["ms_per_step"], st["preprocess"], st["binning"],
        st["compositing"], r["frac"], sp.get("max", 0), sp.get("median", 0)))
d = lj("bench_default")
if d and d.get("with_d2h"):
    leg = d["with_d2h"]["uint8_rgb"].get("leg_alone_ms")
    out += ["", "Frame copy leg alone (nothing else on the device): %s" % json.dumps(leg)]
for name in ("bench_default", "bench_real"):
    d = lj(name)
    live = ((d or {}).get("roofline") or {}).get("traffic_live")
    if live:
        r = d["roofline"]
        out += ["", "`%s`: `roofline.traffic` measured inside the run (two rocprofv3 --pmc child passes, %s s): %s GB per launch of %s "
                "(2 x FETCH_SIZE %.2f GB + WRITE_SIZE %.2f GB; %s dispatches); `frac_on_counter_traffic` %s%s" % (
                    name, live.get("seconds"), ("%.2f" % (live["bytes_per_launch"] / 1e9)) if live.get("bytes_per_launch") else "null",
                    live.get("kernel"), (live.get("fetch_size_bytes_x2") or 0) / 1e9, (live.get("write_size_bytes") or 0) / 1e9,
                    live.get("dispatches_averaged"), ("%.3f" % r["frac_on_counter_traffic"]) if r.get("frac_on_counter_traffic") else "null",
                    ("; error: " + live["error"]) if live.get("error") else "")]
for name in ("bench_dropin", "bench_c4_fp32"):
    d = lj(name)
    if d:
        out += ["", "`%s`: value %.0f views/s, %.3f ms/step; %s" % (name, d["value"], d["ms_per_step"], json.dumps({k: v for k, v in d.items() if k.startswith("value_")}))]
for name in ("bench_c5", "bench_c5_lockstep"):
    d = lj(name)
    if d:
        out += ["", "`%s` (C5: 1 M Gaussians, 32 views @512^2, forward + backward; backward kernel %s): %.0f views/s, %.2f ms/step, stages %s" % (
            name, d["roofline"]["kernel"], d["value"], d["ms_per_step"], json.dumps({k: round(v, 3) for k, v in d["stage_ms_per_step"].items()}))]
open(os.path.join(dst, "summary.md"), "w").write("\n".join(out) + "\n")

for f in glob.glob(os.path.join(src, "stats", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "bench_kernel_stats.csv"))
for f, g in (("pytest_gpu.log", "pytest_gpu_final.log"), ("parity_report.md", "parity_report.md"), ("scan_dist_probe.log", "scan_dist_probe.log")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, g))

# ---- traffic.json (C2 default) for bench.py's profiles_record
p = pmc("pmc_c2_fast")
d = lj("bench_default")
if p and d:
    k, e = p
    c = e["counters"]
    t = {"kernel": k[:70], "config": {"gaussians": 196608, "views": 120, "resolution": 256, "views_per_call": 120, "render_mode": "fast", "tile_cull": 1, "sigma0": 0.01},
         "FETCH_SIZE_KB_per_launch": c.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": c.get("WRITE_SIZE"), "traffic_bytes_per_launch": e["hbm_bytes"]["total"],
         "valu": {"lane_utilisation": e.get("lane_utilisation"), "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"),
                  "kernel_us_rocprof": e["kernel_stats"]["avg_us_all_calls"], "effective_clock_ghz": e.get("effective_clock_ghz"),
                  "valu_issue_frac": e.get("valu_issue_frac"), "lds_busy_frac": e.get("lds_busy_frac"),
                  "note": "means over the dispatches with the largest grid of the kernel name (tools/pmc_kernel.py)"},
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled: calibrated for this kernel's 64-byte record gathers by "
                 "tools/micro/gather64.hip (a miss is one 128-byte line request tallied at 64 bytes: profiles/r06_final/gather64_pmc.txt)"}
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

# ---- compositing_scan.md
L = ["# Split-pixel compositing (`render5_fwd_kernel`, F3DG_FLAG_SCAN) against the default kernel (`render3s_fwd_kernel`), fast arithmetic", "",
     "Bench lines of the same run (`bench.py [--data real] [--channels rgb_depth_alpha] [--scan 1]`; compositing = HIP events of the stage):", "",
     "| workload | default ms | scan ms | default counters | scan counters |", "|---|---:|---:|---|---|"]
for label, a, b in (("real merged set x 128 views, nine channels", "bench_real", "bench_real_scan"), ("real, rgb + depth + alpha", "bench_real_lean", "bench_real_lean_scan"),
                    ("C2", "bench_default", "bench_default_scan")):
    da, db = lj(a), lj(b)
    if da and db:
        ca, cb = da["roofline"].get("kernel_counters") or {}, db["roofline"].get("kernel_counters") or {}
        L.append("| %s | %.2f | %.2f | trips %.3g at %.3f lanes busy, slides %.3g | fused trips %.3g at %.3f + %.3g batches (%.3g pairs, %.3f lanes busy), %.3g pixels compacted |" % (
            label, da["roofline"]["stage_ms_per_step"]["compositing"], db["roofline"]["stage_ms_per_step"]["compositing"],
            ca.get("phase2_wave_trips", 0), ca.get("phase2_lane_utilisation") or 0, ca.get("slides", 0),
            cb.get("fused_trips", 0), cb.get("fused_trip_lane_utilisation") or 0, cb.get("dense_batches", 0), cb.get("pairs_in_dense_batches", 0),
            cb.get("dense_batch_lane_utilisation") or 0, cb.get("pixels_compacted", 0)))
L += ["", "PMC of the timed launches (`tools/pmc_kernel.sh`: one rocprofv3 --pmc pass per counter set, means over the largest-grid dispatches):", "",
      "| launch | kernel | avg us | SQ_INSTS_VALU | SQ_INSTS_SALU | lane utilisation (all VALU) | VALU issue frac | LDS busy | 2 x FETCH_SIZE + WRITE_SIZE |", "|---|---|---:|---:|---:|---:|---:|---:|---:|"]
for label, tag in (("C2", "pmc_c2_fast"), ("C2, scan", "pmc_c2_scan"), ("real x 128", "pmc_real_fast"), ("real x 128, scan", "pmc_real_scan")):
    p = pmc(tag)
    if p:
        k, e = p
        c = e["counters"]
        L.append("| %s | %s | %.0f | %.3e | %.3e | %.3f | %.3f | %.3f | %.2f GB |" % (label, k.split("(")[0][5:45], e["kernel_stats"]["avg_us_all_calls"], c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"],
                                                                                e.get("lane_utilisation", 0), e.get("valu_issue_frac", 0), e.get("lds_busy_frac", 0), e["hbm_bytes"]["total"] / 1e9))
open(os.path.join(dst, "compositing_scan.md"), "w").write("\n".join(L) + "\n")

# ---- backward_dense.md + c5_traffic.json
B = ["# Compositing backward: lock-step walk (`render3_bwd_kernel`, option bwd_dense 0) against dense batches (`render5_bwd_kernel`, the default)", ""]
for name in ("bench_c5_lockstep", "bench_c5"):
    d = lj(name)
    if d:
        B.append("* `%s`: %.0f views/s, %.2f ms/step, stages (ms) %s" % (name, d["value"], d["ms_per_step"], json.dumps({k: round(v, 3) for k, v in d["stage_ms_per_step"].items()})))
for o in (0, 1):
    for f in ("real_train_dense%d.log" % o, "one_view_train_dense%d.log" % o):
        p = os.path.join(src, f)
        if os.path.exists(p):
            B.append("* `%s` (bwd_dense %d): %s" % (f, o, " / ".join(x.strip() for x in open(p).read().strip().splitlines())))
B += ["", "PMC of the C5 launches:", "", "| kernel | avg us | SQ_INSTS_VALU | SQ_INSTS_SALU | SQ_INSTS_LDS | lane utilisation | VALU issue frac | LDS busy | 2 x FETCH_SIZE + WRITE_SIZE |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
for tag in ("pmc_c5_bwd3", "pmc_c5_bwd5"):
    p = pmc(tag)
    if p:
        k, e = p
        c = e["counters"]
        B.append("| %s | %.0f | %.3e | %.3e | %.3e | %.3f | %.3f | %.3f | %.2f GB |" % (k.split("(")[0][5:40], e["kernel_stats"]["avg_us_all_calls"], c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"],
                                                                                  c.get("SQ_INSTS_LDS", 0), e.get("lane_utilisation", 0), e.get("valu_issue_frac", 0), e.get("lds_busy_frac", 0), e["hbm_bytes"]["total"] / 1e9))
        if tag == "pmc_c5_bwd5":
            json.dump({"kernel": k.split("(")[0][5:40], "config": {"gaussians": 1000000, "views": 32, "resolution": 512},
                       "FETCH_SIZE_KB_per_launch": c.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": c.get("WRITE_SIZE"), "traffic_bytes_per_launch": e["hbm_bytes"]["total"],
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --workload c5`; FETCH_SIZE doubled per MI355X_MICROARCH.md",
                       "kernel_us_rocprof": e["kernel_stats"]["avg_us_all_calls"],
                       "valu": {"SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "lane_utilisation": e.get("lane_utilisation"), "valu_issue_frac": e.get("valu_issue_frac")}},
                      open(os.path.join(dst, "c5_traffic.json"), "w"), indent=1)
open(os.path.join(dst, "backward_dense.md"), "w").write("\n".join(B) + "\n")
print("\n".join(out[:16]))
