"""Builds the whole committed profile directory from one `tools/collect_r03.sh <tag>` run:
    python tools/assemble_profile.py gpurun_out/<tag> profiles/<name>
= tools/make_profile.py (summary.md, bench_kernel_stats.csv, traffic.json) + c2_sq_counters.md (second SQ pass), c5.md and
c5_kernel_stats.csv (C5 forward + backward: bench line, kernel table, PMC passes), bench_lines.md (every bench line of the run),
parity_report.md and the tail of the GPU test log."""
import csv
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_profile import counters, last_json, short  # noqa: E402


def mean(v):
    return sum(v) / len(v) if v else 0.0


def kernel_table(stats_csv, limit=24):
    rows = list(csv.DictReader(open(stats_csv)))
    out = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:limit]:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    subprocess.check_call([sys.executable, os.path.join(HERE, "make_profile.py"), src, dst], stdout=subprocess.DEVNULL)

    # ---- second SQ pass of the C2 bench
    sq2 = counters(os.path.join(src, "pmc_sq2"))
    names = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
             "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VMEM_RD"]
    out = ["# C2 bench, second SQ counter pass (rocprofv3 --pmc, PMC only), per dispatch (mean)", "",
           "| kernel | " + " | ".join(names) + " |", "|---|" + "---:|" * len(names)]
    for k in sorted(sq2, key=lambda k: -mean(sq2[k].get("SQ_ACTIVE_INST_ANY", [0])))[:10]:
        out.append(f"| `{k}` | " + " | ".join(f"{mean(sq2[k].get(n, [])):.3g}" for n in names) + " |")
    open(os.path.join(dst, "c2_sq_counters.md"), "w").write("\n".join(out) + "\n")

    # ---- effective clock of the compositing launch (GRBM_GUI_ACTIVE / duration) and the exact-arithmetic kernel
    grbm = counters(os.path.join(src, "pmc_grbm"))
    st_fast = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
    st_exact = glob.glob(os.path.join(src, "stats_exact", "*kernel_stats.csv"))
    extra = ["", "# Compositing kernel, both arithmetic modes (rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-d2h`)", ""]
    for label, st in (("fast (default)", st_fast), ("exact (--render-mode exact)", st_exact)):
        if st:
            for r in csv.DictReader(open(st[0])):
                if short(r["Name"]).startswith(("render3s_fwd_kernel", "render3_fwd_kernel")):
                    extra.append(f"* {label}: `{short(r['Name'])}` {r['Calls']} launches, average {float(r['AverageNs']) / 1e3:.1f} us")
    for k, c in grbm.items():
        if k.startswith(("render3s_fwd_kernel", "render3_fwd_kernel")) and c.get("GRBM_GUI_ACTIVE"):
            cyc = mean(c["GRBM_GUI_ACTIVE"])
            us = [float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(st_fast[0])) if short(r["Name"]) == k]
            if us:
                extra.append(f"* `{k}`: GRBM_GUI_ACTIVE {cyc:.3g} cycles per launch (summed over the 8 XCDs) / 8 / {us[0]:.1f} us = "
                             f"{cyc / 8 / us[0] / 1e3:.2f} GHz effective clock during the launch (2.4 GHz peak: the chip clocks to its power budget)")
    open(os.path.join(dst, "c2_sq_counters.md"), "a").write("\n".join(extra) + "\n")

    # ---- sigma0 = 0.05 (SURVEY 8d's second sweep), the larger set, the small-call path: bench line + kernel table each
    for tag, title, log, stats in (
            ("sigma005", "C2 at sigma0 = 0.05 -- `python bench.py --sigma0 0.05`", "bench_sigma005.log", "s005_stats"),
            ("larger_set", "589,824 Gaussians x 128 views @256x256 -- `python bench.py --gaussians 589824 --views 128`", "bench_589k.log", "l589_stats"),
            ("small_calls", "One view of 65,536 Gaussians per call (the small-call path) -- `python tools/prof_small.py 65536 1`", "small_calls.log", "small_stats")):
        st = glob.glob(os.path.join(src, stats, "*kernel_stats.csv"))
        path = os.path.join(src, log)
        out = ["# " + title, ""]
        if os.path.exists(path):
            line = last_json(path)
            out += ["```", line if line else "\n".join(l for l in open(path).read().splitlines() if l.startswith("P=")), "```", ""]
        if st:
            out += ["rocprofv3 --kernel-trace --stats (3-5 steps):", ""] + kernel_table(st[0], 28) + [""]
        if tag == "sigma005":
            fetch, write, sq = (counters(os.path.join(src, d)) for d in ("s005_pmc_fetch", "s005_pmc_write", "s005_pmc_sq"))
            out += ["PMC, separate passes (per dispatch, mean; FETCH x2 = gfx950 correction, WRITE_SIZE uncalibrated):", "",
                    "| kernel | FETCH x2 MB | WRITE MB | SQ_INSTS_VALU | lane utilisation |", "|---|---:|---:|---:|---:|"]
            for k in sorted(fetch, key=lambda k: -sum(fetch[k]["FETCH_SIZE"]))[:8]:
                m = {n: mean(sq.get(k, {}).get(n, [])) for n in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU")}
                lanes = m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if m["SQ_ACTIVE_INST_VALU"] else 0.0
                out.append(f"| `{k}` | {2 * mean(fetch[k]['FETCH_SIZE']) * 1024 / 1e6:.1f} | {mean(write.get(k, {}).get('WRITE_SIZE', [])) * 1024 / 1e6:.1f} | "
                           f"{m['SQ_INSTS_VALU']:.3g} | {lanes:.2f} |")
        open(os.path.join(dst, tag + ".md"), "w").write("\n".join(out) + "\n")

    # ---- C5
    c5_stats = glob.glob(os.path.join(src, "c5_stats", "*kernel_stats.csv"))
    if c5_stats:
        shutil.copy(c5_stats[0], os.path.join(dst, "c5_kernel_stats.csv"))
        fetch, write, sq = (counters(os.path.join(src, d)) for d in ("c5_pmc_fetch", "c5_pmc_write", "c5_pmc_sq"))
        out = ["# C5 (1,000,000 Gaussians, 32 views @512x512, forward with auxiliary planes + backward) -- `python bench.py --workload c5 "
               "--steps 3 --warmup 1`", "", "Bench line:", "", "```", last_json(os.path.join(src, "bench_c5.log")) or "(missing)", "```", "",
               "rocprofv3 --kernel-trace --stats of the same command:", ""] + kernel_table(c5_stats[0], 20)
        out += ["", "PMC, separate passes (per dispatch, mean; FETCH x2 = gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE uncalibrated):", "",
                "| kernel | FETCH x2 MB | WRITE MB | SQ_INSTS_VALU | lane utilisation | SQ_INSTS_LDS | SQ_WAVE_CYCLES |", "|---|---:|---:|---:|---:|---:|---:|"]
        for k in sorted(fetch, key=lambda k: -sum(fetch[k]["FETCH_SIZE"]) - sum(write.get(k, {}).get("WRITE_SIZE", [0])))[:12]:
            f_mb = 2 * mean(fetch[k]["FETCH_SIZE"]) * 1024 / 1e6
            w_mb = mean(write.get(k, {}).get("WRITE_SIZE", [])) * 1024 / 1e6
            m = {n: mean(sq.get(k, {}).get(n, [])) for n in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES")}
            lanes = m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if m["SQ_ACTIVE_INST_VALU"] else 0.0
            out.append(f"| `{k}` | {f_mb:.1f} | {w_mb:.1f} | {m['SQ_INSTS_VALU']:.3g} | {lanes:.2f} | {m['SQ_INSTS_LDS']:.3g} | {m['SQ_WAVE_CYCLES']:.3g} |")
        open(os.path.join(dst, "c5.md"), "w").write("\n".join(out) + "\n")
        # the same counters as a record bench.py --workload c5 can quote beside the formula (roofline.traffic_from_profiles)
        kb = next((k for k in fetch if k.startswith("render3_bwd_kernel")), None)
        if kb and write.get(kb):
            import json
            stk = next((r for r in csv.DictReader(open(c5_stats[0])) if short(r["Name"]) == kb), None)
            rec = {"kernel": kb, "config": {"gaussians": 1000000, "views": 32, "resolution": 512},
                   "FETCH_SIZE_KB_per_launch": mean(fetch[kb]["FETCH_SIZE"]), "WRITE_SIZE_KB_per_launch": mean(write[kb]["WRITE_SIZE"]),
                   "traffic_bytes_per_launch": 2 * mean(fetch[kb]["FETCH_SIZE"]) * 1024 + mean(write[kb]["WRITE_SIZE"]) * 1024,
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --workload c5`; FETCH_SIZE doubled per MI355X_MICROARCH.md"}
            if stk:
                rec["kernel_us_rocprof"] = float(stk["AverageNs"]) / 1e3
            sqk = sq.get(kb)
            if sqk and sqk.get("SQ_ACTIVE_INST_VALU"):
                rec["valu"] = {"SQ_INSTS_VALU": mean(sqk["SQ_INSTS_VALU"]),
                               "lane_utilisation": round(mean(sqk["SQ_THREAD_CYCLES_VALU"]) / (64 * mean(sqk["SQ_ACTIVE_INST_VALU"])), 3)}
                if stk:
                    rec["valu"]["issue_utilisation_lower_bound"] = round(mean(sqk["SQ_INSTS_VALU"]) * 2 / (1024 * float(stk["AverageNs"]) * 1e-9 * 2.4e9), 3)
            json.dump(rec, open(os.path.join(dst, "c5_traffic.json"), "w"), indent=1)

    # ---- integrate
    ilog = os.path.join(src, "bench_integrate.log")
    if os.path.exists(ilog):
        out = ["# integrate (Gaussians -> points), 1 M points @256x256 -- `python tools/bench_integrate.py`", "", "```"] + \
              [l for l in open(ilog).read().splitlines() if l.startswith(("integrate P", "  pass 1", "mesh-extraction", "AlphaSweep"))] + ["```", ""]
        for c, label in ((0, "196,608 Gaussians, sigma0 = 0.01"), (1, "589,824 Gaussians, sigma0 = 0.01")):
            st = glob.glob(os.path.join(src, f"int_stats{c}", "*kernel_stats.csv"))
            if st:
                out += [f"rocprofv3 --kernel-trace --stats of `CFG={c} python tools/bench_integrate.py` ({label}; 7 calls):", ""] + kernel_table(st[0], 12) + [""]
        st = glob.glob(os.path.join(src, "int_sweep_stats", "*kernel_stats.csv"))
        if st:
            out += ["rocprofv3 --kernel-trace --stats of `V=16 python tools/bench_integrate.py` (the whole script incl. the AlphaSweep preparations of 16 cameras, "
                    "1 / 4 / 16 per call: the 16-camera launch of integrate_pass1_rays_kernel is its maximum):", ""] + kernel_table(st[0], 8) + [""]
        # the per-pixel pass of a 16-camera preparation on its own (tools/prof_pass1.py, tools/pmc_pass1.sh): kernel time and SQ counters
        st = glob.glob(os.path.join(src, "p1_stats", "*kernel_stats.csv"))
        if st:
            out += ["rocprofv3 --kernel-trace --stats of `python tools/prof_pass1.py` (three preparations of 16 cameras of 589,824 Gaussians; "
                    "integrate_pass1_cull_kernel behind the ray kernel only computes tiles that reached 1,024 contributors: none here):", ""] + kernel_table(st[0], 6) + [""]
        rows = {}
        for d in ("sq1", "sq2"):
            f = glob.glob(os.path.join(src, "p1_pmc", d, "*counter_collection.csv"))
            if f:
                for r in csv.DictReader(open(f[0])):
                    if "integrate_pass1_rays" in r["Kernel_Name"]:
                        rows.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if rows:
            m = {k: sum(v) / len(v) for k, v in rows.items()}
            out += ["`integrate_pass1_rays_kernel`, SQ counters per 16-camera launch (rocprofv3 --pmc, PMC-only passes; round 2's "
                    "integrate_pass1_cull_kernel on the same input: SQ_INSTS_VALU 9.0e9, SQ_INSTS_VALU_MUL_F64 2.46e8, lane utilisation 0.58, 13.4 ms):", "",
                    "| counter | value |", "|---|---:|"] + [f"| {k} | {v:.4g} |" for k, v in sorted(m.items())]
            if m.get("SQ_ACTIVE_INST_VALU"):
                out += [f"| VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) | {m.get('SQ_THREAD_CYCLES_VALU', 0) / (64 * m['SQ_ACTIVE_INST_VALU']):.2f} |"]
            out += [""]
        open(os.path.join(dst, "integrate.md"), "w").write("\n".join(out))

    # ---- every bench line of the run
    out = ["# Bench lines of the evidence run (tools/collect_r0N.sh), one MI355X", ""]
    for log, cmd in (("bench_default", "python bench.py` (C2; fast arithmetic, the reference arithmetic and the D2H-inclusive rate in one line)"),
                     ("bench_sigma005", "python bench.py --sigma0 0.05 --no-cpu-baseline` (SURVEY 8d's second sweep)"),
                     ("bench_nocull", "python bench.py --tile-cull 0 --no-cpu-baseline` (the reference's tile lists)"),
                     ("bench_589k", "python bench.py --gaussians 589824 --views 128 --no-cpu-baseline` (the merged set of a final orbit)"),
                     ("bench_real", "python bench.py --data real` (the real image's merged set, 589,824 Gaussians x 128 views)"),
                     ("bench_dropin", "python bench.py --workload dropin --views 60` (render_predicted_more_v2_gof, one view per call, 65,536 Gaussians)"),
                     ("bench_dropin_589k", "python bench.py --workload dropin --views 60 --gaussians 589824`"),
                     ("bench_c5", "python bench.py --workload c5 --steps 3 --warmup 1`"),
                     ("bench_c4_fp32", "python bench.py --workload c4 --images 16 --steps 2 --warmup 1`"),
                     ("bench_c4_bf16", "python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone bf16`"),
                     ("bench_c4_bf16_64", "python bench.py --workload c4 --images 64 --steps 1 --warmup 1 --backbone bf16`"),
                     ("bench_c4_fp32_nchw", "python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone-layout nchw`"),
                     ("bench_c4_bf16_nchw", "python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone bf16 --backbone-layout nchw`")):
        path = os.path.join(src, log + ".log")
        if os.path.exists(path):
            out += ["`" + cmd + ":", "", "```", last_json(path) or "(no line: " + open(path).read()[-300:] + ")", "```", ""]
    small = os.path.join(src, "small_calls.log")
    if os.path.exists(small):
        out += ["`python tools/bench_small_calls.py` (per-call cost of small batched renders):", "", "```"] + \
               [l for l in open(small).read().splitlines() if l.startswith("P=")] + ["```", ""]
    log = os.path.join(src, "pytest_gpu.log")
    if os.path.exists(log):
        out += ["`python -m pytest tests -m gpu -q` on the same box:", "", "```"] + open(log).read().splitlines()[-3:] + ["```", ""]
    open(os.path.join(dst, "bench_lines.md"), "w").write("\n".join(out))
    if os.path.exists(os.path.join(src, "parity_report.md")):
        shutil.copy(os.path.join(src, "parity_report.md"), os.path.join(dst, "parity_report.md"))
    print("\n".join(sorted(os.listdir(dst))))


if __name__ == "__main__":
    main()
