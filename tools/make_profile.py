"""Builds a committed profile directory from one `tools/collect_profiles.sh <tag>` run (gpurun_out/<tag>/):
    python tools/make_profile.py gpurun_out/<tag> profiles/<name>
writes summary.md (kernel table of the rocprofv3 --kernel-trace --stats run, FETCH_SIZE / WRITE_SIZE and SQ counter tables
of the separate PMC passes, the two bench lines), bench_kernel_stats.csv and traffic.json (what bench.py reads back for
roofline.traffic / roofline.valu on the same configuration)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

SIMDS = 1024          # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
RENDER_KERNELS = ("render3s_fwd_kernel", "render3_fwd_kernel", "render2_fwd_kernel", "render_fwd_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def counters(d):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, "bench_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    out = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for r in rows:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    fetch, write, sq = counters(os.path.join(src, "pmc_fetch")), counters(os.path.join(src, "pmc_write")), counters(os.path.join(src, "pmc_sq"))
    out += ["", "PMC (separate passes; KB per dispatch as reported by rocprofv3; `FETCH x2` applies the gfx950 correction of "
            "MI355X_MICROARCH.md section HBM for wide coalesced reads; WRITE_SIZE is uncalibrated):", "",
            "| kernel | dispatches | FETCH_SIZE KB (mean) | FETCH x2 MB | WRITE_SIZE KB (mean) | WRITE MB |", "|---|---:|---:|---:|---:|---:|"]
    for k in sorted(fetch, key=lambda k: -sum(fetch[k]["FETCH_SIZE"])):
        fv, wv = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", [0.0])
        fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
        out.append(f"| `{k}` | {len(fv)} | {fm:.1f} | {2 * fm * 1024 / 1e6:.1f} | {wm:.1f} | {wm * 1024 / 1e6:.1f} |")
    names = ["SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES"]
    out += ["", "SQ counters per dispatch (mean), PMC-only pass:", "", "| kernel | " + " | ".join(names) + " | VALU lane utilisation |",
            "|---|" + "---:|" * (len(names) + 1)]
    valu = None
    for k in sorted(sq, key=lambda k: -sum(sq[k].get("SQ_INSTS_VALU", [0]))):
        m = {n: (sum(sq[k][n]) / len(sq[k][n]) if sq[k].get(n) else 0.0) for n in names}
        if m["SQ_INSTS_VALU"] < 1e6:
            continue
        lanes = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64) if m["SQ_ACTIVE_INST_VALU"] else 0.0
        out.append(f"| `{k}` | " + " | ".join(f"{m[n]:.3g}" for n in names) + f" | {lanes:.2f} |")
        if k.startswith(RENDER_KERNELS) and valu is None:
            valu = {"lane_utilisation": round(lanes, 3), "SQ_INSTS_VALU": m["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": m["SQ_ACTIVE_INST_VALU"],
                    "SQ_BUSY_CYCLES": m["SQ_BUSY_CYCLES"], "SQ_WAVE_CYCLES": m["SQ_WAVE_CYCLES"],
                    "note": "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); issue_utilisation_lower_bound = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz)"}
    default_line = last_json(os.path.join(src, "bench_default.log"))
    prof_line = last_json(os.path.join(src, "bench_under_rocprof.log"))
    out += ["", "Default `python bench.py` line of the same build (N=1, steps 10, warmup 3):", "", "```", default_line or "(missing)", "```", "",
            "Bench line under `rocprofv3 --kernel-trace --stats` (steps 5, warmup 2):", "", "```", prof_line or "(missing)", "```"]
    open(os.path.join(dst, "summary.md"), "w").write("\n".join(out) + "\n")

    cfg = json.loads(default_line)["config"]
    rk = sorted((k for k in fetch if k.startswith(RENDER_KERNELS)), key=lambda k: -sum(fetch[k]["FETCH_SIZE"]))[0]
    f_kb = sum(fetch[rk]["FETCH_SIZE"]) / len(fetch[rk]["FETCH_SIZE"])
    w_kb = sum(write[rk]["WRITE_SIZE"]) / len(write[rk]["WRITE_SIZE"])
    kernel_us = [float(r["AverageNs"]) / 1e3 for r in rows if short(r["Name"]) == rk][0]
    if valu:
        cycles = kernel_us * 1e-6 * 2.4e9          # 2.4 GHz peak engine clock
        # every wave64 VALU instruction occupies its SIMD for >= 2 cycles (32 lanes/cycle, MI355X_MICROARCH.md "Wave
        # scheduling"); float64 and transcendental ones for longer, so this is a LOWER bound of the issue-slot utilisation
        valu["issue_utilisation_lower_bound"] = round(valu["SQ_INSTS_VALU"] * 2 / (SIMDS * cycles), 3)
        valu["kernel_us_rocprof"] = kernel_us
        # the same at the clock the chip actually held during the launch (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8), with the
        # quarter-rate transcendentals (v_exp / v_rcp / v_rsq: 8 cycles instead of 2) counted: the fraction of all SIMD cycles of the
        # launch in which a VALU instruction was issuing
        try:
            grbm = counters(os.path.join(src, "pmc_grbm"))
            sq2 = counters(os.path.join(src, "pmc_sq2"))
            g = [v for k, v in grbm.items() if k == rk and v.get("GRBM_GUI_ACTIVE")]
            t2 = [v for k, v in sq2.items() if k == rk and v.get("SQ_INSTS_VALU_TRANS_F32")]
            if g:
                cyc = sum(g[0]["GRBM_GUI_ACTIVE"]) / len(g[0]["GRBM_GUI_ACTIVE"]) / 8.0
                trans = (sum(t2[0]["SQ_INSTS_VALU_TRANS_F32"]) / len(t2[0]["SQ_INSTS_VALU_TRANS_F32"])) if t2 else 0.0
                valu["effective_clock_ghz"] = round(cyc / (kernel_us * 1e3), 3)
                valu["valu_issue_frac"] = round((valu["SQ_INSTS_VALU"] * 2 + trans * 6) / (SIMDS * cyc), 3)
        except Exception as ex:      # (older runs have no GRBM pass)
            valu["valu_issue_frac_error"] = str(ex)
    # HBM traffic of the other two stages per forward call: every kernel of the call that is not the projection / compositing kernel is
    # binning; bytes = 2 x FETCH_SIZE + WRITE_SIZE summed over a call's dispatches (calls = dispatches of preprocess_kernel)
    n_calls = max(sum(len(v.get("FETCH_SIZE", [])) for k, v in fetch.items() if k.startswith("preprocess_kernel")), 1)
    def stage_bytes(pred):
        tot = 0.0
        for k in fetch:
            if pred(k):
                tot += 2 * sum(fetch[k]["FETCH_SIZE"]) * 1024 + sum(write.get(k, {}).get("WRITE_SIZE", [])) * 1024
        return tot / n_calls
    own = lambda k: not k.startswith(("at::", "__amd", "void at", "elementwise", "vectorized", "Cijk", "measured"))
    stages = {"preprocess": {"bytes_per_launch": stage_bytes(lambda k: k.startswith("preprocess_kernel")), "note": "PMC, 2 x FETCH_SIZE + WRITE_SIZE"},
              "binning": {"bytes_per_launch": stage_bytes(lambda k: own(k) and not k.startswith(RENDER_KERNELS + ("preprocess_kernel", "pack_frames"))),
                          "note": "PMC, 2 x FETCH_SIZE + WRITE_SIZE summed over the binning kernels of one call"}}
    t = {"kernel": rk, "stages": stages,
         "config": {"gaussians": cfg["gaussians"], "views": cfg["views"], "resolution": cfg["resolution"], "views_per_call": cfg["views_per_call"],
                    "render_mode": cfg.get("render_mode", "exact"), "tile_cull": cfg.get("tile_cull", 0), "sigma0": cfg.get("sigma0", 0.01)},
         "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
         "traffic_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024, "valu": valu,
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncalibrated"}
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "summary.md")).read()[:3000])
    print(json.dumps(t, indent=1))


if __name__ == "__main__":
    main()
