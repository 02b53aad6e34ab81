"""Builds the whole committed profile directory from one `tools/collect_r02.sh <tag>` run:
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

    # ---- integrate
    ilog = os.path.join(src, "bench_integrate.log")
    if os.path.exists(ilog):
        out = ["# integrate (Gaussians -> points), 1 M points @256x256 -- `python tools/bench_integrate.py`", "", "```"] + \
              [l for l in open(ilog).read().splitlines() if l.startswith(("integrate P", "  pass 1", "mesh-extraction"))] + ["```", ""]
        for c, label in ((0, "196,608 Gaussians, sigma0 = 0.01"), (1, "589,824 Gaussians, sigma0 = 0.01")):
            st = glob.glob(os.path.join(src, f"int_stats{c}", "*kernel_stats.csv"))
            if st:
                out += [f"rocprofv3 --kernel-trace --stats of `CFG={c} python tools/bench_integrate.py` ({label}; 7 calls):", ""] + kernel_table(st[0], 12) + [""]
        open(os.path.join(dst, "integrate.md"), "w").write("\n".join(out))

    # ---- every bench line of the run
    out = ["# Bench lines of the evidence run (tools/collect_r02.sh), one MI355X", ""]
    for log, cmd in (("bench_default", "python bench.py` (C2, fast arithmetic = the default)"),
                     ("bench_exact", "python bench.py --render-mode exact --no-cpu-baseline`"),
                     ("bench_nocull", "python bench.py --tile-cull 0 --no-cpu-baseline` (the reference's tile lists)"),
                     ("bench_c5", "python bench.py --workload c5 --steps 3 --warmup 1`"),
                     ("bench_c4_fp32", "python bench.py --workload c4 --images 16 --steps 2 --warmup 1`"),
                     ("bench_c4_bf16", "python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone bf16`"),
                     ("bench_c4_bf16_64", "python bench.py --workload c4 --images 64 --steps 1 --warmup 1 --backbone bf16`")):
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
