"""Round-4 additions to a profile directory assembled by tools/assemble_profile.py from a tools/collect_r04.sh run:
    python tools/assemble_r04.py gpurun_out/<tag> profiles/<name> [gpurun_out/<ab tag>]
compositing_before_after.md (round 3's compositing arithmetic rebuilt and profiled in the same run), projection_valu_split.md (float64 /
transcendental instruction split of preprocess_kernel), real_data.md (the real image's merged set: bench line, kernel table, counters),
culled.md, unet.md (copied), ab_tail_fused.md (the A/B lines of tools/ab_r04b.sh when given)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_profile import short  # noqa: E402


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def stats(d):
    f = glob.glob(os.path.join(d, "*kernel_stats.csv"))
    return {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])) for r in csv.DictReader(open(f[0]))} if f else {}


def last_json(path):
    if not os.path.exists(path):
        return None
    lines = [x for x in open(path) if x.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def pick(d, prefix):
    return next((v for k, v in d.items() if k.startswith(prefix)), None)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    # ---- compositing before / after
    out = ["# Compositing kernel, round 3's arithmetic (`-DF3DG_FAST_R03=1 -DF3DG_R3_B128=0`) and round 4's, same box, same run", "",
           "| build | avg us (rocprofv3 --kernel-trace --stats) | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | lane utilisation | SQ_INSTS_LDS | SQ_LDS_BANK_CONFLICT | SQ_LDS_IDX_ACTIVE | SQ_INSTS_VALU_TRANS_F32 |",
           "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for label, st, p1, p2 in (("before (round 3)", "before_stats", "before_pmc_sq", "before_pmc_sq2"), ("after (round 4)", "stats", "pmc_sq", "pmc_sq2")):
        s, a, b = pick(stats(os.path.join(src, st)), "render3s_fwd_kernel"), pick(counters(os.path.join(src, p1)), "render3s_fwd_kernel"), pick(counters(os.path.join(src, p2)), "render3s_fwd_kernel")
        if s and a and b:
            out.append(f"| {label} | {s[1]:.1f} | {a['SQ_INSTS_VALU']:.4g} | {a['SQ_ACTIVE_INST_VALU']:.4g} | {a['SQ_THREAD_CYCLES_VALU'] / (64 * a['SQ_ACTIVE_INST_VALU']):.3f} | "
                       f"{a['SQ_INSTS_LDS']:.4g} | {b['SQ_LDS_BANK_CONFLICT']:.4g} | {b['SQ_LDS_IDX_ACTIVE']:.4g} | {b['SQ_INSTS_VALU_TRANS_F32']:.4g} |")
    open(os.path.join(dst, "compositing_before_after.md"), "w").write("\n".join(out) + "\n")
    # ---- projection: float64 / transcendental split
    f64 = pick(counters(os.path.join(src, "pmc_f64")), "preprocess_kernel")
    sq = pick(counters(os.path.join(src, "pmc_sq")), "preprocess_kernel")
    s = pick(stats(os.path.join(src, "stats")), "preprocess_kernel")
    if f64 and sq and s:
        waves = sq["SQ_WAVES"]
        out = ["# `preprocess_kernel<false>` at C2 (23.6 M (view, Gaussian) pairs per launch): where the VALU instructions go", "",
               f"* {s[1]:.1f} us per launch; SQ_INSTS_VALU {sq['SQ_INSTS_VALU']:.4g} = {sq['SQ_INSTS_VALU'] / waves:.0f} per wave (= per pair), SALU {sq['SQ_INSTS_SALU'] / waves:.0f}, "
               f"lane utilisation {sq['SQ_THREAD_CYCLES_VALU'] / (64 * sq['SQ_ACTIVE_INST_VALU']):.3f}",
               "", "| counter | per launch | per pair |", "|---|---:|---:|"]
        for k in sorted(f64):
            out.append(f"| {k} | {f64[k]:.4g} | {f64[k] / waves:.1f} |")
        tot64 = sum(f64.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64"))
        out += ["", f"float64 add + fma + mul = {tot64 / waves:.0f} of {sq['SQ_INSTS_VALU'] / waves:.0f} instructions per pair ({100 * tot64 / sq['SQ_INSTS_VALU']:.0f} %); "
                    f"float64 transcendentals (v_rcp_f64 / v_rsq_f64 / v_sqrt_f64) {f64.get('SQ_INSTS_VALU_FLOPS_FP64_TRANS', 0) / waves:.1f}, float32 {f64.get('SQ_INSTS_VALU_FLOPS_FP32_TRANS', 0) / waves:.1f} per pair."]
        open(os.path.join(dst, "projection_valu_split.md"), "w").write("\n".join(out) + "\n")
    # ---- real data
    d = last_json(os.path.join(src, "bench_real.log"))
    if d:
        rf, kc = d["roofline"], d["roofline"].get("kernel_counters") or {}
        out = ["# The real image's merged set (fixture F6's image at 256^2 through the build's predictor + cycle aggregation: 589,824 Gaussians, 128-view orbit)", "",
               f"* `bench.py --data real`: **{d['value']:.0f} views/s** ({d['value_in_hbm']:.0f} in HBM, {d.get('value_exact', 0):.0f} in the reference's arithmetic), {d['ms_per_step']:.2f} ms per step: "
               + ", ".join(f"{k} {v:.2f}" for k, v in rf["stage_ms_per_step"].items()) + " ms",
               f"* compositing: `frac` {rf['frac']:.3f} on the handed entries, {rf.get('frac_on_staged_entries', 0):.3f} on the staged ones; counters of one step: "
               + ", ".join(f"{k} {v:.4g}" if isinstance(v, (int, float)) else "" for k, v in kc.items() if k != "note"), ""]
        st = stats(os.path.join(src, "real_stats"))
        out += ["| kernel | calls | avg us | % |", "|---|---:|---:|---:|"] + [f"| `{k}` | {v[0]} | {v[1]:.1f} | {v[2]:.2f} |" for k, v in list(st.items())[:14]]
        sq = pick(counters(os.path.join(src, "real_pmc_sq")), "render3s_fwd_kernel")
        fe, wr = pick(counters(os.path.join(src, "real_pmc_fetch")), "render3s_fwd_kernel"), pick(counters(os.path.join(src, "real_pmc_write")), "render3s_fwd_kernel")
        if sq:
            out += ["", f"compositing kernel PMC: SQ_INSTS_VALU {sq['SQ_INSTS_VALU']:.4g}, lane utilisation {sq['SQ_THREAD_CYCLES_VALU'] / (64 * sq['SQ_ACTIVE_INST_VALU']):.3f}, "
                        f"SQ_WAVE_CYCLES {sq['SQ_WAVE_CYCLES']:.4g}, SQ_BUSY_CYCLES {sq['SQ_BUSY_CYCLES']:.4g}"
                        + (f"; FETCH_SIZE x 2 = {2 * fe['FETCH_SIZE'] / 1e6:.2f} GB, WRITE_SIZE {wr['WRITE_SIZE'] / 1e6:.2f} GB per launch" if fe and wr else "")]
        open(os.path.join(dst, "real_data.md"), "w").write("\n".join(out) + "\n")
    for f in ("culled.md", "unet.md"):
        if os.path.exists(os.path.join(src, f)):
            txt = [x for x in open(os.path.join(src, f)) if "amdgpu.ids" not in x]
            open(os.path.join(dst, f), "w").write("".join(txt))
    if len(sys.argv) > 3:
        ab = sys.argv[3]
        out = ["# A/B lines of round 4's second GPU session (`tools/ab_r04b.sh`): per-stage ms min / median / max over the steps, frame hash", ""]
        for f, title in (("ab_fused.log", "## option sort_fused_rects (the rectangle gather inside a view's last depth pass)"),
                         ("ab_tail.log", "## option render_tail = N (the tail schedule of the one-wave compositing kernel from at most N live pixels on)"),
                         ("../r04c/ab_pre_order.log", "## option pre_order (1: chunk-major projection grid, 2: streaming stores for the record + ellipse, 3: both) -- `tools/ab_r04c.sh`")):
            p = os.path.join(ab, f)
            if os.path.exists(p):
                out += [title, "", "```"] + [x.rstrip() for x in open(p) if "amdgpu.ids" not in x] + ["```", ""]
        open(os.path.join(dst, "ab_tail_fused.md"), "w").write("\n".join(out) + "\n")
    shutil.copy(os.path.join(HERE, "collect_r04.sh"), os.path.join(dst, "collect_r04.sh.txt"))


if __name__ == "__main__":
    main()
