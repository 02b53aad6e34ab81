"""Condenses the rocprofv3 passes of tools/pmc_kernel.sh: per kernel whose name contains <substring>, the mean counter values over the
dispatches with the largest grid of that name (the timed launches), the average duration of the same dispatches from the kernel
stats, and the derived figures (VALU issue fraction, lane utilisation, LDS busy fraction, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE with
the gfx950 correction of MI355X_MICROARCH.md).   usage: python tools/pmc_kernel.py <dir> <kernel substring>"""
import collections
import csv
import glob
import json
import os
import sys

d, sub = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values of the largest-grid dispatches
for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    big = collections.defaultdict(int)
    for r in rows:
        big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) == big[r["Kernel_Name"]]:
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {}
for f in glob.glob(os.path.join(d, "stats", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
out = {}
for k, cs in vals.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    e = {"dispatches_averaged": n, "counters": {c: round(v) for c, v in sorted(m.items())}}
    if k in stats:
        e["kernel_stats"] = {"calls": stats[k][0], "avg_us_all_calls": round(stats[k][1], 1), "max_us": round(stats[k][2], 1)}
    if "GRBM_GUI_ACTIVE" in m and k in stats:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; max_us is the duration of the largest launch when set-up launches exist
        e["effective_clock_ghz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / (stats[k][2] * 1e3), 3)
    if "SQ_INSTS_VALU" in m and "SQ_BUSY_CYCLES" in m:
        trans = m.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0            # kernel cycles
            e["valu_issue_frac"] = round((m["SQ_INSTS_VALU"] * 2 + trans * 6) / 1024.0 / cyc, 3)      # wave64 on SIMD-32: 2 cycles, transcendentals 8
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"]:
        e["lane_utilisation"] = round(m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]), 3)
    if "SQ_LDS_IDX_ACTIVE" in m and "GRBM_GUI_ACTIVE" in m:
        e["lds_busy_frac"] = round(m["SQ_LDS_IDX_ACTIVE"] / (256.0 * m["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
        e["hbm_bytes"] = {"fetch_x2": round(2 * m.get("FETCH_SIZE", 0) * 1024), "write": round(m.get("WRITE_SIZE", 0) * 1024)}
        e["hbm_bytes"]["total"] = e["hbm_bytes"]["fetch_x2"] + e["hbm_bytes"]["write"]
    out[k.replace("(anonymous namespace)::", "")[:110]] = e
print(json.dumps(out, indent=1))
