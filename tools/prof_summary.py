"""Condenses rocprofv3 output (kernel stats csv + optional PMC counter csvs) into a small markdown summary that is
committed under profiles/. Usage: python tools/prof_summary.py <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>] > profiles/x.md"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    stats_dir = sys.argv[1]
    f = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    if len(sys.argv) >= 4:
        print()
        print("PMC (separate passes; KB per dispatch as reported by rocprofv3; `FETCH x2` applies the gfx950 correction of "
              "MI355X_MICROARCH.md section HBM for wide coalesced reads; WRITE_SIZE is uncalibrated):")
        print()
        print("| kernel | dispatches | FETCH_SIZE KB (mean) | FETCH x2 MB | WRITE_SIZE KB (mean) | WRITE MB |")
        print("|---|---:|---:|---:|---:|---:|")
        agg = {}
        for key, d in (("fetch", sys.argv[2]), ("write", sys.argv[3])):
            f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
            a = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                a[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            agg[key] = a
        for k in sorted(agg["fetch"], key=lambda k: -sum(agg["fetch"][k])):
            fv, wv = agg["fetch"][k], agg["write"].get(k, [0.0])
            fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
            print(f"| `{k}` | {len(fv)} | {fm:.1f} | {2 * fm * 1024 / 1e6:.1f} | {wm:.1f} | {wm * 1024 / 1e6:.1f} |")


if __name__ == "__main__":
    main()
