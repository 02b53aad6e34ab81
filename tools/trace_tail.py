"""Last N kernel dispatches of a rocprofv3 kernel trace: name, duration, gap to the previous kernel's end (us).  python tools/trace_tail.py trace.csv [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-90s %8.1f us  gap %6.1f us  grid %s" % (r["Kernel_Name"][:90], (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev = e
