"""Prints value / compositing ms / frac / kernel counters of every bench log (one JSON line each) in a directory."""
import glob
import json
import os
import sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.log"))):
    line = [l for l in open(f) if l.startswith("{")]
    if not line:
        continue
    j = json.loads(line[-1])
    rf = j.get("roofline") or {}
    kc = rf.get("kernel_counters") or {}
    keep = ("fused_trips", "packed_batches", "blend_trips", "phase2_wave_trips", "slides", "stateless_lane_utilisation", "phase2_lane_utilisation")
    st = rf.get("stage_ms_per_step") or {}
    print("%-26s value %7.0f  step %.3f ms  compositing %.3f ms  frac %.3f  %s" % (
        os.path.basename(f)[:-4], j["value"], j["ms_per_step"], rf.get("ms_per_launch", 0), rf.get("frac", 0),
        {k: (round(v, 3) if isinstance(v, float) else v) for k, v in kc.items() if k in keep}))
