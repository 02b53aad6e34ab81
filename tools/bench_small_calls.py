"""Per-call cost of small batched renders (the reference's one-view-per-call loops and the cycle-aggregation regime of 8 views
per call): wall time per call in a back-to-back loop (check=False: no host sync) against the host's issue time. Every
configuration runs in its own process (tools/prof_small.py): in one process the allocator state left by the previous
configuration made single lines 2-3x slower from run to run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for V, P in ((1, 65536), (2, 65536), (1, 131072), (1, 196608), (3, 196608), (8, 65536), (8, 196608), (8, 589824), (12, 196608)):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_small.py"), str(P), str(V)], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("P=")]
    print(line[-1] if line else "P=%d V=%d: FAILED %s" % (P, V, r.stderr[-300:]), flush=True)
