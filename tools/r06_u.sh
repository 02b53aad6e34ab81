#!/bin/bash
cd /root/repo; export TMPDIR=/tmp F3DG_BENCH_PMC=0; ulimit -c 0
for a in "" "--data real" "--sigma0 0.05"; do echo "== $a"; timeout 300 python bench.py --no-cpu-baseline --no-exact --no-d2h --steps 3 --warmup 1 $a 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_counters']; print(k['phase2_wave_trips'], k['phase2_lane_utilisation'], k['two_pixels_per_lane_emulation'])"; done
