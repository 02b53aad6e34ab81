#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06d; mkdir -p $O; rm -rf $O/*
line() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('kernel_counters') or {}
print(round(d['value_in_hbm']), 'ms/step', round(d['ms_per_step_in_hbm'],3), {k:round(v,3) for k,v in r['stage_ms_per_step'].items()}, r['kernel'][:70], {k:c.get(k) for k in ('list_entries_staged','staged_entries_reaching_a_live_pixel','list_entries_scanned','slides','phase2_wave_trips')})"; }
B="timeout 400 python bench.py --no-cpu-baseline --no-d2h --no-exact --steps 6 --warmup 2"
$B --data real > $O/real_base.log 2>&1; echo "real base: $(line $O/real_base.log)"
$B > $O/c2_base.log 2>&1; echo "c2 base: $(line $O/c2_base.log)"
$B --sigma0 0.05 > $O/s005_base.log 2>&1; echo "sigma 0.05 base: $(line $O/s005_base.log)"
F3DG_OPTIONS="render_wpb=4" $B --data real > $O/real_wpb4.log 2>&1; echo "real wpb4: $(line $O/real_wpb4.log)"
