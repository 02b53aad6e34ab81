#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06c; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_scan_mode_gpu.py -m gpu -x -q > $O/pytest_scan.log 2>&1; tail -15 $O/pytest_scan.log
line() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('kernel_counters') or {}
print(round(d['value_in_hbm']), 'ms/step', round(d['ms_per_step_in_hbm'],3), {k:round(v,3) for k,v in r['stage_ms_per_step'].items()}, r['kernel'][:80], {k:c.get(k) for k in ('fused_trips','dense_batches','pairs_in_dense_batches','pixels_compacted','dense_batch_lane_utilisation','fused_trip_lane_utilisation','phase2_wave_trips','phase2_lane_utilisation','slides')})"; }
B="timeout 400 python bench.py --no-cpu-baseline --no-d2h --no-exact --steps 6 --warmup 2"
for th in 10; do
  F3DG_OPTIONS="render_scan_th=$th" $B --data real --scan 1 > $O/real_scan_$th.log 2>&1; echo "real scan th=$th: $(line $O/real_scan_$th.log)"
done
$B --data real > $O/real_base.log 2>&1; echo "real base: $(line $O/real_base.log)"
for th in 10; do
  F3DG_OPTIONS="render_scan_th=$th" $B --data real --scan 1 --channels rgb_depth_alpha > $O/real_lean_scan_$th.log 2>&1; echo "real lean scan th=$th: $(line $O/real_lean_scan_$th.log)"
done
$B --data real --channels rgb_depth_alpha > $O/real_lean_base.log 2>&1; echo "real lean base: $(line $O/real_lean_base.log)"
for th in 10; do
  F3DG_OPTIONS="render_scan_th=$th" $B --scan 1 > $O/c2_scan_$th.log 2>&1; echo "c2 scan th=$th: $(line $O/c2_scan_$th.log)"
done
$B > $O/c2_base.log 2>&1; echo "c2 base: $(line $O/c2_base.log)"
