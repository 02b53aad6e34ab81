#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06f; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_scan_mode_gpu.py -m gpu -x -q > $O/pytest_scan.log 2>&1; tail -3 $O/pytest_scan.log
line() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('kernel_counters') or {}
print(round(d['value_in_hbm']), 'ms/step', round(d['ms_per_step_in_hbm'],3), {k:round(v,3) for k,v in r['stage_ms_per_step'].items()}, r['kernel'][:60], {k:c.get(k) for k in ('fused_trips','dense_batches','pairs_in_dense_batches','pixels_compacted','dense_batch_lane_utilisation')})"; }
B="timeout 400 python bench.py --no-cpu-baseline --no-d2h --no-exact --steps 6 --warmup 2"
$B --data real > $O/real_base.log 2>&1; echo "real base: $(line $O/real_base.log)"
for cfg in "12 0" "12 4" "12 8" "16 6" "20 8" "24 10"; do set -- $cfg
  F3DG_OPTIONS="render_scan_th=$1,render_scan_min=$2" $B --data real --scan 1 > $O/real_scan_$1_$2.log 2>&1; echo "real scan th=$1 min=$2: $(line $O/real_scan_$1_$2.log)"
done
$B > $O/c2_base.log 2>&1; echo "c2 base: $(line $O/c2_base.log)"
for cfg in "12 4" "12 8"; do set -- $cfg
  F3DG_OPTIONS="render_scan_th=$1,render_scan_min=$2" $B --scan 1 > $O/c2_scan_$1_$2.log 2>&1; echo "c2 scan th=$1 min=$2: $(line $O/c2_scan_$1_$2.log)"
done
