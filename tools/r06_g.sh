#!/bin/bash
# dense backward (option bwd_dense) and the next-slide prefetch (render_prefetch): parity, then A/B
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06g; mkdir -p $O; rm -rf $O/*
F3DG_OPTIONS="bwd_dense=1" timeout 900 python -m pytest tests/test_raster_backward_gpu.py -m gpu -x -q > $O/pytest_bwd_dense.log 2>&1; tail -12 $O/pytest_bwd_dense.log
c5() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"; }
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 > $O/c5_base.log 2>&1; echo "c5 base: $(c5 $O/c5_base.log)"
F3DG_OPTIONS="bwd_dense=1" timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 > $O/c5_dense.log 2>&1; echo "c5 dense: $(c5 $O/c5_dense.log)"
F3DG_OPTIONS="bwd_dense=0" timeout 400 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -3 > $O/real_train_base.log; cat $O/real_train_base.log
F3DG_OPTIONS="bwd_dense=1" timeout 400 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -3 > $O/real_train_dense.log; cat $O/real_train_dense.log
line() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value_in_hbm']), 'ms/step', round(d['ms_per_step_in_hbm'],3), {k:round(v,3) for k,v in r['stage_ms_per_step'].items()}, r['ms_per_launch_spread'])"; }
B="timeout 400 python bench.py --no-cpu-baseline --no-d2h --no-exact --steps 10 --warmup 3"
for i in 1 2; do
$B > $O/c2_base_$i.log 2>&1; echo "c2 base: $(line $O/c2_base_$i.log)"
F3DG_OPTIONS="render_prefetch=1" $B > $O/c2_pref_$i.log 2>&1; echo "c2 prefetch: $(line $O/c2_pref_$i.log)"
done
$B --data real > $O/real_base.log 2>&1; echo "real base: $(line $O/real_base.log)"
F3DG_OPTIONS="render_prefetch=1" $B --data real > $O/real_pref.log 2>&1; echo "real prefetch: $(line $O/real_pref.log)"
F3DG_OPTIONS="render_prefetch=1" timeout 600 python -m pytest tests/test_raster_forward_gpu.py -m gpu -x -q -k "stagewise or random or packed" > $O/pytest_fwd_pref.log 2>&1; tail -3 $O/pytest_fwd_pref.log
