#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06j; mkdir -p $O; rm -rf $O/*
c5() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"; }
for o in "bwd_dense=0" "bwd_dense=1" "bwd_dense=0,small_debug=9" "bwd_dense=1,small_debug=9"; do
F3DG_OPTIONS="$o" timeout 300 python bench.py --workload c5 --steps 3 --warmup 1 > $O/c5.log 2>&1; echo "c5 $o: $(c5 $O/c5.log)"
done
for o in "bwd_dense=0,small_debug=9" "bwd_dense=1,small_debug=9"; do
F3DG_OPTIONS="$o" timeout 300 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1
done
