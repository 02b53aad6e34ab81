#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06i; mkdir -p $O; rm -rf $O/*
F3DG_OPTIONS="bwd_dense=1" timeout 300 python -m pytest tests/test_raster_backward_gpu.py -m gpu -x -q > $O/pytest_bwd_dense.log 2>&1; tail -5 $O/pytest_bwd_dense.log
c5() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"; }
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 > $O/c5_base.log 2>&1; echo "c5 base: $(c5 $O/c5_base.log)"
F3DG_OPTIONS="bwd_dense=1" timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 > $O/c5_dense.log 2>&1; echo "c5 dense: $(c5 $O/c5_dense.log)"
F3DG_OPTIONS="bwd_dense=0" timeout 400 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1 > $O/real_train_base.log; cat $O/real_train_base.log
F3DG_OPTIONS="bwd_dense=1" timeout 400 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1 > $O/real_train_dense.log; cat $O/real_train_dense.log
