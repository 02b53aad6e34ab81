#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06k; mkdir -p $O; rm -rf $O/*
for o in "bwd_dense=0" "bwd_dense=1"; do echo "== one view train, $o"; F3DG_OPTIONS="$o" timeout 300 python tools/bench_one_view_train.py 2>&1 | grep -v amdgpu.ids | tail -4; done
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
