#!/bin/bash
cd /root/repo; export TMPDIR=/tmp F3DG_BENCH_PMC=0; ulimit -c 0
timeout 900 python -m pytest tests/test_raster_backward_gpu.py tests/test_baseline_configs_gpu.py tests/test_boundary_gpu.py -m gpu -x -q 2>&1 | tail -4
python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
for o in 1; do F3DG_OPTIONS="bwd_dense=$o" timeout 300 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 300 python tools/bench_one_view_train.py 2>&1 | grep -v amdgpu.ids | tail -2
