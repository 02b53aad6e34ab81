#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run2; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_raster_forward_gpu.py -m gpu -x -q > $O/pytest_fwd.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fwd.log
tail -15 $O/pytest_fwd.log
python bench.py --no-cpu-baseline --render-mode fast > $O/bench_fast.log 2>&1; grep '^{' $O/bench_fast.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_step'])"
python bench.py --no-cpu-baseline --render-mode exact > $O/bench_exact.log 2>&1; grep '^{' $O/bench_exact.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_step'])"
