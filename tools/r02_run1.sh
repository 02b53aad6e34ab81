#!/bin/bash
# round-2 GPU run 1: forward parity in both modes, bench fast/exact, SQ counters (fast)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run1; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_raster_forward_gpu.py -m gpu -x -q > $O/pytest_fwd.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fwd.log
tail -5 $O/pytest_fwd.log
python bench.py --no-cpu-baseline --render-mode fast > $O/bench_fast.log 2>&1; grep '^{' $O/bench_fast.log | tail -1 | cut -c1-400
python bench.py --no-cpu-baseline --render-mode exact > $O/bench_exact.log 2>&1; grep '^{' $O/bench_exact.log | tail -1 | cut -c1-400
bash tools/pmc_render.sh > $O/pmc_fast.log 2>&1; cat $O/pmc_fast.log
