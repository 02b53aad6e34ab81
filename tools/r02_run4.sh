#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run4; mkdir -p $O
cd $R
python tools/debug_r2.py 3 2>&1 | tail -4
timeout 2400 python -m pytest tests/test_raster_forward_gpu.py -m gpu -q -x > $O/pytest_fwd.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fwd.log
tail -8 $O/pytest_fwd.log
for k in 4 3; do for m in fast exact; do F3DG_RENDER_KERNEL=$k python bench.py --no-cpu-baseline --render-mode $m > $O/bench_${m}_k$k.log 2>&1; echo "kernel $k $m"; grep '^{' $O/bench_${m}_k$k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_step'])"; done; done
