#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run3; mkdir -p $O
cd $R
python tools/debug_r2.py 2>&1 | tail -4
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/pytest_all.log
tail -25 $O/pytest_all.log
for m in fast exact; do python bench.py --no-cpu-baseline --render-mode $m > $O/bench_$m.log 2>&1; grep '^{' $O/bench_$m.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_step'])"; done
