#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run5; mkdir -p $O
cd $R
for k in 2 3; do F3DG_DEBUG_SKIP_ALL=1 F3DG_RENDER_KERNEL=$k python bench.py --no-cpu-baseline > $O/bench_skip_k$k.log 2>&1; echo "staging-only kernel $k"; grep '^{' $O/bench_skip_k$k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_step'])"; done
