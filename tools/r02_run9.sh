#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run9; mkdir -p $O
cd $R
for k in 2 3; do for v in 120 32 16; do F3DG_RENDER_KERNEL=$k python bench.py --no-cpu-baseline --views $v --views-per-call $v > $O/b.log 2>&1; echo "kernel $k views $v"; grep '^{' $O/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['stage_ms_per_step']['compositing']; print(c, 'per view us', 1e3*c/$v)"; done; done
