#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run8; mkdir -p $O
cd $R
python tools/debug_r2.py 2 2>&1 | tail -4
for occ in 6 5; do for m in fast exact; do F3DG_RENDER_OCC=$occ F3DG_RENDER_KERNEL=2 python bench.py --no-cpu-baseline --render-mode $m > $O/b.log 2>&1; echo "render2 occ $occ $m"; grep '^{' $O/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['stage_ms_per_step']['compositing'])"; done; done
