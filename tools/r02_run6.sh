#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run6; mkdir -p $O
cd $R
# render2: 24072 B static. blocks/CU = floor(163840 / (24072 + pad))
for pad in 0 8500 16500 30000 57000; do F3DG_RENDER_LDS_PAD=$pad F3DG_RENDER_KERNEL=2 python bench.py --no-cpu-baseline > $O/b.log 2>&1; echo "render2 pad $pad"; grep '^{' $O/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['stage_ms_per_step']['compositing'])"; done
for pad in 0 8500 22000 49000; do F3DG_RENDER_LDS_PAD=$pad F3DG_RENDER_KERNEL=3 python bench.py --no-cpu-baseline > $O/b.log 2>&1; echo "render3 pad $pad"; grep '^{' $O/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['stage_ms_per_step']['compositing'])"; done
