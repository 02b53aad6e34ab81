#!/bin/bash
# round 4, second GPU session: the tail schedule and the fused rectangle gather -- tests first, then A/B lines
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_raster_forward_gpu.py -x -q -k "tail or pretest or fused or batched_views or stagewise" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
for f in 1 0; do
  F3DG_OPT_SORT_FUSED_RECTS=$f python tools/ab_render.py --steps 8 --label "fused_rects=$f"
  F3DG_OPT_SORT_FUSED_RECTS=$f python tools/ab_render.py --steps 6 --gaussians 589824 --views 128 --label "fused_rects=$f"
done 2>&1 | tee $O/ab_fused.log
STEPS=6 bash tools/ab_tail.sh 2>&1 | tee $O/ab_tail.log
