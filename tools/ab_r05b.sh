#!/bin/bash
# Round 5: render3s against the rank-packed kernel at several thresholds, fast and reference arithmetic, C2 and the real merged set
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_raster_forward_gpu.py -x -q -m gpu -k "packed" > $O/pytest_packed.log 2>&1
tail -3 $O/pytest_packed.log
B="--no-cpu-baseline --no-exact --no-d2h --steps 10 --warmup 3"
CFGS=${CFGS:-3:32 4:64 4:40 4:32 4:24 4:16 4:0}
for cfg in $CFGS; do
  k=${cfg%:*}; th=${cfg#*:}
  F3DG_RENDER_KERNEL=$k F3DG_RENDER_PACK_TH=$th python bench.py $B > $O/c2_k${k}_th$th.log 2>&1
  F3DG_RENDER_KERNEL=$k F3DG_RENDER_PACK_TH=$th python bench.py $B --data real > $O/real_k${k}_th$th.log 2>&1
  [ -n "$NOEXACT" ] || F3DG_RENDER_KERNEL=$k F3DG_RENDER_PACK_TH=$th python bench.py $B --render-mode exact > $O/c2x_k${k}_th$th.log 2>&1
  [ -n "$NOEXACT" ] || F3DG_RENDER_KERNEL=$k F3DG_RENDER_PACK_TH=$th python bench.py $B --render-mode exact --data real > $O/realx_k${k}_th$th.log 2>&1
done
python tools/ab_summary.py $O
