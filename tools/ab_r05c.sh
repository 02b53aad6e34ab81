#!/bin/bash
# experiment: render4's fused loop with and without its population test (rebuilds f3dg_render4.hip on the box with -DF3DG_R4_NOPACK)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-exact --no-d2h --steps 10 --warmup 3"
F3DG_RENDER_KERNEL=3 python bench.py $B > $O/c2_k3.log 2>&1
F3DG_RENDER_KERNEL=4 F3DG_RENDER_PACK_TH=0 python bench.py $B > $O/c2_k4_th0.log 2>&1
touch f3d-gaus_amd/csrc/f3dg_render4.hip
F3DG_EXTRA_F3DG_RENDER4="-fno-slp-vectorize -DF3DG_R4_NOPACK" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build()" > $O/build.log 2>&1
F3DG_RENDER_KERNEL=4 F3DG_RENDER_PACK_TH=0 python bench.py $B > $O/c2_k4_nopack.log 2>&1
F3DG_RENDER_KERNEL=4 F3DG_RENDER_PACK_TH=0 python bench.py $B --data real > $O/real_k4_nopack.log 2>&1
python tools/ab_summary.py $O
