#!/bin/bash
# experiment: render4's variants rebuilt on the box with macro sets ($VARIANTS: "name:flags;name:flags"), C2 + real, fast arithmetic
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05i}; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-exact --no-d2h --steps 10 --warmup 3"
F3DG_RENDER_KERNEL=3 python bench.py $B > $O/c2_k3.log 2>&1
F3DG_RENDER_KERNEL=3 python bench.py $B --data real > $O/real_k3.log 2>&1
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  touch f3d-gaus_amd/csrc/f3dg_render4.hip
  F3DG_EXTRA_F3DG_RENDER4="-fno-slp-vectorize $flags" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build()" > $O/build_$name.log 2>&1
  for th in $THS; do
    F3DG_RENDER_KERNEL=4 F3DG_RENDER_PACK_TH=$th python bench.py $B > $O/c2_${name}_th$th.log 2>&1
    F3DG_RENDER_KERNEL=4 F3DG_RENDER_PACK_TH=$th python bench.py $B --data real > $O/real_${name}_th$th.log 2>&1
  done
done
python tools/ab_summary.py $O
