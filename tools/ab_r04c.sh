#!/bin/bash
# round 4, third GPU session: backbone with folded biases / residual join in both layouts; projection grid order and streaming stores
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_python_ops_gpu.py -x -q -s -k "group_norm or backbone or conv_bias" 2>&1 | grep -v amdgpu | tail -8 | tee $O/pytest_sel.log
for cfg in "8 fp32" "8 fp32_nhwc" "8 bf16" "8 bf16_nhwc" "64 bf16" "64 bf16_nhwc"; do python tools/prof_unet.py $cfg 2>/dev/null | grep -v "^$" | head -60; done > $O/unet.md; grep "^###" $O/unet.md
for o in 0 1 2 3; do
  F3DG_OPT_PRE_ORDER=$o python tools/ab_render.py --steps 8 --label "pre_order=$o"
  F3DG_OPT_PRE_ORDER=$o python tools/ab_render.py --steps 6 --gaussians 589824 --views 128 --label "pre_order=$o"
done 2>&1 | grep -v amdgpu | tee $O/ab_pre_order.log
for l in nchw nhwc; do for b in fp32 bf16; do python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone $b --backbone-layout $l 2>&1 | grep "^{" ; done; done | tee $O/bench_c4.log | cut -c1-200
