#!/bin/bash
# round 4: the backbone at small batches in both layouts (auto layout, cached filter copies): tests, pass times, small C4-shaped steps
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r04f; mkdir -p $O
timeout 600 python -m pytest tests/test_python_ops_gpu.py -x -q -k "backbone or group_norm or conv_bias" 2>&1 | grep -v amdgpu | tail -8 | tee $O/pytest_sel.log
for cfg in "1 fp32" "1 fp32_nhwc" "1 bf16" "1 bf16_nhwc" "2 fp32" "2 fp32_nhwc" "2 bf16" "2 bf16_nhwc" "8 fp32_nhwc" "8 bf16_nhwc" "8 bf16"; do
  python tools/prof_unet.py $cfg 2>/dev/null | grep "^###" | grep "per pass"; done | tee $O/unet_graph.md
for b in fp32 bf16; do for n in 1 4; do python bench.py --workload c4 --images $n --steps 3 --warmup 2 --backbone $b 2>&1 | grep "^{" ; done; done | tee $O/bench_c4_small.log | cut -c1-220
