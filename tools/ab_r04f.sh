#!/bin/bash
# round 4: small backbone passes as HIP graphs -- test, pass times at 1 / 2 / 4 images, the single-image C4-shaped step with and without
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r04f; mkdir -p $O
timeout 600 python -m pytest tests/test_python_ops_gpu.py -x -q -k "graph or backbone" 2>&1 | grep -v amdgpu | tail -8 | tee $O/pytest_sel.log
for cfg in "1 fp32_nhwc" "1 fp32_nhwc_graph" "1 bf16_nhwc" "1 bf16_nhwc_graph" "2 fp32_nhwc_graph" "4 fp32_nhwc" "4 fp32_nhwc_graph" "4 bf16_nhwc" "4 bf16_nhwc_graph"; do
  python tools/prof_unet.py $cfg 2>/dev/null | grep "^###" | grep "per pass"; done | tee $O/unet_graph.md
for b in fp32 bf16; do for n in 1 4; do python bench.py --workload c4 --images $n --steps 3 --warmup 2 --backbone $b 2>&1 | grep "^{" ; done; done | tee $O/bench_c4_small.log | cut -c1-220
