#!/bin/bash
# round 4, fourth GPU session: the whole GPU suite with the channels-last backbone as the default, then the backbone tables and C4 lines
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for cfg in "8 fp32_nhwc" "8 bf16_nhwc" "64 bf16_nhwc" "8 fp32" "8 bf16" "64 bf16"; do python tools/prof_unet.py $cfg 2>/dev/null | grep -v "^$" | head -60; done > $O/unet.md; grep "^###" $O/unet.md
for l in nhwc nchw; do for b in fp32 bf16; do python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone $b --backbone-layout $l 2>&1 | grep "^{" ; done; done | tee $O/bench_c4.log | cut -c1-200
python bench.py --workload c4 --images 64 --steps 1 --warmup 1 --backbone bf16 2>&1 | grep "^{" | tee $O/bench_c4_64.log | cut -c1-200
