#!/bin/bash
# Round-5 evidence run on the GPU box (everything under gpurun_out/<tag>/; every step under its own timeout). Bench lines of the
# configurations DESIGN.md section 5 quotes, kernel stats of the default command, PMC passes restricted to the timed launches
# (tools/pmc_kernel.sh) for C2 and the real merged set in both arithmetic modes with and without the packed kernel, the backbone's
# GroupNorm kernel times, MIOpen's run-to-run reproducibility, and the full GPU suite.
TAG=${1:-r05}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
T="timeout 400"
$T python bench.py > $O/bench_default.log 2>&1
$T python bench.py --data real > $O/bench_real.log 2>&1
$T python bench.py --data real --channels rgb_depth_alpha --no-cpu-baseline > $O/bench_real_lean.log 2>&1
$T python bench.py --channels rgb_depth_alpha --no-cpu-baseline > $O/bench_lean.log 2>&1
F3DG_RENDER_PACK=0 $T python bench.py --no-cpu-baseline > $O/bench_default_pack0.log 2>&1
F3DG_RENDER_PACK=0 $T python bench.py --data real --no-cpu-baseline > $O/bench_real_pack0.log 2>&1
$T python bench.py --sigma0 0.05 --no-cpu-baseline > $O/bench_sigma005.log 2>&1
$T python bench.py --gaussians 589824 --views 128 --no-cpu-baseline > $O/bench_589k.log 2>&1
$T python bench.py --workload dropin --views 60 > $O/bench_dropin.log 2>&1
$T python bench.py --workload dropin --views 60 --render-mode exact > $O/bench_dropin_exact.log 2>&1
[ -x tools/micro/issue_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rate.hip -o tools/micro/issue_rate > /dev/null 2>&1; timeout 60 tools/micro/issue_rate > $O/issue_rate.log 2>&1
( for a in "2" "3" "2 exact" "3 exact"; do echo "== render_split $a"; timeout 120 python tools/r3q_clocks.py $a 2>&1 | grep -v amdgpu.ids; done ) > $O/r3q_clocks.log 2>&1
$T python bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5.log 2>&1
$T python bench.py --workload c4 --images 16 --steps 2 --warmup 1 > $O/bench_c4_fp32.log 2>&1
$T python bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone-chunk 0 > $O/bench_c4_fp32_chunk0.log 2>&1
# kernel stats of the default command
cd /tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-d2h --no-exact"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_under_rocprof.log 2>&1
rm -f $O/stats/bench_kernel_trace.csv
cd $R
# PMC of the timed launches
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c2_fast render3s > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c2_exact render4 --render-mode exact > /dev/null 2>&1
F3DG_RENDER_PACK=0 timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c2_exact_pack0 render3s --render-mode exact > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_real_fast render3s --data real > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_real_exact render4 --data real --render-mode exact > /dev/null 2>&1
F3DG_RENDER_PACK=0 timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_real_exact_pack0 render3s --data real --render-mode exact > /dev/null 2>&1
F3DG_RENDER_PACK=1 F3DG_RENDER_PACK_TH=16 timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_real_fast_pack1 render4 --data real > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c2_preprocess preprocess_kernel > /dev/null 2>&1
# backbone: GroupNorm kernel times (8 images, channels-last), reproducibility
cd /tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_stats -o u -- python $R/tools/unet_first_use.py fp32 0 8 > $O/unet_first_use.log 2>&1
rm -f $O/unet_stats/u_kernel_trace.csv
cd $R
$T python tools/unet_determinism.py 8 2>&1 | grep -v amdgpu.ids > $O/unet_determinism.log
python tests/tools/parity_report.py > $O/parity_report.md 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q --durations=10 ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
for f in bench_default bench_real bench_real_lean bench_lean bench_default_pack0 bench_real_pack0 bench_sigma005 bench_589k bench_dropin bench_c5 bench_c4_fp32 bench_c4_fp32_chunk0; do echo "$f: $(grep '^{' $O/$f.log | tail -1 | cut -c1-260)"; done
