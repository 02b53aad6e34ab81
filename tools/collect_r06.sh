#!/bin/bash
# Round-6 evidence run on the GPU box (default library; everything under gpurun_out/<tag>/, every step under its own timeout): the
# bench lines DESIGN.md section 5 quotes, the frame copy issued by either thread, the split-pixel mode against the default kernel
# (bench lines, kernel counters, PMC of the timed launches), kernel stats of the default command, the GPU suite.
TAG=${1:-r06}
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R; ulimit -c 0
T="timeout 400"
export F3DG_BENCH_PMC=0        # only the two full bench lines below measure roofline.traffic live (two rocprofv3 child runs each)
python -c "import f3dgaus_amd; from f3dgaus_amd import _lib; print(_lib.lib().f3dg_version().decode())" 2>/dev/null > $O/version.txt
F3DG_BENCH_PMC=1 timeout 900 python bench.py > $O/bench_default.log 2>&1
$T python bench.py --no-cpu-baseline --d2h-issue thread > $O/bench_default_thread.log 2>&1
$T python bench.py --no-cpu-baseline --d2h-issue main > $O/bench_default_main2.log 2>&1
F3DG_BENCH_PMC=1 timeout 900 python bench.py --data real > $O/bench_real.log 2>&1
$T python bench.py --data real --scan 1 --no-cpu-baseline > $O/bench_real_scan.log 2>&1
$T python bench.py --data real --channels rgb_depth_alpha --no-cpu-baseline > $O/bench_real_lean.log 2>&1
$T python bench.py --data real --channels rgb_depth_alpha --scan 1 --no-cpu-baseline > $O/bench_real_lean_scan.log 2>&1
$T python bench.py --scan 1 --no-cpu-baseline > $O/bench_default_scan.log 2>&1
$T python bench.py --channels rgb_depth_alpha --no-cpu-baseline > $O/bench_lean.log 2>&1
$T python bench.py --sigma0 0.05 --no-cpu-baseline > $O/bench_sigma005.log 2>&1
$T python bench.py --gaussians 589824 --views 128 --no-cpu-baseline > $O/bench_589k.log 2>&1
$T python bench.py --workload dropin --views 60 > $O/bench_dropin.log 2>&1
$T python bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5.log 2>&1
$T python bench.py --workload c4 --images 16 --steps 2 --warmup 1 > $O/bench_c4_fp32.log 2>&1
python tools/scan_dist_probe.py 2>&1 | grep -v amdgpu.ids > $O/scan_dist_probe.log
# kernel stats of the default command
cd /tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-d2h --no-exact"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_under_rocprof.log 2>&1
rm -f $O/stats/bench_kernel_trace.csv
cd $R
# PMC of the timed launches: the default kernel and the split-pixel kernel on the same sets
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c2_fast render3s > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_real_fast render3s --data real > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_real_scan render5 --data real --scan 1 > /dev/null 2>&1
timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c2_scan render5 --scan 1 > /dev/null 2>&1
# the compositing backward: lock-step walk against the dense batches (C5, the real set's training step, one-view training calls)
F3DG_OPTIONS="bwd_dense=0" $T python bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5_lockstep.log 2>&1
for o in 0 1; do F3DG_OPTIONS="bwd_dense=$o" $T python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1 > $O/real_train_dense$o.log; done
for o in 0 1; do F3DG_OPTIONS="bwd_dense=$o" $T python tools/bench_one_view_train.py 2>&1 | grep -v amdgpu.ids | tail -2 > $O/one_view_train_dense$o.log; done
F3DG_OPTIONS="bwd_dense=0" timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c5_bwd3 render3_bwd --workload c5 > /dev/null 2>&1
F3DG_OPTIONS="bwd_dense=1" timeout 700 bash tools/pmc_kernel.sh $TAG/pmc_c5_bwd5 render5_bwd --workload c5 > /dev/null 2>&1
python tests/tools/parity_report.py > $O/parity_report.md 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
for f in bench_default bench_default_thread bench_default_main2 bench_real bench_real_scan bench_real_lean bench_real_lean_scan bench_default_scan bench_lean bench_sigma005 bench_589k bench_dropin bench_c5 bench_c5_lockstep bench_c4_fp32; do echo "$f: $(grep '^{' $O/$f.log | tail -1 | cut -c1-200)"; done
for d in pmc_c2_fast pmc_real_fast pmc_real_scan pmc_c2_scan pmc_c5_bwd3 pmc_c5_bwd5; do echo "== $d"; cat $O/$d/summary.txt 2>/dev/null | head -60; done
