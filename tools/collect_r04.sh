#!/bin/bash
# Round-4 evidence run on the GPU box (everything under gpurun_out/<tag>/). As tools/collect_r03.sh, plus: the compositing kernel
# BEFORE (round 3's fast arithmetic and 12-byte LDS reads, rebuilt with -DF3DG_FAST_R03=1 -DF3DG_R3_B128=0) and AFTER in the same run
# on the same box with kernel stats and SQ counters; the projection kernel's float64 / transcendental instruction split; the real-image
# bench line (--data real) with kernel stats; the drop-in line with the deferred status; the backbone's operator tables
# (tools/prof_unet.py); the fraction of (view, Gaussian) pairs in no list (tools/culled_fraction.py).
TAG=${1:-r04}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD"
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $O/avail_valu_counters.txt
python $R/bench.py > $O/bench_default.log 2>&1
python $R/bench.py --sigma0 0.05 --no-cpu-baseline > $O/bench_sigma005.log 2>&1
python $R/bench.py --tile-cull 0 --no-cpu-baseline > $O/bench_nocull.log 2>&1
python $R/bench.py --gaussians 589824 --views 128 --no-cpu-baseline > $O/bench_589k.log 2>&1
python $R/bench.py --data real > $O/bench_real.log 2>&1
python $R/bench.py --workload dropin --views 60 > $O/bench_dropin.log 2>&1
python $R/bench.py --workload dropin --views 60 --gaussians 589824 > $O/bench_dropin_589k.log 2>&1
python $R/tools/bench_small_calls.py > $O/small_calls.log 2>&1
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-d2h --no-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_sq -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc $SQ2 --output-format csv -d $O/pmc_sq2 -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o bench -- $B > /dev/null 2>&1
# float64 / transcendental split of the VALU work (the counters this box offers are listed in avail_valu_counters.txt)
F64=$(grep -E "F64|TRANS" $O/avail_valu_counters.txt | head -8 | tr '\n' ' ')
[ -n "$F64" ] && rocprofv3 --pmc $F64 --output-format csv -d $O/pmc_f64 -o bench -- $B > /dev/null 2>&1
rm -f $O/stats/bench_kernel_trace.csv
# the exact-arithmetic kernel of the same workload
BX="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-d2h --render-mode exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_exact -o bench -- $BX > /dev/null 2>&1
rm -f $O/stats_exact/bench_kernel_trace.csv
# BEFORE: round 3's compositing arithmetic and LDS reads, same box, same run
cd $R; touch f3d-gaus_amd/csrc/f3dg_render.hip
F3DG_EXTRA_F3DG_RENDER="-fno-slp-vectorize -DF3DG_FAST_R03=1 -DF3DG_R3_B128=0" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build()" > $O/build_before.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/before_stats -o bench -- $B > $O/before_under_rocprof.log 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/before_pmc_sq -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc $SQ2 --output-format csv -d $O/before_pmc_sq2 -o bench -- $B > /dev/null 2>&1
rm -f $O/before_stats/bench_kernel_trace.csv
cd $R; touch f3d-gaus_amd/csrc/f3dg_render.hip; python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build()" >> $O/build_before.log 2>&1; cd /tmp
# sigma0 = 0.05
BS="python $R/bench.py --sigma0 0.05 --steps 3 --warmup 1 --no-cpu-baseline --no-d2h --no-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s005_stats -o bench -- $BS > $O/s005_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/s005_pmc_fetch -o bench -- $BS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/s005_pmc_write -o bench -- $BS > /dev/null 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/s005_pmc_sq -o bench -- $BS > /dev/null 2>&1
rm -f $O/s005_stats/bench_kernel_trace.csv
# larger set and the real merged set: kernel stats (+ SQ counters of the real one)
BL="python $R/bench.py --gaussians 589824 --views 128 --steps 3 --warmup 1 --no-cpu-baseline --no-d2h --no-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/l589_stats -o bench -- $BL > /dev/null 2>&1
rm -f $O/l589_stats/bench_kernel_trace.csv
BR="python $R/bench.py --data real --steps 3 --warmup 1 --no-cpu-baseline --no-d2h --no-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/real_stats -o bench -- $BR > $O/real_under_rocprof.log 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/real_pmc_sq -o bench -- $BR > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/real_pmc_fetch -o bench -- $BR > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/real_pmc_write -o bench -- $BR > /dev/null 2>&1
rm -f $O/real_stats/bench_kernel_trace.csv
# small call: kernel stats of the one-view path
rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_stats -o s -- python $R/tools/prof_small.py 65536 1 > /dev/null 2>&1
rm -f $O/small_stats/s_kernel_trace.csv
# C5
C5="python $R/bench.py --workload c5 --steps 3 --warmup 1"
$C5 > $O/bench_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o c5 -- $C5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_pmc_fetch -o c5 -- $C5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5_pmc_write -o c5 -- $C5 > /dev/null 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/c5_pmc_sq -o c5 -- $C5 > /dev/null 2>&1
rm -f $O/c5_stats/c5_kernel_trace.csv
# C4 shape (C3 at one GPU)
python $R/bench.py --workload c4 --images 16 --steps 2 --warmup 1 > $O/bench_c4_fp32.log 2>&1
python $R/bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone bf16 > $O/bench_c4_bf16.log 2>&1
python $R/bench.py --workload c4 --images 64 --steps 1 --warmup 1 --backbone bf16 > $O/bench_c4_bf16_64.log 2>&1
python $R/bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone-layout nchw > $O/bench_c4_fp32_nchw.log 2>&1
python $R/bench.py --workload c4 --images 16 --steps 2 --warmup 1 --backbone bf16 --backbone-layout nchw > $O/bench_c4_bf16_nchw.log 2>&1
# backbone operator tables
# (the channels-last layout is the default of the predictor; plain "fp32" / "bf16" = torch's NCHW layout with the same fused kernels)
for cfg in "8 fp32_nhwc" "8 bf16_nhwc" "64 bf16_nhwc" "8 fp32" "8 bf16" "64 bf16"; do python $R/tools/prof_unet.py $cfg 2>> $O/unet.err | grep -v "^$" | head -64 >> $O/unet.md; done
python $R/tools/ubench_conv_layout.py 2>/dev/null | grep -v amdgpu > $O/conv_layout.md
python $R/tools/culled_fraction.py > $O/culled.md 2>&1
# integrate (Gaussians -> points)
python $R/tools/bench_integrate.py > $O/bench_integrate.log 2>&1
for c in 0 1; do CFG=$c rocprofv3 --kernel-trace --stats --output-format csv -d $O/int_stats$c -o int -- python $R/tools/bench_integrate.py > $O/int_stats$c.log 2>&1; rm -f $O/int_stats$c/int_kernel_trace.csv; done
V=16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/int_sweep_stats -o int -- python $R/tools/bench_integrate.py > /dev/null 2>&1; rm -f $O/int_sweep_stats/int_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1_stats -o p1 -- python $R/tools/prof_pass1.py > $O/p1_stats.log 2>&1; rm -f $O/p1_stats/p1_kernel_trace.csv
PASS=1 bash $R/tools/pmc_pass1.sh $TAG/p1_pmc > /dev/null 2>&1; PASS=2 bash $R/tools/pmc_pass1.sh $TAG/p1_pmc > /dev/null 2>&1
cd $R
python tests/tools/parity_report.py > $O/parity_report.md 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
for f in bench_default bench_sigma005 bench_nocull bench_589k bench_real bench_dropin bench_c5 bench_c4_fp32 bench_c4_bf16 bench_c4_bf16_64 bench_c4_fp32_nchw bench_c4_bf16_nchw; do grep '^{' $O/$f.log | tail -1 | cut -c1-300; done
