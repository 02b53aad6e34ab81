#!/bin/bash
# Run on the GPU box: bench line (with cpu_baseline), rocprofv3 kernel stats, and the FETCH_SIZE / WRITE_SIZE PMC passes
# (separate runs, PMC only) of the same bench command. Results under gpurun_out/<tag>/.
TAG=${1:-prof}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
python $R/bench.py > $O/bench_default.log 2>&1
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_sq -o bench -- $B > /dev/null 2>&1
rm -f $O/stats/bench_kernel_trace.csv
grep '^{' $O/bench_default.log | tail -1
