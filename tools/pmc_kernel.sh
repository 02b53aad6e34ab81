#!/bin/bash
# PMC passes (counters only, one rocprofv3 run per counter set) over one bench.py command, condensed to per-kernel means of the
# dispatches with the LARGEST grid of each kernel name (the timed launches: set-up launches of the same kernel on fewer views are
# left out).   usage: tools/pmc_kernel.sh <out_tag> <kernel substring> <bench.py args...>     (env F3DG_* pass through)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; KSUB=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-d2h --no-exact --no-pmc $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $B > $O/bench_under_rocprof.log 2>&1
rm -f $O/stats/*kernel_trace.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/sq1 -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o b -- $B > /dev/null 2>&1
python $R/tools/pmc_kernel.py $O "$KSUB" | tee $O/summary.txt
find $O -name "*counter_collection.csv" -size +2M -delete
