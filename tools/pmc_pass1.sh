#!/bin/bash
# SQ counters of the integrate pass-1 kernels in a 16-camera preparation (one PMC-only pass per invocation: PASS=1|2|3) on the GPU box.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/tools/prof_pass1.py"
TAG=${1:-pass1}; PASS=${PASS:-1}
O=$R/gpurun_out/$TAG; mkdir -p $O
case $PASS in
1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU";;
2) C="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU";;
3) C="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE32_INSTS";;
esac
REPS=1 timeout 500 rocprofv3 --pmc $C --output-format csv -d $O/sq$PASS -o b -- $B > $O/log$PASS.txt 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$O/sq$PASS/*counter_collection.csv")
if not f: print("missing")
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "integrate_pass1" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[1].replace("anonymous namespace)::", "")[:40] if "(" in r["Kernel_Name"] else r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
