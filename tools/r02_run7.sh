#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run7; mkdir -p $O
cd /tmp
for pad in 0 8500 16500 30000; do
F3DG_RENDER_LDS_PAD=$pad F3DG_RENDER_KERNEL=2 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES --output-format csv -d $O/pmc_$pad -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$O/pmc_$pad/*counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "render2" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("pad $pad", {c: round(sum(v)/len(v)) for c, v in agg.items()})
PY
done
