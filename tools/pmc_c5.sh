#!/bin/bash
# SQ counters of the C5 stress run (forward + backward), PMC-only passes.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --workload c5 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc5_1 -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc5_2 -o b -- $B > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc5_1", "pmc5_2"):
    f = glob.glob(f"{R}/gpurun_out/{d}/*counter_collection.csv")
    if not f: print(d, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "render" in k:
            agg[k.split("(")[0][-34:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: "%.3g" % (sum(v) / len(v)) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
