#!/bin/bash
# Collects SQ counters for the compositing kernel (separate passes, PMC only) on the GPU box.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-d2h $PMC_BENCH_ARGS"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_sq1 -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_sq2 -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $R/gpurun_out/pmc_sq3 -o b -- $B > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    f = glob.glob(f"{R}/gpurun_out/{d}/*counter_collection.csv")
    if not f: print(d, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "render" in k or "radix_scatter" in k:
            agg[("render3s" if "render3s" in k else "render3" if "render3" in k else "render2" if "render2" in k else "render_fwd" if "render_fwd" in k else "radix_scatter")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
