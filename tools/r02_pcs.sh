#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_pcs; mkdir -p $O
cd /tmp
K=${1:-2}
F3DG_RENDER_KERNEL=$K timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1000 --kernel-trace --output-format csv -d $O/k$K -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/log_k$K.txt 2>&1
echo rc=$?; tail -3 $O/log_k$K.txt; ls -la $O/k$K | head; 
f=$(ls $O/k$K/*pc_sampling*.csv 2>/dev/null | head -1); echo $f; head -3 $f; wc -l $f
# keep it small: aggregate here
python - <<PY
import csv, collections, glob
fs = glob.glob("$O/k$K/*pc_sampling*.csv")
if fs:
    c = collections.Counter()
    rows = list(csv.DictReader(open(fs[0])))
    print(rows[0].keys())
    for r in rows:
        c[(r.get("Code_Object_Id"), r.get("Code_Object_Offset"), r.get("Instruction", ""))] += 1
    with open("$O/k${K}_hist.txt", "w") as f:
        for k, v in c.most_common(4000):
            f.write("%s %s %d %s\n" % (k[0], k[1], v, k[2]))
    print("samples", len(rows), "distinct", len(c))
PY
rm -f $O/k$K/*pc_sampling*.csv
