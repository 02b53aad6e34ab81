#!/bin/bash
# rocprofv3 kernel stats of a command on the GPU box: tools/kstats.sh <tag> <command...>; prints the top kernels
TAG=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- "$@" > $O/log.txt 2>&1
rm -f $O/k_kernel_trace.csv
python - <<PY
import csv, glob
for r in list(csv.DictReader(open(glob.glob("$O/*kernel_stats.csv")[0])))[:22]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
    print(f"{n:54s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:10.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} {float(r['Percentage']):6.2f} %")
PY
