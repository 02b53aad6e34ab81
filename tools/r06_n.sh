#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out/r06n; mkdir -p $O; rm -rf $O/*
for cfg in "real:--data real" "s005:--sigma0 0.05"; do tag=${cfg%%:*}; a=${cfg#*:}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-d2h --no-exact $a > $O/$tag.log 2>&1
rm -f $O/$tag/*kernel_trace.csv
python - <<PY
import csv,re,glob
f=glob.glob('$O/$tag/*kernel_stats.csv')[0]
print('== $tag')
for r in list(csv.DictReader(open(f)))[:24]:
    m=re.search(r'(\w+_kernel|__amd\w+)(<[^>]*>)?', r['Name']); n=(m.group(0) if m else r['Name'])[:55]
    print("%-57s calls %4s avg %9.1f us  %6s%%" % (n, r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:6]))
PY
done
