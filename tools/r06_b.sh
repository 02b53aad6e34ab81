#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06b; mkdir -p $O; rm -rf $O/*
cd /tmp
timeout 120 /root/repo/tools/micro/gather64 > $O/gather64.log 2>&1
timeout 120 /root/repo/tools/micro/gather64 4718592 > $O/gather64_302MB.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=$O/pmc_$(echo $c | tr ' ' '_'); timeout 200 rocprofv3 --pmc $c --output-format csv -d $d -o g -- /root/repo/tools/micro/gather64 > /dev/null 2>&1
  d=$O/pmcbig_$(echo $c | tr ' ' '_'); timeout 200 rocprofv3 --pmc $c --output-format csv -d $d -o g -- /root/repo/tools/micro/gather64 4718592 > /dev/null 2>&1
done
cd /root/repo
python - <<'PY' > $O/gather64_pmc.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/r06b/pmc*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f.split('/')[5], k, ['%.4e' % x for x in v])
PY
cat $O/gather64.log $O/gather64_302MB.log; cat $O/gather64_pmc.txt
