#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06o; mkdir -p $O; rm -rf $O/*
run() { echo "== $1 $2"; env $1 timeout 300 python bench.py --workload dropin --steps 5 --warmup 2 --views 60 $2 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); u=d['us_per_call']
print({k:round(v) for k,v in d.items() if k.startswith('value')}, {k[:30]:(round(v,1) if not isinstance(v,dict) else {a:round(b,1) for a,b in v.items()}) for k,v in u.items()})"; }
run A=1 ""; run F3DG_OPTIONS=render_scan=1 ""
run A=1 "--gaussians 262144 --res 512"; run F3DG_OPTIONS=render_scan=1 "--gaussians 262144 --res 512"
