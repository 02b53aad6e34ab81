#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0
timeout 900 python -m pytest tests/test_scan_mode_gpu.py tests/test_small_path_gpu.py tests/test_percall_flags_gpu.py -m gpu -x -q 2>&1 | tail -5
run() { echo "== $1 $2"; env $1 timeout 300 python bench.py --workload dropin --steps 5 --warmup 2 --views 60 $2 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); u=d['us_per_call']
print({k:round(v) for k,v in d.items() if k.startswith('value')}, {k[:30]:(round(v,1) if not isinstance(v,dict) else {a:round(b,1) for a,b in v.items()}) for k,v in u.items() if 'stages' in k or 'alone' in k})"; }
run A=1 ""; run F3DG_OPTIONS=render_scan=1 ""
run A=1 "--gaussians 262144 --res 512"; run F3DG_OPTIONS=render_scan=1 "--gaussians 262144 --res 512"
