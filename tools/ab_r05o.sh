#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05o; mkdir -p $O; rm -rf $O/*
timeout 400 python -m pytest tests/test_small_path_gpu.py tests/test_python_ops_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { echo "== $1 $2" >> $O/dropin.log; env $1 timeout 300 python bench.py --workload dropin --steps 3 --warmup 1 --views 30 $2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); u=d['us_per_call']
print({k:round(v) for k,v in d.items() if k.startswith('value')}, {k[:24]:(round(v,1) if not isinstance(v,dict) else {a:round(b,1) for a,b in v.items()}) for k,v in u.items()}, d['config']['kernel_launches_per_call'])" >> $O/dropin.log 2>&1; }
run A=1 ""; run F3DG_DROPIN_RANDOM_IDS=1 ""; run A=1 "--gaussians 262144 --res 512"; run A=1 "--gaussians 589824"; run F3DG_DROPIN_RANDOM_IDS=1 "--gaussians 589824"
cat $O/dropin.log
