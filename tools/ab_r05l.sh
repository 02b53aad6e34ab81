#!/bin/bash
# one-view launches with the defaults chosen: the small-launch tests, the small-path tests, the drop-in line in both arithmetic modes
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05l; mkdir -p $O; rm -rf $O/*
timeout 400 python -m pytest tests/test_raster_forward_gpu.py tests/test_small_path_gpu.py tests/test_python_ops_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { echo "== $1 $2" >> $O/dropin.log; F3DG_OPTIONS=$1 timeout 200 python bench.py --workload dropin --steps 5 --warmup 2 $2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); u=d['us_per_call']
print(round(d['value']), round(d['value_deferred_status']), {k[:24]:(round(v,1) if not isinstance(v,dict) else {a:round(b,1) for a,b in v.items()}) for k,v in u.items()})" >> $O/dropin.log 2>&1; }
run "" ""; run "render_split=0,render_unroll=1" ""
run "" "--render-mode exact"; run "render_split=0,render_unroll=1" "--render-mode exact"
run "" "--gaussians 262144 --res 512 --views 8"; run "render_split=0,render_unroll=1" "--gaussians 262144 --res 512 --views 8"
cat $O/dropin.log
