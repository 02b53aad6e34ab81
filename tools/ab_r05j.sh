#!/bin/bash
# one-view calls (drop-in loop): entries per phase-2 trip of the small-launch kernel (option render_unroll)
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05j; mkdir -p $O; rm -rf $O/*
timeout 300 python -m pytest tests/test_raster_forward_gpu.py -m gpu -x -q -k "small_launch_unrolled" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { echo "== $1" >> $O/dropin.log; F3DG_OPTIONS=$1 timeout 200 python bench.py --workload dropin --steps 5 --warmup 2 $2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); u=d['us_per_call']
print(round(d['value']), round(d['value_deferred_status']), {k[:24]:(round(v,1) if not isinstance(v,dict) else {a:round(b,1) for a,b in v.items()}) for k,v in u.items()})" >> $O/dropin.log 2>&1; }
for o in "render_unroll=1" "render_split=1,render_unroll=2" "render_split=2" "render_split=3"; do run $o; done
for o in "render_unroll=1" "render_split=2" "render_split=3"; do run $o "--render-mode exact"; done
cat $O/dropin.log
