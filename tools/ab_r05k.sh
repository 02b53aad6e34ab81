#!/bin/bash
# one-view calls: what the compositing kernel costs WITHOUT phase 2 (debug_skip_all: no entry passes the ellipse test)
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05k; mkdir -p $O; rm -rf $O/*
run() { echo "== $1 $2" >> $O/dropin.log; F3DG_OPTIONS=$1 timeout 200 python bench.py --workload dropin --steps 5 --warmup 2 $2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); u=d['us_per_call']
print(round(d['value']), round(d['value_deferred_status']), {k[:24]:(round(v,1) if not isinstance(v,dict) else {a:round(b,1) for a,b in v.items()}) for k,v in u.items()})" >> $O/dropin.log 2>&1; }
run "render_unroll=2" "--tile-cull 0"
run "render_unroll=2,debug_skip_all=1" "--tile-cull 0"
run "render_unroll=1,debug_skip_all=1" "--tile-cull 0"
run "render_lowocc=0,debug_skip_all=1" "--tile-cull 0"
cat $O/dropin.log
