#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06l; mkdir -p $O
c5() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"; }
timeout 400 python bench.py --workload c5 --steps 4 --warmup 1 > $O/c5_dense.log 2>&1; echo "c5 dense: $(c5 $O/c5_dense.log)"
timeout 400 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1
